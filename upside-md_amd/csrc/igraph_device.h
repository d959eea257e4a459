// Device-side building blocks shared by the interaction-graph kernels (kernels_igraph.hip, kernels_rotamer.hip).
//
// Work decomposition on gfx950:
//   * one workgroup (up to 16 wavefronts) serves ONE system at a time; it first stages into LDS
//       - the whole spline/parameter table of the graph (6-99 KB), and
//       - the packed coordinates of every element of both sides (8 floats per element, <= ~60 KB),
//     so the per-pair gathers (4 coefficient windows + neighbour coordinates) hit LDS instead of issuing
//     64-line scattered global loads per wave instruction;
//   * one wavefront owns a few consecutive rows of the cached Verlet list at a time; candidates are distance-tested
//     64 at a time and the survivors of ALL its rows are compacted (ballot + popcount) into one small per-wave LDS
//     queue, so that the expensive pair functor always runs with 64 busy lanes;
//   * row results are recovered by a segmented wavefront reduction into LDS accumulators; nothing is scattered.
#pragma once
#include "device_math.h"
#include "../../include/upside_hip_kernels.h"

#include <cstdlib>
// total workgroups an LDS-staged pair kernel aims at (each re-stages its system, so fewer + fatter is cheaper;
// 2 x 16-wave workgroups fill a CU)
#define DR_CHUNK 8              // most rows per work item (accumulated in LDS, then flushed)
#define DR_QUEUE 128
#define DR_WAVE_LDS (DR_QUEUE + DR_CHUNK * 8)   // 32-bit words per wave
// rows per work item: 1 for a single system (latency), a quarter of a wave's share as the chip fills up (~4096 waves
// in flight) so that the LDS counter can still balance the waves, 6 at most: measured on the 300-residue benchmark,
// system-steps/s at 1024 systems by rows per item: 1: 61.2 k, 2: 63.2 k, 4: 64.2 k, 6: 64.7 k, 8: 64.4 k, 16: 63.4 k
static inline int dr_chunk_rows(int n_system, int n_rows) {
    static int forced = -1;   // UPSIDE_HIP_DR_CHUNK=n pins it (experiments)
    if (forced < 0) { const char* e = getenv("UPSIDE_HIP_DR_CHUNK"); forced = e ? atoi(e) : 0; }
    if (forced > 0) return forced > DR_CHUNK ? DR_CHUNK : forced;
    long c = ((long)n_system * n_rows) / (4096 * 4);
    return c < 1 ? 1 : (c > 6 ? 6 : (int)c);
}
static inline int ig_target_wgs() {
    static int v = 0;
    if (!v) { const char* e = getenv("UPSIDE_HIP_IG_WGS"); v = e ? atoi(e) : 256; if (v < 1) v = 256; }
    return v;
}

namespace up {


__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wavefront execute in order; this only stops the compiler from reordering them and
    // makes the data written by other lanes of the SAME wave visible to subsequent reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// uniform cubic B-spline in basis form: value and derivative from the 4-coefficient window starting at c[bin-1]
// (same interpolant as spline.h:136-174; de Boor's recurrence re-associated into the 4 basis polynomials)
__device__ __forceinline__ void bspline_basis(float y, float b[4], float d[4]) {
    // Horner forms with the 1/6 folded into the constants: 18 operations for the 8 numbers (the expanded polynomials
    // cost 30, and three bases are evaluated per pair)
    const float y2 = y * y, omy = 1.f - y, omy2 = omy * omy;
    const float s = 1.f / 6.f;
    b[0] = omy2 * (omy * s);
    b[1] = fmaf(fmaf(0.5f, y, -1.f), y2, 2.f / 3.f);
    b[2] = fmaf(fmaf(fmaf(-0.5f, y, 0.5f), y, 0.5f), y, s);
    b[3] = y2 * (y * s);
    d[0] = -0.5f * omy2;
    d[1] = y * fmaf(1.5f, y, -2.f);
    d[2] = fmaf(fmaf(-1.5f, y, 1.f), y, 0.5f);
    d[3] = 0.5f * y2;
}
template <typename P>
__device__ __forceinline__ void bspline_vd(float& val, float& der, P c, int bin, const float b[4], const float d[4]) {
    const float c0 = c[bin - 1], c1 = c[bin], c2 = c[bin + 1], c3 = c[bin + 2];
    val = c0 * b[0] + c1 * b[1] + c2 * b[2] + c3 * b[3];
    der = c0 * d[0] + c1 * d[1] + c2 * d[2] + c3 * d[3];
}

struct QuadShape { int ka, k; float inv_dx, inv_dtheta; };

// bead_interaction.h:30-84 with the B-splines in basis form and hardware rsqrt (1 ulp).
// WANT_D: 0 value only, 1 derivative w.r.t. the first element, 2 w.r.t. the second.
template <int WANT_D, typename P>
__device__ __forceinline__ float quadspline2(const QuadShape& Q, P p, const float* x1, const float* x2, float* d) {
    const f3 displace = mk3(x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]);
    const f3 rvec1 = mk3(x1[3], x1[4], x1[5]), rvec2 = mk3(x2[3], x2[4], x2[5]);
    const float dist2 = mag2(displace), inv_dist = rsqrtf(dist2);
    const float dist_coord = dist2 * (inv_dist * Q.inv_dx);
    const f3 u = inv_dist * displace;
    const float cos1 = dot(rvec1, u), cos2 = -dot(rvec2, u);
    float b[4], db[4];
    // angular splines (unclamped, spline.h:228-242)
    float a1, da1, a2, da2;
    {
        const float x = (cos1 + 1.f) * Q.inv_dtheta + 1.f; const int bin = (int)x;
        bspline_basis(x - (float)bin, b, db); bspline_vd(a1, da1, p, bin, b, db);
    }
    {
        const float x = (cos2 + 1.f) * Q.inv_dtheta + 1.f; const int bin = (int)x;
        bspline_basis(x - (float)bin, b, db); bspline_vd(a2, da2, p + Q.ka, bin, b, db);
    }
    // radial splines share one coordinate; clamped ends (spline.h:275-310)
    float wide, dwide, narrow, dnarrow;
    {
        const bool too_small = dist_coord < 1.f, too_big = (float)(Q.k - 2) <= dist_coord;
        const float xc = (too_small || too_big) ? 1.f : dist_coord;
        const int bin = (int)xc;
        bspline_basis(xc - (float)bin, b, db);
        bspline_vd(wide, dwide, p + 2 * Q.ka, bin, b, db);
        bspline_vd(narrow, dnarrow, p + 2 * Q.ka + Q.k, bin, b, db);
        if (too_small || too_big) {
            dwide = 0.f; dnarrow = 0.f;
            const int o = too_small ? 0 : Q.k - 3;
            P pw = p + 2 * Q.ka, pn = p + 2 * Q.ka + Q.k;
            wide = (1.f / 6.f) * pw[o] + (2.f / 3.f) * pw[o + 1] + (1.f / 6.f) * pw[o + 2];
            narrow = (1.f / 6.f) * pn[o] + (2.f / 3.f) * pn[o + 1] + (1.f / 6.f) * pn[o + 2];
        }
    }
    const float angular_weight = a1 * a2;
    if (WANT_D) {
        const float radial_deriv = Q.inv_dx * (dwide + angular_weight * dnarrow);
        const float angular_deriv1 = Q.inv_dtheta * da1 * a2 * narrow;
        const float angular_deriv2 = Q.inv_dtheta * a1 * da2 * narrow;
        const f3 rXX = angular_deriv1 * rvec1 - angular_deriv2 * rvec2;
        const f3 deriv_dir = inv_dist * (rXX - dot(u, rXX) * u);
        const f3 dd = radial_deriv * u + deriv_dir;
        if (WANT_D == 1) {
            d[0] = -dd.x; d[1] = -dd.y; d[2] = -dd.z;
            d[3] = angular_deriv1 * u.x; d[4] = angular_deriv1 * u.y; d[5] = angular_deriv1 * u.z;
        } else {
            d[0] = dd.x; d[1] = dd.y; d[2] = dd.z;
            d[3] = -angular_deriv2 * u.x; d[4] = -angular_deriv2 * u.y; d[5] = -angular_deriv2 * u.z;
        }
    }
    return wide + angular_weight * narrow;
}

// Parameter derivative of one quadspline pair (bead_interaction.h:86-130): the value is linear in the spline
// coefficients, so d(value)/d(coefficient) is the basis weight of that coefficient (spline.h:318-336, 375-392) times
// the other factors of  wide + a1*a2*narrow.  `scale` (pair sensitivity) * those weights is added to the parameter
// row `out` of the pair's type combination.  Not on the MD path: plain global atomics.
template <typename P>
__device__ __forceinline__ void quadspline_param_accum(const QuadShape& Q, P p, const float* x1, const float* x2, float scale, float* out) {
    const f3 displace = mk3(x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]);
    const f3 rvec1 = mk3(x1[3], x1[4], x1[5]), rvec2 = mk3(x2[3], x2[4], x2[5]);
    const float dist2 = mag2(displace), inv_dist = rsqrtf(dist2);
    const float dist_coord = dist2 * (inv_dist * Q.inv_dx);
    const f3 u = inv_dist * displace;
    const float cos1 = dot(rvec1, u), cos2 = -dot(rvec2, u);
    float b1[4], b2[4], br[4], db[4];
    const float xa1 = (cos1 + 1.f) * Q.inv_dtheta + 1.f, xa2 = (cos2 + 1.f) * Q.inv_dtheta + 1.f;
    const int bin1 = (int)xa1, bin2 = (int)xa2;
    float a1, a2, narrow, unused;
    bspline_basis(xa1 - (float)bin1, b1, db); bspline_vd(a1, unused, p, bin1, b1, db);
    bspline_basis(xa2 - (float)bin2, b2, db); bspline_vd(a2, unused, p + Q.ka, bin2, b2, db);
    int rbin;   // first coefficient of the radial window
    if (dist_coord <= 1.f) { rbin = 0; br[0] = 1.f / 6.f; br[1] = 2.f / 3.f; br[2] = 1.f / 6.f; br[3] = 0.f; }
    else if (dist_coord >= (float)(Q.k - 2)) { rbin = Q.k - 4; br[0] = 0.f; br[1] = 1.f / 6.f; br[2] = 2.f / 3.f; br[3] = 1.f / 6.f; }
    else { const int bin = (int)dist_coord; rbin = bin - 1; bspline_basis(dist_coord - (float)bin, br, db); }
    {
        P pn = p + 2 * Q.ka + Q.k;
        narrow = pn[rbin] * br[0] + pn[rbin + 1] * br[1] + pn[rbin + 2] * br[2] + pn[rbin + 3] * br[3];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        atomicAdd(out + 2 * Q.ka + rbin + i, scale * br[i]);
        atomicAdd(out + 2 * Q.ka + Q.k + rbin + i, scale * (a1 * a2 * br[i]));
        atomicAdd(out + bin1 - 1 + i, scale * (a2 * narrow * b1[i]));
        atomicAdd(out + Q.ka + bin2 - 1 + i, scale * (a1 * narrow * b2[i]));
    }
}

// cooperative staging of one system's packed elements: element i of `node` (gathered through `loc`) becomes the
// 8-float LDS row  [0,dim) coordinates | [6] aux0 | [7] aux1  where the aux words carry per-element metadata so
// that the pair loop never touches global memory for them:
//   aux1 = integer `meta1[i]` (e.g. the element type) as raw bits,
//   aux0 = sens[i*sens_stride] (a per-element pair sensitivity) when dim <= 6 and sens != nullptr,
//          else integer `meta0[i]` as raw bits when meta0 != nullptr.
__device__ __forceinline__ void stage_rows(float* lds, const upk_coord_t& node, int s, const int* __restrict__ loc, int n, int dim,
                                           const int* __restrict__ meta1, const int* __restrict__ meta0,
                                           const float* __restrict__ sens, int sens_stride) {
    const float* base = node.out + (size_t)s * node.n_elem * node.stride;
    for (int t = threadIdx.x; t < n * 8; t += blockDim.x) {
        const int i = t >> 3, c = t & 7;
        float v = 0.f;
        if (c < dim) v = base[(size_t)loc[i] * node.stride + c];
        else if (c == 7 && meta1) v = __int_as_float(meta1[i]);
        else if (c == 6 && sens) v = sens[(size_t)i * sens_stride];
        else if (c == 6 && meta0) v = __int_as_float(meta0[i]);
        lds[t] = v;
    }
}
__device__ __forceinline__ void stage_coords(float* lds, const upk_coord_t& node, int s, const int* __restrict__ loc, int n, int dim) {
    stage_rows(lds, node, s, loc, n, dim, nullptr, nullptr, nullptr, 0);
}
__device__ __forceinline__ void stage_table(float* lds, const float* __restrict__ tab, int n) {
    for (int t = threadIdx.x; t < n; t += blockDim.x) lds[t] = tab[t];
}

// ---- dense-lane pair loop over a contiguous CHUNK of rows owned by one wavefront -----------------------------
// A row has 10-90 in-range neighbours, so running the pair functor row by row leaves about half of the 64 lanes
// idle (a 66-hit row costs two functor passes).  Here the hits of consecutive rows share one queue: the functor
// always runs on 64 queued (row, neighbour) entries, whatever rows they belong to, and the per-row sums are
// recovered by a segmented wave reduction into a small per-wave LDS accumulator (rows are contiguous in the queue,
// a batch spans ~1-6 of them).  Per wave: DR_QUEUE queue words, DR_CHUNK x 8 floats.

//   row_xyz(row, x[3]): position of the row element (wave-uniform; kept in registers while its list is scanned)
//   test(x, row, k, j, payload&) -> is cached neighbour k (= element j) of `row` in range?  payload < 2^28
//   batch(row_local, payload, valid): called with ALL lanes converged on 64 (or, at the end of the chunk, fewer) entries
// One queue word per hit: row_local << 28 | payload (DR_CHUNK <= 16).
template <typename RowFn, typename TestFn, typename BatchFn>
__device__ __forceinline__ void dense_row_loop(int cb, int ce, const int* __restrict__ cnt_arr, const int* __restrict__ nbr_base, int cap,
                                               int lane, int* q, RowFn row_xyz, TestFn test, BatchFn batch) {
    int nq = 0;
    const int my_cnt = lane < ce - cb ? cnt_arr[cb + lane] : 0;       // the chunk's list lengths in one load
    for (int row = cb; row < ce; ++row) {
        const int cnt = __builtin_amdgcn_readlane(my_cnt, row - cb);      // wave-uniform lane index: v_readlane, not a shuffle
        const int* __restrict__ nbr = nbr_base + (size_t)row * cap;
        float x[3];
        row_xyz(row, x);
        for (int k0 = 0; k0 < cnt; k0 += 256) {
            // the list scan is a chain of dependent global loads per wave: keep four of them in flight
            int jj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int k = k0 + u * 64 + lane; jj[u] = k < cnt ? nbr[k] : -1; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + u * 64 >= cnt) break;
                const int k = k0 + u * 64 + lane;
                int pay = 0; bool hit = false;
                if (jj[u] >= 0) hit = test(x, row, k, jj[u], pay);
                const unsigned long long m = __ballot(hit);
                if (hit) q[nq + __popcll(m & ((1ull << lane) - 1ull))] = ((row - cb) << 28) | pay;
                nq += __popcll(m);
                wave_lds_fence();
                if (nq >= 64) {
                    const int w = q[lane];
                    const bool more = lane + 64 < nq;
                    const int keep = more ? q[lane + 64] : 0;
                    wave_lds_fence();
                    batch((int)((unsigned)w >> 28), w & 0x0FFFFFFF, true);
                    if (more) q[lane] = keep;
                    nq -= 64;
                    wave_lds_fence();
                }
            }
        }
    }
    if (nq > 0) {
        const bool valid = lane < nq;
        const int w = valid ? q[lane] : 0;
        wave_lds_fence();
        batch((int)((unsigned)w >> 28), w & 0x0FFFFFFF, valid);
    }
}
// acc[rl*8 + c] += sum over the lanes of row rl of v[c]; lanes of one row are adjacent, invalid lanes carry nothing
template <int N>
__device__ __forceinline__ void seg_accumulate(float* acc, int rl, bool valid, const float v[8], int lane) {
    unsigned long long pending = __ballot(valid);
    while (pending) {
        const int r0 = __builtin_amdgcn_readlane(rl, __builtin_ctzll(pending));
        const bool mine = valid && rl == r0;
        if (N == 1) {
            const float t = wave_sum(mine ? v[0] : 0.f);
            if (lane == 0) acc[r0 * 8] += t;
        } else {
            float w[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) w[c] = mine ? v[c] : 0.f;
            const float t = wave_sum8(w, lane);           // lane 8*c holds component c
            if ((lane & 7) == 0) acc[r0 * 8 + (lane >> 3)] += t;
        }
        pending &= ~__ballot(mine);
    }
    wave_lds_fence();
}
// Work distribution: the rows of a system are cut into chunks of DR_CHUNK; the workgroups of the system take
// contiguous shares and, inside a workgroup, the waves pull chunks from an LDS counter (rows differ a lot in cost --
// in the symmetric "partner index above the row" pass the first rows have all the work -- so a static split leaves
// waves idle).  The result does not depend on which wave runs a chunk.
__device__ __forceinline__ void workgroup_row_range(int n_rows, int chunk, int& g0, int& g1) {
    const int n_chunk = (n_rows + chunk - 1) / chunk;
    const int per_wg = (n_chunk + gridDim.x - 1) / gridDim.x;
    g0 = blockIdx.x * per_wg * chunk; g1 = g0 + per_wg * chunk;
    if (g1 > n_rows) g1 = n_rows;
    if (g0 > g1) g0 = g1;
}
__device__ __forceinline__ int next_chunk(int* counter, int lane) {
    int c = 0;
    if (lane == 0) c = atomicAdd(counter, 1);
    return __builtin_amdgcn_readfirstlane(c);
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32, 1 ulp

}  // namespace up
