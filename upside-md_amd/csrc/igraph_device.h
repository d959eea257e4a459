// Device-side building blocks shared by the interaction-graph kernels (kernels_igraph.hip, kernels_pair.hip,
// kernels_rotamer.hip).
//
// Work decomposition on gfx950 (round 2):
//   * list upkeep (side streams): k_pairlist_check -> k_pairlist_build (flagged systems) -> k_pairlist_refine, which
//     applies this step's `d2 < cutoff2` test to the cached lists ONCE per graph side and step and leaves the survivors as
//     per-row "hit lists" in global memory (4 bytes per in-range pair, list order) -> k_pairlist_order, which sorts the
//     rows of a system by hit count.  Every pair pass of the step then reads hit lists only: no distance test, no ballot
//     compaction, no queue in the compute kernels;
//   * pair passes: one workgroup serves one system; it stages the spline/parameter table, the packed elements of both
//     sides (8 floats per element) and the sorted row order in LDS.  A wavefront claims a BATCH of 8 consecutive rows of
//     the sorted order; each of its eight 8-lane groups owns one of them, strides over that row's hit list, and keeps the
//     row sums in registers (3 DPP steps per component at the end of the batch instead of a segmented 64-lane reduction per
//     64 pairs).  Rows of a batch have (nearly) the same length, so the groups run in lock step and the batch bookkeeping is
//     uniform code, amortised over the batch's trips;
//   * forward passes are gathers with a fixed summation order; backward passes visit a pair ONCE and hand the partner
//     element's share to 64-bit fixed-point integer LDS atomics (to_fixed32 below), whose sums do not depend on the order of
//     arrival -- no floating-point atomic decides a result, so trajectories are reproducible bit for bit.  (LDS float
//     atomics run at 3 cycles per LANE on gfx950 -- tools/ubench/lds_atomics.hip: 192 cycles per wave instruction against
//     <= 20 for integer ones -- which is what sank the float version of the one-visit pass that was tried first.)
#pragma once
#include "device_math.h"
#include "../../include/upside_hip_kernels.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdlib>

// block coordinates of a kernel body: those of its own launch, or its range of a merged launch (kernels_batch.h)
struct BX { int bx, gx, by, gy; };
#define BX_REAL (BX{(int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)gridDim.y})

namespace up {

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
__device__ __forceinline__ QuadShape quad_shape(const upk_igraph_t& G) { QuadShape Q; Q.ka = G.n_knot_angular; Q.k = G.n_knot; Q.inv_dx = G.inv_dx; Q.inv_dtheta = G.inv_dtheta; return Q; }

// bead_interaction.h:30-84 with the B-splines in basis form and hardware rsqrt (1 ulp).
// WANT_D: 0 value only; 3: both elements' derivatives in compact form --
//   d(value)/d(x1) = (-dd, g1),  d(value)/d(x2) = (dd, g2)      (positions, then direction vectors)
// off1 / off2: where the angular coefficients of x1 / x2 start in the parameter row (0 / ka as stored; ka / 0 when a row of
// the type pair (t2, t1) is used for (t1, t2), see stage_table_sym)
//
// POLY: p is a row of the per-interval polynomial table (quadspline_poly_row, packed on the host from the same spline
// coefficients): every knot interval of every spline holds its cubic as 4 monomial coefficients of the offset y from the
// interval's left knot, 16-byte aligned -- one ds_read_b128 and 3 + 3 FMAs per spline for value and slope instead of four
// scattered coefficient reads, the 18-operation basis evaluation and 8 multiply-adds.  Row layout:
//   [angular of x1: (ka-3) x 4] [angular of x2: (ka-3) x 4] [radial: (k-1) x (wide 4, narrow 4)]
// The radial part carries one extra constant interval at each end (the clamped values of spline.h:275-310), so clamping is
// an index clamp: no selects, no second set of reads.  off1 / off2 are then offsets in this layout (0 / 4 (ka-3) as stored).
template <int WANT_D, bool POLY = false, typename P>
__device__ __forceinline__ float quadspline_pair(const QuadShape& Q, P p, const float* x1, const float* x2, float* dd, float* g1, float* g2,
                                                 int off1 = 0, int off2 = -1) {
    if (off2 < 0) off2 = POLY ? 4 * (Q.ka - 3) : Q.ka;
    const f3 displace = mk3(x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]);
    const f3 rvec1 = mk3(x1[3], x1[4], x1[5]), rvec2 = mk3(x2[3], x2[4], x2[5]);
    const float dist2 = mag2(displace), inv_dist = rsqrtf(dist2);
    const float dist_coord = dist2 * (inv_dist * Q.inv_dx);
    const f3 u = inv_dist * displace;
    const float cos1 = dot(rvec1, u), cos2 = -dot(rvec2, u);
    float a1, da1, a2, da2, wide, dwide, narrow, dnarrow;
    if constexpr (POLY) {
        auto cubic = [](const float* c4, float y, float& v, float& d) {
            const float4 c = *(const float4*)c4;
            v = fmaf(fmaf(fmaf(c.w, y, c.z), y, c.y), y, c.x);
            d = fmaf(fmaf(3.f * c.w, y, c.z + c.z), y, c.y);
        };
        {   // angular splines (unclamped, spline.h:228-242): |cos| can pass 1 by an ulp, the end intervals extrapolate
            const float x = (cos1 + 1.f) * Q.inv_dtheta; const int i = min(max((int)x, 0), Q.ka - 4);
            cubic(p + off1 + 4 * i, x - (float)i, a1, da1);
        }
        {
            const float x = (cos2 + 1.f) * Q.inv_dtheta; const int i = min(max((int)x, 0), Q.ka - 4);
            cubic(p + off2 + 4 * i, x - (float)i, a2, da2);
        }
        {   // radial splines: interval 0 (dist_coord < 1) and interval k-2 (beyond the last knot) are the clamped constants
            const int i = min((int)dist_coord, Q.k - 2);
            const float y = dist_coord - (float)i;
            const float* r = p + 8 * (Q.ka - 3) + 8 * i;
            cubic(r, y, wide, dwide); cubic(r + 4, y, narrow, dnarrow);
        }
    } else {
    float b[4], db[4];
    // angular splines (unclamped, spline.h:228-242)
    {
        const float x = (cos1 + 1.f) * Q.inv_dtheta + 1.f; const int bin = (int)x;
        bspline_basis(x - (float)bin, b, db); bspline_vd(a1, da1, p + off1, bin, b, db);
    }
    {
        const float x = (cos2 + 1.f) * Q.inv_dtheta + 1.f; const int bin = (int)x;
        bspline_basis(x - (float)bin, b, db); bspline_vd(a2, da2, p + off2, bin, b, db);
    }
    // radial splines share one coordinate; clamped ends (spline.h:275-310)
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
    }
    const float angular_weight = a1 * a2;
    if (WANT_D) {
        const float radial_deriv = Q.inv_dx * (dwide + angular_weight * dnarrow);
        const float angular_deriv1 = Q.inv_dtheta * da1 * a2 * narrow;
        const float angular_deriv2 = Q.inv_dtheta * a1 * da2 * narrow;
        const f3 rXX = angular_deriv1 * rvec1 - angular_deriv2 * rvec2;
        const f3 deriv_dir = inv_dist * (rXX - dot(u, rXX) * u);
        const f3 d = radial_deriv * u + deriv_dir;
        dd[0] = d.x; dd[1] = d.y; dd[2] = d.z;
        g1[0] = angular_deriv1 * u.x; g1[1] = angular_deriv1 * u.y; g1[2] = angular_deriv1 * u.z;
        g2[0] = -angular_deriv2 * u.x; g2[1] = -angular_deriv2 * u.y; g2[2] = -angular_deriv2 * u.z;
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

// environment.cpp:27-60.  d1: CB frame (6), d2: weighted side-chain bead (x, y, z, probability)
__device__ __forceinline__ float environment_edge(const float* __restrict__ p, const float* cb, const float* sc, float* d1, float* d2) {
    const f3 displace = mk3(sc[0] - cb[0], sc[1] - cb[1], sc[2] - cb[2]);
    const f3 rvec1 = mk3(cb[3], cb[4], cb[5]);
    const float prob = sc[3];
    const float dist2 = mag2(displace), inv_dist = rsqrt_(dist2), dist = dist2 * inv_dist;
    const f3 u = inv_dist * displace;
    const float dp = dot(u, rvec1);
    float rs, drs, as, das;
    compact_sigmoid(rs, drs, dist - p[0], p[1]);
    compact_sigmoid(as, das, p[2] - dp, p[3]);
    const f3 dd = prob * ((drs * as) * u - (rs * das * inv_dist) * (rvec1 - dp * u));
    const float k = -prob * rs * das;
    d1[3] = k * u.x; d1[4] = k * u.y; d1[5] = k * u.z;
    d1[0] = -dd.x; d1[1] = -dd.y; d1[2] = -dd.z;
    d2[0] = dd.x; d2[1] = dd.y; d2[2] = dd.z;
    const float score = rs * as;
    d2[3] = score;
    return prob * score;
}

// hbond.cpp:128-148, 166-230.  The angular cut-off is applied per pair (the reference applies it per group of
// 4 SIMD edges, which lets pairs outside the cone pick up a value below 1.3e-6).
__device__ __forceinline__ float protein_hbond_edge(const float* __restrict__ p, const float* x1, const float* x2, float* d1, float* d2) {
    const f3 H = mk3(x1[0], x1[1], x1[2]), O = mk3(x2[0], x2[1], x2[2]);
    const f3 rHN = mk3(x1[3], x1[4], x1[5]), rOC = mk3(x2[3], x2[4], x2[5]);
    const f3 HO = H - O;
    const float magHO2 = mag2(HO) + 1e-6f, invHOmag = rsqrt_(magHO2), magHO = magHO2 * invHOmag;
    const f3 rHO = invHOmag * HO;
    const float dotHOC = dot(rHO, rOC), dotOHN = -dot(rHO, rHN);
    f3 dH = mk3(0.f, 0.f, 0.f), drHN = dH, drOC = dH;
    float hb = 0.f;
    if ((0.f < dotHOC) && (0.f < dotOHN)) {
        float os, dos, is, dis, g1, dg1, g2, dg2;
        sigmoid(os, dos, (p[2] - magHO) * p[3]);
        sigmoid(is, dis, (magHO - p[0]) * p[1]);
        const float radial = os * is;
        const float dradial = -p[3] * dos * is + p[1] * dis * os;
        sigmoid(g1, dg1, (dotHOC - p[4]) * p[5]); dg1 *= p[5];
        sigmoid(g2, dg2, (dotOHN - p[4]) * p[5]); dg2 *= p[5];
        hb = radial * g1 * g2;
        const float c0 = dradial * g1 * g2, c1 = radial * dg1 * g2, c2 = -radial * g1 * dg2;
        drOC = c1 * rHO;
        drHN = c2 * rHO;
        dH = c0 * rHO + (c1 * invHOmag) * (rOC - dotHOC * rHO) + (c2 * invHOmag) * (rHN + dotOHN * rHO);
    }
    const float hb_log = (1.f <= hb) ? 100.f : -logf(1.f - hb);
    const float pref = fminf(rcp(1.f - hb), 1e5f);
    d1[0] = dH.x * pref; d1[1] = dH.y * pref; d1[2] = dH.z * pref;
    d1[3] = drHN.x * pref; d1[4] = drHN.y * pref; d1[5] = drHN.z * pref;
    d2[0] = -dH.x * pref; d2[1] = -dH.y * pref; d2[2] = -dH.z * pref;
    d2[3] = drOC.x * pref; d2[4] = drOC.y * pref; d2[5] = drOC.z * pref;
    return hb_log;
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
    // one lane per ELEMENT (row index and metadata in one round of loads, the row as one or two 16-byte loads, two 16-byte LDS stores)
    // instead of one lane per word, which made nine trips of two dependent global loads per lane for 1200 elements
    if ((node.stride & 3) == 0) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int l = loc[i];
            const int m1 = meta1 ? meta1[i] : 0, m0 = meta0 ? meta0[i] : 0;
            const float sv = sens ? sens[(size_t)i * sens_stride] : 0.f;
            const float4* row = (const float4*)(base + (size_t)l * node.stride);
            const float4 r0 = row[0];
            float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dim > 4) r1 = row[1];
            const float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = c < dim ? r[c] : 0.f;
            if (dim <= 7 && meta1) v[7] = __int_as_float(m1);
            if (dim <= 6) { if (sens) v[6] = sv; else if (meta0) v[6] = __int_as_float(m0); }
            ((float4*)lds)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
            ((float4*)lds)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
        }
        return;
    }
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
__device__ __forceinline__ void stage_table(float* lds, const float* __restrict__ tab, int n) {
    const int n4 = (((size_t)tab & 15) == 0) ? n >> 2 : 0;      // (hipMalloc'ed tables are aligned; the LDS side always is)
    for (int t = threadIdx.x; t < n4; t += blockDim.x) ((float4*)lds)[t] = ((const float4*)tab)[t];
    for (int t = 4 * n4 + threadIdx.x; t < n; t += blockDim.x) lds[t] = tab[t];
}
// A symmetric pair table ([nt][nt][n_param] with row (t2, t1) = row (t1, t2) with the two angular blocks exchanged --
// is_compatible, bead_interaction.h:209-218, checked when the node is built) is kept as its upper triangle only (the host
// packs it, upk_rotamer_t::param_tri): half the LDS
__device__ __forceinline__ int tri_row(int lo, int hi, int nt) { return lo * nt - ((lo * (lo - 1)) >> 1) + (hi - lo); }   // lo <= hi
// fixed-point image of a float (|v| < 2^31): v * 2^32 as a 64-bit two's-complement integer, exact down to 2^-31.  Integer adds commute, so
// sums accumulated through LDS atomics in any order are the EXACT sum of the contributions (and bit-reproducible); LDS integer
// atomics run at full rate on gfx950, float ones at 3 cycles per lane (tools/ubench/lds_atomics.hip)
__device__ __forceinline__ unsigned long long to_fixed32(float v) {
    const float h = rintf(v);                                     // nearest integer; v - h is exact (|v| < 0.5: h = 0; else h within a factor 2 of v)
    const int hi = (int)h, half = (int)((v - h) * 2147483648.f);  // |v - h| <= 0.5: the product is exact, |half| <= 2^30 (resolution 2^-31)
    return ((unsigned long long)(unsigned)(hi + (half >> 31)) << 32) | (unsigned)(half << 1);   // hi * 2^32 + sign-extended 2 * half: 8 instructions
}
__device__ __forceinline__ float from_fixed32(unsigned long long a) { return (float)((double)(long long)a * 2.3283064365386963e-10); }
__device__ __forceinline__ void lds_add_fixed(unsigned long long* p, float v) {
    __hip_atomic_fetch_add(p, to_fixed32(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void load_row8(float* x, const float* p) {   // one 32-byte packed element
    const float4 lo = *(const float4*)p, hi = *(const float4*)(p + 4);
    x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
}

// ---- 8-lane groups ------------------------------------------------------------------------------------------
#ifndef PG_LANES
#define PG_LANES 8
#endif
#define PG_PER_WAVE (UP_WAVE / PG_LANES)
// pair kernels run one 1024-lane workgroup per CU (LDS): 4 wavefronts per SIMD, so a lane may use 128 VGPRs -- tell the
// register allocator, or it keeps to 64 in the hope of an occupancy the LDS footprint rules out
#ifndef PG_WAVES_PER_EU
#define PG_WAVES_PER_EU 4
#endif
#define PG_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(PG_WAVES_PER_EU, PG_WAVES_PER_EU)))
#ifndef PG_CHUNK
#define PG_CHUNK 4      // trips whose list words are loaded together, one chunk ahead
#endif
// sum over the lanes of a group of LANES, left in all of them
template <int LANES> __device__ __forceinline__ float group_sum_n(float v) {
    static_assert(LANES == 2 || LANES == 4 || LANES == 8 || LANES == 16, "group_sum_n: DPP steps for groups of 2, 4, 8 or 16 lanes");
    if (LANES == 16) v += dpp_mov<UP_DPP_ROW_MIRROR>(v);
    if (LANES >= 8) v += dpp_mov<UP_DPP_HALF_MIRROR>(v);
    if (LANES >= 4) v += dpp_mov<UP_DPP_XOR2>(v);
    v += dpp_mov<UP_DPP_XOR1>(v);
    return v;
}
__device__ __forceinline__ float group_sum(float v) {
    static_assert(PG_LANES == 4 || PG_LANES == 8 || PG_LANES == 16, "group_sum: DPP steps for groups of 4, 8 or 16 lanes");
    if (PG_LANES == 16) v += dpp_mov<UP_DPP_ROW_MIRROR>(v);
    if (PG_LANES >= 8) v += dpp_mov<UP_DPP_HALF_MIRROR>(v);
    v += dpp_mov<UP_DPP_XOR2>(v);
    v += dpp_mov<UP_DPP_XOR1>(v);
    return v;
}

// List words.  A cached or hit list holds one word per pair: the partner's element index, 16 bits wide (the refine moves one word
// per cached pair per step and is bound by those bytes), except in the rotamer graph, whose words also carry the residue-pair
// slot (bead | slot << UPK_ROT_J_BITS, 32 bits).  The arrays are allocated as 32-bit words either way; a 16-bit graph uses the
// first half, rows `cap` words apart.  upk_igraph_t::word16 says which (set by the host: itype != UPK_IT_ROTAMER).
template <int IT> struct list_word { typedef unsigned short type; };
template <> struct list_word<UPK_IT_ROTAMER> { typedef int type; };
// host side: the word width the device templates will read (list_word<IT>) against what the host allocated (upk_igraph_t::word16): a mismatch
// would reinterpret the lists silently -- the launchers of the build, refine and pair kernels refuse it (code 9010)
static inline bool list_words_match(const upk_igraph_t* G) { return (G->word16 != 0) == (G->itype != UPK_IT_ROTAMER); }
__device__ __forceinline__ int list_word_at(const int* lists, size_t idx, int word16) {
    return word16 ? (int)((const unsigned short*)lists)[idx] : lists[idx];
}

// per-row hit range staged in LDS: first | end << 16 (list positions; the launchers check that capacities fit 16 bits), and
// the sorted row order
__device__ __forceinline__ void stage_ranges(int* lds_range, unsigned short* lds_ord, const int* __restrict__ hcnt, const int* __restrict__ hlo,
                                             const unsigned short* __restrict__ ord, int n_rows) {
    for (int t = threadIdx.x; t < n_rows; t += blockDim.x) {
        lds_range[t] = (hlo ? hlo[t] : 0) | (hcnt[t] << 16);
        lds_ord[t] = ord[t];
    }
}
#define PG_WALK_LDS_WORDS(n_rows) ((n_rows) + ((n_rows) + 1) / 2)   // 32-bit words of LDS behind stage_ranges

// Batches b = batch_first, batch_first + batch_step, ... of 8 consecutive rows of the sorted order `ord` (LDS), claimed by
// the wavefronts of this workgroup through the LDS counter `counter` (zeroed and barrier-published by the caller).  Op provides
//   begin(row)      -- load the row element, reset the row accumulators
//   body(row, word, live) -- one hit-list word per lane; !live: the lane is past the end of its row (word 0), discard
//   flush(row)      -- reduce over the group and write the row's results (also for rows without hits)
// All control flow is wave-uniform except the predication of lanes past the end of their row.
// (LANES: lanes per row group -- 8 by default; graphs whose rows hold two or three pairs take 2, so that a trip is not 3/4 idle lanes)
// (W: the graph's list word type -- list_word<IT> below; `hit` and `cap` count words of that type)
template <typename Op, int LANES = PG_LANES, typename W = int>
__device__ __forceinline__ void group_batch_loop(Op& op, int n_rows, const unsigned short* ord, const int* range,
                                                 const W* __restrict__ hit, int cap, int* counter, int batch_first, int batch_step) {
    const int lane = threadIdx.x & 63, gl = lane & (LANES - 1), g = lane / LANES;
    const int n_batch = (n_rows + (UP_WAVE / LANES) - 1) / (UP_WAVE / LANES);
    auto claim = [&]() { int v = 0; if (lane == 0) v = atomicAdd(counter, 1); return __builtin_amdgcn_readfirstlane(v); };
    // one batch = (row, list range, first chunk of list words) per group; the NEXT batch's is fetched while the current one
    // is processed, so a batch switch waits neither for the claim nor for the first global loads
    struct Batch { int row, n_mine; const W* hrow; int w[PG_CHUNK]; bool valid; };
    auto fetch = [&](int i, Batch& B) -> bool {                   // returns false past the last batch (wave-uniform)
        const int b = batch_first + i * batch_step;
        if (b >= n_batch) return false;
        const int ri = b * (UP_WAVE / LANES) + g;
        B.valid = ri < n_rows;
        B.row = B.valid ? (int)ord[ri] : 0;
        const int rg = B.valid ? range[B.row] : 0;
        const int first = rg & 0xffff, end = (int)((unsigned)rg >> 16);                         // the row's hits [first, end)

        B.hrow = hit + (size_t)B.row * cap + first + gl;
        B.n_mine = end - first - gl;                              // this lane's words sit at hrow[0], hrow[8], ...: k < n_mine
#pragma unroll
        for (int u = 0; u < PG_CHUNK; ++u) B.w[u] = u * LANES < B.n_mine ? B.hrow[u * LANES] : 0;
        return true;
    };
    Batch cur, nxt;
    bool have = fetch(claim(), cur);
    while (have) {
        const bool have_next = fetch(claim(), nxt);
        // (the longest row of the batch decides the trips: a wave-wide maximum, so that a row order computed a few steps ago --
        //  nodes.cpp: IGraphHost::refine -- only costs balance, never pairs; counts are < 2^24, exact in fp32)
        const int n_trip = __builtin_amdgcn_readfirstlane((int)wave_max((float)((cur.n_mine > 0 ? cur.n_mine : 0) + LANES - 1) ) / LANES);
        op.begin(cur.row);      // (groups past the last row take row 0: their lanes are never live, but evaluate like everyone else's)
        for (int t0 = 0; t0 < n_trip; t0 += PG_CHUNK) {
            int wn[PG_CHUNK];
#pragma unroll
            for (int u = 0; u < PG_CHUNK; ++u) { const int k = (t0 + PG_CHUNK + u) * LANES; wn[u] = k < cur.n_mine ? cur.hrow[k] : 0; }
            // Bodies are branch-free (a lane past the end of its row evaluates list word 0 and discards the result); which of
            // the PG_CHUNK unrolled copies of the functor a pair runs through depends only on its position in its row, never on
            // what else is in the batch, so results do not depend on how rows of equal length were dealt to the wavefronts.
#pragma unroll
            for (int u = 0; u < PG_CHUNK; ++u) {
                if (t0 + u >= n_trip) break;        // (wave-uniform)
                op.body(cur.row, cur.w[u], (t0 + u) * LANES < cur.n_mine);
            }
#pragma unroll
            for (int u = 0; u < PG_CHUNK; ++u) cur.w[u] = wn[u];
        }
        if (cur.valid) op.flush(cur.row);
        cur = nxt; have = have_next;
    }
}

// launch geometry of an LDS-staged pair pass: workgroups per system and lanes per workgroup.  A large batch runs one
// 1024-lane workgroup (16 wavefronts = 128 rows in flight) per system; small batches spread a system over several smaller
// workgroups (each stages the system again, so only as many as it takes to give every CU work).
static inline void pair_geometry(int n_system, int n_rows, int& wgs_per_system, int& threads) {
    static int target = 0;
    if (!target) { const char* e = getenv("UPSIDE_HIP_IG_WGS"); target = e ? atoi(e) : 256; if (target < 1) target = 256; }
    int bps = (target + n_system - 1) / n_system;
    // every workgroup stages the whole system (table, elements, row order): at least one batch of 8 rows per wavefront of a
    // 1024-lane workgroup, i.e. 128 rows each -- a single system is served by ~10 fat workgroups, not by 40 thin ones
    const int max_bps = (n_rows + 127) / 128;
    if (bps > max_bps) bps = max_bps;
    if (bps < 1) bps = 1;
    const int rows_per_wg = (n_rows + bps - 1) / bps;
    int t = ((rows_per_wg * PG_LANES + 63) / 64) * 64;           // one batch per wavefront
    threads = t < 256 ? 256 : (t > 1024 ? 1024 : t);
    wgs_per_system = bps;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32, 1 ulp

}  // namespace up
