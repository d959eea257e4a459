// Packed-fp32 pair passes (round 3): every lane evaluates TWO partners of its row at once, as the two halves of 64-bit
// register pairs, so that the geometry, the spline polynomials, the derivative assembly and the sensitivity products of
// /root/reference/src/bead_interaction.h:30-84 issue as v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 -- one wave64 VALU
// instruction per TWO pair evaluations.  The scalar formulation of igraph_device.h sits at the issue ceiling of a dependent
// fp32 chain (one instruction per 4 cycles and SIMD), which is half of what the vector unit can do.
//
// What stays per pair (not packable on gfx950): v_rsq_f32, the float -> interval-index conversions, the LDS reads of the
// partner row and of the four cubic pieces, the gather of the pair sensitivity, the fixed-point conversion and the LDS atomics.
//
// Work decomposition: a wavefront serves a batch of 16 rows of the sorted row order; each 4-lane group owns one row and takes
// 8 consecutive words of its hit list per trip (lane l: words l and l+4, so that each half-evaluation covers 4 consecutive
// partners: contiguous pair-matrix stores, distinct accumulator banks), i.e. the same 8 pairs per row and trip as the
// 8-lane groups of the scalar passes, so the tail waste per row is unchanged.  A lane past the end of its row evaluates the
// SENTINEL element (a far-away, finite dummy row staged behind the real ones) with sensitivity zero: every body is
// branch-free and the discarded halves contribute an exact +0.
//
// Gradient accumulators: 64-bit integer LDS atomics as before (order-independent, bit-reproducible), but a contribution is
// converted as floor(v * 2^22 + 0.5) -- a packed multiply shared by two values, v_cvt_rpi_i32_f32 and a sign extension instead of the 8
// instructions of the 2^32 split (igraph_device.h: to_fixed32): resolution
// 2.4e-7, below the rounding of the fp32 sums it replaces; |v| >= 512 saturates (v_cvt_i32_f32), sums cannot overflow.
#pragma once
#include "igraph_device.h"

namespace up {

typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 mk2(float a, float b) { v2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2 bc2(float a) { v2 r; r.x = a; r.y = a; return r; }
__device__ __forceinline__ v2 fma2(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }

#define P2_FIX_BITS 22
#define P2_FIX_SCALE ((float)(1 << P2_FIX_BITS))
// vs = value * 2^22 (the callers scale two values per v_pk_mul_f32): v_cvt_rpi_i32_f32 = floor(vs + 0.5), saturating; v_ashrrev 31
__device__ __forceinline__ unsigned long long to_fixed22_scaled(float vs) {
    int q;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(vs));
    return (unsigned long long)(long long)q;
}
__device__ __forceinline__ unsigned long long to_fixed22(float v) { return to_fixed22_scaled(v * P2_FIX_SCALE); }
__device__ __forceinline__ void lds_add_fixed22_scaled(unsigned long long* p, float vs) {
    __hip_atomic_fetch_add(p, to_fixed22_scaled(vs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// a ROW's own total (one conversion per row and component, not per pair): the full 64-bit range instead of the saturating
// 32-bit conversion above -- a badly clashing start structure can push a row total past 512
__device__ __forceinline__ unsigned long long to_fixed22_wide(float v) { return (unsigned long long)__float2ll_rn(v * P2_FIX_SCALE); }
__device__ __forceinline__ void lds_add_fixed22_wide(unsigned long long* p, float v) {
    __hip_atomic_fetch_add(p, to_fixed22_wide(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ float from_fixed22(unsigned long long a) { return (float)((double)(long long)a * (1.0 / (double)(1 << P2_FIX_BITS))); }
__device__ __forceinline__ void lds_add_fixed22(unsigned long long* p, float v) {
    __hip_atomic_fetch_add(p, to_fixed22(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Staged elements of the packed passes: TWO 16-byte planes per side -- [0, 4 n') the words 0..3 of every row (position, first
// direction component), [4 n', 8 n') the words 4..7 (rest of the direction, the two metadata words), n' = n + 1 rows: the last
// one is the SENTINEL, a far-away dummy element that is finite everywhere in the functors and beyond every cutoff.
// A ds_read_b128 is served in lane groups of 16, one 16-byte slot of the 256-byte bank row per lane: with 32-byte rows the
// first halves of all rows share the 8 even slots (>= 2-way conflicts for any 16 partners), with 16-byte planes row j sits in
// slot j mod 16 and the consecutive partners a row group reads are conflict free.
// meta1_mul: the staged word is meta1[i] * meta1_mul (the coverage passes stage an element's share of its table row's OFFSET instead of its
// type: the row of a pair is then one addition, not a multiply-add and a multiply per pair)
__device__ __forceinline__ void stage_rows_planes(float* lds, const upk_coord_t& node, int s, const int* __restrict__ loc, int n, int dim,
                                                  const int* __restrict__ meta1, const int* __restrict__ meta0,
                                                  const float* __restrict__ sens, int sens_stride, float sentinel6, float sentinel7, int meta1_mul = 1) {
    const float* base = node.out + (size_t)s * node.n_elem * node.stride;
    const int np = n + 1;
    // one lane per ELEMENT: its row index and metadata in one round of loads, its row as one or two 16-byte loads, two 16-byte LDS stores.
    // (The first form dealt the 8 words of an element to 8 lanes: nine trips per lane for 1200 elements, each a chain of two dependent
    //  global loads in front of the pair loop.)  Rows of a coordinate node are padded to multiples of 4 floats.
    if ((node.stride & 3) == 0) {
        for (int i = threadIdx.x; i < np; i += blockDim.x) {
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (i == n) { v[0] = v[1] = v[2] = 1.0e4f; v[3] = 1.f; v[6] = sentinel6; v[7] = sentinel7; }
            else {
                const int l = loc[i];
                const int m1 = meta1 ? meta1[i] * meta1_mul : 0, m0 = meta0 ? meta0[i] : 0;
                const float sv = sens ? sens[(size_t)i * sens_stride] : 0.f;
                const float4* row = (const float4*)(base + (size_t)l * node.stride);
                const float4 r0 = row[0];
                float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dim > 4) r1 = row[1];
                const float r[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = c < dim ? r[c] : 0.f;
                if (dim <= 7 && meta1) v[7] = __int_as_float(m1);
                if (dim <= 6) { if (sens) v[6] = sv; else if (meta0) v[6] = __int_as_float(m0); }
            }
            ((float4*)lds)[i] = make_float4(v[0], v[1], v[2], v[3]);
            ((float4*)lds)[np + i] = make_float4(v[4], v[5], v[6], v[7]);
        }
        return;
    }
    for (int t = threadIdx.x; t < np * 8; t += blockDim.x) {
        const int i = t >> 3, c = t & 7;
        float v = 0.f;
        if (i == n) v = c < 3 ? 1.0e4f : (c == 3 ? 1.f : (c == 6 ? sentinel6 : (c == 7 ? sentinel7 : 0.f)));
        else if (c < dim) v = base[(size_t)loc[i] * node.stride + c];
        else if (c == 7 && meta1) v = __int_as_float(meta1[i] * meta1_mul);
        else if (c == 6 && sens) v = sens[(size_t)i * sens_stride];
        else if (c == 6 && meta0) v = __int_as_float(meta0[i]);
        lds[(c >> 2) * np * 4 + i * 4 + (c & 3)] = v;
    }
}
__device__ __forceinline__ void load_row8_planes(float* x, const float* planes, int np, int j) {
    const float4 lo = *(const float4*)(planes + j * 4), hi = *(const float4*)(planes + (np + j) * 4);
    x[0] = lo.x; x[1] = lo.y; x[2] = lo.z; x[3] = lo.w; x[4] = hi.x; x[5] = hi.y; x[6] = hi.z; x[7] = hi.w;
}

// one cubic piece for two pairs: value and slope of  c0 + c1 y + c2 y^2 + c3 y^3  with 5 packed FMAs
// (s = c1 + t y, t = c2 + c3 y are the Horner intermediates; slope = s + y (t + c3 y))
__device__ __forceinline__ void cubic2(const float* cA, const float* cB, v2 y, v2& v, v2& d) {
    const float4 a = *(const float4*)cA, b = *(const float4*)cB;
    const v2 c0 = mk2(a.x, b.x), c1 = mk2(a.y, b.y), c2 = mk2(a.z, b.z), c3 = mk2(a.w, b.w);
    const v2 t = fma2(c3, y, c2), s = fma2(t, y, c1);
    v = fma2(s, y, c0);
    d = fma2(fma2(c3, y, t), y, s);
}
__device__ __forceinline__ v2 cubic2_value(const float* cA, const float* cB, v2 y) {
    const float4 a = *(const float4*)cA, b = *(const float4*)cB;
    return fma2(fma2(fma2(mk2(a.w, b.w), y, mk2(a.z, b.z)), y, mk2(a.y, b.y)), y, mk2(a.x, b.x));
}
// value of one cubic piece for ONE pair, plain Horner on the registers the 16-byte read filled: 3 FMAs and no packing move.  (A packed
// evaluation of two pairs is 3 v_pk_fma_f32 plus the 6-7 moves that interleave the two pieces' coefficients into register pairs, and a
// wave64 VALU instruction holds its SIMD for one quad-cycle whether it is packed or a move: the value-only passes evaluate per pair;
// the passes with derivatives keep cubic2, whose five packed FMAs per two pairs outweigh the moves.)
// (The FMAs are written as instructions: left as fmaf() calls, the SLP vectoriser pairs the Horner steps of the two pairs a lane evaluates
//  back into v_pk_fma_f32 behind eight moves -- 246 instead of 230 vector instructions per trip of the coverage forward pass.)
__device__ __forceinline__ float fma_one(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float cubic1_value(const float* c4, float y) {
    const float4 c = *(const float4*)c4;
    return fma_one(fma_one(fma_one(c.w, y, c.z), y, c.y), y, c.x);
}
// knot interval i and offset y in [0, 1) of a spline coordinate x clamped to [0, hi]: v_med3_f32, v_cvt_i32_f32 (truncation = floor for
// x >= 0), v_fract_f32 -- three instructions where (int)x, max, min, (float)i and x - i are four and a half.  hi is the last
// representable coordinate inside the last interval (angular splines: |cos| may pass 1 by an ulp, which the integer clamp answered by
// extrapolating the end piece by that ulp; here the coordinate stops an ulp short of the last knot -- the cubic is continuous, the
// difference is O(1e-7) of the slope) or anywhere inside the constant end piece (radial splines).
__device__ __forceinline__ float spline_interval(float x, float hi, int& i) {
    const float xc = __builtin_amdgcn_fmed3f(x, 0.f, hi);
    i = (int)xc;
    return __builtin_amdgcn_fractf(xc);
}
// the four basis polynomials of a uniform cubic B-spline and their slopes, for two pairs (igraph_device.h: bspline_basis)
__device__ __forceinline__ void bspline_basis2(v2 y, v2 b[4], v2 d[4]) {
    const v2 y2 = y * y, omy = bc2(1.f) - y, omy2 = omy * omy;
    const float s = 1.f / 6.f;
    b[0] = omy2 * (omy * bc2(s));
    b[1] = fma2(fma2(bc2(0.5f), y, bc2(-1.f)), y2, bc2(2.f / 3.f));
    b[2] = fma2(fma2(fma2(bc2(-0.5f), y, bc2(0.5f)), y, bc2(0.5f)), y, bc2(s));
    b[3] = y2 * (y * bc2(s));
    d[0] = bc2(-0.5f) * omy2;
    d[1] = y * fma2(bc2(1.5f), y, bc2(-2.f));
    d[2] = fma2(fma2(bc2(-1.5f), y, bc2(1.f)), y, bc2(0.5f));
    d[3] = bc2(0.5f) * y2;
}
// value and slope from the 4-coefficient windows starting at cA[0] / cB[0]
template <bool WANT_D>
__device__ __forceinline__ void window2(const float* cA, const float* cB, const v2 b[4], const v2 d[4], v2& val, v2& der) {
    const v2 c0 = mk2(cA[0], cB[0]), c1 = mk2(cA[1], cB[1]), c2 = mk2(cA[2], cB[2]), c3 = mk2(cA[3], cB[3]);
    val = fma2(c3, b[3], fma2(c2, b[2], fma2(c1, b[1], c0 * b[0])));
    if (WANT_D) der = fma2(c3, d[3], fma2(c2, d[2], fma2(c1, d[1], c0 * d[0])));
}

// bead_interaction.h:30-84 for two pairs.  x1 / x2: components 0..5 of the two elements as packed values (the row element is
// a broadcast, the partners are packed by the caller); pA / pB: parameter rows of the two type pairs; off1 / off2: where the
// angular pieces of x1 / x2 start in each row (quadspline_pair).  WANT_D: derivatives in the compact form
//   d(value)/d(x1) = (-dd, g1),  d(value)/d(x2) = (dd, g2).
// POLY: rows of the per-interval polynomial table, else spline coefficients [ang1: ka][ang2: ka][wide: k][narrow: k].
template <bool WANT_D, bool POLY>
__device__ __forceinline__ v2 quadspline_pair2(const QuadShape& Q, const float* pA, const float* pB, const v2* x1, const v2* x2,
                                               v2* dd, v2* g1, v2* g2, int off1A, int off2A, int off1B, int off2B) {
    const v2 dx = x2[0] - x1[0], dy = x2[1] - x1[1], dz = x2[2] - x1[2];
    const v2 dist2 = fma2(dz, dz, fma2(dy, dy, dx * dx));
    const v2 inv_dist = mk2(__builtin_amdgcn_rsqf(dist2.x), __builtin_amdgcn_rsqf(dist2.y));   // v_rsq_f32, 1 ulp (no denormal rescaling: dist2 is O(1..100))
    const v2 dist_coord = dist2 * (inv_dist * bc2(Q.inv_dx));
    const v2 ux = inv_dist * dx, uy = inv_dist * dy, uz = inv_dist * dz;
    const v2 cos1 = fma2(x1[5], uz, fma2(x1[4], uy, x1[3] * ux));
    const v2 ncos2 = fma2(x2[5], uz, fma2(x2[4], uy, x2[3] * ux));       // = -cos2
    v2 a1, da1, a2, da2, wide, dwide, narrow, dnarrow;
    if constexpr (POLY) {
        const v2 xa = fma2(cos1, bc2(Q.inv_dtheta), bc2(Q.inv_dtheta));    // (cos1 + 1) inv_dtheta
        const v2 xb = fma2(-ncos2, bc2(Q.inv_dtheta), bc2(Q.inv_dtheta));
        // last coordinate inside the last angular interval (ka - 3 intervals), any coordinate inside the constant radial end piece k - 2
        const float hi_a = __uint_as_float(__float_as_uint((float)(Q.ka - 3)) - 1u), hi_r = (float)(Q.k - 2) + 0.5f;
        int iaA, iaB, ibA, ibB, irA, irB;
        const float yaA = spline_interval(xa.x, hi_a, iaA), yaB = spline_interval(xa.y, hi_a, iaB);
        const float ybA = spline_interval(xb.x, hi_a, ibA), ybB = spline_interval(xb.y, hi_a, ibB);
        const float yrA = spline_interval(dist_coord.x, hi_r, irA), yrB = spline_interval(dist_coord.y, hi_r, irB);
        const v2 ya = mk2(yaA, yaB), yb = mk2(ybA, ybB), yr = mk2(yrA, yrB);
        const float* rA = pA + 8 * (Q.ka - 3) + 8 * irA; const float* rB = pB + 8 * (Q.ka - 3) + 8 * irB;
        if (WANT_D) {
            cubic2(pA + off1A + 4 * iaA, pB + off1B + 4 * iaB, ya, a1, da1);
            cubic2(pA + off2A + 4 * ibA, pB + off2B + 4 * ibB, yb, a2, da2);
            cubic2(rA, rB, yr, wide, dwide);
            cubic2(rA + 4, rB + 4, yr, narrow, dnarrow);
        } else {
            // value only: each pair on its own registers (cubic1_value), the two results land in a register pair for free
            a1 = mk2(cubic1_value(pA + off1A + 4 * iaA, ya.x), cubic1_value(pB + off1B + 4 * iaB, ya.y));
            a2 = mk2(cubic1_value(pA + off2A + 4 * ibA, yb.x), cubic1_value(pB + off2B + 4 * ibB, yb.y));
            wide = mk2(cubic1_value(rA, yr.x), cubic1_value(rB, yr.y));
            narrow = mk2(cubic1_value(rA + 4, yr.x), cubic1_value(rB + 4, yr.y));
        }
    } else {
        v2 b[4], db[4];
        {
            const v2 x = fma2(cos1, bc2(Q.inv_dtheta), bc2(Q.inv_dtheta + 1.f));
            const int binA = (int)x.x, binB = (int)x.y;
            bspline_basis2(x - mk2((float)binA, (float)binB), b, db);
            window2<WANT_D>(pA + off1A + binA - 1, pB + off1B + binB - 1, b, db, a1, da1);
        }
        {
            const v2 x = fma2(-ncos2, bc2(Q.inv_dtheta), bc2(Q.inv_dtheta + 1.f));
            const int binA = (int)x.x, binB = (int)x.y;
            bspline_basis2(x - mk2((float)binA, (float)binB), b, db);
            window2<WANT_D>(pA + off2A + binA - 1, pB + off2B + binB - 1, b, db, a2, da2);
        }
        {   // radial splines share one coordinate.  Clamped ends (spline.h:275-310): the window at the first / last interval
            // evaluated AT the knot gives exactly (c0 + 4 c1 + c2) / 6, and the slope is masked to zero there
            const float kmax = (float)(Q.k - 2);
            const v2 xc = mk2(fminf(fmaxf(dist_coord.x, 1.f), kmax), fminf(fmaxf(dist_coord.y, 1.f), kmax));
            const int binA = min((int)xc.x, Q.k - 3), binB = min((int)xc.y, Q.k - 3);
            bspline_basis2(xc - mk2((float)binA, (float)binB), b, db);
            const float* wA = pA + 2 * Q.ka + binA - 1; const float* wB = pB + 2 * Q.ka + binB - 1;
            window2<WANT_D>(wA, wB, b, db, wide, dwide);
            window2<WANT_D>(wA + Q.k, wB + Q.k, b, db, narrow, dnarrow);
            if (WANT_D) {
                const v2 inside = mk2((dist_coord.x >= 1.f && dist_coord.x < kmax) ? 1.f : 0.f, (dist_coord.y >= 1.f && dist_coord.y < kmax) ? 1.f : 0.f);
                dwide *= inside; dnarrow *= inside;
            }
        }
    }
    const v2 angular_weight = a1 * a2;
    if (WANT_D) {
        const v2 radial_deriv = bc2(Q.inv_dx) * fma2(angular_weight, dnarrow, dwide);
        const v2 ad1 = (bc2(Q.inv_dtheta) * da1) * (a2 * narrow);
        const v2 ad2 = (bc2(Q.inv_dtheta) * a1) * (da2 * narrow);
        // rXX = ad1 rvec1 - ad2 rvec2; deriv_dir = inv_dist (rXX - (u . rXX) u); dd = radial_deriv u + deriv_dir
        const v2 rx = fma2(ad1, x1[3], -(ad2 * x2[3])), ry = fma2(ad1, x1[4], -(ad2 * x2[4])), rz = fma2(ad1, x1[5], -(ad2 * x2[5]));
        const v2 ur = fma2(uz, rz, fma2(uy, ry, ux * rx));
        const v2 k = fma2(-ur, inv_dist, radial_deriv);                    // coefficient of u:  radial_deriv - inv_dist (u . rXX)
        dd[0] = fma2(k, ux, inv_dist * rx); dd[1] = fma2(k, uy, inv_dist * ry); dd[2] = fma2(k, uz, inv_dist * rz);
        g1[0] = ad1 * ux; g1[1] = ad1 * uy; g1[2] = ad1 * uz;
        g2[0] = -(ad2 * ux); g2[1] = -(ad2 * uy); g2[2] = -(ad2 * uz);
    }
    return fma2(angular_weight, narrow, wide);
}

// ---- 4-lane groups, two hit-list words per lane and trip ---------------------------------------------------------------
#ifndef P2_LANES
#define P2_LANES 4      // (measured at 4096 systems: 8 lanes within 0.5 %, 16 lanes -4.5 %: the LDS bank conflicts of the partner gathers --
                        //  two thirds of the LDS-active cycles -- are not what bounds the passes)
#endif
#define P2_ROWS (UP_WAVE / P2_LANES)
#ifndef P2_CHUNK
#define P2_CHUNK 2      // trips whose list words are loaded together, one chunk ahead (a trip = 8 words of a row)
#endif
__device__ __forceinline__ float group_sum4(float v) { return group_sum_n<P2_LANES>(v); }     // (sum over the P2_LANES lanes of a row group)
// As group_batch_loop (igraph_device.h), for batches of 16 rows.  Op provides
//   begin(row)                       -- load the row element, reset the row accumulators
//   body(row, wA, wB, liveA, liveB)  -- two hit-list words per lane; a dead half carries `dead_word` (the sentinel element)
//   flush(row)                       -- reduce over the group and write the row's results (also for rows without hits)
template <typename Op, typename W>
__device__ __forceinline__ void group2_batch_loop(Op& op, int n_rows, const unsigned short* ord, const int* range,
                                                  const W* __restrict__ hit, int cap, int* counter, int batch_first, int batch_step, int dead_word) {
    const int lane = threadIdx.x & 63, gl = lane & (P2_LANES - 1), g = lane / P2_LANES;
    const int n_batch = (n_rows + P2_ROWS - 1) / P2_ROWS;
    auto claim = [&]() { int v = 0; if (lane == 0) v = atomicAdd(counter, 1); return __builtin_amdgcn_readfirstlane(v); };
    struct Batch { int row, n_mine; const W* hrow; int wa[P2_CHUNK], wb[P2_CHUNK]; bool valid; };
    auto load2 = [&](const Batch& B, int t, int& a, int& b) {        // this lane's two words of trip t: positions l and l + 4 of the trip's 8
        const int k = t * 2 * P2_LANES;
        a = k < B.n_mine ? B.hrow[k] : dead_word;
        b = k + P2_LANES < B.n_mine ? B.hrow[k + P2_LANES] : dead_word;
    };
    auto fetch = [&](int i, Batch& B) -> bool {
        const int b = batch_first + i * batch_step;
        if (b >= n_batch) return false;
        const int ri = b * P2_ROWS + g;
        B.valid = ri < n_rows;
        B.row = B.valid ? (int)ord[ri] : 0;
        const int rg = B.valid ? range[B.row] : 0;
        const int first = rg & 0xffff, end = (int)((unsigned)rg >> 16);
        B.hrow = hit + (size_t)B.row * cap + first + gl;
        B.n_mine = end - first - gl;                             // word k of this lane (k = 8 t, 8 t + 4) is live iff k < n_mine
#pragma unroll
        for (int u = 0; u < P2_CHUNK; ++u) load2(B, u, B.wa[u], B.wb[u]);
        return true;
    };
    Batch cur, nxt;
    bool have = fetch(claim(), cur);
    while (have) {
        const bool have_next = fetch(claim(), nxt);
        // (wave-wide maximum: the row order may be a few steps old, igraph_device.h)
        const int n_trip = __builtin_amdgcn_readfirstlane((int)wave_max((float)((cur.n_mine > 0 ? cur.n_mine : 0) + 2 * P2_LANES - 1)) / (2 * P2_LANES));
        op.begin(cur.row);
        for (int t0 = 0; t0 < n_trip; t0 += P2_CHUNK) {
            int na[P2_CHUNK], nb[P2_CHUNK];
#pragma unroll
            for (int u = 0; u < P2_CHUNK; ++u) load2(cur, t0 + P2_CHUNK + u, na[u], nb[u]);
#pragma unroll
            for (int u = 0; u < P2_CHUNK; ++u) {
                if (t0 + u >= n_trip) break;        // (wave-uniform)
                const int k = (t0 + u) * 2 * P2_LANES;
                op.body(cur.row, cur.wa[u], cur.wb[u], k < cur.n_mine, k + P2_LANES < cur.n_mine);
            }
#pragma unroll
            for (int u = 0; u < P2_CHUNK; ++u) { cur.wa[u] = na[u]; cur.wb[u] = nb[u]; }
        }
        if (cur.valid) op.flush(cur.row);
        cur = nxt; have = have_next;
    }
}

// launch geometry as pair_geometry, for 16-row batches
static inline void pair2_geometry(int n_system, int n_rows, int& wgs_per_system, int& threads) {
    static int target = 0;
    if (!target) { const char* e = getenv("UPSIDE_HIP_IG_WGS"); target = e ? atoi(e) : 256; if (target < 1) target = 256; }
    int bps = (target + n_system - 1) / n_system;
    const int max_bps = (n_rows + 255) / 256;                    // one 16-row batch per wavefront of a 1024-lane workgroup = 256 rows
    if (bps > max_bps) bps = max_bps;
    if (bps < 1) bps = 1;
    const int rows_per_wg = (n_rows + bps - 1) / bps;
    int t = ((rows_per_wg * P2_LANES + 63) / 64) * 64;
    threads = t < 256 ? 256 : (t > 1024 ? 1024 : t);
    wgs_per_system = bps;
}

}  // namespace up
