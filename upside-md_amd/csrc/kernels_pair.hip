// Pair passes of the asymmetric interaction graphs (hbond_coverage, environment_coverage, protein_hbond) over this
// step's hit lists: replaces the compute_edges / propagate_derivatives loops of /root/reference/src/interaction_graph.h:443-556
// together with the node-side accumulation loops of hbond.cpp:387-397, environment.cpp:85-101 and hbond.cpp:316-365.
//
// See igraph_device.h for the decomposition.  Forward passes are per-row gathers over the row's in-range partners, the
// backward pass visits each pair once (row share in registers, partner share through fixed-point LDS atomics); pair
// gradients are re-evaluated in the backward pass instead of being stored (the reference keeps 12-13 floats of edge_deriv per
// edge, interaction_graph.h:294-296: ~150 flop against 100+ bytes of HBM traffic per edge).
#include "igraph_device.h"
#include "pair2_device.h"
#include <cstring>

using namespace up;
// list words of the graphs served here (every graph but the rotamer's, which has its own passes in kernels_rotamer.hip): igraph_device.h
typedef unsigned short PW;

#define ST(L) ((hipStream_t)(L)->stream)
#ifndef UPK_LAUNCH_STATUS_DEFINED
#define UPK_LAUNCH_STATUS_DEFINED
static inline int launch_status() { return (int)hipGetLastError(); }
#endif
#define C_SENS(c, s) ((c).sens + (size_t)(s) * (c).n_elem * (c).stride)

struct PairArgs {
    float* out; long out_sys_stride; int out_stride, out_comp, out_row0, out_row0_2;        // modes 0, 1
    float* own_grad;                                                                         // mode 1
    int sens_mode; const float* sens1; const float* sens2; long sens_sys_stride; int sens_stride;   // mode 2
    int tab_floats;
    int atomic_sens;      // mode 2, both row sets in one launch, the sides share elements of one node: the row "+=" must be atomic
};

// value of one pair and, if GRAD, its derivative w.r.t. the ROW element in d[0..8); x1 is always the side-1 element
// POLY (hbond_coverage only): `tab` is the per-interval polynomial table (upk_igraph_t::param_poly), else the spline coefficients
template <int IT, int ROW_SIDE, bool GRAD, bool POLY>
__device__ __forceinline__ float pair_functor(const upk_igraph_t& G, const QuadShape& Q, const float* tab, int t1, int t2,
                                              const float* x1, const float* x2, float* d) {
    const float* p = tab + (t1 * G.n_type2 + t2) * (POLY ? G.n_poly : G.n_param);
    if (IT == UPK_IT_HBOND_COVERAGE) {                               // hbond.cpp:261-276
        float dd[3], g1[3], g2[3];
        const float coverage = quadspline_pair<GRAD ? 3 : 0, POLY>(Q, p, x1, x2, dd, g1, g2);
        const float one_m = 1.f - x1[6], prefactor = one_m * one_m;
        if (GRAD) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                d[c] = ROW_SIDE == 1 ? -prefactor * dd[c] : prefactor * dd[c];
                d[3 + c] = prefactor * (ROW_SIDE == 1 ? g1[c] : g2[c]);
            }
            d[6] = ROW_SIDE == 1 ? -coverage * one_m * 2.f : 0.f;
        }
        return prefactor * coverage;
    } else {
        float d1[8], d2[8];
        const float v = IT == UPK_IT_ENVIRONMENT ? environment_edge(p, x1, x2, d1, d2) : protein_hbond_edge(p, x1, x2, d1, d2);
        if (GRAD) {
            constexpr int n = ROW_SIDE == 1 ? 6 : (IT == UPK_IT_ENVIRONMENT ? 4 : 6);
#pragma unroll
            for (int c = 0; c < 8; ++c) d[c] = c < n ? (ROW_SIDE == 1 ? d1[c] : d2[c]) : 0.f;
        }
        return v;
    }
}

// value of one pair with BOTH elements' derivatives: d1[0..7) w.r.t. the side-1 element, d2[0..6) w.r.t. the side-2 element
template <int IT, bool POLY>
__device__ __forceinline__ float pair_functor_both(const upk_igraph_t& G, const QuadShape& Q, const float* tab, int t1, int t2,
                                                   const float* x1, const float* x2, float* d1, float* d2) {
    const float* p = tab + (t1 * G.n_type2 + t2) * (POLY ? G.n_poly : G.n_param);
    if (IT == UPK_IT_HBOND_COVERAGE) {                               // hbond.cpp:261-276
        float dd[3], g1[3], g2[3];
        const float coverage = quadspline_pair<3, POLY>(Q, p, x1, x2, dd, g1, g2);
        const float one_m = 1.f - x1[6], prefactor = one_m * one_m;
#pragma unroll
        for (int c = 0; c < 3; ++c) { d1[c] = -prefactor * dd[c]; d1[3 + c] = prefactor * g1[c]; d2[c] = prefactor * dd[c]; d2[3 + c] = prefactor * g2[c]; }
        d1[6] = -coverage * one_m * 2.f;
        return prefactor * coverage;
    } else if (IT == UPK_IT_ENVIRONMENT) return environment_edge(p, x1, x2, d1, d2);
    else return protein_hbond_edge(p, x1, x2, d1, d2);
}
template <int IT> struct PairDims;
template <> struct PairDims<UPK_IT_HBOND_COVERAGE> { static constexpr int d1 = 7, d2 = 6; };
template <> struct PairDims<UPK_IT_ENVIRONMENT>    { static constexpr int d1 = 6, d2 = 4; };
template <> struct PairDims<UPK_IT_PROTEIN_HBOND>  { static constexpr int d1 = 6, d2 = 6; };

struct PairLds { float *tab, *c1, *c2; int* range; unsigned short* ord; int* counter; };
__device__ __forceinline__ PairLds pair_lds(float* lds, const upk_igraph_t& G, int tab_floats) {
    PairLds L;
    L.tab = lds;
    L.c1 = lds + ((tab_floats + 3) & ~3);
    L.c2 = L.c1 + G.n1 * 8;
    const int n_max = G.n1 > G.n2 ? G.n1 : G.n2;
    L.range = (int*)(L.c2 + G.n2 * 8);
    L.ord = (unsigned short*)(L.range + n_max);
    L.counter = L.range + PG_WALK_LDS_WORDS(n_max);
    return L;
}

// MODE 0: row sums of the value; 1: value and the UNWEIGHTED sum of d(value)/d(row element); 2: sum of sens(pair) * d(value)/d(row element)
template <int IT, int ROW_SIDE, int MODE, bool POLY, int LANES = PG_LANES>
struct RowOp {
    static constexpr int NV = MODE == 0 ? 1 : 8;
    const upk_igraph_t& G; const QuadShape Q; const PairLds& L; const PairArgs& A;
    const int s; const bool row_has, oth_has;
    float xr[8], acc[NV];
    __device__ __forceinline__ RowOp(const upk_igraph_t& G_, const PairLds& L_, const PairArgs& A_, int s_)
        : G(G_), Q(quad_shape(G_)), L(L_), A(A_), s(s_),
          row_has(MODE == 2 && (A_.sens_mode == 3 || A_.sens_mode == ROW_SIDE)), oth_has(MODE == 2 && (A_.sens_mode == 3 || A_.sens_mode == 3 - ROW_SIDE)) {}
    __device__ __forceinline__ void begin(int row) {
        load_row8(xr, (ROW_SIDE == 1 ? L.c1 : L.c2) + row * 8);
#pragma unroll
        for (int c = 0; c < NV; ++c) acc[c] = 0.f;
    }
    __device__ __forceinline__ void body(int, int j, bool live) {
        float xo[8], d[8];
        load_row8(xo, (ROW_SIDE == 1 ? L.c2 : L.c1) + j * 8);
        const int tr = __float_as_int(xr[7]), to = __float_as_int(xo[7]);
        const float v = ROW_SIDE == 1 ? pair_functor<IT, 1, MODE != 0, POLY>(G, Q, L.tab, tr, to, xr, xo, d)
                                      : pair_functor<IT, 2, MODE != 0, POLY>(G, Q, L.tab, to, tr, xo, xr, d);
        if (MODE == 0) acc[0] += live ? v : 0.f;
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 7; ++c) acc[c] += live ? d[c] : 0.f;
            acc[7] += live ? v : 0.f;
        }
        if (MODE == 2) {   // pair sensitivity = (row part) + (other part); the parts ride in slot 6 of the staged rows
            const float ps = (row_has ? xr[6] : 0.f) + (oth_has ? xo[6] : 0.f);
#pragma unroll
            for (int c = 0; c < 7; ++c) acc[c] = live ? fmaf(ps, d[c], acc[c]) : acc[c];
        }
    }
    __device__ __forceinline__ void flush(int row) {
        float t[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) t[c] = group_sum_n<LANES>(acc[c]);
        if ((threadIdx.x & (LANES - 1)) != 0) return;
        const int n_rows = ROW_SIDE == 1 ? G.n1 : G.n2;
        float* out_p = A.out + (size_t)s * A.out_sys_stride + (size_t)((ROW_SIDE == 1 ? A.out_row0 : A.out_row0_2) + row) * A.out_stride + A.out_comp;
        if (MODE == 0) *out_p = t[0];
        if (MODE == 1) {
            *out_p = t[7];
            float4* o = (float4*)(A.own_grad + ((size_t)s * n_rows + row) * 8);
            o[0] = make_float4(t[0], t[1], t[2], t[3]); o[1] = make_float4(t[4], t[5], t[6], 0.f);
        }
        if (MODE == 2) {
            const upk_coord_t& node = ROW_SIDE == 1 ? G.node1 : G.node2;
            const int dim = ROW_SIDE == 1 ? G.dim1 : G.dim2;
            float* o = C_SENS(node, s) + (size_t)(ROW_SIDE == 1 ? G.loc1 : G.loc2)[row] * node.stride;
            // a row has one owner in this launch, so the "+=" is a plain read-modify-write (16.6 M device-scope float atomics per
            // launch of the backbone-hbond pass cost 0.23 of its 0.64 ms); the one exception is a graph whose two sides share
            // elements of one node while both row sets run in one launch (sens_overlap, set by the host): atomics there
            if (A.atomic_sens) {
#pragma unroll
                for (int c = 0; c < 7; ++c) if (c < dim) unsafeAtomicAdd(o + c, t[c]);
            } else {
#pragma unroll
                for (int c = 0; c < 7; ++c) if (c < dim) o[c] += t[c];
            }
        }
    }
};

// SIDES: 1 or 2 = the rows of that side; 3 = side 1, then side 2 (protein_hbond: both row sets in one launch)
template <int IT, int SIDES, int MODE, bool POLY>
__device__ __forceinline__ void d_pair_rows(const upk_igraph_t& G, const PairArgs& A, const BX B, float* lds) {
        const int s = B.by;
    const PairLds L = pair_lds(lds, G, A.tab_floats);
    const float* S1 = A.sens1 ? A.sens1 + (size_t)s * A.sens_sys_stride : nullptr;
    const float* S2 = A.sens2 ? A.sens2 + (size_t)s * A.sens_sys_stride : nullptr;
    stage_table(L.tab, POLY ? G.param_poly : G.param, A.tab_floats);
    // rows: [0,dim) coordinates, [6] per-element pair sensitivity (mode 2, sides with dim <= 6), [7] element type
    stage_rows(L.c1, G.node1, s, G.loc1, G.n1, G.dim1, G.type1, nullptr, (MODE == 2 && G.dim1 <= 6) ? S1 : nullptr, A.sens_stride);
    stage_rows(L.c2, G.node2, s, G.loc2, G.n2, G.dim2, G.type2, nullptr, (MODE == 2 && G.dim2 <= 6) ? S2 : nullptr, A.sens_stride);
    // (backbone hydrogen bonds: a donor or acceptor has two or three partners in range -- groups of 2 lanes, 32 rows per wavefront
    //  batch, instead of 8-lane groups of which six lanes would evaluate dead pairs: 0.28 + 0.41 -> 0.23 + 0.29 ms per step at 4096 systems)
    constexpr int LANES = IT == UPK_IT_PROTEIN_HBOND ? 2 : PG_LANES;
    if (SIDES & 1) {
        if (threadIdx.x == 0) *L.counter = 0;
        stage_ranges(L.range, L.ord, G.hcnt1 + (size_t)s * G.n1, nullptr, G.ord1 + (size_t)s * G.n1, G.n1);
        __syncthreads();
        RowOp<IT, 1, MODE, POLY, LANES> op(G, L, A, s);
        group_batch_loop<RowOp<IT, 1, MODE, POLY, LANES>, LANES>(op, G.n1, L.ord, L.range, (const PW*)G.hit1 + (size_t)s * G.n1 * G.cap1, G.cap1, L.counter, B.bx, B.gx);
    }
    if (SIDES & 2) {
        if (SIDES == 3) __syncthreads();
        if (threadIdx.x == 0) *L.counter = 0;
        stage_ranges(L.range, L.ord, G.hcnt2 + (size_t)s * G.n2, nullptr, G.ord2 + (size_t)s * G.n2, G.n2);
        __syncthreads();
        RowOp<IT, 2, MODE, POLY, LANES> op(G, L, A, s);
        group_batch_loop<RowOp<IT, 2, MODE, POLY, LANES>, LANES>(op, G.n2, L.ord, L.range, (const PW*)G.hit2 + (size_t)s * G.n2 * G.cap2, G.cap2, L.counter, B.bx, B.gx);
    }
}
template <int IT, int SIDES, int MODE, bool POLY>
__global__ void __launch_bounds__(1024) PG_KERNEL_ATTR k_pair_rows(upk_igraph_t G, PairArgs A)  {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn_[];
    d_pair_rows<IT, SIDES, MODE, POLY>(G, A, BX_REAL, lds_dyn_);
}

// End of a backward pass whose workgroup is the system's only one: the other side's totals leave the LDS accumulators ELEMENT by element --
// one lane loads the element's row index, its DO sums and its sens row (16-byte accesses: rows of a coordinate node are padded to
// multiples of 4 floats), adds and stores.  (The first form walked the DO * n accumulators one by one: six or seven trips per lane, each
// a chain of two dependent global loads, 12-24 us at the end of a 94 us workgroup.)  get(c, i): the total of component c of element i.
template <int DO, typename Get>
__device__ __forceinline__ void flush_other_side(float* __restrict__ osens, const int* __restrict__ oloc, int stride, int n_other, Get get) {
    for (int i = threadIdx.x; i < n_other; i += blockDim.x) {
        float add[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) add[c] = c < DO ? get(c, i) : 0.f;
        float4* row = (float4*)(osens + (size_t)oloc[i] * stride);
        float4 r0 = row[0];
        r0.x += add[0]; r0.y += add[1]; r0.z += add[2]; r0.w += add[3];
        if (DO > 4) { float4 r1 = row[1]; r1.x += add[4]; r1.y += add[5]; r1.z += add[6]; r1.w += add[7]; row[1] = r1; }
        row[0] = r0;
    }
}
// ---- backward, ONE visit per pair over the rows of side RS (the side the forward pass ran over): the row element's gradient
// accumulates in registers; the other element's goes through 64-bit integer LDS atomics as exact fixed point (igraph_device.h:
// to_fixed32), so its total does not depend on the order in which the pairs arrive -- results stay bit-reproducible
template <int IT, int RS, bool POLY>
struct BackwardOp {
    static constexpr int DR = RS == 1 ? PairDims<IT>::d1 : PairDims<IT>::d2;     // components of the row / the other element
    static constexpr int DO = RS == 1 ? PairDims<IT>::d2 : PairDims<IT>::d1;
    const upk_igraph_t& G; const QuadShape Q; const PairLds& L; unsigned long long* oacc;
    const bool row_has, oth_has;
    float xr[8], acc[DR];
    float* row_sens; const int* row_loc; int row_stride;
    __device__ __forceinline__ BackwardOp(const upk_igraph_t& G_, const PairLds& L_, unsigned long long* oacc_, int sens_mode, int s)
        : G(G_), Q(quad_shape(G_)), L(L_), oacc(oacc_), row_has(sens_mode == 3 || sens_mode == RS), oth_has(sens_mode == 3 || sens_mode == 3 - RS) {
        const upk_coord_t& node = RS == 1 ? G.node1 : G.node2;
        row_sens = C_SENS(node, s); row_loc = RS == 1 ? G.loc1 : G.loc2; row_stride = node.stride;
    }
    __device__ __forceinline__ void begin(int row) {
        load_row8(xr, (RS == 1 ? L.c1 : L.c2) + row * 8);
#pragma unroll
        for (int c = 0; c < DR; ++c) acc[c] = 0.f;
    }
    __device__ __forceinline__ void body(int, int j, bool live) {
        float xo[8], d1[8], d2[8];
        load_row8(xo, (RS == 1 ? L.c2 : L.c1) + j * 8);
        const int tr = __float_as_int(xr[7]), to = __float_as_int(xo[7]);
        if (RS == 1) pair_functor_both<IT, POLY>(G, Q, L.tab, tr, to, xr, xo, d1, d2);
        else pair_functor_both<IT, POLY>(G, Q, L.tab, to, tr, xo, xr, d1, d2);
        // pair sensitivity = (row part) + (other part); the parts ride in slot 6 of the staged rows
        const float ps = (row_has ? xr[6] : 0.f) + (oth_has ? xo[6] : 0.f);
        const float* dr = RS == 1 ? d1 : d2; const float* dv = RS == 1 ? d2 : d1;
#pragma unroll
        for (int c = 0; c < DR; ++c) acc[c] = live ? fmaf(ps, dr[c], acc[c]) : acc[c];
        if (live) {
#pragma unroll
            for (int c = 0; c < DO; ++c) lds_add_fixed(oacc + j * DO + c, ps * dv[c]);
        }
    }
    __device__ __forceinline__ void flush(int row) {
        float t[DR];
#pragma unroll
        for (int c = 0; c < DR; ++c) t[c] = group_sum(acc[c]);
        if ((threadIdx.x & (PG_LANES - 1)) != 0) return;
        float* o = row_sens + (size_t)row_loc[row] * row_stride;     // the row has one owner in this launch: a plain "+="
#pragma unroll
        for (int c = 0; c < DR; ++c) o[c] += t[c];      // (the element's other writer, the epilogue below / the finish kernel, runs after a barrier)
    }
};

template <int IT, int RS, bool POLY>
__device__ __forceinline__ void d_pair_backward(const upk_igraph_t& G, const PairArgs& A, const BX B, float* lds) {
        constexpr int DO = BackwardOp<IT, RS, POLY>::DO;
    const int s = B.by;
    const int n_rows = RS == 1 ? G.n1 : G.n2, n_other = RS == 1 ? G.n2 : G.n1;
    const PairLds L = pair_lds(lds, G, A.tab_floats);
    // (offset arithmetic on the LDS base, not an integer round trip of the pointer: the accumulators must stay ds_add_u64, not flat atomics)
    unsigned long long* oacc = (unsigned long long*)lds + (((size_t)((float*)(L.counter + 1) - lds) + 1) >> 1);
    const float* S1 = A.sens1 ? A.sens1 + (size_t)s * A.sens_sys_stride : nullptr;
    const float* S2 = A.sens2 ? A.sens2 + (size_t)s * A.sens_sys_stride : nullptr;
    stage_table(L.tab, POLY ? G.param_poly : G.param, A.tab_floats);
    // rows: [0,dim) coordinates, [6] per-element pair sensitivity (sides with dim <= 6), [7] element type
    stage_rows(L.c1, G.node1, s, G.loc1, G.n1, G.dim1, G.type1, nullptr, G.dim1 <= 6 ? S1 : nullptr, A.sens_stride);
    stage_rows(L.c2, G.node2, s, G.loc2, G.n2, G.dim2, G.type2, nullptr, G.dim2 <= 6 ? S2 : nullptr, A.sens_stride);
    for (int t = threadIdx.x; t < n_other * DO; t += blockDim.x) oacc[t] = 0ull;
    if (threadIdx.x == 0) *L.counter = 0;
    stage_ranges(L.range, L.ord, (RS == 1 ? G.hcnt1 : G.hcnt2) + (size_t)s * n_rows, nullptr, (RS == 1 ? G.ord1 : G.ord2) + (size_t)s * n_rows, n_rows);
    __syncthreads();
    {
        BackwardOp<IT, RS, POLY> op(G, L, oacc, A.sens_mode, s);
        const int cap = RS == 1 ? G.cap1 : G.cap2;
        group_batch_loop(op, n_rows, L.ord, L.range, (const PW*)(RS == 1 ? G.hit1 : G.hit2) + (size_t)s * n_rows * cap, cap, L.counter, B.bx, B.gx);
    }
    __syncthreads();
    const upk_coord_t& onode = RS == 1 ? G.node2 : G.node1;
    const int* oloc = RS == 1 ? G.loc2 : G.loc1;
    float* osens = C_SENS(onode, s);
    unsigned long long* gacc = G.gacc ? G.gacc + (size_t)s * n_other * 8 : nullptr;
    const bool alone = B.gx == 1;        // the system's only workgroup: its accumulators are the totals
    if (alone && (onode.stride & 3) == 0) { flush_other_side<DO>(osens, oloc, onode.stride, n_other, [&](int c, int i) { return from_fixed32(oacc[i * DO + c]); }); return; }
    for (int t = threadIdx.x; t < n_other * DO; t += blockDim.x) {
        const unsigned long long a = oacc[t];
        if (!a) continue;
        const int i = t / DO, c = t - i * DO;
        if (alone) osens[(size_t)oloc[i] * onode.stride + c] += from_fixed32(a);
        else atomicAdd(gacc + i * 8 + c, a);   // exact partial sums of the system's workgroups; k_pair_backward_finish converts
    }
}
template <int IT, int RS, bool POLY>
__global__ void __launch_bounds__(1024) PG_KERNEL_ATTR k_pair_backward(upk_igraph_t G, PairArgs A)  {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn_[];
    d_pair_backward<IT, RS, POLY>(G, A, BX_REAL, lds_dyn_);
}
// ---- packed passes of the hbond_coverage graphs (pair2_device.h): two partners per lane, 4-lane row groups ----------------
// x1 is always the side-1 element (H-bond site, 7 components: [6] = its bond probability), x2 the side-2 element (bead).
// Staged rows carry a sentinel element behind each side (lanes past the end of their row evaluate it with weight zero).
__device__ __forceinline__ PairLds pair2_lds(float* lds, const upk_igraph_t& G, int tab_floats) {
    PairLds L;
    L.tab = lds;
    L.c1 = lds + ((tab_floats + 3) & ~3);
    L.c2 = L.c1 + (G.n1 + 1) * 8;
    const int n_max = G.n1 > G.n2 ? G.n1 : G.n2;
    L.range = (int*)(L.c2 + (G.n2 + 1) * 8);
    L.ord = (unsigned short*)(L.range + n_max);
    L.counter = L.range + PG_WALK_LDS_WORDS(n_max);
    return L;
}
// value of two hbond_coverage pairs (hbond.cpp:261-276) and, if WANT_D, the derivatives: d1[0..7) w.r.t. the sites, d2[0..6) w.r.t. the beads
// (o1 / o2: the elements' shares of the table row's offset in floats, staged in place of their types -- site: type1 * n_type2 * row length,
//  bead: type2 * row length, d_cov_rows2 / d_cov_backward2)
template <bool WANT_D, bool POLY>
__device__ __forceinline__ v2 coverage_pair2(const upk_igraph_t& G, const QuadShape& Q, const float* tab, int o1A, int o1B, int o2A, int o2B,
                                             const v2* x1, v2 hb1, const v2* x2, v2* d1, v2* d2) {
    const float* pA = tab + (o1A + o2A); const float* pB = tab + (o1B + o2B);
    const int o2 = POLY ? 4 * (Q.ka - 3) : Q.ka;
    v2 dd[3], g1[3], g2[3];
    const v2 coverage = quadspline_pair2<WANT_D, POLY>(Q, pA, pB, x1, x2, dd, g1, g2, 0, o2, 0, o2);
    const v2 one_m = bc2(1.f) - hb1, prefactor = one_m * one_m;
    if (WANT_D) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { d2[c] = prefactor * dd[c]; d1[c] = -d2[c]; d1[3 + c] = prefactor * g1[c]; d2[3 + c] = prefactor * g2[c]; }
        d1[6] = coverage * (one_m * bc2(-2.f));
    }
    return prefactor * coverage;
}

// forward: row sums of the pair value over the rows of side RS
template <int RS, bool POLY>
struct CovRowOp2 {
    const upk_igraph_t& G; const QuadShape Q; const PairLds& L; const PairArgs& A; const int s;
    v2 xr[6], hbr, acc; int tr;
    __device__ __forceinline__ CovRowOp2(const upk_igraph_t& G_, const PairLds& L_, const PairArgs& A_, int s_) : G(G_), Q(quad_shape(G_)), L(L_), A(A_), s(s_) {}
    __device__ __forceinline__ void begin(int row) {
        float x[8]; load_row8_planes(x, RS == 1 ? L.c1 : L.c2, (RS == 1 ? G.n1 : G.n2) + 1, row);
#pragma unroll
        for (int c = 0; c < 6; ++c) xr[c] = bc2(x[c]);
        hbr = bc2(x[6]); tr = __float_as_int(x[7]); acc = bc2(0.f);
    }
    __device__ __forceinline__ void body(int, int jA, int jB, bool liveA, bool liveB) {
        float xa[8], xb[8];
        load_row8_planes(xa, RS == 1 ? L.c2 : L.c1, (RS == 1 ? G.n2 : G.n1) + 1, jA); load_row8_planes(xb, RS == 1 ? L.c2 : L.c1, (RS == 1 ? G.n2 : G.n1) + 1, jB);
        v2 xo[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) xo[c] = mk2(xa[c], xb[c]);
        const int toA = __float_as_int(xa[7]), toB = __float_as_int(xb[7]);
        const v2 v = RS == 1 ? coverage_pair2<false, POLY>(G, Q, L.tab, tr, tr, toA, toB, xr, hbr, xo, nullptr, nullptr)
                             : coverage_pair2<false, POLY>(G, Q, L.tab, toA, toB, tr, tr, xo, mk2(xa[6], xb[6]), xr, nullptr, nullptr);
        acc += mk2(liveA ? v.x : 0.f, liveB ? v.y : 0.f);
    }
    __device__ __forceinline__ void flush(int row) {
        const float t = group_sum4(acc.x + acc.y);
        if ((threadIdx.x & (P2_LANES - 1)) != 0) return;
        A.out[(size_t)s * A.out_sys_stride + (size_t)((RS == 1 ? A.out_row0 : A.out_row0_2) + row) * A.out_stride + A.out_comp] = t;
    }
};
template <int RS, bool POLY>
__device__ __forceinline__ void d_cov_rows2(const upk_igraph_t& G, const PairArgs& A, const BX B, float* lds) {
        const int s = B.by;
    const PairLds L = pair2_lds(lds, G, A.tab_floats);
    stage_table(L.tab, POLY ? G.param_poly : G.param, A.tab_floats);
    const int row_len = POLY ? G.n_poly : G.n_param;
    stage_rows_planes(L.c1, G.node1, s, G.loc1, G.n1, G.dim1, G.type1, nullptr, nullptr, 0, 0.f, __int_as_float(0), G.n_type2 * row_len);
    stage_rows_planes(L.c2, G.node2, s, G.loc2, G.n2, G.dim2, G.type2, nullptr, nullptr, 0, 0.f, __int_as_float(0), row_len);
    if (threadIdx.x == 0) *L.counter = 0;
    const int n_rows = RS == 1 ? G.n1 : G.n2, cap = RS == 1 ? G.cap1 : G.cap2;
    stage_ranges(L.range, L.ord, (RS == 1 ? G.hcnt1 : G.hcnt2) + (size_t)s * n_rows, nullptr, (RS == 1 ? G.ord1 : G.ord2) + (size_t)s * n_rows, n_rows);
    __syncthreads();
    CovRowOp2<RS, POLY> op(G, L, A, s);
    group2_batch_loop(op, n_rows, L.ord, L.range, (const PW*)(RS == 1 ? G.hit1 : G.hit2) + (size_t)s * n_rows * cap, cap, L.counter, B.bx, B.gx, RS == 1 ? G.n2 : G.n1);
}
template <int RS, bool POLY>
__global__ void __launch_bounds__(1024) PG_KERNEL_ATTR k_cov_rows2(upk_igraph_t G, PairArgs A)  {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn_[];
    d_cov_rows2<RS, POLY>(G, A, BX_REAL, lds_dyn_);
}

// backward over the rows of side RS, ONE visit per pair: the row element's gradient in registers, the partner's through the
// fixed-point LDS accumulators (k_pair_backward above, two partners per lane)
template <int RS, bool POLY>
struct CovBackwardOp2 {
    static constexpr int DR = RS == 1 ? 7 : 6, DO = RS == 1 ? 6 : 7;
    const upk_igraph_t& G; const QuadShape Q; const PairLds& L; unsigned long long* oacc;
    const bool row_has, oth_has;
    v2 xr[6], hbr, acc[DR]; float sr; int tr;
    float* row_sens; const int* row_loc; int row_stride;
    __device__ __forceinline__ CovBackwardOp2(const upk_igraph_t& G_, const PairLds& L_, unsigned long long* oacc_, int sens_mode, int s)
        : G(G_), Q(quad_shape(G_)), L(L_), oacc(oacc_), row_has(sens_mode == 3 || sens_mode == RS), oth_has(sens_mode == 3 || sens_mode == 3 - RS) {
        const upk_coord_t& node = RS == 1 ? G.node1 : G.node2;
        row_sens = node.sens + (size_t)s * node.n_elem * node.stride; row_loc = RS == 1 ? G.loc1 : G.loc2; row_stride = node.stride;
    }
    // staged rows: sites [0,7) coordinates, beads [0,6) coordinates + [6] the bead's pair sensitivity; [7] the type.  The sites' pair
    // sensitivity (sens_mode 1 / 3) rides in a separate LDS array in front of the accumulators: L-side detail of the kernel below.
    const float* site_sens = nullptr;
    __device__ __forceinline__ void begin(int row) {
        float x[8]; load_row8_planes(x, RS == 1 ? L.c1 : L.c2, (RS == 1 ? G.n1 : G.n2) + 1, row);
#pragma unroll
        for (int c = 0; c < 6; ++c) xr[c] = bc2(x[c]);
        hbr = bc2(x[6]); tr = __float_as_int(x[7]);
        sr = !row_has ? 0.f : (RS == 1 ? site_sens[row] : x[6]);
#pragma unroll
        for (int c = 0; c < DR; ++c) acc[c] = bc2(0.f);
    }
    __device__ __forceinline__ void body(int, int jA, int jB, bool liveA, bool liveB) {
        float xa[8], xb[8];
        load_row8_planes(xa, RS == 1 ? L.c2 : L.c1, (RS == 1 ? G.n2 : G.n1) + 1, jA); load_row8_planes(xb, RS == 1 ? L.c2 : L.c1, (RS == 1 ? G.n2 : G.n1) + 1, jB);
        v2 xo[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) xo[c] = mk2(xa[c], xb[c]);
        const int toA = __float_as_int(xa[7]), toB = __float_as_int(xb[7]);
        v2 d1[7], d2[6];
        if (RS == 1) coverage_pair2<true, POLY>(G, Q, L.tab, tr, tr, toA, toB, xr, hbr, xo, d1, d2);
        else coverage_pair2<true, POLY>(G, Q, L.tab, toA, toB, tr, tr, xo, mk2(xa[6], xb[6]), xr, d1, d2);
        // pair sensitivity = (row part) + (other part), zero for the dead halves
        float soA = 0.f, soB = 0.f;
        if (oth_has) { soA = RS == 1 ? xa[6] : site_sens[jA]; soB = RS == 1 ? xb[6] : site_sens[jB]; }
        const v2 ps = mk2(liveA ? sr + soA : 0.f, liveB ? sr + soB : 0.f);
        const v2* dr = RS == 1 ? d1 : d2; const v2* dv = RS == 1 ? d2 : d1;
#pragma unroll
        for (int c = 0; c < DR; ++c) acc[c] = fma2(ps, dr[c], acc[c]);
        v2 od[DO];
        const v2 pss = ps * bc2(P2_FIX_SCALE);              // (fixed-point scale folded into the sensitivity: two values per instruction)
#pragma unroll
        for (int c = 0; c < DO; ++c) od[c] = pss * dv[c];
        const int n_o = RS == 1 ? G.n2 : G.n1;          // accumulators as DO planes [component][element]
        if (liveA) {
#pragma unroll
            for (int c = 0; c < DO; ++c) lds_add_fixed22_scaled(oacc + c * n_o + jA, od[c].x);
        }
        if (liveB) {
#pragma unroll
            for (int c = 0; c < DO; ++c) lds_add_fixed22_scaled(oacc + c * n_o + jB, od[c].y);
        }
    }
    __device__ __forceinline__ void flush(int row) {
        float t[DR];
#pragma unroll
        for (int c = 0; c < DR; ++c) t[c] = group_sum4(acc[c].x + acc[c].y);
        if ((threadIdx.x & (P2_LANES - 1)) != 0) return;
        float* o = row_sens + (size_t)row_loc[row] * row_stride;     // the row has one owner in this launch: a plain "+="
#pragma unroll
        for (int c = 0; c < DR; ++c) o[c] += t[c];
    }
};
template <int RS, bool POLY>
__device__ __forceinline__ void d_cov_backward2(const upk_igraph_t& G, const PairArgs& A, const BX B, float* lds) {
        constexpr int DO = CovBackwardOp2<RS, POLY>::DO;
    const int s = B.by;
    const int n_rows = RS == 1 ? G.n1 : G.n2, n_other = RS == 1 ? G.n2 : G.n1;
    const PairLds L = pair2_lds(lds, G, A.tab_floats);
    // (offset arithmetic on the LDS base, not an integer round trip of the pointer: the accumulators must stay ds_add_u64, not flat atomics)
    unsigned long long* oacc = (unsigned long long*)lds + (((size_t)((float*)(L.counter + 1) - lds) + 1) >> 1);
    float* site_sens = (float*)(oacc + (size_t)n_other * DO);            // [n1] (only when the sites carry a pair sensitivity)
    const float* S1 = A.sens1 ? A.sens1 + (size_t)s * A.sens_sys_stride : nullptr;
    const float* S2 = A.sens2 ? A.sens2 + (size_t)s * A.sens_sys_stride : nullptr;
    stage_table(L.tab, POLY ? G.param_poly : G.param, A.tab_floats);
    const int row_len = POLY ? G.n_poly : G.n_param;
    stage_rows_planes(L.c1, G.node1, s, G.loc1, G.n1, G.dim1, G.type1, nullptr, nullptr, 0, 0.f, __int_as_float(0), G.n_type2 * row_len);
    stage_rows_planes(L.c2, G.node2, s, G.loc2, G.n2, G.dim2, G.type2, nullptr, S2, A.sens_stride, 0.f, __int_as_float(0), row_len);
    if (S1) for (int t = threadIdx.x; t < G.n1; t += blockDim.x) site_sens[t] = S1[(size_t)t * A.sens_stride];
    for (int t = threadIdx.x; t < n_other * DO; t += blockDim.x) oacc[t] = 0ull;
    if (threadIdx.x == 0) *L.counter = 0;
    stage_ranges(L.range, L.ord, (RS == 1 ? G.hcnt1 : G.hcnt2) + (size_t)s * n_rows, nullptr, (RS == 1 ? G.ord1 : G.ord2) + (size_t)s * n_rows, n_rows);
    __syncthreads();
    {
        CovBackwardOp2<RS, POLY> op(G, L, oacc, A.sens_mode, s);
        op.site_sens = site_sens;
        const int cap = RS == 1 ? G.cap1 : G.cap2;
        group2_batch_loop(op, n_rows, L.ord, L.range, (const PW*)(RS == 1 ? G.hit1 : G.hit2) + (size_t)s * n_rows * cap, cap, L.counter, B.bx, B.gx, n_other);
    }
    __syncthreads();
    const upk_coord_t& onode = RS == 1 ? G.node2 : G.node1;
    const int* oloc = RS == 1 ? G.loc2 : G.loc1;
    float* osens = C_SENS(onode, s);
    unsigned long long* gacc = G.gacc ? G.gacc + (size_t)s * n_other * 8 : nullptr;
    const bool alone = B.gx == 1;        // the system's only workgroup: its accumulators are the totals
    if (alone && (onode.stride & 3) == 0) { flush_other_side<DO>(osens, oloc, onode.stride, n_other, [&](int c, int i) { return from_fixed22(oacc[c * n_other + i]); }); return; }
    for (int t = threadIdx.x; t < n_other * DO; t += blockDim.x) {
        const int i = t / DO, c = t - i * DO;
        const unsigned long long a = oacc[c * n_other + i];
        if (!a) continue;
        if (alone) osens[(size_t)oloc[i] * onode.stride + c] += from_fixed22(a);
        else atomicAdd(gacc + i * 8 + c, a);   // exact partial sums of the system's workgroups; k_pair_backward_finish converts
    }
}
template <int RS, bool POLY>
__global__ void __launch_bounds__(1024) PG_KERNEL_ATTR k_cov_backward2(upk_igraph_t G, PairArgs A)  {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn_[];
    d_cov_backward2<RS, POLY>(G, A, BX_REAL, lds_dyn_);
}
static bool pair2_enabled() {          // UPSIDE_HIP_PAIR2=0: the scalar passes (one partner per lane) -- A/B and tests
    static int v = -1;
    if (v < 0) { const char* e = getenv("UPSIDE_HIP_PAIR2"); v = (e && !atoi(e)) ? 0 : 1; }
    return v != 0;
}

// global accumulators (several workgroups per system) -> the other side's sens; cleared for the next evaluation
__device__ __forceinline__ void d_pair_backward_finish(const upk_igraph_t& G, int other_side, double unit, const BX B, float* lds_unused) {   // unit: value of one accumulator count
    const int s = B.by;
    const int n_other = other_side == 1 ? G.n1 : G.n2, dim = other_side == 1 ? G.dim1 : G.dim2;
    const upk_coord_t& onode = other_side == 1 ? G.node1 : G.node2;
    const int* oloc = other_side == 1 ? G.loc1 : G.loc2;
    float* osens = C_SENS(onode, s);
    unsigned long long* gacc = G.gacc + (size_t)s * n_other * 8;
    for (int t = B.bx * blockDim.x + threadIdx.x; t < n_other * 8; t += B.gx * blockDim.x) {
        const int i = t >> 3, c = t & 7;
        if (c >= dim) continue;
        const unsigned long long a = gacc[t];
        if (a) { osens[(size_t)oloc[i] * onode.stride + c] += (float)((double)(long long)a * unit); gacc[t] = 0ull; }
    }
}
__global__ void k_pair_backward_finish(upk_igraph_t G, int other_side, double unit)  { d_pair_backward_finish(G, other_side, unit, BX_REAL, nullptr); }

// LDS bytes of a staged pair pass; false when the system does not fit (callers fall back to the list-walking kernels)
static bool pair_lds_bytes(const upk_igraph_t* G, bool poly, int& tab_floats, size_t& bytes) {
    tab_floats = G->n_type1 * G->n_type2 * (poly ? G->n_poly : G->n_param);
    if (poly && !(G->itype == UPK_IT_HBOND_COVERAGE && G->param_poly)) return false;
    const int n_max = G->n1 > G->n2 ? G->n1 : G->n2;
    bytes = ((size_t)((tab_floats + 3) & ~3) + (size_t)(G->n1 + G->n2) * 8 + PG_WALK_LDS_WORDS(n_max) + 4) * sizeof(float);
    static int force_unstaged = -1;   // UPSIDE_HIP_IG_UNSTAGED=1 exercises the path taken by systems too large for LDS staging
    if (force_unstaged < 0) { const char* e = getenv("UPSIDE_HIP_IG_UNSTAGED"); force_unstaged = (e && atoi(e)) ? 1 : 0; }
    return bytes <= 158 * 1024 && !force_unstaged && n_max < 65536 && G->cap1 < 65536 && G->cap2 < 65536;
}

template <int IT, int SIDES, bool POLY>
static void rows_launch(const upk_launch_t* L, const upk_igraph_t* G, int mode, dim3 grid, dim3 block, size_t lds, const PairArgs& A) {
    if (mode == 0) hipLaunchKernelGGL((k_pair_rows<IT, SIDES, 0, POLY>), grid, block, lds, ST(L), *G, A);
    else if (mode == 1) hipLaunchKernelGGL((k_pair_rows<IT, SIDES, 1, POLY>), grid, block, lds, ST(L), *G, A);
    else hipLaunchKernelGGL((k_pair_rows<IT, SIDES, 2, POLY>), grid, block, lds, ST(L), *G, A);
}
template <int IT, bool POLY = false>
static void rows_launch_sides(const upk_launch_t* L, const upk_igraph_t* G, int side, int mode, dim3 grid, dim3 block, size_t lds, const PairArgs& A) {
    if (side == 1) rows_launch<IT, 1, POLY>(L, G, mode, grid, block, lds, A);
    else if (side == 2) rows_launch<IT, 2, POLY>(L, G, mode, grid, block, lds, A);
    else rows_launch<IT, 3, POLY>(L, G, mode, grid, block, lds, A);
}
// the polynomial table when it fits LDS next to the elements (and extra bytes), else the spline coefficients, else nothing fits
static int pair_table_choice(const upk_igraph_t* G, size_t extra, int& tab_floats, size_t& lds) {
    static int no_poly = -1;      // UPSIDE_HIP_IG_POLY=0: always the spline-coefficient table (A/B and the large-table path)
    if (no_poly < 0) { const char* e = getenv("UPSIDE_HIP_IG_POLY"); no_poly = (e && !atoi(e)) ? 1 : 0; }
    if (!no_poly && pair_lds_bytes(G, true, tab_floats, lds) && lds + extra <= 158 * 1024) return 2;
    if (pair_lds_bytes(G, false, tab_floats, lds) && lds + extra <= 158 * 1024) return 1;
    return 0;
}

extern "C" int upk_igraph_passes_staged(const upk_launch_t* L, const upk_igraph_t* G, int row_side) {
    UPK_FLUSH(L);
    (void)L;
    if (row_side != 1 && row_side != 2) return 0;
    const int n_other = row_side == 1 ? G->n2 : G->n1;
    int tab_floats; size_t lds;
    if (!pair_table_choice(G, 0, tab_floats, lds)) return 0;
    return pair_table_choice(G, 8 + (size_t)n_other * 8 * sizeof(unsigned long long), tab_floats, lds) ? 1 : 0;   // (an upper bound of the accumulators)
}

extern "C" int upk_igraph_rows(const upk_launch_t* L, const upk_igraph_t* G, int side, int mode, float* out, long out_sys_stride,
                               int out_stride, int out_comp, int out_row0, int out_row0_2, float* own_grad, int sens_mode,
                               const float* sens1, const float* sens2, long sens_sys_stride, int sens_stride) {
    if (side < 1 || side > 3 || mode < 0 || mode > 2 || (mode == 1 && (!own_grad || side == 3))) return 9007;
    if (!list_words_match(G)) return 9010;
    PairArgs A; memset(&A, 0, sizeof(A));
    A.out = out; A.out_sys_stride = out_sys_stride; A.out_stride = out_stride; A.out_comp = out_comp;
    A.out_row0 = side == 2 ? 0 : out_row0; A.out_row0_2 = side == 2 ? out_row0 : out_row0_2;
    A.own_grad = own_grad;
    A.sens_mode = sens_mode; A.sens1 = sens1; A.sens2 = sens2; A.sens_sys_stride = sens_sys_stride; A.sens_stride = sens_stride;
    A.atomic_sens = side == 3 && G->sens_overlap;
    size_t lds;
    const int table = pair_table_choice(G, 0, A.tab_floats, lds);
    if (!table) {                                      // list-walking kernels, one side at a time
        UPK_FLUSH(L);
        int r = 0;
        for (int sd = 1; sd <= 2 && !r; ++sd) {
            if (!(side & sd)) continue;
            const int row0 = sd == 1 ? A.out_row0 : A.out_row0_2;
            if (mode == 2) r = upk_igraph_grad(L, G, sd, sens_mode, sens1, sens2, sens_sys_stride, sens_stride);
            else r = upk_igraph_rowsum(L, G, sd, out, out_sys_stride, out_stride, out_comp, row0, mode == 1 ? own_grad : nullptr);
        }
        return r;
    }
    const int n_rows = side == 1 ? G->n1 : (side == 2 ? G->n2 : (G->n1 > G->n2 ? G->n1 : G->n2));
    int bps, threads;
    if (G->itype == UPK_IT_HBOND_COVERAGE && mode == 0 && side != 3 && pair2_enabled() && lds + 64 <= 158 * 1024) {   // packed pass
        pair2_geometry(L->n_system, n_rows, bps, threads);
        if (side == 2 && batch_add(L, table == 2 ? BK_COV_ROWS2_POLY : BK_COV_ROWS2, bps, L->n_system, lds + 64, G, sizeof(*G), &A, sizeof(A))) return 0;
        UPK_FLUSH(L);
        const dim3 grid2(bps, L->n_system), block2(threads);
        if (side == 1) { if (table == 2) hipLaunchKernelGGL((k_cov_rows2<1, true>), grid2, block2, lds + 64, ST(L), *G, A); else hipLaunchKernelGGL((k_cov_rows2<1, false>), grid2, block2, lds + 64, ST(L), *G, A); }
        else { if (table == 2) hipLaunchKernelGGL((k_cov_rows2<2, true>), grid2, block2, lds + 64, ST(L), *G, A); else hipLaunchKernelGGL((k_cov_rows2<2, false>), grid2, block2, lds + 64, ST(L), *G, A); }
        return launch_status();
    }
    pair_geometry(L->n_system, n_rows, bps, threads);
    // (a few hundred rows of two or three pairs: the pass is the latency of staging the system, which several small workgroups
    //  per CU overlap: 1024 lanes 0.29 + 0.23 ms, 512: 0.25 + 0.18, 256: 0.23 + 0.17, 128: 0.29 + 0.20; the environment graph, 300 rows of ~40 pairs, is best left at 1024)
    if (G->itype == UPK_IT_PROTEIN_HBOND && bps == 1) {
        threads = 256;
    }
    {   // merged launch (kernels_batch.h): the instances the README force field uses
        const int bk = (G->itype == UPK_IT_PROTEIN_HBOND && side == 3 && mode == 0) ? BK_ROWS_HB_FWD : (G->itype == UPK_IT_PROTEIN_HBOND && side == 3 && mode == 2) ? BK_ROWS_HB_BWD
                     : (G->itype == UPK_IT_ENVIRONMENT && side == 1 && mode == 0) ? BK_ROWS_ENV_FWD : 0;
        if (bk && batch_add(L, bk, bps, L->n_system, lds, G, sizeof(*G), &A, sizeof(A))) return 0;
    }
    UPK_FLUSH(L);
    const dim3 grid(bps, L->n_system), block(threads);
    switch (G->itype) {
        case UPK_IT_HBOND_COVERAGE:
            if (table == 2) rows_launch_sides<UPK_IT_HBOND_COVERAGE, true>(L, G, side, mode, grid, block, lds, A);
            else rows_launch_sides<UPK_IT_HBOND_COVERAGE, false>(L, G, side, mode, grid, block, lds, A);
            break;
        case UPK_IT_ENVIRONMENT: rows_launch_sides<UPK_IT_ENVIRONMENT>(L, G, side, mode, grid, block, lds, A); break;
        case UPK_IT_PROTEIN_HBOND: rows_launch_sides<UPK_IT_PROTEIN_HBOND>(L, G, side, mode, grid, block, lds, A); break;
        default: return 9008;
    }
    return launch_status();
}

// the finish pass of a backward item that DID join a merged launch but could not join itself (kind masked out, or the queue flush inside
// batch_add failed): what the batch holds runs first, then the conversion is launched on its own -- never dropped
static int finish_alone(const upk_launch_t* L, const upk_igraph_t* G, int n_other, int other_side, double unit) {
    const int r = upk_batch_run(L);
    if (r) return r;
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_pair_backward_finish, dim3((n_other * 8 + 255) / 256, L->n_system), dim3(256), 0, ST(L), *G, other_side, unit);
    return launch_status();
}

extern "C" int upk_igraph_backward(const upk_launch_t* L, const upk_igraph_t* G, int row_side, int sens_mode, const float* sens1,
                                   const float* sens2, long sens_sys_stride, int sens_stride) {
    if (row_side != 1 && row_side != 2) return 9007;
    if (!list_words_match(G)) return 9010;
    PairArgs A; memset(&A, 0, sizeof(A));
    A.sens_mode = sens_mode; A.sens1 = sens1; A.sens2 = sens2; A.sens_sys_stride = sens_sys_stride; A.sens_stride = sens_stride;
    const int n_rows = row_side == 1 ? G->n1 : G->n2, n_other = row_side == 1 ? G->n2 : G->n1;
    const int d_other = G->itype == UPK_IT_HBOND_COVERAGE ? (row_side == 1 ? PairDims<UPK_IT_HBOND_COVERAGE>::d2 : PairDims<UPK_IT_HBOND_COVERAGE>::d1)
                      : G->itype == UPK_IT_ENVIRONMENT    ? (row_side == 1 ? PairDims<UPK_IT_ENVIRONMENT>::d2 : PairDims<UPK_IT_ENVIRONMENT>::d1)
                                                          : (row_side == 1 ? PairDims<UPK_IT_PROTEIN_HBOND>::d2 : PairDims<UPK_IT_PROTEIN_HBOND>::d1);
    const size_t acc_bytes = 8 + (size_t)n_other * d_other * sizeof(unsigned long long);    // the other side's accumulators (BackwardOp::DO per element)
    size_t lds;
    const int table = pair_table_choice(G, acc_bytes, A.tab_floats, lds);
    lds += acc_bytes;
    if (!table) {      // list-walking kernels, one side at a time (they need no hit lists)
        UPK_FLUSH(L);
        int r = upk_igraph_grad(L, G, 1, sens_mode, sens1, sens2, sens_sys_stride, sens_stride);
        if (!r) r = upk_igraph_grad(L, G, 2, sens_mode, sens1, sens2, sens_sys_stride, sens_stride);
        return r;
    }
    int bps, threads;
    if (G->itype == UPK_IT_HBOND_COVERAGE && pair2_enabled()) {            // packed pass: + sentinel rows and the sites' sensitivities
        const size_t extra = 64 + (size_t)G->n1 * sizeof(float);
        size_t lds2; int tf;
        const int table2 = pair_table_choice(G, acc_bytes + extra, tf, lds2);
        if (table2) {
            A.tab_floats = tf; lds2 += acc_bytes + extra;
            pair2_geometry(L->n_system, n_rows, bps, threads);
            if (!G->gacc) bps = 1;
            if (row_side == 2 && batch_add(L, table2 == 2 ? BK_COV_BWD2_POLY : BK_COV_BWD2, bps, L->n_system, lds2, G, sizeof(*G), &A, sizeof(A))) {
                if (bps > 1 && !batch_add(L, BK_BWD_FINISH, (n_other * 8 + 1023) / 1024, L->n_system, 0, G, sizeof(*G), nullptr, 0, 3 - row_side, 0, 1.0 / (double)(1 << P2_FIX_BITS)))
                    return finish_alone(L, G, n_other, 3 - row_side, 1.0 / (double)(1 << P2_FIX_BITS));
                return 0;
            }
            UPK_FLUSH(L);
            const dim3 grid2(bps, L->n_system), block2(threads);
            if (row_side == 1) { if (table2 == 2) hipLaunchKernelGGL((k_cov_backward2<1, true>), grid2, block2, lds2, ST(L), *G, A); else hipLaunchKernelGGL((k_cov_backward2<1, false>), grid2, block2, lds2, ST(L), *G, A); }
            else { if (table2 == 2) hipLaunchKernelGGL((k_cov_backward2<2, true>), grid2, block2, lds2, ST(L), *G, A); else hipLaunchKernelGGL((k_cov_backward2<2, false>), grid2, block2, lds2, ST(L), *G, A); }
            if (bps > 1) hipLaunchKernelGGL(k_pair_backward_finish, dim3((n_other * 8 + 255) / 256, L->n_system), dim3(256), 0, ST(L), *G, 3 - row_side, 1.0 / (double)(1 << P2_FIX_BITS));
            return launch_status();
        }
    }
    pair_geometry(L->n_system, n_rows, bps, threads);
    if (!G->gacc) bps = 1;
    if (G->itype == UPK_IT_ENVIRONMENT && row_side == 1 && batch_add(L, BK_ENV_BWD, bps, L->n_system, lds, G, sizeof(*G), &A, sizeof(A))) {
        if (bps > 1 && !batch_add(L, BK_BWD_FINISH, (n_other * 8 + 1023) / 1024, L->n_system, 0, G, sizeof(*G), nullptr, 0, 3 - row_side, 0, 1.0 / 4294967296.0))
            return finish_alone(L, G, n_other, 3 - row_side, 1.0 / 4294967296.0);
        return 0;
    }
    UPK_FLUSH(L);
    const dim3 grid(bps, L->n_system), block(threads);
    switch (G->itype) {
        case UPK_IT_HBOND_COVERAGE:
            if (table == 2) {
                if (row_side == 1) hipLaunchKernelGGL((k_pair_backward<UPK_IT_HBOND_COVERAGE, 1, true>), grid, block, lds, ST(L), *G, A);
                else hipLaunchKernelGGL((k_pair_backward<UPK_IT_HBOND_COVERAGE, 2, true>), grid, block, lds, ST(L), *G, A);
            } else {
                if (row_side == 1) hipLaunchKernelGGL((k_pair_backward<UPK_IT_HBOND_COVERAGE, 1, false>), grid, block, lds, ST(L), *G, A);
                else hipLaunchKernelGGL((k_pair_backward<UPK_IT_HBOND_COVERAGE, 2, false>), grid, block, lds, ST(L), *G, A);
            }
            break;
        case UPK_IT_ENVIRONMENT:
            if (row_side == 1) hipLaunchKernelGGL((k_pair_backward<UPK_IT_ENVIRONMENT, 1, false>), grid, block, lds, ST(L), *G, A);
            else hipLaunchKernelGGL((k_pair_backward<UPK_IT_ENVIRONMENT, 2, false>), grid, block, lds, ST(L), *G, A);
            break;
        case UPK_IT_PROTEIN_HBOND:
            if (row_side == 1) hipLaunchKernelGGL((k_pair_backward<UPK_IT_PROTEIN_HBOND, 1, false>), grid, block, lds, ST(L), *G, A);
            else hipLaunchKernelGGL((k_pair_backward<UPK_IT_PROTEIN_HBOND, 2, false>), grid, block, lds, ST(L), *G, A);
            break;
        default: return 9008;
    }
    if (bps > 1) hipLaunchKernelGGL(k_pair_backward_finish, dim3((n_other * 8 + 255) / 256, L->n_system), dim3(256), 0, ST(L), *G, 3 - row_side, 1.0 / 4294967296.0);
    return launch_status();
}

// backward pass of the row side when the pair sensitivity is the row element's own: sens[row] * own_grad[row]
__global__ void k_igraph_apply_own_grad(upk_igraph_t G, int side, const float* __restrict__ own_grad, const float* __restrict__ sens,
                                        long sens_sys_stride, int sens_stride) {
    const int s = blockIdx.y;
    const int n_rows = side == 1 ? G.n1 : G.n2, dim = side == 1 ? G.dim1 : G.dim2;
    const upk_coord_t& node = side == 1 ? G.node1 : G.node2;
    const int* loc = side == 1 ? G.loc1 : G.loc2;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_rows * 8; t += gridDim.x * blockDim.x) {
        const int row = t >> 3, c = t & 7;
        if (c >= dim) continue;
        const float v = sens[(size_t)s * sens_sys_stride + (size_t)row * sens_stride] * own_grad[((size_t)s * n_rows + row) * 8 + c];
        C_SENS(node, s)[(size_t)loc[row] * node.stride + c] += v;
    }
}
extern "C" int upk_igraph_apply_own_grad(const upk_launch_t* L, const upk_igraph_t* G, int side, const float* own_grad,
                                         const float* sens, long sens_sys_stride, int sens_stride) {
    UPK_FLUSH(L);
    const int n_rows = side == 1 ? G->n1 : G->n2;
    hipLaunchKernelGGL(k_igraph_apply_own_grad, dim3((n_rows * 8 + 255) / 256, L->n_system), dim3(256), 0, ST(L), *G, side, own_grad,
                       sens, sens_sys_stride, sens_stride);
    return launch_status();
}

// ---- VALU issue ceiling of this device, measured (bench.py: roofline.igraph) ---------------------------------------------
// The pair functors are long dependent chains of fp32 operations evaluated by one 1024-lane workgroup per CU (4 wavefronts
// per SIMD).  k_valu_chain runs exactly that shape with a known instruction count: ILP independent fused multiply-adds per
// lane, 16 * ILP per iteration.  ILP 1 = what scalar dependent code can issue (one VALU instruction per ~4.3 cycles per
// SIMD on gfx950, whatever the occupancy); ILP 4 lets the compiler pack pairs (v_pk_fma_f32) and approaches the
// 2-cycle-per-instruction peak the data sheet's 157 TFLOP/s assumes.
template <int ILP>
__global__ void __launch_bounds__(1024) k_valu_chain(float* out, int iters, float a, float b) {
    float x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// rates in wave-level fp32 FMA instructions per second for the whole device: [0] dependent scalar chain, [1] four independent chains
extern "C" int upk_calibrate_valu(double* rates) {
    const int cu = upk_device_cu_count(), iters = 20000;
    float* out = nullptr;
    if (hipMalloc((void**)&out, (size_t)cu * 1024 * sizeof(float)) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v) {
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {     // (the first launch warms the clocks)
            (void)hipEventRecord(e0, nullptr);
            if (v == 0) hipLaunchKernelGGL(k_valu_chain<1>, dim3(cu), dim3(1024), 0, nullptr, out, iters, 1.0001f, 0.5f);
            else hipLaunchKernelGGL(k_valu_chain<4>, dim3(cu), dim3(1024), 0, nullptr, out, iters, 1.0001f, 0.5f);
            (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        rates[v] = (double)cu * 16 /* wavefronts */ * iters * 16.0 * (v == 0 ? 1 : 4) / (ms * 1e-3);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
    return launch_status();
}
