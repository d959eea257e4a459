// Interaction-graph kernels for gfx950 (replaces /root/reference/src/interaction_graph.h).
//
// Layout/algorithm (MI355X-first, not the reference's edge-list design):
//   * Cached Verlet lists are stored per ROW (ELL, [row][k] ascending) for BOTH sides of an asymmetric
//     graph, so every pass is a pure gather: one wavefront owns one row, its 64 lanes stride over the row's
//     cached neighbours (coalesced index loads), evaluate the pair functor, and reduce with wavefront
//     shuffles.  No edge_value / edge_deriv / edge_sensitivity arrays, no atomics, no scatter.
//   * Pair gradients are RE-EVALUATED in the backward pass instead of being stored (12-13 floats per edge in
//     the reference, interaction_graph.h:294-296): ~150 flop against 100+ bytes of HBM traffic per edge.
//   * `dist2 < cutoff2` is evaluated with explicitly rounded operations (device_math.h dist2_exact) so
//     pair-list membership is bit-identical to the CPU oracle on identical coordinates.
#include "device_math.h"
#include "../../include/upside_hip_kernels.h"
#include <cstring>

using namespace up;

#define ST(L) ((hipStream_t)(L)->stream)
#define ROWS_PER_BLOCK 4
#define IG_BLOCK (ROWS_PER_BLOCK * UP_WAVE)
static inline int launch_status() { return (int)hipGetLastError(); }
static inline unsigned rows_grid(int n_rows) { return n_rows > 0 ? (unsigned)((n_rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK) : 1u; }   // an empty side still needs a non-zero grid

#define C_OUT(c, s)  ((c).out  + (size_t)(s) * (c).n_elem * (c).stride)
#define C_SENS(c, s) ((c).sens + (size_t)(s) * (c).n_elem * (c).stride)

// ------------------------------------------------------------------------------------------------
// pair functors.  x1/x2: element coordinates in registers; p: parameter row of the type pair.
// d1/d2 receive d(value)/d(x1), d(value)/d(x2).

// bead_interaction.h:30-84
__device__ __forceinline__ float quadspline(const upk_igraph_t& G, const float* __restrict__ p, const float* x1, const float* x2,
                                            float* d1, float* d2) {
    const int ka = G.n_knot_angular, k = G.n_knot;
    const float inv_dx = G.inv_dx, inv_dtheta = G.inv_dtheta;
    const f3 displace = mk3(x2[0] - x1[0], x2[1] - x1[1], x2[2] - x1[2]);
    const f3 rvec1 = mk3(x1[3], x1[4], x1[5]), rvec2 = mk3(x2[3], x2[4], x2[5]);
    const float dist2 = mag2(displace), inv_dist = rsqrt_(dist2);
    const float dist_coord = dist2 * (inv_dist * inv_dx);
    const f3 u = inv_dist * displace;
    const float cos1 = dot(rvec1, u), cos2 = -dot(rvec2, u);
    float a1, da1, a2, da2, wide, dwide, narrow, dnarrow;
    deBoor_vd(a1, da1, p, (cos1 + 1.f) * inv_dtheta + 1.f);
    deBoor_vd(a2, da2, p + ka, (cos2 + 1.f) * inv_dtheta + 1.f);
    clamped_deBoor_vd(wide, dwide, p + 2 * ka, dist_coord, k);
    clamped_deBoor_vd(narrow, dnarrow, p + 2 * ka + k, dist_coord, k);
    const float angular_weight = a1 * a2;
    const float radial_deriv = inv_dx * (dwide + angular_weight * dnarrow);
    const float angular_deriv1 = inv_dtheta * da1 * a2 * narrow;
    const float angular_deriv2 = inv_dtheta * a1 * da2 * narrow;
    const f3 rXX = angular_deriv1 * rvec1 - angular_deriv2 * rvec2;
    const f3 deriv_dir = inv_dist * (rXX - dot(u, rXX) * u);
    const f3 dd = radial_deriv * u + deriv_dir;
    d1[0] = -dd.x; d1[1] = -dd.y; d1[2] = -dd.z;
    d1[3] = angular_deriv1 * u.x; d1[4] = angular_deriv1 * u.y; d1[5] = angular_deriv1 * u.z;
    d2[0] = dd.x; d2[1] = dd.y; d2[2] = dd.z;
    d2[3] = -angular_deriv2 * u.x; d2[4] = -angular_deriv2 * u.y; d2[5] = -angular_deriv2 * u.z;
    return wide + angular_weight * narrow;
}

// hbond.cpp:261-276
__device__ __forceinline__ float hbond_coverage_edge(const upk_igraph_t& G, const float* __restrict__ p, const float* x1,
                                                     const float* x2, float* d1, float* d2) {
    const float coverage = quadspline(G, p, x1, x2, d1, d2);
    const float one_m = 1.f - x1[6];
    const float prefactor = one_m * one_m;
#pragma unroll
    for (int c = 0; c < 6; ++c) { d1[c] *= prefactor; d2[c] *= prefactor; }
    d1[6] = -coverage * one_m * 2.f;
    return prefactor * coverage;
}

// environment.cpp:27-60
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

// sidechain_radial.cpp:46-61: clamped cubic spline of the distance; p[0] = 1/dx, 16 coefficients follow
__device__ __forceinline__ float radial_edge(const float* __restrict__ p, const float* x1, const float* x2, float* d1, float* d2) {
    const float inv_dx = p[0];
    const f3 disp = mk3(x1[0] - x2[0], x1[1] - x2[1], x1[2] - x2[2]);
    const float dist2 = mag2(disp), inv_dist = rsqrt_(dist2 + 1e-7f);
    const float dist_coord = dist2 * (inv_dist * inv_dx);
    float v, dv;
    clamped_deBoor_vd(v, dv, p + 1, dist_coord, 16);
    const float k = inv_dist * inv_dx * dv;
    d1[0] = disp.x * k; d1[1] = disp.y * k; d1[2] = disp.z * k;
    d2[0] = -d1[0]; d2[1] = -d1[1]; d2[2] = -d1[2];
    return v;
}

__device__ __forceinline__ float pair_eval(const upk_igraph_t& G, const float* __restrict__ p, const float* x1, const float* x2,
                                           float* d1, float* d2) {
    switch (G.itype) {
        case UPK_IT_RADIAL: case UPK_IT_HBOND_SC_RADIAL: return radial_edge(p, x1, x2, d1, d2);
        case UPK_IT_HBOND_COVERAGE: return hbond_coverage_edge(G, p, x1, x2, d1, d2);
        case UPK_IT_ENVIRONMENT: return environment_edge(p, x1, x2, d1, d2);
        case UPK_IT_PROTEIN_HBOND: return protein_hbond_edge(p, x1, x2, d1, d2);
        default: return quadspline(G, p, x1, x2, d1, d2);
    }
}

__device__ __forceinline__ bool acceptable_id_pair(int itype, int id1, int id2) {
    switch (itype) {
        case UPK_IT_ROTAMER: return ((unsigned)id1 >> 4) != ((unsigned)id2 >> 4);      // bead_interaction.h:195-197
        case UPK_IT_HBOND_COVERAGE:                                                    // hbond.cpp:254-259
        case UPK_IT_RADIAL: case UPK_IT_HBOND_SC_RADIAL:                               // sidechain_radial.cpp:41-44
        case UPK_IT_ENVIRONMENT: return (2 < id1 - id2) || (2 < id2 - id1);            // environment.cpp:22-25
        default: return true;                                                          // hbond.cpp:162-164
    }
}

__device__ __forceinline__ void load_elem(float* x, const upk_coord_t& node, int s, int loc, int dim) {
    const float* p = C_OUT(node, s) + (size_t)loc * node.stride;
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = (c < dim) ? p[c] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// K1: one workgroup per system decides whether the cached lists are still valid (interaction_graph.h:57-90).  A
// system that moved too far is appended to this step's flagged list and its reference positions are refreshed
// right here (x, y, z and the element id in the 4th word), so the rebuild streams one float4 array per side.
__global__ void k_pairlist_check(upk_igraph_t G) {
    __shared__ int moved;
    const int s = blockIdx.y;
    if (threadIdx.x == 0) moved = 0;
    __syncthreads();
    const float lim = sqr(0.5f * (G.cache_cutoff - G.cutoff));
    const int n_tot = G.symmetric ? G.n1 : G.n1 + G.n2;
    bool m = false;
    for (int e = threadIdx.x; e < n_tot; e += blockDim.x) {
        const bool side1 = e < G.n1;
        const int i = side1 ? e : e - G.n1;
        const upk_coord_t& node = side1 ? G.node1 : G.node2;
        const float* x = C_OUT(node, s) + (size_t)(side1 ? G.loc1[i] : G.loc2[i]) * node.stride;
        const float* c = (side1 ? G.cache_pos1 + (size_t)s * G.n1 * 4 : G.cache_pos2 + (size_t)s * G.n2 * 4) + (size_t)i * 4;
        const float dx = x[0] - c[0], dy = x[1] - c[1], dz = x[2] - c[2];
        m |= lim < dx * dx + dy * dy + dz * dz;
    }
    if (m) moved = 1;   // benign race: every writer stores 1
    __syncthreads();
    if (moved) {
        for (int e = threadIdx.x; e < n_tot; e += blockDim.x) {
            const bool side1 = e < G.n1;
            const int i = side1 ? e : e - G.n1;
            const upk_coord_t& node = side1 ? G.node1 : G.node2;
            const float* x = C_OUT(node, s) + (size_t)(side1 ? G.loc1[i] : G.loc2[i]) * node.stride;
            float4* c = (float4*)(side1 ? G.cache_pos1 + (size_t)s * G.n1 * 4 : G.cache_pos2 + (size_t)s * G.n2 * 4) + i;
            *c = make_float4(x[0], x[1], x[2], __int_as_float(side1 ? G.id1[i] : G.id2[i]));
        }
    }
    if (threadIdx.x == 0) {
        G.rebuild_flag[s] = moved;
        int* fl = UPK_FLAG_LIST(G);
        if (moved) fl[1 + atomicAdd(&fl[0], 1)] = s;
        if (s == 0) G.flagged[(size_t)(G.parity ^ 1) * G.flag_stride] = 0;
    }
}
extern "C" int upk_pairlist_check(const upk_launch_t* L, const upk_igraph_t* G) {
    hipLaunchKernelGGL(k_pairlist_check, dim3(1, L->n_system), dim3(256), 0, ST(L), *G);
    return launch_status();
}

// K2: rebuild the row lists of the flagged systems from the refreshed reference positions.  A workgroup stages the
// whole other side (16 bytes per element) in LDS and serves plb_rows() rows; one wavefront per row tests 64
// candidates at a time and compacts hits with a ballot + popcount prefix, so each row comes out in ascending order.
// rows per workgroup: every workgroup stages the whole other side first, so more rows per workgroup amortise that copy
// (128 rows: +0.4 % of the benchmark at 4096 systems; 256 leaves too few workgroups per system: -0.4 %), while a small
// batch wants its few rebuilds spread over many workgroups (64 rows: +1 % at 1 and 64 systems)
static inline int plb_rows(int n_system) { return n_system >= upk_device_cu_count() ? 128 : 64; }
// IT: pair functor (compile time, so the id rule is straight-line code); the inner loop is written without
// short-circuit tests: uniform branches and nested exec masks cost as much as the arithmetic here (measured: the
// branchy form spent ~half of each 64-candidate trip in scalar control flow).  The staged copy is padded to a multiple
// of 64 with far-away sentinels, so the trip needs no bounds test.
template <int IT>
__device__ __forceinline__ bool plb_id_ok(int id_row_side1, int id_other_side2) {
    if (IT == UPK_IT_ROTAMER) return ((unsigned)(id_row_side1 ^ id_other_side2)) > 15u;     // different residue: ids differ above bit 4
    if (IT == UPK_IT_HBOND_COVERAGE || IT == UPK_IT_ENVIRONMENT || IT == UPK_IT_RADIAL || IT == UPK_IT_HBOND_SC_RADIAL) { const int d = id_row_side1 - id_other_side2; return (d > 2) | (d < -2); }
    return true;
}
// node of a side-chain bead from its id (rotamer.cpp:812-816), straight-line
__device__ __forceinline__ int plb_node_of(const upk_igraph_t& G, int id) {
    const int nr = (id >> 4) & 15;
    return (id >> 8) + ((-(int)(nr == 6) & G.mark_start6) | (-(int)(nr == 3) & G.mark_start3));
}
template <bool STAGED, int IT>
__global__ void __launch_bounds__(1024) k_pairlist_build(upk_igraph_t G, int blocks1, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float plb_lds[];
    float4* oth = (float4*)plb_lds;
    constexpr bool SYM = IT == UPK_IT_ROTAMER || IT == UPK_IT_RADIAL;
    const int* fl = UPK_FLAG_LIST(G);
    const int n_flagged = fl[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_wave = blockDim.x >> 6;
    const bool side1 = (int)blockIdx.x < blocks1;             // workgroups [0, blocks1) serve the side-1 rows
    const int rb = side1 ? blockIdx.x : blockIdx.x - blocks1;
    const int n_my = side1 ? G.n1 : G.n2, n_other = side1 ? G.n2 : G.n1;
    const int n_pad = (n_other + 63) & ~63;
    const int cap = side1 ? G.cap1 : G.cap2;
    const float cut2 = G.cache_cutoff * G.cache_cutoff;
    for (int fi = blockIdx.y; fi < n_flagged; fi += gridDim.y) {
        const int s = fl[1 + fi];
        const float4* src = (const float4*)((side1 ? G.cache_pos2 : G.cache_pos1) + (size_t)s * n_other * 4);
        const float4* mine = (const float4*)((side1 ? G.cache_pos1 : G.cache_pos2) + (size_t)s * n_my * 4);
        if (STAGED) {
            __syncthreads();                                   // the previous system's copy is no longer read
            for (int j = threadIdx.x; j < n_pad; j += blockDim.x) oth[j] = j < n_other ? src[j] : make_float4(1e18f, 1e18f, 1e18f, 0.f);
            __syncthreads();
        }
        for (int r = wave; r < rows_per_wg; r += n_wave) {
            const int i = rb * rows_per_wg + r;
            if (i >= n_my) break;
            const float4 x = mine[i];
            const int my_id = __float_as_int(x.w);
            int* nbr = (side1 ? G.nbr1 + (size_t)s * G.n1 * G.cap1 : G.nbr2 + (size_t)s * G.n2 * G.cap2) + (size_t)i * cap;
            int count = 0;
            const int my_node = SYM ? plb_node_of(G, my_id) : 0;
            for (int j0 = 0; j0 < n_pad; j0 += 64) {
                const int j = j0 + lane;
                float4 y;
                if (STAGED) y = oth[j];
                else y = j < n_other ? src[j] : make_float4(1e18f, 1e18f, 1e18f, 0.f);
                const float d2 = dist2_exact(x.x, x.y, x.z, y.x, y.y, y.z);
                const int oid = __float_as_int(y.w);
                bool hit = (d2 < cut2) & (side1 ? plb_id_ok<IT>(my_id, oid) : plb_id_ok<IT>(oid, my_id));
                if (SYM) hit = hit & (j != i);
                const unsigned long long b = __ballot(hit);
                const int pos = count + __popcll(b & ((1ull << lane) - 1ull));
                if (hit & (pos < cap)) nbr[pos] = j;
                count += __popcll(b);
                if (SYM && G.mark_table) {   // residue pairs owning a cached bead pair (rotamer slots); benign race.  Beads of one
                                             // residue are adjacent: only the first hit lane of each residue run stores
                    const bool up = hit & (j > i);
                    const int key = up ? plb_node_of(G, oid) : -1;                 // -1: no mark from this lane
                    const int key_prev = __builtin_amdgcn_update_dpp(-1, key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    if (up & (key_prev != key)) {
                        unsigned char* mt = G.mark_table + (size_t)s * G.mark_stride;
                        mt[my_node * G.mark_n + key] = 1; mt[key * G.mark_n + my_node] = 1;
                    }
                }
            }
            if (lane == 0) {
                (side1 ? G.cnt1 + (size_t)s * G.n1 : G.cnt2 + (size_t)s * G.n2)[i] = count < cap ? count : cap;
                if (count > cap) *G.error_flag = 1;
            }
        }
    }
}
template <int IT>
static void plb_launch(const upk_launch_t* L, const upk_igraph_t* G, dim3 grid, size_t lds, bool staged, int blocks1) {
    const int rows = plb_rows(L->n_system);
    if (staged) hipLaunchKernelGGL((k_pairlist_build<true, IT>), grid, dim3(1024), lds, ST(L), *G, blocks1, rows);
    else hipLaunchKernelGGL((k_pairlist_build<false, IT>), grid, dim3(1024), 0, ST(L), *G, blocks1, rows);
}
extern "C" int upk_pairlist_build(const upk_launch_t* L, const upk_igraph_t* G) {
    const int rows = plb_rows(L->n_system);
    const int blocks1 = (G->n1 + rows - 1) / rows;
    const int blocks2 = G->symmetric ? 0 : (G->n2 + rows - 1) / rows;
    const int n_max = G->n1 > G->n2 ? G->n1 : G->n2;
    const size_t lds = (size_t)((n_max + 63) & ~63) * 16;
    const dim3 grid(blocks1 + blocks2, UPK_FLAG_GRID(L->n_system));
    static int force_unstaged = -1;   // UPSIDE_HIP_PLB_UNSTAGED=1 exercises the path of systems whose elements do not fit LDS
    if (force_unstaged < 0) { const char* e = getenv("UPSIDE_HIP_PLB_UNSTAGED"); force_unstaged = (e && atoi(e)) ? 1 : 0; }
    const bool staged = lds <= 150 * 1024 && !force_unstaged;
    switch (G->itype) {
        case UPK_IT_ROTAMER: plb_launch<UPK_IT_ROTAMER>(L, G, grid, lds, staged, blocks1); break;
        case UPK_IT_HBOND_COVERAGE: plb_launch<UPK_IT_HBOND_COVERAGE>(L, G, grid, lds, staged, blocks1); break;
        case UPK_IT_ENVIRONMENT: plb_launch<UPK_IT_ENVIRONMENT>(L, G, grid, lds, staged, blocks1); break;
        case UPK_IT_RADIAL: plb_launch<UPK_IT_RADIAL>(L, G, grid, lds, staged, blocks1); break;
        case UPK_IT_HBOND_SC_RADIAL: plb_launch<UPK_IT_HBOND_SC_RADIAL>(L, G, grid, lds, staged, blocks1); break;
        default: plb_launch<UPK_IT_PROTEIN_HBOND>(L, G, grid, lds, staged, blocks1); break;
    }
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// K3 forward: row sums of the pair value
__global__ void k_igraph_rowsum(upk_igraph_t G, int side, float* __restrict__ out, long out_sys_stride, int out_stride, int out_comp,
                                int out_row0, float* __restrict__ own_grad) {
    const int s = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int n_rows = side == 1 ? G.n1 : G.n2;
    const float cut2 = G.cutoff * G.cutoff;
    for (int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); row < n_rows; row += gridDim.x * ROWS_PER_BLOCK) {
        const int cap = side == 1 ? G.cap1 : G.cap2;
        const int* nbr = (side == 1 ? G.nbr1 + (size_t)s * G.n1 * G.cap1 : G.nbr2 + (size_t)s * G.n2 * G.cap2) + (size_t)row * cap;
        const int cnt = (side == 1 ? G.cnt1 + (size_t)s * G.n1 : G.cnt2 + (size_t)s * G.n2)[row];
        float xr[8];
        if (side == 1) load_elem(xr, G.node1, s, G.loc1[row], G.dim1); else load_elem(xr, G.node2, s, G.loc2[row], G.dim2);
        const int tr = side == 1 ? G.type1[row] : G.type2[row];
        float acc = 0.f, og[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) og[c] = 0.f;
        for (int k = lane; k < cnt; k += 64) {
            const int j = nbr[k];
            float xo[8], d1[8], d2[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) { d1[c] = 0.f; d2[c] = 0.f; }
            if (side == 1) load_elem(xo, G.node2, s, G.loc2[j], G.dim2); else load_elem(xo, G.node1, s, G.loc1[j], G.dim1);
            if (!(dist2_exact(xr[0], xr[1], xr[2], xo[0], xo[1], xo[2]) < cut2)) continue;
            const int t1 = side == 1 ? tr : G.type1[j], t2 = side == 1 ? G.type2[j] : tr;
            const float* p = G.param + (size_t)(t1 * G.n_type2 + t2) * G.n_param;
            acc += side == 1 ? pair_eval(G, p, xr, xo, d1, d2) : pair_eval(G, p, xo, xr, d1, d2);
#pragma unroll
            for (int c = 0; c < 8; ++c) og[c] += side == 1 ? d1[c] : d2[c];
        }
        acc = wave_sum(acc);
        if (lane == 0) out[(size_t)s * out_sys_stride + (size_t)(out_row0 + row) * out_stride + out_comp] = acc;
        if (own_grad) {
#pragma unroll
            for (int c = 0; c < 8; ++c) og[c] = wave_sum(og[c]);
            if (lane == 0) {
                float* o = own_grad + ((size_t)s * n_rows + row) * 8;
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = og[c];
            }
        }
    }
}
static int igraph_rowsum_v1(const upk_launch_t* L, const upk_igraph_t* G, int side, float* out, long out_sys_stride,
                                 int out_stride, int out_comp, int out_row0, float* own_grad) {
    const int n_rows = side == 1 ? G->n1 : G->n2;
    hipLaunchKernelGGL(k_igraph_rowsum, dim3(rows_grid(n_rows), L->n_system), dim3(IG_BLOCK), 0, ST(L), *G, side,
                       out, out_sys_stride, out_stride, out_comp, out_row0, own_grad);
    return launch_status();
}

// K8 backward: per-row gather of sens(pair) * d(value)/d(row coordinates)
__global__ void k_igraph_grad(upk_igraph_t G, int side, int sens_mode, const float* __restrict__ sens1, const float* __restrict__ sens2,
                              long sens_sys_stride, int sens_stride) {
    const int s = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int n_rows = side == 1 ? G.n1 : G.n2;
    const int dim_row = side == 1 ? G.dim1 : G.dim2;
    const float cut2 = G.cutoff * G.cutoff;
    const float* S1 = sens1 ? sens1 + (size_t)s * sens_sys_stride : nullptr;
    const float* S2 = sens2 ? sens2 + (size_t)s * sens_sys_stride : nullptr;
    for (int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); row < n_rows; row += gridDim.x * ROWS_PER_BLOCK) {
        const int cap = side == 1 ? G.cap1 : G.cap2;
        const int* nbr = (side == 1 ? G.nbr1 + (size_t)s * G.n1 * G.cap1 : G.nbr2 + (size_t)s * G.n2 * G.cap2) + (size_t)row * cap;
        const int cnt = (side == 1 ? G.cnt1 + (size_t)s * G.n1 : G.cnt2 + (size_t)s * G.n2)[row];
        float xr[8];
        if (side == 1) load_elem(xr, G.node1, s, G.loc1[row], G.dim1); else load_elem(xr, G.node2, s, G.loc2[row], G.dim2);
        const int tr = side == 1 ? G.type1[row] : G.type2[row];
        float srow = 0.f;
        if (sens_mode == 1 && side == 1) srow = S1[(size_t)row * sens_stride];
        if (sens_mode == 2 && side == 2) srow = S2[(size_t)row * sens_stride];
        if (sens_mode == 3) srow = side == 1 ? S1[(size_t)row * sens_stride] : S2[(size_t)row * sens_stride];
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
        for (int k = lane; k < cnt; k += 64) {
            const int j = nbr[k];
            float xo[8], d1[8], d2[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) { d1[c] = 0.f; d2[c] = 0.f; }
            if (side == 1) load_elem(xo, G.node2, s, G.loc2[j], G.dim2); else load_elem(xo, G.node1, s, G.loc1[j], G.dim1);
            if (!(dist2_exact(xr[0], xr[1], xr[2], xo[0], xo[1], xo[2]) < cut2)) continue;
            const int t1 = side == 1 ? tr : G.type1[j], t2 = side == 1 ? G.type2[j] : tr;
            const float* p = G.param + (size_t)(t1 * G.n_type2 + t2) * G.n_param;
            if (side == 1) pair_eval(G, p, xr, xo, d1, d2); else pair_eval(G, p, xo, xr, d1, d2);
            float ps;
            if (sens_mode == 0) ps = 1.f;
            else if (sens_mode == 1) ps = side == 1 ? srow : S1[(size_t)j * sens_stride];
            else if (sens_mode == 2) ps = side == 2 ? srow : S2[(size_t)j * sens_stride];
            else ps = srow + (side == 1 ? S2[(size_t)j * sens_stride] : S1[(size_t)j * sens_stride]);
            const float* dr = side == 1 ? d1 : d2;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += ps * dr[c];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = wave_sum(acc[c]);
        if (lane == 0) {
            const upk_coord_t& node = side == 1 ? G.node1 : G.node2;
            float* t = C_SENS(node, s) + (size_t)(side == 1 ? G.loc1[row] : G.loc2[row]) * node.stride;
#pragma unroll
            for (int c = 0; c < 8; ++c) if (c < dim_row) t[c] += acc[c];
        }
    }
}
static int igraph_grad_v1(const upk_launch_t* L, const upk_igraph_t* G, int side, int sens_mode, const float* sens1,
                               const float* sens2, long sens_sys_stride, int sens_stride) {
    const int n_rows = side == 1 ? G->n1 : G->n2;
    hipLaunchKernelGGL(k_igraph_grad, dim3(rows_grid(n_rows), L->n_system), dim3(IG_BLOCK), 0, ST(L), *G, side,
                       sens_mode, sens1, sens2, sens_sys_stride, sens_stride);
    return launch_status();
}


// ================================================================================================
// LDS-staged kernels (the normal path): see igraph_device.h for the decomposition
#include "igraph_device.h"

__device__ __forceinline__ QuadShape quad_shape(const upk_igraph_t& G) { QuadShape Q; Q.ka = G.n_knot_angular; Q.k = G.n_knot; Q.inv_dx = G.inv_dx; Q.inv_dtheta = G.inv_dtheta; return Q; }

// value and (optionally) the derivative w.r.t. the ROW element; x1 is always the side-1 element.
// ROW_SIDE: 1 or 2.  GRAD: derivative wanted.  d has 8 entries.
template <int IT, int ROW_SIDE, bool GRAD>
__device__ __forceinline__ float pair_eval2(const upk_igraph_t& G, const QuadShape& Q, const float* tab, int t1, int t2,
                                            const float* x1, const float* x2, float* d) {
    const float* p = tab + (t1 * G.n_type2 + t2) * G.n_param;
    if (IT == UPK_IT_HBOND_COVERAGE) {
        const float coverage = quadspline2<GRAD ? ROW_SIDE : 0>(Q, p, x1, x2, d);
        const float one_m = 1.f - x1[6], prefactor = one_m * one_m;
        if (GRAD) {
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] *= prefactor;
            if (ROW_SIDE == 1) d[6] = -coverage * one_m * 2.f;
        }
        return prefactor * coverage;
    } else if (IT == UPK_IT_ENVIRONMENT) {
        float d1[8], d2[8];
        const float v = environment_edge(p, x1, x2, d1, d2);
        if (GRAD) {
#pragma unroll
            for (int c = 0; c < 8; ++c) d[c] = ROW_SIDE == 1 ? (c < 6 ? d1[c] : 0.f) : (c < 4 ? d2[c] : 0.f);
        }
        return v;
    } else {
        float d1[8], d2[8];
        const float v = protein_hbond_edge(p, x1, x2, d1, d2);
        if (GRAD) {
#pragma unroll
            for (int c = 0; c < 8; ++c) d[c] = c < 6 ? (ROW_SIDE == 1 ? d1[c] : d2[c]) : 0.f;
        }
        return v;
    }
}

struct Ig2Args {
    float* out; long out_sys_stride; int out_stride, out_comp, out_row0;      // rowsum
    float* own_grad;                                                          // rowsum: [S][n_rows][8] sum of d(value)/d(row element)
    int sens_mode; const float* sens1; const float* sens2; long sens_sys_stride; int sens_stride;   // grad
    int tab_floats, chunk_rows;
};

// MODE 0: row sums of the value; 1: value and the UNWEIGHTED sum of d(value)/d(row element) (so that a backward
// pass whose pair sensitivity depends on the row element only is a per-element product, upk_igraph_apply_own_grad);
// 2: sum of sens(pair) * d(value)/d(row element)
template <int IT, int ROW_SIDE, int MODE>
__global__ void __launch_bounds__(1024) k_ig2(upk_igraph_t G, Ig2Args A) {
    constexpr bool GRAD = MODE == 2;
    constexpr bool WANT_D = MODE != 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_wave = blockDim.x >> 6;
    float* tab = lds;
    float* c1 = lds + ((A.tab_floats + 3) & ~3);
    float* c2 = c1 + G.n1 * 8;
    int* q = (int*)(c2 + G.n2 * 8) + wave * DR_WAVE_LDS; float* acc = (float*)(q + DR_QUEUE);
    int* chunk_counter = (int*)(c2 + G.n2 * 8) + n_wave * DR_WAVE_LDS;
    if (threadIdx.x == 0) *chunk_counter = 0;
    const float* S1 = A.sens1 ? A.sens1 + (size_t)s * A.sens_sys_stride : nullptr;
    const float* S2 = A.sens2 ? A.sens2 + (size_t)s * A.sens_sys_stride : nullptr;
    stage_table(tab, G.param, A.tab_floats);
    // rows: [0,dim) coordinates, [6] per-element pair sensitivity (sides with dim <= 6), [7] element type
    stage_rows(c1, G.node1, s, G.loc1, G.n1, G.dim1, G.type1, nullptr, (GRAD && G.dim1 <= 6) ? S1 : nullptr, A.sens_stride);
    stage_rows(c2, G.node2, s, G.loc2, G.n2, G.dim2, G.type2, nullptr, (GRAD && G.dim2 <= 6) ? S2 : nullptr, A.sens_stride);
    __syncthreads();
    const QuadShape Q = quad_shape(G);
    const int n_rows = ROW_SIDE == 1 ? G.n1 : G.n2;
    const float cut2 = G.cutoff * G.cutoff;
    const float* crow = ROW_SIDE == 1 ? c1 : c2;
    const float* coth = ROW_SIDE == 1 ? c2 : c1;
    const int cap = ROW_SIDE == 1 ? G.cap1 : G.cap2;
    const int* nbr_base = ROW_SIDE == 1 ? G.nbr1 + (size_t)s * G.n1 * G.cap1 : G.nbr2 + (size_t)s * G.n2 * G.cap2;
    const int* cnt_arr = ROW_SIDE == 1 ? G.cnt1 + (size_t)s * G.n1 : G.cnt2 + (size_t)s * G.n2;
    // pair sensitivity = (row part) + (other part); a part is 0 when that side does not contribute
    const bool row_has = GRAD && ((A.sens_mode == 3) || (A.sens_mode == ROW_SIDE));
    const bool oth_has = GRAD && ((A.sens_mode == 3) || (A.sens_mode == 3 - ROW_SIDE));
    const int dim_row = ROW_SIDE == 1 ? G.dim1 : G.dim2;
    const upk_coord_t& row_node = ROW_SIDE == 1 ? G.node1 : G.node2;
    const int* row_loc = ROW_SIDE == 1 ? G.loc1 : G.loc2;
    int g0, g1;
    const int chunk = A.chunk_rows;
    workgroup_row_range(n_rows, chunk, g0, g1);
    for (;;) {
        const int cb = g0 + next_chunk(chunk_counter, lane) * chunk;
        if (cb >= g1) break;
        const int ce = cb + chunk < g1 ? cb + chunk : g1;
        for (int t = lane; t < DR_CHUNK * 8; t += 64) acc[t] = 0.f;
        wave_lds_fence();
        dense_row_loop(cb, ce, cnt_arr, nbr_base, cap, lane, q,
            [&](int row, float* x) { const float* p = crow + row * 8; x[0] = p[0]; x[1] = p[1]; x[2] = p[2]; },
            [&](const float* x, int, int, int j, int& pay) {
                const float* y = coth + j * 8;
                pay = j;
                return dist2_exact(x[0], x[1], x[2], y[0], y[1], y[2]) < cut2;
            },
            [&](int rl, int j, bool valid) {
                float v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = 0.f;
                if (valid) {
                    float xr[8], xo[8], d[8];
                    const float* pr = crow + (cb + rl) * 8; const float* po = coth + j * 8;
                    const float4 rlo = *(const float4*)pr, rhi = *(const float4*)(pr + 4), lo = *(const float4*)po, hi = *(const float4*)(po + 4);
                    xr[0] = rlo.x; xr[1] = rlo.y; xr[2] = rlo.z; xr[3] = rlo.w; xr[4] = rhi.x; xr[5] = rhi.y; xr[6] = rhi.z; xr[7] = rhi.w;
                    xo[0] = lo.x; xo[1] = lo.y; xo[2] = lo.z; xo[3] = lo.w; xo[4] = hi.x; xo[5] = hi.y; xo[6] = hi.z; xo[7] = hi.w;
#pragma unroll
                    for (int c = 0; c < 8; ++c) d[c] = 0.f;
                    const int tr = __float_as_int(xr[7]), to = __float_as_int(xo[7]);
                    const float val = ROW_SIDE == 1 ? pair_eval2<IT, 1, WANT_D>(G, Q, tab, tr, to, xr, xo, d)
                                                    : pair_eval2<IT, 2, WANT_D>(G, Q, tab, to, tr, xo, xr, d);
                    if (MODE == 0) v[0] = val;
                    if (MODE == 1) {
#pragma unroll
                        for (int c = 0; c < 7; ++c) v[c] = d[c];
                        v[7] = val;
                    }
                    if (MODE == 2) {
                        const float ps = (row_has ? xr[6] : 0.f) + (oth_has ? xo[6] : 0.f);
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] = ps * d[c];
                    }
                }
                seg_accumulate<MODE == 0 ? 1 : 8>(acc, rl, valid, v, lane);
            });
        wave_lds_fence();
        // flush the chunk: lane = (row, component)
        for (int t = lane; t < (ce - cb) * 8; t += 64) {
            const int row = cb + (t >> 3), c = t & 7;
            const float val = acc[t];
            float* out_p = A.out + (size_t)s * A.out_sys_stride + (size_t)(A.out_row0 + row) * A.out_stride + A.out_comp;
            if (MODE == 0) { if (c == 0) *out_p = val; }
            else if (MODE == 1) { if (c == 7) *out_p = val; else A.own_grad[((size_t)s * n_rows + row) * 8 + c] = val; }
            else if (c < dim_row) C_SENS(row_node, s)[(size_t)row_loc[row] * row_node.stride + c] += val;
        }
        wave_lds_fence();
    }
}

static bool ig2_geometry(const upk_launch_t* L, const upk_igraph_t* G, int n_rows, int& tab_floats, int& chunk_rows, size_t& lds_bytes, dim3& grid, dim3& block) {
    tab_floats = G->n_type1 * G->n_type2 * G->n_param;
    chunk_rows = dr_chunk_rows(L->n_system, n_rows);
    // a small graph (a few hundred rows) cannot feed 16 waves with several chunks each: smaller workgroups, more of
    // them per CU (they overlap each other's staging and list latency)
    const int n_chunk = (n_rows + chunk_rows - 1) / chunk_rows;
    int waves = n_chunk / 4;
    waves = waves < 4 ? 4 : (waves > 16 ? 16 : waves);
    lds_bytes = ((size_t)((tab_floats + 3) & ~3) + (size_t)(G->n1 + G->n2) * 8 + (size_t)waves * DR_WAVE_LDS + 4) * sizeof(float);
    static int force_unstaged = -1;   // UPSIDE_HIP_IG_UNSTAGED=1 exercises the path taken by systems too large for LDS staging
    if (force_unstaged < 0) { const char* e = getenv("UPSIDE_HIP_IG_UNSTAGED"); force_unstaged = (e && atoi(e)) ? 1 : 0; }
    if (lds_bytes > 158 * 1024 || force_unstaged) return false;
    int bps = (ig_target_wgs() + L->n_system - 1) / L->n_system;   // workgroups in flight across systems
    const int max_bps = (n_rows + waves * chunk_rows - 1) / (waves * chunk_rows);   // at least one chunk per wave
    if (bps > max_bps) bps = max_bps;
    if (bps < 1) bps = 1;
    grid = dim3(bps, L->n_system); block = dim3(waves * 64);
    return true;
}

template <int IT, int SIDE>
static int ig2_launch(const upk_launch_t* L, const upk_igraph_t* G, int mode, const Ig2Args& A0) {
    Ig2Args A = A0; size_t lds; dim3 grid, block;
    if (!ig2_geometry(L, G, SIDE == 1 ? G->n1 : G->n2, A.tab_floats, A.chunk_rows, lds, grid, block)) return -1;
    if (mode == 0) hipLaunchKernelGGL((k_ig2<IT, SIDE, 0>), grid, block, lds, ST(L), *G, A);
    else if (mode == 1) hipLaunchKernelGGL((k_ig2<IT, SIDE, 1>), grid, block, lds, ST(L), *G, A);
    else hipLaunchKernelGGL((k_ig2<IT, SIDE, 2>), grid, block, lds, ST(L), *G, A);
    return launch_status();
}
static int ig2_dispatch(const upk_launch_t* L, const upk_igraph_t* G, int side, int mode, const Ig2Args& A) {
    switch (G->itype) {
        case UPK_IT_HBOND_COVERAGE: return side == 1 ? ig2_launch<UPK_IT_HBOND_COVERAGE, 1>(L, G, mode, A) : ig2_launch<UPK_IT_HBOND_COVERAGE, 2>(L, G, mode, A);
        case UPK_IT_ENVIRONMENT: return side == 1 ? ig2_launch<UPK_IT_ENVIRONMENT, 1>(L, G, mode, A) : ig2_launch<UPK_IT_ENVIRONMENT, 2>(L, G, mode, A);
        case UPK_IT_PROTEIN_HBOND: return side == 1 ? ig2_launch<UPK_IT_PROTEIN_HBOND, 1>(L, G, mode, A) : ig2_launch<UPK_IT_PROTEIN_HBOND, 2>(L, G, mode, A);
        default: return -1;
    }
}

extern "C" int upk_igraph_rowsum(const upk_launch_t* L, const upk_igraph_t* G, int side, float* out, long out_sys_stride,
                                 int out_stride, int out_comp, int out_row0, float* own_grad) {
    Ig2Args A; memset(&A, 0, sizeof(A));
    A.out = out; A.out_sys_stride = out_sys_stride; A.out_stride = out_stride; A.out_comp = out_comp; A.out_row0 = out_row0;
    A.own_grad = own_grad;
    const int r = ig2_dispatch(L, G, side, own_grad ? 1 : 0, A);
    if (r >= 0) return r;
    return igraph_rowsum_v1(L, G, side, out, out_sys_stride, out_stride, out_comp, out_row0, own_grad);   // system too large for LDS staging
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
    const int n_rows = side == 1 ? G->n1 : G->n2;
    hipLaunchKernelGGL(k_igraph_apply_own_grad, dim3((n_rows * 8 + 255) / 256, L->n_system), dim3(256), 0, ST(L), *G, side, own_grad,
                       sens, sens_sys_stride, sens_stride);
    return launch_status();
}
extern "C" int upk_igraph_grad(const upk_launch_t* L, const upk_igraph_t* G, int side, int sens_mode, const float* sens1,
                               const float* sens2, long sens_sys_stride, int sens_stride) {
    Ig2Args A; memset(&A, 0, sizeof(A));
    A.sens_mode = sens_mode; A.sens1 = sens1; A.sens2 = sens2; A.sens_sys_stride = sens_sys_stride; A.sens_stride = sens_stride;
    const int r = ig2_dispatch(L, G, side, 2, A);
    if (r >= 0) return r;
    return igraph_grad_v1(L, G, side, sens_mode, sens1, sens2, sens_sys_stride, sens_stride);
}

// parity / diagnostics: which cached neighbours of the side-1 rows are in range this step
__global__ void k_igraph_inrange(upk_igraph_t G, unsigned char* __restrict__ flags) {
    const int s = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const float cut2 = G.cutoff * G.cutoff;
    for (int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); row < G.n1; row += gridDim.x * ROWS_PER_BLOCK) {
        const int* nbr = G.nbr1 + ((size_t)s * G.n1 + row) * G.cap1;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        const float* x = C_OUT(G.node1, s) + (size_t)G.loc1[row] * G.node1.stride;
        unsigned char* f = flags + ((size_t)s * G.n1 + row) * G.cap1;
        for (int k = lane; k < cnt; k += 64) {
            const float* y = C_OUT(G.node2, s) + (size_t)G.loc2[nbr[k]] * G.node2.stride;
            f[k] = dist2_exact(x[0], x[1], x[2], y[0], y[1], y[2]) < cut2 ? 1 : 0;
        }
    }
}
extern "C" int upk_igraph_inrange(const upk_launch_t* L, const upk_igraph_t* G, unsigned char* flags) {
    hipLaunchKernelGGL(k_igraph_inrange, dim3(rows_grid(G->n1), L->n_system), dim3(IG_BLOCK), 0, ST(L), *G, flags);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// Parameter derivative of one system's pair potential (interaction_graph.h:404-416, 497-503, 537-543): every
// in-range pair adds  pair_sensitivity * d(pair value)/d(interaction_param[type1][type2][:])  to `table`
// ([n_type1][n_type2][n_param], zeroed by the caller).  sens_mode as in k_igraph_grad.  Off the MD path.
__global__ void k_igraph_param_deriv(upk_igraph_t G, int s, int sens_mode, const float* __restrict__ sens1, const float* __restrict__ sens2,
                                     long sens_sys_stride, int sens_stride, float* __restrict__ table) {
    const int lane = threadIdx.x & 63;
    const float cut2 = G.cutoff * G.cutoff;
    const float* S1 = sens1 ? sens1 + (size_t)s * sens_sys_stride : nullptr;
    const float* S2 = sens2 ? sens2 + (size_t)s * sens_sys_stride : nullptr;
    const QuadShape Q = quad_shape(G);
    for (int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); row < G.n1; row += gridDim.x * ROWS_PER_BLOCK) {
        const int* nbr = G.nbr1 + ((size_t)s * G.n1 + row) * G.cap1;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        float xr[8];
        load_elem(xr, G.node1, s, G.loc1[row], G.dim1);
        const int t1 = G.type1[row];
        for (int k = lane; k < cnt; k += 64) {
            const int j = nbr[k];
            float xo[8];
            load_elem(xo, G.node2, s, G.loc2[j], G.dim2);
            if (!(dist2_exact(xr[0], xr[1], xr[2], xo[0], xo[1], xo[2]) < cut2)) continue;
            float ps = sens_mode == 0 ? 1.f : 0.f;
            if (sens_mode == 1 || sens_mode == 3) ps += S1[(size_t)row * sens_stride];
            if (sens_mode == 2 || sens_mode == 3) ps += S2[(size_t)j * sens_stride];
            const size_t prow = (size_t)(t1 * G.n_type2 + G.type2[j]) * G.n_param;
            float* out = table + prow;
            if (G.itype == UPK_IT_HBOND_COVERAGE) {                      // hbond.cpp:278-283
                const float one_m = 1.f - xr[6];
                quadspline_param_accum(Q, G.param + prow, xr, xo, ps * (one_m * one_m), out);
            } else if (G.itype == UPK_IT_RADIAL || G.itype == UPK_IT_HBOND_SC_RADIAL) {   // sidechain_radial.cpp:63-77
                if (G.symmetric && j <= row) continue;                                  // each pair once (i1 < i2)
                const float* p = G.param + prow;
                const float dist = sqrtf(sqr(xr[0] - xo[0]) + sqr(xr[1] - xo[1]) + sqr(xr[2] - xo[2]));
                const float x = p[0] * dist;
                float v, dv;
                clamped_deBoor_vd_scalar(v, dv, p + 1, x, 16);
                atomicAdd(out, ps * dv * dist);
                int bin; float w[4];
                if (x <= 1.f) { bin = 0; w[0] = 1.f / 6.f; w[1] = 2.f / 3.f; w[2] = 1.f / 6.f; w[3] = 0.f; }
                else if (x >= 14.f) { bin = 12; w[0] = 0.f; w[1] = 1.f / 6.f; w[2] = 2.f / 3.f; w[3] = 1.f / 6.f; }
                else { const int xb = (int)x; bin = xb - 1; float db[4]; bspline_basis(x - (float)xb, w, db); }
                for (int k = 0; k < 4; ++k) atomicAdd(out + 1 + bin + k, ps * w[k]);
            }   // environment.cpp:62-65: "not implemented" = zeros; protein_hbond has no get_param_deriv in the reference
        }
    }
}
extern "C" int upk_igraph_param_deriv(const upk_launch_t* L, const upk_igraph_t* G, int system, int sens_mode, const float* sens1,
                                      const float* sens2, long sens_sys_stride, int sens_stride, float* table) {
    if (system < 0 || system >= L->n_system) return 9101;
    if (G->itype == UPK_IT_ROTAMER) return 9102;   // upk_rotamer_param_deriv owns the pair sensitivities of that graph
    hipLaunchKernelGGL(k_igraph_param_deriv, dim3(rows_grid(G->n1)), dim3(IG_BLOCK), 0, ST(L), *G, system,
                       sens_mode, sens1, sens2, sens_sys_stride, sens_stride, table);
    return launch_status();
}
