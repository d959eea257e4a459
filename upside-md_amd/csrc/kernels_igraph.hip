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
#include "igraph_device.h"
#include <cstring>

using namespace up;

#define ST(L) ((hipStream_t)(L)->stream)
#define ROWS_PER_BLOCK 4
#define IG_BLOCK (ROWS_PER_BLOCK * UP_WAVE)
#ifndef UPK_LAUNCH_STATUS_DEFINED
#define UPK_LAUNCH_STATUS_DEFINED
static inline int launch_status() { return (int)hipGetLastError(); }
#endif
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
__device__ __forceinline__ void d_pairlist_check(const upk_igraph_t& G, const BX B, float* lds_unused) {
    __shared__ int moved;
    __shared__ float top[2][16];             // per wavefront: the largest and second largest squared displacement
    const int s = B.by;
    if (threadIdx.x == 0) moved = 0;
    __syncthreads();
    const int n_tot = G.symmetric ? G.n1 : G.n1 + G.n2;
    float d1 = 0.f, d2 = 0.f;                // this lane's two largest squared displacements
    for (int e = threadIdx.x; e < n_tot; e += blockDim.x) {
        const bool side1 = e < G.n1;
        const int i = side1 ? e : e - G.n1;
        const upk_coord_t& node = side1 ? G.node1 : G.node2;
        const float* x = C_OUT(node, s) + (size_t)(side1 ? G.loc1[i] : G.loc2[i]) * node.stride;
        const float* c = (side1 ? G.cache_pos1 + (size_t)s * G.n1 * 4 : G.cache_pos2 + (size_t)s * G.n2 * 4) + (size_t)i * 4;
        const float4 xv = *(const float4*)x, cv = *(const float4*)c;     // (rows of a coordinate node are padded to multiples of 4 floats)
        const float x0 = xv.x, x1 = xv.y, x2 = xv.z;
        const float dx = x0 - cv.x, dy = x1 - cv.y, dz = x2 - cv.z;
        const float dd = dx * dx + dy * dy + dz * dz;
        d2 = fmaxf(d2, fminf(d1, dd)); d1 = fmaxf(d1, dd);
        // this step's positions, packed: what upk_pairlist_refine tests against the cutoff (the same bits the pair passes read)
        ((float4*)(side1 ? G.cur_pos1 + (size_t)s * G.n1 * 4 : G.cur_pos2 + (size_t)s * G.n2 * 4))[i] = make_float4(x0, x1, x2, 0.f);
    }
    // A cached list stays valid while no PAIR of elements has approached by more than the margin: |dr_i| + |dr_j| <= margin for
    // the two largest displacements of the system.  (The reference rebuilds when ANY element has moved margin / 2,
    // interaction_graph.h:57-90 -- sufficient, not necessary; the in-range pairs do not depend on when the cache is rebuilt, and
    // one fast-moving bead no longer forces the rebuild that two would.)
    {
        // top two of the wavefront, then of the workgroup (4 wavefronts)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float o1 = __shfl_xor(d1, off, UP_WAVE), o2 = __shfl_xor(d2, off, UP_WAVE);
            const float n1 = fmaxf(d1, o1), n2 = fmaxf(fminf(d1, o1), fmaxf(d2, o2));
            d1 = n1; d2 = n2;
        }
        if ((threadIdx.x & 63) == 0) { top[0][threadIdx.x >> 6] = d1; top[1][threadIdx.x >> 6] = d2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f, b = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
                const float o1 = top[0][w], o2 = top[1][w];
                const float n1 = fmaxf(a, o1), n2 = fmaxf(fminf(a, o1), fmaxf(b, o2));
                a = n1; b = n2;
            }
            // (first build: the reference positions sit at 1e10, a is huge)  sqrt in fp32, compared with a slightly tightened margin
            const float margin = G.cache_cutoff - G.cutoff;
            if (sqrtf(a) + sqrtf(b) > 0.999f * margin) moved = 1;
        }
    }
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
        // side-chain graph: the node x node mark table of a system that rebuilds is cleared here (the list build marks it next;
        // round 4: this was a launch of its own, upk_rotamer_clear_slots)
        if (G.mark_table) {
            uint4* m = (uint4*)(G.mark_table + (size_t)s * G.mark_stride);
            const int n16 = G.mark_stride / 16;
            for (int i = threadIdx.x; i < n16; i += blockDim.x) m[i] = make_uint4(0, 0, 0, 0);
        }
    }
    if (threadIdx.x == 0) {
        G.rebuild_flag[s] = moved;
        int* fl = UPK_FLAG_LIST(G);
        if (moved) fl[1 + atomicAdd(&fl[0], 1)] = s;
        if (s == 0) G.flagged[(size_t)(G.parity ^ 1) * G.flag_stride] = 0;
    }
}
__global__ void k_pairlist_check(upk_igraph_t G)  { d_pairlist_check(G, BX_REAL, nullptr); }
// Workgroup size of the list-upkeep kernels (rebuild test, list build) when they are launched on their own.  A batch that fills the
// device runs them on side streams next to the pair passes of the main stream, whose 1024-lane workgroups hold ~100 registers per lane
// and 55-106 KB of LDS: what a CU has left beside one of those is 4 wavefront slots and ~96 registers per SIMD -- room for workgroups of
// 256 lanes at <= 40 registers, none for one of 1024.  With 256 lanes the vector-bound list build and the memory-bound rebuild test run
// BESIDE the LDS-bound pair passes instead of taking turns with them at workgroup granularity (4096 systems: 214.8 -> 218.2 k system-steps/s,
// three alternating runs; the build alone +0.8 %; the slot-numbering kernel at 320 lanes as well: no further change; 1024 systems +0.9 %,
// 512 x 150 residues +0.4 %, 256 systems -1.4 %: from two systems per CU on).  Smaller batches keep the large workgroups: there a
// system's latency is what a step waits for.
static inline int upkeep_block(int n_system, int large) { return n_system >= 2 * upk_device_cu_count() ? 256 : large; }
extern "C" int upk_pairlist_check(const upk_launch_t* L, const upk_igraph_t* G) {
    if (batch_add(L, BK_CHECK, 1, L->n_system, 0, G, sizeof(*G))) return 0;
    UPK_FLUSH(L);
    // (latency bound -- every element is a chain load position -> load reference -> store packed copy: as many lanes as the
    //  system has elements, up to a full workgroup, so that a lane walks one or two elements instead of seven)
    const int n_tot = G->symmetric ? G->n1 : G->n1 + G->n2;
    const int threads = upkeep_block(L->n_system, n_tot <= 256 ? 256 : (n_tot <= 512 ? 512 : 1024));
    hipLaunchKernelGGL(k_pairlist_check, dim3(1, L->n_system), dim3(threads), 0, ST(L), *G);
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
__device__ __forceinline__ void d_pairlist_build(const upk_igraph_t& G, int blocks1, int rows_per_wg, const BX B, float* plb_lds) {
        float4* oth = (float4*)plb_lds;
    constexpr bool SYM = IT == UPK_IT_ROTAMER || IT == UPK_IT_RADIAL;
    constexpr bool UPPER = IT == UPK_IT_ROTAMER;   // each bead pair once (partner above the row, i1 < i2 as in the reference's edge list): the
                                                   // rotamer passes visit a pair once and give both beads their share
    const int* fl = UPK_FLAG_LIST(G);
    const int n_flagged = fl[0];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_wave = blockDim.x >> 6;
    const bool side1 = (int)B.bx < blocks1;             // workgroups [0, blocks1) serve the side-1 rows
    const int rb = side1 ? B.bx : B.bx - blocks1;
    const int n_my = side1 ? G.n1 : G.n2, n_other = side1 ? G.n2 : G.n1;
    const int n_pad = (n_other + 63) & ~63;
    const int cap = side1 ? G.cap1 : G.cap2;
    const float cut2 = G.cache_cutoff * G.cache_cutoff;
    for (int fi = B.by; fi < n_flagged; fi += B.gy) {
        const int s = fl[1 + fi];
        const float4* src = (const float4*)((side1 ? G.cache_pos2 : G.cache_pos1) + (size_t)s * n_other * 4);
        const float4* mine = (const float4*)((side1 ? G.cache_pos1 : G.cache_pos2) + (size_t)s * n_my * 4);
        if (STAGED) {
            __syncthreads();                                   // the previous system's copy is no longer read
            for (int j = threadIdx.x; j < n_pad; j += blockDim.x) oth[j] = j < n_other ? src[j] : make_float4(1e18f, 1e18f, 1e18f, 0.f);
            __syncthreads();
        }
        float4 x_next = make_float4(0.f, 0.f, 0.f, 0.f);       // the next row's element is fetched while this row is tested
        { const int i0 = rb * rows_per_wg + wave; if (wave < rows_per_wg && i0 < n_my) x_next = mine[i0]; }
        for (int r = wave; r < rows_per_wg; r += n_wave) {
            const int i = rb * rows_per_wg + r;
            if (i >= n_my) break;
            const float4 x = x_next;
            { const int in = i + n_wave; if (r + n_wave < rows_per_wg && in < n_my) x_next = mine[in]; }
            const int my_id = __float_as_int(x.w);
            typedef typename list_word<IT>::type W;      // (igraph_device.h: 16-bit partner indices, 32-bit words in the rotamer graph)
            W* nbr = (side1 ? (W*)G.nbr1 + (size_t)s * G.n1 * G.cap1 : (W*)G.nbr2 + (size_t)s * G.n2 * G.cap2) + (size_t)i * cap;
            int count = 0;
            const int my_node = SYM ? plb_node_of(G, my_id) : 0;
            // (tried in round 3: 128 candidates per trip, two per lane with packed distances -- 1.40 instead of 1.08 ms for the coverage
            //  graphs: most 64-candidate trips leave after one ballot, a 128-candidate trip rarely does)
            for (int j0 = UPPER ? ((i + 1) & ~63) : 0; j0 < n_pad; j0 += 64) {
                const int j = j0 + lane;
                float4 y;
                if (STAGED) y = oth[j];
                else y = j < n_other ? src[j] : make_float4(1e18f, 1e18f, 1e18f, 0.f);
                const float d2 = dist2_exact(x.x, x.y, x.z, y.x, y.y, y.z);
                const int oid = __float_as_int(y.w);
                bool hit = (d2 < cut2) & (side1 ? plb_id_ok<IT>(my_id, oid) : plb_id_ok<IT>(oid, my_id));
                if (UPPER) hit = hit & (j > i);
                else if (SYM) hit = hit & (j != i);
                const unsigned long long b = __ballot(hit);
                if (!b) continue;                                  // (wave-uniform: most trips of a row find nobody within reach)
                const int pos = count + __popcll(b & ((1ull << lane) - 1ull));
                if (hit & (pos < cap)) nbr[pos] = (W)j;
                count += __popcll(b);
                if (SYM && G.mark_table) {   // residue pairs owning a cached bead pair (rotamer slots); benign race.  Beads of one
                                             // residue are adjacent: only the first hit lane of each residue run stores
                    const bool up = hit & (j > i);
                    const int key = up ? plb_node_of(G, oid) : -1;                 // -1: no mark from this lane
                    const int key_prev = __builtin_amdgcn_update_dpp(-1, key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    if (up & (key_prev != key)) {
                        unsigned char* mt = G.mark_table + (size_t)s * G.mark_stride;
                        mt[my_node * G.mark_ld + key] = 1; mt[key * G.mark_ld + my_node] = 1;
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
template <bool STAGED, int IT>
__global__ void __launch_bounds__(1024) k_pairlist_build(upk_igraph_t G, int blocks1, int rows_per_wg)  {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn_[];
    d_pairlist_build<STAGED, IT>(G, blocks1, rows_per_wg, BX_REAL, lds_dyn_);
}
template <int IT>
static void plb_launch(const upk_launch_t* L, const upk_igraph_t* G, dim3 grid, size_t lds, bool staged, int blocks1) {
    const int rows = plb_rows(L->n_system);
    if (staged) hipLaunchKernelGGL((k_pairlist_build<true, IT>), grid, dim3(upkeep_block(L->n_system, 1024)), lds, ST(L), *G, blocks1, rows);
    else hipLaunchKernelGGL((k_pairlist_build<false, IT>), grid, dim3(1024), 0, ST(L), *G, blocks1, rows);
}
extern "C" int upk_pairlist_build(const upk_launch_t* L, const upk_igraph_t* G) { return upk_pairlist_build_sides(L, G, 3); }
extern "C" int upk_pairlist_build_sides(const upk_launch_t* L, const upk_igraph_t* G, int sides) {
    if (!list_words_match(G)) return 9010;
    const int rows = plb_rows(L->n_system);
    const int blocks1 = (sides & 1) ? (G->n1 + rows - 1) / rows : 0;
    const int blocks2 = (G->symmetric || !(sides & 2)) ? 0 : (G->n2 + rows - 1) / rows;
    if (blocks1 + blocks2 == 0) return 0;
    const int n_max = G->n1 > G->n2 ? G->n1 : G->n2;
    const size_t lds = (size_t)((n_max + 63) & ~63) * 16;
    const dim3 grid(blocks1 + blocks2, UPK_FLAG_GRID(L->n_system));
    static int force_unstaged = -1;   // UPSIDE_HIP_PLB_UNSTAGED=1 exercises the path of systems whose elements do not fit LDS
    if (force_unstaged < 0) { const char* e = getenv("UPSIDE_HIP_PLB_UNSTAGED"); force_unstaged = (e && atoi(e)) ? 1 : 0; }
    const bool staged = lds <= 150 * 1024 && !force_unstaged;
    if (staged) {       // merged launch (kernels_batch.h)
        const int bk = G->itype == UPK_IT_ROTAMER ? BK_BUILD_ROT : G->itype == UPK_IT_HBOND_COVERAGE ? BK_BUILD_COV : G->itype == UPK_IT_ENVIRONMENT ? BK_BUILD_ENV
                     : G->itype == UPK_IT_PROTEIN_HBOND ? BK_BUILD_HB : 0;
        if (bk && batch_add(L, bk, (int)grid.x, (int)grid.y, lds, G, sizeof(*G), nullptr, 0, blocks1, rows)) return 0;
    }
    UPK_FLUSH(L);
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
// K2b: this step's in-range pairs (the refine of interaction_graph.h:201-257, once per step and graph side).  A workgroup
// stages the other side's positions (16 bytes per element) in LDS and serves rows_per_wg rows; one wavefront per row tests
// 64 cached neighbours per trip and appends the survivors' list words to the row's hit list in list order (ballot +
// popcount prefix).  Latency bound by design -- it runs on the upkeep streams next to the VALU-bound pair passes -- so a
// wavefront fetches the positions and list lengths of ALL its rows with one load each and keeps the first list words of
// the next PLR_DEPTH row pairs in flight.
#ifndef PLR_BLOCK
#define PLR_BLOCK 256
#endif
#ifndef PLR_ROWS
#define PLR_ROWS 256
#endif
#ifndef PLR_DEPTH
#define PLR_DEPTH 2
#endif
static_assert(PLR_ROWS <= 64 * (PLR_BLOCK / 64), "a wavefront of the refine keeps the positions and list lengths of its rows one per lane: at most 64 rows per wavefront");
// squared distances of two pairs at once, every operation rounded separately (no contraction): the same bits as dist2_exact,
// from v_pk_add_f32 / v_pk_mul_f32 (3 + 3 + 2 packed instructions for the two candidates of a lane)
typedef float plr_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ plr_v2 dist2_exact2(plr_v2 ax, plr_v2 ay, plr_v2 az, plr_v2 bx, plr_v2 by, plr_v2 bz) {
#pragma clang fp contract(off)
    const plr_v2 dx = ax - bx, dy = ay - by, dz = az - bz;
    const plr_v2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const plr_v2 s = xx + yy;
    return s + zz;
}
// TWO rows per wavefront trip (rows 2a and 2a+1 of the wavefront's run: one candidate of each per lane), so that the distance
// arithmetic issues packed and the per-trip bookkeeping (loop control, ballots, list-length broadcasts) is shared: the kernel
// is bound by instruction issue, not by the 4 bytes it reads per cached pair.
template <bool SYM, typename W>
__device__ __forceinline__ void d_pairlist_refine(const upk_igraph_t& G, int side, int rows_per_wg, const BX B, float* plr_lds) {
        float4* oth = (float4*)plr_lds;
    const int s = B.by;
    // (the wavefront index is uniform, which the compiler cannot see: made scalar, row numbers, list lengths and row base addresses
    //  stay in scalar registers and the loop control runs on the scalar unit)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_wave = blockDim.x >> 6;
    const bool rows1 = side == 1;
    const int n_rows = rows1 ? G.n1 : G.n2, n_other = rows1 ? G.n2 : G.n1;
    const int cap = rows1 ? G.cap1 : G.cap2;
    const float4* src = (const float4*)((rows1 ? G.cur_pos2 : G.cur_pos1) + (size_t)s * n_other * 4);
    const float4* mine = (const float4*)((rows1 ? G.cur_pos1 : G.cur_pos2) + (size_t)s * n_rows * 4);
    for (int j = threadIdx.x; j <= n_other; j += blockDim.x) oth[j] = j < n_other ? src[j] : make_float4(1e18f, 1e18f, 1e18f, 0.f);
    __syncthreads();
    // (lanes past the end of their row carry the word n_other, which names the far-away sentinel behind the staged elements: such a
    //  lane fails the distance test by itself and the trip needs no validity masks)
    const int dead = n_other;
    const W* nbr_base = (const W*)(rows1 ? G.nbr1 : G.nbr2) + (size_t)s * n_rows * cap;
    const int* cnt_arr = (rows1 ? G.cnt1 : G.cnt2) + (size_t)s * n_rows;
    W* hit_base = (W*)(rows1 ? G.hit1 : G.hit2) + (size_t)s * n_rows * cap;
    int* hcnt = (rows1 ? G.hcnt1 : G.hcnt2) + (size_t)s * n_rows;
    int* hlo = (SYM && G.hlo1) ? G.hlo1 + (size_t)s * n_rows : nullptr;
    const float cut2 = G.cutoff * G.cutoff;
    const int jmask = G.nbr_j_bits ? (1 << G.nbr_j_bits) - 1 : 0x7fffffff;
    // this wavefront's rows: a contiguous run of at most 64 (lane r holds row r0 + r's position and list length)
    const int per_wave = (rows_per_wg + n_wave - 1) / n_wave;     // <= 64 (launcher)
    const int r0 = B.bx * rows_per_wg + wave * per_wave;
    int r1 = r0 + per_wave; { const int wg_end = (B.bx + 1) * rows_per_wg; if (r1 > wg_end) r1 = wg_end; if (r1 > n_rows) r1 = n_rows; }
    if (r0 >= r1) return;
    const bool have = r0 + lane < r1;
    const float4 my_x = have ? mine[r0 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int my_cnt = have ? cnt_arr[r0 + lane] : 0;
    int my_n = 0, my_lo = 0;                                      // results of row r0 + lane
    auto bcast = [&](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    // The first 64 list words of the next PLR_DEPTH row pairs are in flight while this pair is tested: a wavefront's share of the memory
    // system is what it keeps in flight, and one pair (two 256-byte rows) per wavefront left the refine at 2.4 TB/s of list traffic
    // with neither the vector unit nor the LDS half busy (round 5: two pairs ahead).
    auto fetch_pair = [&](int ra, int& a, int& b) {      // words [0, 64) of rows ra, ra + 1 (dead past the rows or their ends)
        a = dead; b = dead;
        if (ra < r1) {
            const int la = ra - r0;
            const int cA = __builtin_amdgcn_readlane(my_cnt, la & 63), cB = ra + 1 < r1 ? __builtin_amdgcn_readlane(my_cnt, (la + 1) & 63) : 0;
            if (lane < cA) a = nbr_base[(size_t)ra * cap + lane];
            if (lane < cB) b = nbr_base[(size_t)(ra + 1) * cap + lane];
        }
    };
    int qA[PLR_DEPTH], qB[PLR_DEPTH];
#pragma unroll
    for (int d = 0; d < PLR_DEPTH; ++d) fetch_pair(r0 + 2 * d, qA[d], qB[d]);
    for (int ra = r0; ra < r1; ra += 2) {
        const int la = ra - r0, lb = la + 1;
        const bool hasB = ra + 1 < r1;
        const int cntA = __builtin_amdgcn_readlane(my_cnt, la), cntB = hasB ? __builtin_amdgcn_readlane(my_cnt, lb & 63) : 0;
        int wA = qA[0], wB = qB[0];
#pragma unroll
        for (int d = 0; d + 1 < PLR_DEPTH; ++d) { qA[d] = qA[d + 1]; qB[d] = qB[d + 1]; }
        fetch_pair(ra + 2 * PLR_DEPTH, qA[PLR_DEPTH - 1], qB[PLR_DEPTH - 1]);
        plr_v2 xx, xy, xz;
        xx.x = bcast(my_x.x, la); xy.x = bcast(my_x.y, la); xz.x = bcast(my_x.z, la);
        xx.y = bcast(my_x.x, lb & 63); xy.y = bcast(my_x.y, lb & 63); xz.y = bcast(my_x.z, lb & 63);
        const W* nbrA = nbr_base + (size_t)ra * cap; const W* nbrB = nbrA + cap;
        W* outA = hit_base + (size_t)ra * cap; W* outB = outA + cap;
        int na = 0, nb = 0, loa = 0, lob = 0;
        const int cmax = cntA > cntB ? cntA : cntB;
        for (int k0 = 0; k0 < cmax; k0 += 64) {
            const int k = k0 + lane;
            if (k0) { wA = k < cntA ? nbrA[k] : dead; wB = k < cntB ? nbrB[k] : dead; }
            const int jA = wA & jmask, jB = wB & jmask;
            const float4 yA = oth[jA], yB = oth[jB];
            plr_v2 yx, yy, yz; yx.x = yA.x; yx.y = yB.x; yy.x = yA.y; yy.y = yB.y; yz.x = yA.z; yz.y = yB.z;
            const plr_v2 d2 = dist2_exact2(xx, xy, xz, yx, yy, yz);
            const bool hitA = d2.x < cut2, hitB = d2.y < cut2;
            const unsigned long long mA = __builtin_amdgcn_ballot_w64(hitA), mB = __builtin_amdgcn_ballot_w64(hitB);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (hitA) outA[na + __popcll(mA & below)] = (W)wA;
            if (hitB) outB[nb + __popcll(mB & below)] = (W)wB;
            na += __popcll(mA); nb += __popcll(mB);
            if (SYM) { loa += __popcll(__builtin_amdgcn_ballot_w64(hitA && jA < ra)); lob += __popcll(__builtin_amdgcn_ballot_w64(hitB && jB < ra + 1)); }
        }
        if (lane == la) { my_n = na; my_lo = loa; }
        if (hasB && lane == lb) { my_n = nb; my_lo = lob; }
    }
    if (have) { hcnt[r0 + lane] = my_n; if (SYM && hlo) hlo[r0 + lane] = my_lo; }
}
template <bool SYM, typename W>
__global__ void __launch_bounds__(PLR_BLOCK) k_pairlist_refine(upk_igraph_t G, int side, int rows_per_wg)  {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn_[];
    d_pairlist_refine<SYM, W>(G, side, rows_per_wg, BX_REAL, lds_dyn_);
}
// Rows of a handful of cached neighbours (backbone hydrogen bonds: three per donor or acceptor): one LANE per row walks its list --
// the row-pair machinery above costs a fixed ~100 instructions and a dependent chain per pair of rows, 0.19 ms per launch for
// 840 cached pairs per system.  Same hit lists, in the same order.
__device__ __forceinline__ void d_pairlist_refine_short(const upk_igraph_t& G, int side, const BX B, float* lds_unused) {
    const int s = B.by, row = B.bx * blockDim.x + threadIdx.x;
    const bool rows1 = side == 1;
    const int n_rows = rows1 ? G.n1 : G.n2, n_other = rows1 ? G.n2 : G.n1;
    if (row >= n_rows) return;
    const int cap = rows1 ? G.cap1 : G.cap2;
    const float4* src = (const float4*)((rows1 ? G.cur_pos2 : G.cur_pos1) + (size_t)s * n_other * 4);
    const float4 x = ((const float4*)((rows1 ? G.cur_pos1 : G.cur_pos2) + (size_t)s * n_rows * 4))[row];
    typedef list_word<UPK_IT_PROTEIN_HBOND>::type W;
    const W* nbr = (const W*)(rows1 ? G.nbr1 : G.nbr2) + ((size_t)s * n_rows + row) * cap;
    W* hit = (W*)(rows1 ? G.hit1 : G.hit2) + ((size_t)s * n_rows + row) * cap;
    const int cnt = ((rows1 ? G.cnt1 : G.cnt2) + (size_t)s * n_rows)[row];
    const float cut2 = G.cutoff * G.cutoff;
    const int jmask = G.nbr_j_bits ? (1 << G.nbr_j_bits) - 1 : 0x7fffffff;
    int n = 0;
    for (int k0 = 0; k0 < cnt; k0 += 4) {          // four words, then their four positions, in flight together
        int w[4]; float4 y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = nbr[k0 + u < cnt ? k0 + u : k0];
#pragma unroll
        for (int u = 0; u < 4; ++u) y[u] = src[w[u] & jmask];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u < cnt && dist2_exact(x.x, x.y, x.z, y[u].x, y[u].y, y[u].z) < cut2) hit[n++] = (W)w[u];
    }
    ((rows1 ? G.hcnt1 : G.hcnt2) + (size_t)s * n_rows)[row] = n;
}
__global__ void __launch_bounds__(256) k_pairlist_refine_short(upk_igraph_t G, int side)  { d_pairlist_refine_short(G, side, BX_REAL, nullptr); }
extern "C" int upk_pairlist_refine(const upk_launch_t* L, const upk_igraph_t* G, int side) {
    const bool rows1 = side == 1;
    const int n_rows = rows1 ? G->n1 : G->n2, n_other = rows1 ? G->n2 : G->n1;
    if (n_rows < 1) return 0;
    if (G->itype == UPK_IT_PROTEIN_HBOND && !G->symmetric) {
        if (batch_add(L, BK_REFINE_SHORT, (n_rows + 1023) / 1024, L->n_system, 0, G, sizeof(*G), nullptr, 0, side)) return 0;
        UPK_FLUSH(L);
        hipLaunchKernelGGL(k_pairlist_refine_short, dim3((n_rows + 255) / 256, L->n_system), dim3(256), 0, ST(L), *G, side);
        return launch_status();
    }
    if (!list_words_match(G)) return 9010;
    if ((G->symmetric != 0) == (G->word16 != 0)) return 9011;   // the two list forms there are: the rotamer graph's (symmetric, 32-bit words) and everyone else's
    const size_t lds = (size_t)((n_other > 0 ? n_other : 0) + 1) * 16;
    if (lds > 150 * 1024) return 9006;   // (callers fall back to the list-walking kernels long before this)
    // every workgroup stages the other side again: few fat workgroups for a large batch, many small ones for a small one
    int rows_per_wg = L->n_system >= upk_device_cu_count() ? PLR_ROWS : (L->n_system >= 16 ? 64 : 16);
    {   // a side of a few hundred rows (environment graph: 300): smaller workgroups, or one of its two would walk 256 rows as 32
        // dependent row pairs per wavefront while the other has 44 (0.30 -> 0.24 ms)
        if (n_rows <= 512 && rows_per_wg > 128) rows_per_wg = 128;
    }
    {   // merged launch: 1024-lane workgroups, four times the wavefronts -- four times the rows, so that a wavefront keeps its run of row pairs
        const int rows_b = rows_per_wg * 4;
        if (batch_add(L, G->symmetric ? BK_REFINE_SYM : BK_REFINE, (n_rows + rows_b - 1) / rows_b, L->n_system, lds, G, sizeof(*G), nullptr, 0, side, rows_b)) return 0;
    }
    UPK_FLUSH(L);
    const dim3 grid((n_rows + rows_per_wg - 1) / rows_per_wg, L->n_system);
    if (G->symmetric) hipLaunchKernelGGL((k_pairlist_refine<true, int>), grid, dim3(PLR_BLOCK), lds, ST(L), *G, side, rows_per_wg);
    else hipLaunchKernelGGL((k_pairlist_refine<false, unsigned short>), grid, dim3(PLR_BLOCK), lds, ST(L), *G, side, rows_per_wg);

    return launch_status();
}

// K2c: rows of a system sorted by descending hit count (counting sort in LDS, one workgroup per system): the pair passes
// give the 8 lane groups of a wavefront 8 consecutive rows of this order, so they run the same number of trips.  Rows of
// equal length keep no particular order (it only decides which wavefront serves them).  Symmetric graphs get a second order
// by the number of partners ABOVE the row (the pair-energy pass visits each pair once).
#define PLO_BINS 1024
template <typename KeyFn>
__device__ __forceinline__ void order_rows(unsigned short* __restrict__ ord, int n_rows, KeyFn key, int* hist, int* scratch) {
    hist[threadIdx.x] = 0;                                         // blockDim.x == PLO_BINS
    __syncthreads();
    for (int r = threadIdx.x; r < n_rows; r += blockDim.x) atomicAdd(&hist[PLO_BINS - 1 - key(r)], 1);   // bin 0 = the longest rows
    __syncthreads();
    // exclusive scan of the histogram, one bin per thread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = hist[threadIdx.x];
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, UP_WAVE); if (lane >= off) incl += t; }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { int acc = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { const int t = scratch[w]; scratch[w] = acc; acc += t; } }
    __syncthreads();
    hist[threadIdx.x] = scratch[wave] + incl - v;
    __syncthreads();
    for (int r = threadIdx.x; r < n_rows; r += blockDim.x) ord[atomicAdd(&hist[PLO_BINS - 1 - key(r)], 1)] = (unsigned short)r;
    __syncthreads();
}
__device__ __forceinline__ void d_pairlist_order(const upk_igraph_t& G, int side, const BX B, float* lds_unused) {
    __shared__ int hist[PLO_BINS], scratch[16];
    const int s = B.bx;
    const bool rows1 = side == 1;
    const int n_rows = rows1 ? G.n1 : G.n2;
    const int* hcnt = (rows1 ? G.hcnt1 : G.hcnt2) + (size_t)s * n_rows;
    order_rows((rows1 ? G.ord1 : G.ord2) + (size_t)s * n_rows, n_rows,
               [&](int r) { const int c = hcnt[r]; return c < PLO_BINS ? c : PLO_BINS - 1; }, hist, scratch);
    if (G.symmetric && G.hlo1) {
        const int* hlo = G.hlo1 + (size_t)s * n_rows;
        order_rows(G.ord1u + (size_t)s * n_rows, n_rows,
                   [&](int r) { const int c = hcnt[r] - hlo[r]; return c < PLO_BINS ? c : PLO_BINS - 1; }, hist, scratch);
    }
}
__global__ void __launch_bounds__(PLO_BINS) k_pairlist_order(upk_igraph_t G, int side)  { d_pairlist_order(G, side, BX_REAL, nullptr); }
extern "C" int upk_pairlist_order(const upk_launch_t* L, const upk_igraph_t* G, int side) {
    const int n_rows = side == 1 ? G->n1 : G->n2;
    if (n_rows < 1) return 0;
    if (n_rows > 65535) return 9009;   // (16-bit row ids; larger systems take the list-walking kernels, which need no order)
    if (batch_add(L, BK_ORDER, L->n_system, 1, 0, G, sizeof(*G), nullptr, 0, side)) return 0;
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_pairlist_order, dim3(L->n_system), dim3(PLO_BINS), 0, ST(L), *G, side);
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
        const int* lists = side == 1 ? G.nbr1 : G.nbr2;
        const size_t row_at = ((size_t)s * n_rows + row) * cap;         // (in list words: list_word_at)
        const int cnt = (side == 1 ? G.cnt1 + (size_t)s * G.n1 : G.cnt2 + (size_t)s * G.n2)[row];
        float xr[8];
        if (side == 1) load_elem(xr, G.node1, s, G.loc1[row], G.dim1); else load_elem(xr, G.node2, s, G.loc2[row], G.dim2);
        const int tr = side == 1 ? G.type1[row] : G.type2[row];
        float acc = 0.f, og[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) og[c] = 0.f;
        for (int k = lane; k < cnt; k += 64) {
            const int j = list_word_at(lists, row_at + k, G.word16);
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
extern "C" int upk_igraph_rowsum(const upk_launch_t* L, const upk_igraph_t* G, int side, float* out, long out_sys_stride,
                                 int out_stride, int out_comp, int out_row0, float* own_grad) {
    UPK_FLUSH(L);
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
        const int* lists = side == 1 ? G.nbr1 : G.nbr2;
        const size_t row_at = ((size_t)s * n_rows + row) * cap;         // (in list words: list_word_at)
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
            const int j = list_word_at(lists, row_at + k, G.word16);
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
extern "C" int upk_igraph_grad(const upk_launch_t* L, const upk_igraph_t* G, int side, int sens_mode, const float* sens1,
                               const float* sens2, long sens_sys_stride, int sens_stride) {
    UPK_FLUSH(L);
    const int n_rows = side == 1 ? G->n1 : G->n2;
    hipLaunchKernelGGL(k_igraph_grad, dim3(rows_grid(n_rows), L->n_system), dim3(IG_BLOCK), 0, ST(L), *G, side,
                       sens_mode, sens1, sens2, sens_sys_stride, sens_stride);
    return launch_status();
}


// parity / diagnostics: which cached neighbours of the side-1 rows are in range this step
__global__ void k_igraph_inrange(upk_igraph_t G, unsigned char* __restrict__ flags) {
    const int s = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const float cut2 = G.cutoff * G.cutoff;
    for (int row = blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6); row < G.n1; row += gridDim.x * ROWS_PER_BLOCK) {
        const size_t row_at = ((size_t)s * G.n1 + row) * G.cap1;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        const float* x = C_OUT(G.node1, s) + (size_t)G.loc1[row] * G.node1.stride;
        unsigned char* f = flags + ((size_t)s * G.n1 + row) * G.cap1;
        for (int k = lane; k < cnt; k += 64) {
            const int w = list_word_at(G.nbr1, row_at + k, G.word16);
            const int j = G.nbr_j_bits ? (w & ((1 << G.nbr_j_bits) - 1)) : w;
            const float* y = C_OUT(G.node2, s) + (size_t)G.loc2[j] * G.node2.stride;
            f[k] = dist2_exact(x[0], x[1], x[2], y[0], y[1], y[2]) < cut2 ? 1 : 0;
        }
    }
}
extern "C" int upk_igraph_inrange(const upk_launch_t* L, const upk_igraph_t* G, unsigned char* flags) {
    UPK_FLUSH(L);
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
        const size_t row_at = ((size_t)s * G.n1 + row) * G.cap1;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        float xr[8];
        load_elem(xr, G.node1, s, G.loc1[row], G.dim1);
        const int t1 = G.type1[row];
        for (int k = lane; k < cnt; k += 64) {
            const int j = list_word_at(G.nbr1, row_at + k, G.word16);
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
    UPK_FLUSH(L);
    if (system < 0 || system >= L->n_system) return 9101;
    if (G->itype == UPK_IT_ROTAMER) return 9102;   // upk_rotamer_param_deriv owns the pair sensitivities of that graph
    hipLaunchKernelGGL(k_igraph_param_deriv, dim3(rows_grid(G->n1)), dim3(IG_BLOCK), 0, ST(L), *G, system,
                       sens_mode, sens1, sens2, sens_sys_stride, sens_stride, table);
    return launch_status();
}
