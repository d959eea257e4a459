// Rotamer free-energy node on gfx950 (replaces /root/reference/src/rotamer.cpp).
//
// Structure (MI355X-first):
//   * side-chain "nodes" (one per residue with 1/3/6 rotamer states) get global ids sorted by state count
//     (1-state, then 3, then 6), so the reference's canonical edge orientation n_rot1 <= n_rot2
//     (rotamer.cpp:837) is simply a < b;
//   * residue-pair "slots" (the BP edges) are assigned on the device when the bead pair list is rebuilt:
//     a dense node x node table replaces the EdgeLocator hash (rotamer.cpp:134-206) and per-node adjacency
//     lists make the belief update a gather (no in-place multiply through a shared node belief);
//   * bead-pair energies are accumulated straight into the 6x6 slot matrices, belief propagation runs as ONE
//     persistent workgroup per system with the node beliefs in LDS, and the derivative pass re-evaluates the
//     pair gradient per bead row and reduces it with wavefront shuffles.
#include "device_math.h"
#include "../../include/upside_hip_kernels.h"
#include "igraph_device.h"

using namespace up;

#define ST(L) ((hipStream_t)(L)->stream)
#define ROWS_PER_BLOCK 4
#define IG_BLOCK (ROWS_PER_BLOCK * UP_WAVE)
#define BP_BLOCK 1024
static inline int launch_status() { return (int)hipGetLastError(); }
#define C_OUT(c, s)  ((c).out  + (size_t)(s) * (c).n_elem * (c).stride)
#define C_SENS(c, s) ((c).sens + (size_t)(s) * (c).n_elem * (c).stride)

// defined in kernels_igraph.hip; duplicated as a static inline copy would be error-prone, so the quadspline is
// re-declared here through a small header-less contract: same translation unit layout, separate copy.
__device__ __forceinline__ float quadspline_r(const upk_igraph_t& G, const float* __restrict__ p, const float* x1, const float* x2,
                                              float* d1) {
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
    if (d1) {
        const float radial_deriv = inv_dx * (dwide + angular_weight * dnarrow);
        const float angular_deriv1 = inv_dtheta * da1 * a2 * narrow;
        const float angular_deriv2 = inv_dtheta * a1 * da2 * narrow;
        const f3 rXX = angular_deriv1 * rvec1 - angular_deriv2 * rvec2;
        const f3 deriv_dir = inv_dist * (rXX - dot(u, rXX) * u);
        const f3 dd = radial_deriv * u + deriv_dir;
        d1[0] = -dd.x; d1[1] = -dd.y; d1[2] = -dd.z;
        d1[3] = angular_deriv1 * u.x; d1[4] = angular_deriv1 * u.y; d1[5] = angular_deriv1 * u.z;
    }
    return wide + angular_weight * narrow;
}

// slot matrices are stored structure-of-arrays: entry e (= ra*6+rb) of slot sl lives at [e*slot_cap + sl], so that
// consecutive lanes working on consecutive slots read consecutive addresses
#define PIDX(R, sl, e) ((size_t)(e) * (R).slot_cap + (sl))

__device__ __forceinline__ void load6(float* x, const upk_coord_t& node, int s, int loc) {
    const float* p = C_OUT(node, s) + (size_t)loc * node.stride;
#pragma unroll
    for (int c = 0; c < 6; ++c) x[c] = p[c];
}

// ------------------------------------------------------------------------------------------------
// slots + adjacency, one workgroup per flagged system
__global__ void __launch_bounds__(BP_BLOCK) k_rotamer_build_slots(upk_rotamer_t R) {
    const int s = blockIdx.y;
    const upk_igraph_t& G = R.G;
    if (!G.rebuild_flag[s]) return;
    __shared__ int row_count[2048];
    __shared__ int row_start[2048];
    __shared__ int total;
    const int NN = R.n_node;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_wave = blockDim.x >> 6;
    int* slot_of = R.slot_of + (size_t)s * NN * NN;
    for (int i = tid; i < NN * NN; i += blockDim.x) slot_of[i] = -1;
    __syncthreads();
    // mark residue pairs that own at least one cached bead pair
    for (int row = wave; row < G.n1; row += n_wave) {
        const int* nbr = G.nbr1 + ((size_t)s * G.n1 + row) * G.cap1;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        const int a = R.bead_node[row];
        for (int k = lane; k < cnt; k += 64) {
            const int j = nbr[k];
            if (j <= row) continue;
            const int b = R.bead_node[j];
            slot_of[a * NN + b] = -2; slot_of[b * NN + a] = -2;    // benign race: all writers store -2
        }
    }
    __syncthreads();
    // count canonical (a<b) marked pairs per row a
    for (int a = wave; a < NN; a += n_wave) {
        int c = 0;
        for (int b0 = a + 1; b0 < NN; b0 += 64) {
            const int b = b0 + lane;
            c += __popcll(__ballot(b < NN && slot_of[a * NN + b] == -2));
        }
        if (lane == 0) row_count[a] = c;
    }
    __syncthreads();
    if (tid == 0) {   // NN is a few hundred: a serial scan is cheaper than another two barriers
        int acc = 0;
        for (int a = 0; a < NN; ++a) { row_start[a] = acc; acc += row_count[a]; }
        total = acc;
        R.n_slot[s] = acc < R.slot_cap ? acc : R.slot_cap;
        if (acc > R.slot_cap) *G.error_flag = 2;
    }
    __syncthreads();
    int* slot_a = R.slot_a + (size_t)s * R.slot_cap;
    int* slot_b = R.slot_b + (size_t)s * R.slot_cap;
    for (int a = wave; a < NN; a += n_wave) {
        int base = row_start[a];
        for (int b0 = a + 1; b0 < NN; b0 += 64) {
            const int b = b0 + lane;
            const bool hit = b < NN && slot_of[a * NN + b] == -2;
            const unsigned long long m = __ballot(hit);
            const int sl = base + __popcll(m & ((1ull << lane) - 1ull));
            if (hit) {
                if (sl < R.slot_cap) { slot_a[sl] = a; slot_b[sl] = b; slot_of[a * NN + b] = sl; slot_of[b * NN + a] = sl; }
                else { slot_of[a * NN + b] = -1; slot_of[b * NN + a] = -1; }
            }
            base += __popcll(m);
        }
    }
    __syncthreads();
    // adjacency: slots touching each node, ascending partner id.  BP slots (both nodes with >1 state) also get
    // an "inbox" position: messages TO node g are stored contiguously at inbox[(bp_start[g]+k)*6 ...], so the node
    // update streams them without index indirection; slot_off[2*slot+side] remembers where each slot writes.
    const int* nrot = R.node_nrot;
    for (int g = wave; g < NN; g += n_wave) {
        int* adj = R.adj_slot + ((size_t)s * NN + g) * R.adj_cap;
        int count = 0, count_bp = 0;
        const bool g_multi = nrot[g] > 1;
        for (int b0 = 0; b0 < NN; b0 += 64) {
            const int b = b0 + lane;
            const int sl = b < NN ? slot_of[g * NN + b] : -1;
            const bool hit = sl >= 0;
            const unsigned long long m = __ballot(hit);
            const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
            if (hit && pos < R.adj_cap) adj[pos] = sl;
            count += __popcll(m);
            count_bp += __popcll(__ballot(hit && g_multi && nrot[b] > 1));
        }
        if (lane == 0) {
            R.adj_cnt[(size_t)s * NN + g] = count < R.adj_cap ? count : R.adj_cap;
            if (count > R.adj_cap) *G.error_flag = 3;
            row_count[g] = count_bp;
        }
    }
    __syncthreads();
    int* bp_start = R.bp_start + (size_t)s * (NN + 1);
    if (tid == 0) {
        int acc = 0;
        for (int g = 0; g < NN; ++g) { bp_start[g] = acc; row_start[g] = acc; acc += row_count[g]; }
        bp_start[NN] = acc;
    }
    __syncthreads();
    int* slot_off = R.slot_off + (size_t)s * R.slot_cap * 2;
    for (int g = wave; g < NN; g += n_wave) {
        if (nrot[g] == 1) continue;
        int base = row_start[g];
        for (int b0 = 0; b0 < NN; b0 += 64) {
            const int b = b0 + lane;
            const int sl = b < NN ? slot_of[g * NN + b] : -1;
            const bool hit = sl >= 0 && nrot[b] > 1;
            const unsigned long long m = __ballot(hit);
            if (hit) slot_off[sl * 2 + (g < b ? 0 : 1)] = (base + __popcll(m & ((1ull << lane) - 1ull))) * 6;
            base += __popcll(m);
        }
    }
}
extern "C" int upk_rotamer_build_slots(const upk_launch_t* L, const upk_rotamer_t* R) {
    if (R->n_node > 2048) return 9003;
    hipLaunchKernelGGL(k_rotamer_build_slots, dim3(1, L->n_system), dim3(BP_BLOCK), 0, ST(L), *R);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// 1-body energies -> node probabilities (rotamer.cpp:811-826, 239-256); also clears the slot accumulators
__global__ void k_rotamer_node_prob(upk_rotamer_t R) {
    const int* __restrict__ nb_start = R.node_bead_start; const int* __restrict__ nb_list = R.node_bead_list;
    const int s = blockIdx.y;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_slot = R.n_slot[s];
    if (tid < R.slot_cap * 36 && (tid % R.slot_cap) < n_slot) R.P[(size_t)s * R.slot_cap * 36 + tid] = 0.f;
    if (tid < n_slot) R.slot_active[(size_t)s * R.slot_cap + tid] = 0;
    if (tid >= R.n_node) return;
    const int g = tid, n_rot = R.node_nrot[g];
    float e[6];
    float off = 0.f;
    for (int r = 0; r < 6; ++r) {
        e[r] = 0.f;
        if (r >= n_rot) continue;
        for (int q = nb_start[g * 6 + r]; q < nb_start[g * 6 + r + 1]; ++q) {
            const int bead = nb_list[q];
            const int loc = R.G.loc1[bead];
            float en = 0.f;
            for (int k = 0; k < R.n_prob; ++k) en += R.prob_out[k][(size_t)s * R.prob_sys_stride[k] + (size_t)loc * R.prob_stride[k]];
            e[r] += en;
        }
        off = r == 0 ? e[0] : fminf(off, e[r]);
    }
    float* pr = R.node_prob + ((size_t)s * R.n_node + g) * 6;
    for (int r = 0; r < 6; ++r) pr[r] = r < n_rot ? expf(off - e[r]) : 0.f;
    R.node_off[(size_t)s * R.n_node + g] = off;
}

extern "C" int upk_rotamer_node_prob(const upk_launch_t* L, const upk_rotamer_t* R) {
    int n = R->slot_cap * 36; if (R->n_node > n) n = R->n_node;
    hipLaunchKernelGGL(k_rotamer_node_prob, dim3((n + 255) / 256, L->n_system), dim3(256), 0, ST(L), *R);
    return launch_status();
}

// bead-pair energies into the slot matrices (interaction_graph.h:470-503 + rotamer.cpp:832-846)
struct RotLds { float* tab; float* coords; int* q; };
__device__ __forceinline__ RotLds rot_stage(const upk_rotamer_t& R, float* lds, int s, int tab_floats) {
    RotLds r;
    r.tab = lds; r.coords = lds + ((tab_floats + 3) & ~3);
    r.q = (int*)(r.coords + R.G.n1 * 8) + (threadIdx.x >> 6) * IG_QUEUE;
    stage_table(r.tab, R.G.param, tab_floats);
    stage_coords(r.coords, R.G.node1, s, R.G.loc1, R.G.n1, 6);
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(1024) k_rotamer_pair_energy(upk_rotamer_t R, int tab_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const upk_igraph_t& G = R.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_wave = blockDim.x >> 6;
    const RotLds L = rot_stage(R, lds, s, tab_floats);
    const int NN = R.n_node;
    const float cut2 = G.cutoff * G.cutoff;
    const int* slot_of = R.slot_of + (size_t)s * NN * NN;
    float* P = R.P + (size_t)s * R.slot_cap * 36;
    int* active = R.slot_active + (size_t)s * R.slot_cap;
    QuadShape Q; Q.ka = G.n_knot_angular; Q.k = G.n_knot; Q.inv_dx = G.inv_dx; Q.inv_dtheta = G.inv_dtheta;
    for (int row = blockIdx.x * n_wave + wave; row < G.n1; row += gridDim.x * n_wave) {
        const int* nbr = G.nbr1 + ((size_t)s * G.n1 + row) * G.cap1;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        float xr[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) xr[c] = L.coords[row * 8 + c];
        const int tr = G.type1[row], a = R.bead_node[row], ra = R.bead_rot[row];
        // each pair once: only partners with a larger bead index (i1 < i2 as in the reference)
        for_each_inrange(nbr, cnt, xr, L.coords, cut2, L.q, lane, row, [&](int j, bool valid) {
            if (!valid) return;
            float xo[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) xo[c] = L.coords[j * 8 + c];
            const float* p = L.tab + (tr * G.n_type2 + G.type1[j]) * G.n_param;
            const float E = quadspline2<0>(Q, p, xr, xo, nullptr);
            const int b = R.bead_node[j], rb = R.bead_rot[j];
            const int sl = slot_of[a * NN + b];
            if (sl < 0) return;                           // only after a capacity overflow (error flag is set)
            const int idx = a < b ? ra * 6 + rb : rb * 6 + ra;
            atomicAdd(&P[PIDX(R, sl, idx)], E);           // one bead per rotamer state => a single contributor
            active[sl] = 1;
        });
    }
}

static bool rot_geometry(const upk_launch_t* L, const upk_rotamer_t* R, int& tab_floats, size_t& lds_bytes, dim3& grid, dim3& block) {
    tab_floats = R->G.n_type1 * R->G.n_type2 * R->G.n_param;
    const int waves = 16;
    lds_bytes = ((size_t)((tab_floats + 3) & ~3) + (size_t)R->G.n1 * 8 + (size_t)waves * IG_QUEUE) * sizeof(float);
    if (lds_bytes > 158 * 1024) return false;
    int bps = (1024 + L->n_system - 1) / L->n_system;
    const int max_bps = (R->G.n1 + waves - 1) / waves;
    if (bps > max_bps) bps = max_bps;
    if (bps < 1) bps = 1;
    grid = dim3(bps, L->n_system); block = dim3(waves * 64);
    return true;
}

extern "C" int upk_rotamer_pair_energy(const upk_launch_t* L, const upk_rotamer_t* R) {
    int tab_floats; size_t lds; dim3 grid, block;
    if (!rot_geometry(L, R, tab_floats, lds, grid, block)) return 9005;   // more beads than LDS can stage
    hipLaunchKernelGGL(k_rotamer_pair_energy, grid, block, lds, ST(L), *R, tab_floats);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// belief propagation: one persistent workgroup per system (rotamer.cpp:1005-1061)
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, scratch[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += scratch[w];
    return r;
}

// one residue-pair edge with compile-time state counts: new messages from the old beliefs
// (update_beliefs, rotamer.cpp:468-499 and the L1 normalisation of 506-521), written in place
template <int NA, int NB>
__device__ __forceinline__ void bp_edge(const float* __restrict__ Ps, int pstride, const float* __restrict__ nba, const float* __restrict__ nbb,
                                        float* __restrict__ ma, float* __restrict__ mb) {
    float va[NA], vb[NB], P[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) va[i] = nba[i] * rcp(1e-10f + ma[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) vb[j] = nbb[j] * rcp(1e-10f + mb[j]);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) P[i][j] = Ps[(size_t)(i * 6 + j) * pstride];
    float ta[NA], tb[NB], sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) { float t = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) t += P[i][j] * vb[j];
        ta[i] = t; sa += t; }
#pragma unroll
    for (int j = 0; j < NB; ++j) { float t = 0.f;
#pragma unroll
        for (int i = 0; i < NA; ++i) t += va[i] * P[i][j];
        tb[j] = t; sb += t; }
    const float ra = rcp(sa), rb = rcp(sb);
#pragma unroll
    for (int i = 0; i < NA; ++i) ma[i] = ta[i] * ra;
#pragma unroll
    for (int j = 0; j < NB; ++j) mb[j] = tb[j] * rb;
}

// pair marginal and (optionally) its Bethe free-energy term (rotamer.cpp:405-451)
template <int NA, int NB>
__device__ __forceinline__ float bp_marginal(const float* __restrict__ Ps, int pstride, const float* __restrict__ nba, const float* __restrict__ nbb,
                                             const float* __restrict__ ma, const float* __restrict__ mb, float* __restrict__ mg_out,
                                             bool want_energy) {
    float Pl[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) Pl[i][j] = Ps[(size_t)(i * 6 + j) * pstride];
    float bc1[NA], bc2[NB], mg[NA][NB], sum = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) bc1[i] = nba[i] * rcp(1e-10f + ma[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) bc2[j] = nbb[j] * rcp(1e-10f + mb[j]);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) { mg[i][j] = Pl[i][j] * bc1[i] * bc2[j]; sum += mg[i][j]; }
    const float rs = rcp(sum);
    float en = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float pm = mg[i][j] * rs;
            mg_out[(size_t)(i * 6 + j) * pstride] = pm;
            if (want_energy) en += pm * logf((1e-10f + pm) * rcp(1e-10f + Pl[i][j] * nba[i] * nbb[j]));
        }
    return en;
}

#define BP_GROUP 16   // lanes cooperating on one node in the node phase

__global__ void __launch_bounds__(BP_BLOCK) k_rotamer_bp(upk_rotamer_t R, int want_energy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int NN = R.n_node;
    float* prob = lds;                 // [NN][6]  node probabilities with the 1-state partners folded in
    float* nb0 = lds + NN * 6;         // [NN][6]
    float* nb1 = lds + NN * 12;        // [NN][6]
    float* scratch = lds + NN * 18;    // [32]
    int* nrot = (int*)(lds + NN * 18 + 32);      // [NN]   state counts
    int* bp_start = nrot + NN;                   // [NN+1] inbox CSR
    const int n_slot = R.n_slot[s];
    const int cap = R.slot_cap;
    const int* slot_a = R.slot_a + (size_t)s * R.slot_cap;
    const int* slot_b = R.slot_b + (size_t)s * R.slot_cap;
    const int* active = R.slot_active + (size_t)s * R.slot_cap;
    const int* adj_cnt = R.adj_cnt + (size_t)s * NN;
    const int* adj_slot = R.adj_slot + (size_t)s * NN * R.adj_cap;
    for (int i = tid; i < NN; i += nt) nrot[i] = R.node_nrot[i];
    for (int i = tid; i <= NN; i += nt) bp_start[i] = R.bp_start[(size_t)s * (NN + 1) + i];
    __syncthreads();
    const int* slot_off = R.slot_off + (size_t)s * R.slot_cap * 2;
    float* P = R.P + (size_t)s * R.slot_cap * 36;
    float* inbox = R.msg_cur + (size_t)s * R.slot_cap * 12;
    float* marg = R.marg + (size_t)s * R.slot_cap * 36;

    // energies -> probabilities (rotamer.cpp:835); old edge beliefs = 1 (rotamer.cpp:1015-1032), also for the
    // slots without an in-range bead pair this step, whose (unit) message then multiplies as an exact 1
    for (int i = tid; i < n_slot * 36; i += nt) {
        const int e = i / n_slot, sl = i % n_slot, ra = e / 6, rb = e % 6;
        const bool used = ra < nrot[slot_a[sl]] && rb < nrot[slot_b[sl]];
        const size_t pi = (size_t)e * cap + sl;
        P[pi] = used ? expf(-P[pi]) : 0.f;
    }
    for (int i = tid; i < bp_start[NN] * 6; i += nt) inbox[i] = 1.f;
    for (int i = tid; i < NN * 6; i += nt) prob[i] = R.node_prob[(size_t)s * NN * 6 + i];
    __syncthreads();
    // fold edges to 1-state partners into the node probabilities (move_edge_prob_to_node2, rotamer.cpp:378-385)
    for (int g = tid; g < NN; g += nt) {
        const int n = nrot[g];
        if (n == 1) continue;
        for (int k = 0; k < adj_cnt[g]; ++k) {
            const int sl = adj_slot[g * R.adj_cap + k];
            const int a = slot_a[sl];
            if (a == g || nrot[a] != 1 || !active[sl]) continue;     // a 1-state partner always has the lower id
            for (int r = 0; r < n; ++r) prob[g * 6 + r] *= P[(size_t)r * cap + sl];
        }
    }
    __syncthreads();
    for (int i = tid; i < NN * 6; i += nt) { nb0[i] = prob[i]; nb1[i] = prob[i]; }   // old node belief = prob (rotamer.cpp:1009-1013)
    __syncthreads();

    float* nb_old = nb0; float* nb_cur = nb1;
    int iter = 0;
    float maxdev = 1e10f;
    const int grp = tid / BP_GROUP, gl = tid % BP_GROUP, n_grp = nt / BP_GROUP;
    // sweep -1 is calculate_new_beliefs(0.f, true): only its messages survive and the "old" node belief becomes
    // prob / max(prob) (rotamer.cpp:1034 with the swap at 995-1001)
    for (int sweep = -1;; ++sweep) {
        // ---- edge phase: every residue pair rewrites its two messages in place from the old node beliefs
        for (int sl = tid; sl < n_slot; sl += nt) {
            const int a = slot_a[sl], b = slot_b[sl];
            const int na = nrot[a];
            if (na == 1 || !active[sl]) continue;
            const int nb = nrot[b];
            const float* Ps = P + sl;
            float* ma = inbox + slot_off[sl * 2];
            float* mb = inbox + slot_off[sl * 2 + 1];
            if (na == 3 && nb == 3) bp_edge<3, 3>(Ps, cap, nb_old + a * 6, nb_old + b * 6, ma, mb);
            else if (na == 3) bp_edge<3, 6>(Ps, cap, nb_old + a * 6, nb_old + b * 6, ma, mb);
            else bp_edge<6, 6>(Ps, cap, nb_old + a * 6, nb_old + b * 6, ma, mb);
        }
        __syncthreads();
        // ---- node phase: BP_GROUP lanes per node stream the node's inbox, multiply, and combine by shuffles
        float dev = 0.f;
        for (int g0 = 0; g0 < NN; g0 += n_grp) {
            const int g = g0 + grp;
            const bool live = g < NN && nrot[g] > 1;
            const int n = live ? nrot[g] : 0;
            float bb[6] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            if (live && sweep >= 0) {
                const int k0 = bp_start[g], k1 = bp_start[g + 1];
                for (int k = k0 + gl; k < k1; k += BP_GROUP) {
                    const float* m = inbox + (size_t)k * 6;
                    float mx = 0.f;
#pragma unroll
                    for (int r = 0; r < 6; ++r) { bb[r] *= (r < n ? m[r] : 1.f); mx = fmaxf(mx, r < n ? bb[r] : 0.f); }
                    const float rm = rcp(mx);           // keep the running product O(1) (rotamer.cpp:489-493 re-normalises too)
#pragma unroll
                    for (int r = 0; r < 6; ++r) bb[r] *= rm;
                }
            }
            // product over the lanes of the group
#pragma unroll
            for (int off = BP_GROUP / 2; off > 0; off >>= 1) {
                float mx = 0.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) { bb[r] *= __shfl_xor(bb[r], off, UP_WAVE); mx = fmaxf(mx, r < n ? bb[r] : 0.f); }
                const float rm = mx > 0.f ? rcp(mx) : 1.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) bb[r] *= rm;
            }
            if (live && gl < n) {
                // lane r of the group finishes rotamer state r: b = prob * product, then standardize (rotamer.cpp:258-273)
                float mine = 0.f, mx = 0.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) if (r < n) { const float v = prob[g * 6 + r] * bb[r]; mx = fmaxf(mx, v); if (r == gl) mine = v; }
                const float o = nb_old[g * 6 + gl];
                const float damp = sweep < 0 ? 0.f : R.damping;
                const float v = damp != 0.f ? (1.f - damp) * rcp(mx) * mine + damp * o : rcp(mx) * mine;
                nb_cur[g * 6 + gl] = v;
                dev = fmaxf(v - o, dev);                               // signed, rotamer.cpp:275-281
            }
        }
        __syncthreads();
        if (sweep >= 0) {
            ++iter;
            if (iter % R.chunk == 0) {
                maxdev = block_max(dev, scratch);
                if (!(maxdev > R.tol && iter < R.max_iter)) break;     // rotamer.cpp:1038
            }
        }
        float* t = nb_old; nb_old = nb_cur; nb_cur = t;                // rotamer.cpp:1040-1044
    }
    if (tid == 0) R.iters[s] = iter;

    // ---- marginals (rotamer.cpp:1053-1059)
    for (int g = tid; g < NN; g += nt) {
        const int n = nrot[g];
        float sum = 0.f;
        if (n == 1) { nb_cur[g * 6] = 1.f; for (int r = 1; r < 6; ++r) nb_cur[g * 6 + r] = 0.f; continue; }
        for (int r = 0; r < n; ++r) sum += nb_cur[g * 6 + r];
        const float rs = rcp(sum);
        for (int r = 0; r < n; ++r) nb_cur[g * 6 + r] *= rs;
    }
    __syncthreads();
    float en = 0.f;
    for (int sl = tid; sl < n_slot; sl += nt) {
        const int a = slot_a[sl], b = slot_b[sl];
        const int na = nrot[a], nb = nrot[b];
        if (!active[sl]) continue;
        const float* Ps = P + sl;
        if (nb == 1) { if (want_energy) en += -logf(Ps[0]); continue; }   // 1-1 edge (rotamer.cpp:861)
        if (na == 1) continue;                                            // folded into node b
        const float* ma = inbox + slot_off[sl * 2];
        const float* mb = inbox + slot_off[sl * 2 + 1];
        float* mo = marg + sl;
        if (na == 3 && nb == 3) en += bp_marginal<3, 3>(Ps, cap, nb_cur + a * 6, nb_cur + b * 6, ma, mb, mo, want_energy);
        else if (na == 3) en += bp_marginal<3, 6>(Ps, cap, nb_cur + a * 6, nb_cur + b * 6, ma, mb, mo, want_energy);
        else en += bp_marginal<6, 6>(Ps, cap, nb_cur + a * 6, nb_cur + b * 6, ma, mb, mo, want_energy);
    }
    if (want_energy) {
        for (int g = tid; g < NN; g += nt) {   // node_free_energy, rotamer.cpp:292-302
            const int n = nrot[g];
            float e = R.node_off[(size_t)s * NN + g];
            for (int r = 0; r < n; ++r) { const float b = nb_cur[g * 6 + r]; e += b * logf((1e-10f + b) * rcp(1e-10f + prob[g * 6 + r])); }
            en += e;
        }
        const float tot = block_sum(en, scratch);
        if (tid == 0) R.energy[s] = tot;
    }
    for (int i = tid; i < NN * 6; i += nt) R.nb_cur[(size_t)s * NN * 6 + i] = nb_cur[i];
}

extern "C" int upk_rotamer_bp(const upk_launch_t* L, const upk_rotamer_t* R, int want_energy) {
    const size_t lds = ((size_t)R->n_node * 20 + 40) * sizeof(float);
    if (lds > 155 * 1024) return 9004;
    hipLaunchKernelGGL(k_rotamer_bp, dim3(1, L->n_system), dim3(BP_BLOCK), lds, ST(L), *R, want_energy);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// derivative push (rotamer.cpp:956-985 + interaction_graph.h:525-555 as a per-bead gather)
__global__ void __launch_bounds__(1024) k_rotamer_grad(upk_rotamer_t R, int tab_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const upk_igraph_t& G = R.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_wave = blockDim.x >> 6;
    const RotLds L = rot_stage(R, lds, s, tab_floats);
    const int NN = R.n_node;
    const float cut2 = G.cutoff * G.cutoff;
    const int* slot_of = R.slot_of + (size_t)s * NN * NN;
    const float* marg = R.marg + (size_t)s * R.slot_cap * 36;
    const float* nbm = R.nb_cur + (size_t)s * NN * 6;
    QuadShape Q; Q.ka = G.n_knot_angular; Q.k = G.n_knot; Q.inv_dx = G.inv_dx; Q.inv_dtheta = G.inv_dtheta;
    for (int row = blockIdx.x * n_wave + wave; row < G.n1; row += gridDim.x * n_wave) {
        const int* nbr = G.nbr1 + ((size_t)s * G.n1 + row) * G.cap1;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        float xr[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) xr[c] = L.coords[row * 8 + c];
        const int tr = G.type1[row], a = R.bead_node[row], ra = R.bead_rot[row], na = R.node_nrot[a];
        float acc[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] = 0.f;
        for_each_inrange(nbr, cnt, xr, L.coords, cut2, L.q, lane, -1, [&](int j, bool valid) {
            if (!valid) return;
            float xo[6], d1[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) xo[c] = L.coords[j * 8 + c];
            const float* p = L.tab + (tr * G.n_type2 + G.type1[j]) * G.n_param;
            quadspline2<1>(Q, p, xr, xo, d1);
            const int b = R.bead_node[j], rb = R.bead_rot[j], nb = R.node_nrot[b];
            float ps;
            if (na == 1 && nb == 1) ps = 1.f;
            else if (na == 1) ps = nbm[b * 6 + rb];
            else if (nb == 1) ps = nbm[a * 6 + ra];
            else {
                const int sl = slot_of[a * NN + b];
                ps = sl < 0 ? 0.f : marg[PIDX(R, sl, a < b ? ra * 6 + rb : rb * 6 + ra)];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) acc[c] += ps * d1[c];
        });
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] = wave_sum(acc[c]);
        if (lane == 0) {
            const int loc = G.loc1[row];
            float* t = C_SENS(G.node1, s) + (size_t)loc * G.node1.stride;
#pragma unroll
            for (int c = 0; c < 6; ++c) t[c] += acc[c];
            const float mg = nbm[a * 6 + ra];
            for (int k = 0; k < R.n_prob; ++k) R.prob_sens[k][(size_t)s * R.prob_sys_stride[k] + (size_t)loc * R.prob_stride[k]] += mg;
        }
    }
}
extern "C" int upk_rotamer_grad(const upk_launch_t* L, const upk_rotamer_t* R) {
    int tab_floats; size_t lds; dim3 grid, block;
    if (!rot_geometry(L, R, tab_floats, lds, grid, block)) return 9005;
    hipLaunchKernelGGL(k_rotamer_grad, grid, block, lds, ST(L), *R, tab_floats);
    return launch_status();
}
