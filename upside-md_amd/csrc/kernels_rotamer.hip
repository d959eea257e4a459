// Rotamer free-energy node on gfx950 (replaces /root/reference/src/rotamer.cpp).
//
// Structure (MI355X-first):
//   * side-chain "nodes" (one per residue with 1/3/6 rotamer states) get global ids sorted by state count
//     (1-state, then 3, then 6), so the reference's canonical edge orientation n_rot1 <= n_rot2
//     (rotamer.cpp:837) is simply a < b;
//   * residue-pair "slots" (the BP edges) are assigned on the device when the bead pair list is rebuilt: the list
//     build marks a dense node x node table (replaces the EdgeLocator hash, rotamer.cpp:134-206), one workgroup
//     per system then numbers the slots GROUPED BY CLASS (3x3, 3x6, 6x6, 1x1, 1xN) so that belief propagation
//     runs divergence-free template instances over contiguous ranges, and every cached bead pair remembers its
//     slot (nbr_slot) so the pair kernels need no table lookup;
//   * slot matrices are structure-of-arrays ([36][slot_cap]); BP messages live in an "inbox" grouped by receiving
//     node so the node update streams them;
//   * belief propagation is ONE persistent workgroup per system with node beliefs in LDS; the bead-pair kernels
//     stage the spline table and every bead (coordinates + packed metadata) of the system in LDS.
#include "device_math.h"
#include "../../include/upside_hip_kernels.h"
#include "igraph_device.h"

using namespace up;

#define ST(L) ((hipStream_t)(L)->stream)
#define BP_BLOCK 1024
static inline int launch_status() { return (int)hipGetLastError(); }
#define C_OUT(c, s)  ((c).out  + (size_t)(s) * (c).n_elem * (c).stride)
#define C_SENS(c, s) ((c).sens + (size_t)(s) * (c).n_elem * (c).stride)
#define PIDX(R, sl, e) ((size_t)(e) * (R).slot_cap + (sl))

// slot classes in storage order
enum { CL33 = 0, CL36 = 1, CL66 = 2, CL11 = 3, CL1X = 4, N_CLASS = 5 };
__device__ __forceinline__ int slot_class(int na, int nb) {   // na <= nb
    if (na == 1) return nb == 1 ? CL11 : CL1X;
    if (na == 3) return nb == 3 ? CL33 : CL36;
    return CL66;
}

// ------------------------------------------------------------------------------------------------
// rebuild step 0: clear the node x node mark table of the flagged systems (before the list build marks it)
__global__ void k_rotamer_clear_slots(upk_rotamer_t R) {
    const int* fl = UPK_FLAG_LIST(R.G);
    const int n_flagged = fl[0];
    const int n16 = (R.n_node * R.n_node + 15) / 16;          // the table is padded to a multiple of 16 bytes
    for (int fi = blockIdx.y; fi < n_flagged; fi += gridDim.y) {
        uint4* m = (uint4*)(R.mark + (size_t)fl[1 + fi] * R.G.mark_stride);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) m[i] = make_uint4(0, 0, 0, 0);
    }
}
extern "C" int upk_rotamer_clear_slots(const upk_launch_t* L, const upk_rotamer_t* R) {
    const int n16 = (R->n_node * R->n_node + 15) / 16;
    int blocks = (n16 + 1023) / 1024; if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(k_rotamer_clear_slots, dim3(blocks, L->n_system < UPK_FLAG_GRID ? L->n_system : UPK_FLAG_GRID), dim3(1024), 0, ST(L), *R);
    return launch_status();
}

// block-wide exclusive prefix sum over blockDim.x = 1024 values (scratch: 17 ints); returns the prefix, *total the sum
__device__ __forceinline__ int block_excl_scan(int v, int* scratch, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, UP_WAVE); if (lane >= off) incl += t; }
    __syncthreads();                                   // scratch may still be read from a previous call
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) { int acc = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { const int t = scratch[w]; scratch[w] = acc; acc += t; } scratch[16] = acc; }
    __syncthreads();
    *total = scratch[16];
    return scratch[wave] + incl - v;
}

// rebuild step 2 (after the list build has marked the table): number the slots by class, build the adjacency and
// the message inbox layout.  One workgroup per flagged system; the marks are packed into an LDS bit matrix
// (one row of 64-bit words per node) and every later pass is one THREAD per node over its row of words.
__global__ void __launch_bounds__(BP_BLOCK) k_rotamer_build_slots(upk_rotamer_t R) {
    extern __shared__ unsigned long long bits[];            // [NN][W]
    __shared__ int row_lo[1024], row_hi[1024], deg1[1024], bp_s[1025], scratch[17], cls_lds[N_CLASS + 1];
    const upk_igraph_t& G = R.G;
    const int* fl = UPK_FLAG_LIST(G);
    const int n_flagged = fl[0];
    const int NN = R.n_node, W = (NN + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_wave = blockDim.x >> 6;
    const int* nrot = R.node_nrot;
    // nodes are sorted by state count: [0, e1) have 1 state, [e1, e3) 3, [e3, NN) 6
    const int e1 = R.n_node1, e3 = R.n_node1 + R.n_node3;
    auto below = [](int x, int c) -> unsigned long long {   // bits of word c whose node id is < x
        const int r = x - c * 64;
        return r <= 0 ? 0ull : (r >= 64 ? ~0ull : ((1ull << r) - 1ull));
    };
    auto rank_below = [&](int g, int x) {                    // partners of g with id < x
        int n = 0;
        for (int c = 0; c * 64 < x && c < W; ++c) n += __popcll(bits[g * W + c] & below(x, c));
        return n;
    };
    for (int fi = blockIdx.y; fi < n_flagged; fi += gridDim.y) {
        const int s = fl[1 + fi];
        const unsigned char* mark = R.mark + (size_t)s * G.mark_stride;
        int* slot_of = R.slot_of + (size_t)s * NN * NN;
        __syncthreads();                                     // LDS of the previous system is no longer read
        // ---- pack the marks: wave item = (row, word); 8 loads in flight per lane
        for (int item0 = wave * 8; item0 < NN * W; item0 += n_wave * 8) {
            unsigned char v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int item = item0 + u, a = item / W, b = (item % W) * 64 + lane;
                v[u] = (item < NN * W && b < NN) ? mark[(size_t)a * NN + b] : (unsigned char)0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned long long m = __ballot(v[u] != 0);
                if (lane == 0 && item0 + u < NN * W) bits[item0 + u] = m;
            }
        }
        __syncthreads();
        // ---- per node: partners above it in its own class / in a higher class, 1-state partners, BP partners
        const int a = tid;
        const int na = a < NN ? nrot[a] : 0;
        const int cend = na == 1 ? e1 : (na == 3 ? e3 : NN);  // end of a's own class
        int c_lo = 0, c_hi = 0, d1 = 0, dbp = 0;
        if (a < NN) {
            for (int c = 0; c < W; ++c) {
                const unsigned long long w = bits[a * W + c];
                const unsigned long long up = w & ~below(a + 1, c);
                c_lo += __popcll(up & below(cend, c));
                c_hi += __popcll(up & ~below(cend, c));
                d1 += __popcll(w & below(e1, c));
                dbp += __popcll(w & ~below(e1, c));
            }
            if (na == 1) dbp = 0;
            deg1[a] = d1;
        }
        int tot_lo, tot_hi, tot_bp;
        const int x_lo = block_excl_scan(c_lo, scratch, &tot_lo);
        const int x_hi = block_excl_scan(c_hi, scratch, &tot_hi);
        const int x_bp = block_excl_scan(dbp, scratch, &tot_bp);
        if (a < NN) { row_lo[a] = x_lo; row_hi[a] = x_hi; bp_s[a] = x_bp; }
        if (tid == 0) bp_s[NN] = tot_bp;
        __syncthreads();
        // class sizes from the scans: class (1,1) = lo counts of the 1-state rows, (1,x) = their hi counts, ...
        if (tid == 0) {
            auto at = [&](const int* x, int i, int tot) { return i < NN ? x[i] : tot; };
            int cnt[N_CLASS];
            cnt[CL11] = at(row_lo, e1, tot_lo);
            cnt[CL1X] = at(row_hi, e1, tot_hi);
            cnt[CL33] = at(row_lo, e3, tot_lo) - at(row_lo, e1, tot_lo);
            cnt[CL36] = at(row_hi, e3, tot_hi) - at(row_hi, e1, tot_hi);
            cnt[CL66] = tot_lo - at(row_lo, e3, tot_lo);
            int acc = 0;
            for (int c = 0; c < N_CLASS; ++c) { cls_lds[c] = acc; acc += cnt[c]; }
            cls_lds[N_CLASS] = acc;
            int* cs = R.class_start + (size_t)s * (N_CLASS + 1);
            for (int c = 0; c <= N_CLASS; ++c) cs[c] = cls_lds[c] < R.slot_cap ? cls_lds[c] : R.slot_cap;
            R.n_slot[s] = acc < R.slot_cap ? acc : R.slot_cap;
            if (acc > R.slot_cap) *G.error_flag = 2;
        }
        __syncthreads();
        int* bp_start = R.bp_start + (size_t)s * (NN + 1);
        for (int g = tid; g <= NN; g += blockDim.x) bp_start[g] = bp_s[g];
        if (a < NN) {
            const int d = na > 1 ? deg1[a] : 0;
            R.adj_cnt[(size_t)s * NN + a] = d < R.adj_cap ? d : R.adj_cap;
            if (d > R.adj_cap) *G.error_flag = 3;
        }
        // ---- assign: node a numbers its pairs (a, b > a) in ascending b; everything a pair needs follows from
        // popcounts of the two rows (messages TO node g sit at inbox rows bp_start[g] + rank among g's multi-state
        // partners, 8 floats each; folded 1-state partners of g are listed in ascending id)
        if (a < NN) {
            int* slot_a = R.slot_a + (size_t)s * R.slot_cap;
            int* slot_b = R.slot_b + (size_t)s * R.slot_cap;
            int* slot_off = R.slot_off + (size_t)s * R.slot_cap * 2;
            const int cl_lo = slot_class(na, na), cl_hi = na == 1 ? CL1X : (na == 3 ? CL36 : CL66);
            const int first = na == 1 ? 0 : (na == 3 ? e1 : e3);          // first row of a's class
            const int base_lo = first < NN ? row_lo[first] : 0, base_hi = first < NN ? row_hi[first] : 0;
            int p_lo = cls_lds[cl_lo] + row_lo[a] - base_lo, p_hi = cls_lds[cl_hi] + row_hi[a] - base_hi;
            int k_multi = na > 1 ? rank_below(a, a) - deg1[a] : 0;        // multi-state partners of a below b so far
            for (int c = a / 64; c < W; ++c) {
                unsigned long long w = bits[a * W + c] & ~below(a + 1, c);
                while (w) {
                    const int b = c * 64 + __builtin_ctzll(w);
                    w &= w - 1;
                    const bool lo = b < cend;
                    const int sl = lo ? p_lo++ : p_hi++;
                    const int nb = nrot[b];
                    if (sl < R.slot_cap) {
                        slot_a[sl] = a; slot_b[sl] = b; slot_of[(size_t)a * NN + b] = sl; slot_of[(size_t)b * NN + a] = sl;
                        if (na == 1 && nb > 1) {
                            const int pos = rank_below(b, a);              // 1-state partners of b below a
                            if (pos < R.adj_cap) R.adj_slot[((size_t)s * NN + b) * R.adj_cap + pos] = sl;
                        } else if (na > 1) {
                            slot_off[sl * 2] = (bp_s[a] + k_multi) * 8;
                            slot_off[sl * 2 + 1] = (bp_s[b] + rank_below(b, a) - deg1[b]) * 8;
                        }
                    } else { slot_of[(size_t)a * NN + b] = -1; slot_of[(size_t)b * NN + a] = -1; }
                    if (na > 1) ++k_multi;
                }
            }
        }
    }
}
extern "C" int upk_rotamer_build_slots(const upk_launch_t* L, const upk_rotamer_t* R) {
    if (R->n_node > 1024) return 9003;
    const size_t lds = (size_t)R->n_node * ((R->n_node + 63) / 64) * 8;
    hipLaunchKernelGGL(k_rotamer_build_slots, dim3(1, L->n_system < 64 ? L->n_system : 64), dim3(BP_BLOCK), lds, ST(L), *R);
    return launch_status();
}

// rebuild step 3: every cached bead pair remembers its slot
__global__ void k_rotamer_nbr_slots(upk_rotamer_t R) {
    const upk_igraph_t& G = R.G;
    const int* fl = UPK_FLAG_LIST(G);
    const int n_flagged = fl[0];
    const int NN = R.n_node;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_wave = blockDim.x >> 6;
    for (int fi = blockIdx.y; fi < n_flagged; fi += gridDim.y) {
        const int s = fl[1 + fi];
        const int* slot_of = R.slot_of + (size_t)s * NN * NN;
        for (int row = blockIdx.x * n_wave + wave; row < G.n1; row += gridDim.x * n_wave) {
            const size_t base = ((size_t)s * G.n1 + row) * G.cap1;
            const int cnt = G.cnt1[(size_t)s * G.n1 + row];
            const int a = R.bead_node[row];
            for (int k = lane; k < cnt; k += 64) R.nbr_slot[base + k] = slot_of[a * NN + R.bead_node[G.nbr1[base + k]]];
        }
    }
}
extern "C" int upk_rotamer_nbr_slots(const upk_launch_t* L, const upk_rotamer_t* R) {
    int blocks = (R->G.n1 + 3) / 4;
    hipLaunchKernelGGL(k_rotamer_nbr_slots, dim3(blocks, L->n_system < UPK_FLAG_GRID ? L->n_system : UPK_FLAG_GRID), dim3(256), 0, ST(L), *R);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// 1-body energies -> node probabilities (rotamer.cpp:811-826, 239-256)
__global__ void k_rotamer_node_prob(upk_rotamer_t R) {
    const int* __restrict__ nb_start = R.node_bead_start; const int* __restrict__ nb_list = R.node_bead_list;
    const int s = blockIdx.y;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= R.n_node) return;
    const int n_rot = R.node_nrot[g];
    float e[6];
    float off = 0.f;
    for (int r = 0; r < 6; ++r) {
        e[r] = 0.f;
        if (r >= n_rot) continue;
        for (int q = nb_start[g * 6 + r]; q < nb_start[g * 6 + r + 1]; ++q) {
            const int loc = R.G.loc1[nb_list[q]];
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
    hipLaunchKernelGGL(k_rotamer_node_prob, dim3((R->n_node + 255) / 256, L->n_system), dim3(256), 0, ST(L), *R);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// bead-pair kernels: table + all beads of the system in LDS.  Bead row: [0,6) pos+dir, [6] type | rot<<8 |
// nrot<<12, [7] node id (raw int bits).
struct RotLds { float* tab; float* rows; int* q; };
__device__ __forceinline__ RotLds rot_stage(const upk_rotamer_t& R, float* lds, int s, int tab_floats) {
    RotLds r;
    r.tab = lds; r.rows = lds + ((tab_floats + 3) & ~3);
    r.q = (int*)(r.rows + R.G.n1 * 8) + (threadIdx.x >> 6) * IG_QUEUE;
    stage_table(r.tab, R.G.param, tab_floats);
    stage_rows(r.rows, R.G.node1, s, R.G.loc1, R.G.n1, 6, R.bead_node, R.bead_meta, nullptr, 0);
    __syncthreads();
    return r;
}

// bead-pair energies into the slot matrices (interaction_graph.h:470-503 + rotamer.cpp:832-846)
__global__ void __launch_bounds__(1024) k_rotamer_pair_energy(upk_rotamer_t R, int tab_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const upk_igraph_t& G = R.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_wave = blockDim.x >> 6;
    const RotLds L = rot_stage(R, lds, s, tab_floats);
    const float cut2 = G.cutoff * G.cutoff;
    float* P = R.P + (size_t)s * R.slot_cap * 36;
    int* active = R.slot_active + (size_t)s * R.slot_cap;
    QuadShape Q; Q.ka = G.n_knot_angular; Q.k = G.n_knot; Q.inv_dx = G.inv_dx; Q.inv_dtheta = G.inv_dtheta;
    for (int row = blockIdx.x * n_wave + wave; row < G.n1; row += gridDim.x * n_wave) {
        const size_t base = ((size_t)s * G.n1 + row) * G.cap1;
        const int* nbr = G.nbr1 + base;
        const int* nsl = R.nbr_slot + base;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        float xr[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) xr[c] = L.rows[row * 8 + c];
        const int mr = __float_as_int(xr[6]), a = __float_as_int(xr[7]);
        const int tr = mr & 0xFF, ra = (mr >> 8) & 0xF;
        // each pair once: only partners with a larger bead index (i1 < i2 as in the reference)
        for_each_inrange(nbr, cnt, xr, L.rows, cut2, L.q, lane, row, [&](int j, int k, bool valid) {
            if (!valid) return;
            float xo[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) xo[c] = L.rows[j * 8 + c];
            const int mo = __float_as_int(xo[6]), b = __float_as_int(xo[7]);
            const float* p = L.tab + (tr * G.n_type2 + (mo & 0xFF)) * G.n_param;
            const float E = quadspline2<0>(Q, p, xr, xo, nullptr);
            const int sl = nsl[k];
            if (sl < 0) return;                           // only after a capacity overflow (error flag is set)
            const int rb = (mo >> 8) & 0xF;
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
    int bps = (ig_target_wgs() + L->n_system - 1) / L->n_system;
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

// derivative push (rotamer.cpp:956-985 + interaction_graph.h:525-555 as a per-bead gather)
__global__ void __launch_bounds__(1024) k_rotamer_grad(upk_rotamer_t R, int tab_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const upk_igraph_t& G = R.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_wave = blockDim.x >> 6;
    const RotLds L = rot_stage(R, lds, s, tab_floats);
    const int NN = R.n_node;
    const float cut2 = G.cutoff * G.cutoff;
    const float* marg = R.marg + (size_t)s * R.slot_cap * 36;
    const float* nbm = R.nb_cur + (size_t)s * NN * 6;
    QuadShape Q; Q.ka = G.n_knot_angular; Q.k = G.n_knot; Q.inv_dx = G.inv_dx; Q.inv_dtheta = G.inv_dtheta;
    for (int row = blockIdx.x * n_wave + wave; row < G.n1; row += gridDim.x * n_wave) {
        const size_t base = ((size_t)s * G.n1 + row) * G.cap1;
        const int* nbr = G.nbr1 + base;
        const int* nsl = R.nbr_slot + base;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        float xr[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) xr[c] = L.rows[row * 8 + c];
        const int mr = __float_as_int(xr[6]), a = __float_as_int(xr[7]);
        const int tr = mr & 0xFF, ra = (mr >> 8) & 0xF, na = (mr >> 12) & 0xF;
        float acc[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] = 0.f;
        for_each_inrange(nbr, cnt, xr, L.rows, cut2, L.q, lane, -1, [&](int j, int k, bool valid) {
            if (!valid) return;
            float xo[8], d1[6];
#pragma unroll
            for (int c = 0; c < 8; ++c) xo[c] = L.rows[j * 8 + c];
            const int mo = __float_as_int(xo[6]), b = __float_as_int(xo[7]);
            const float* p = L.tab + (tr * G.n_type2 + (mo & 0xFF)) * G.n_param;
            quadspline2<1>(Q, p, xr, xo, d1);
            const int rb = (mo >> 8) & 0xF, nb = (mo >> 12) & 0xF;
            float ps;
            if (na == 1 && nb == 1) ps = 1.f;
            else if (na == 1) ps = nbm[b * 6 + rb];
            else if (nb == 1) ps = nbm[a * 6 + ra];
            else {
                const int sl = nsl[k];
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

struct BpCtx {
    const int *slot_a, *slot_b, *active, *slot_off;
    float *P, *inbox, *marg;
    int cap;
};

// edge phase over one class range: new messages from the old beliefs (update_beliefs, rotamer.cpp:468-499 and the
// L1 normalisation of 506-521), rewritten in place.  1-ulp hardware reciprocals: the reference itself uses the
// 12-bit rcpps here (Float4.h:199-212).
template <int NA, int NB>
__device__ __forceinline__ void bp_edge_range(const BpCtx& C, int lo, int hi, const float* __restrict__ nb_old, int tid, int nt) {
    for (int sl = lo + tid; sl < hi; sl += nt) {
        if (!C.active[sl]) continue;
        const int a = C.slot_a[sl], b = C.slot_b[sl];
        float* ma = C.inbox + C.slot_off[sl * 2];
        float* mb = C.inbox + C.slot_off[sl * 2 + 1];
        float P[NA][NB], va[NA], vb[NB];
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) P[i][j] = C.P[(size_t)(i * 6 + j) * C.cap + sl];
#pragma unroll
        for (int i = 0; i < NA; ++i) va[i] = nb_old[a * 6 + i] * fast_rcp(1e-10f + ma[i]);
#pragma unroll
        for (int j = 0; j < NB; ++j) vb[j] = nb_old[b * 6 + j] * fast_rcp(1e-10f + mb[j]);
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
        const float ra = fast_rcp(sa), rb = fast_rcp(sb);
#pragma unroll
        for (int i = 0; i < NA; ++i) ma[i] = ta[i] * ra;
#pragma unroll
        for (int j = 0; j < NB; ++j) mb[j] = tb[j] * rb;
    }
}

// pair marginals and (optionally) their Bethe free-energy terms (rotamer.cpp:405-451)
template <int NA, int NB>
__device__ __forceinline__ float bp_marginal_range(const BpCtx& C, int lo, int hi, const float* __restrict__ nbm, int tid, int nt,
                                                   bool want_energy) {
    float en = 0.f;
    for (int sl = lo + tid; sl < hi; sl += nt) {
        if (!C.active[sl]) continue;
        const int a = C.slot_a[sl], b = C.slot_b[sl];
        const float* ma = C.inbox + C.slot_off[sl * 2];
        const float* mb = C.inbox + C.slot_off[sl * 2 + 1];
        float P[NA][NB], bc1[NA], bc2[NB], mg[NA][NB], sum = 0.f;
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) P[i][j] = C.P[(size_t)(i * 6 + j) * C.cap + sl];
#pragma unroll
        for (int i = 0; i < NA; ++i) bc1[i] = nbm[a * 6 + i] * rcp(1e-10f + ma[i]);
#pragma unroll
        for (int j = 0; j < NB; ++j) bc2[j] = nbm[b * 6 + j] * rcp(1e-10f + mb[j]);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) { mg[i][j] = P[i][j] * bc1[i] * bc2[j]; sum += mg[i][j]; }
        const float rs = rcp(sum);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float pm = mg[i][j] * rs;
                C.marg[(size_t)(i * 6 + j) * C.cap + sl] = pm;
                if (want_energy) en += pm * logf((1e-10f + pm) * rcp(1e-10f + P[i][j] * nbm[a * 6 + i] * nbm[b * 6 + j]));
            }
    }
    return en;
}

#define BP_GROUP 4   // lanes cooperating on one node in the node phase

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
    int* cls = bp_start + NN + 1;                // [N_CLASS+1]
    const int n_slot = R.n_slot[s];
    long long tr_t0 = 0, tr_edge = 0, tr_node = 0, tr_pro = 0, tr_loop = 0;
    const bool trace = R.bp_trace != nullptr && tid == 0;
    if (trace) tr_t0 = wall_clock64();
    BpCtx C;
    C.cap = R.slot_cap;
    C.slot_a = R.slot_a + (size_t)s * R.slot_cap;
    C.slot_b = R.slot_b + (size_t)s * R.slot_cap;
    C.active = R.slot_active + (size_t)s * R.slot_cap;
    C.slot_off = R.slot_off + (size_t)s * R.slot_cap * 2;
    C.P = R.P + (size_t)s * R.slot_cap * 36;
    C.inbox = R.msg_cur + (size_t)s * R.slot_cap * 16;
    C.marg = R.marg + (size_t)s * R.slot_cap * 36;
    const int* adj_cnt = R.adj_cnt + (size_t)s * NN;
    const int* adj_slot = R.adj_slot + (size_t)s * NN * R.adj_cap;
    for (int i = tid; i < NN; i += nt) nrot[i] = R.node_nrot[i];
    for (int i = tid; i <= NN; i += nt) bp_start[i] = R.bp_start[(size_t)s * (NN + 1) + i];
    if (tid <= N_CLASS) cls[tid] = R.class_start[(size_t)s * (N_CLASS + 1) + tid];
    for (int i = tid; i < NN * 6; i += nt) prob[i] = R.node_prob[(size_t)s * NN * 6 + i];
    __syncthreads();

    // energies -> probabilities for the entries each class uses (rotamer.cpp:835)
    for (int i = tid; i < n_slot * 36; i += nt) {
        const int e = i / n_slot, sl = i % n_slot, ra = e / 6, rb = e % 6;
        const int c = sl < cls[1] ? CL33 : (sl < cls[2] ? CL36 : (sl < cls[3] ? CL66 : (sl < cls[4] ? CL11 : CL1X)));
        const int na = c == CL66 ? 6 : (c == CL33 || c == CL36 ? 3 : 1);
        const int nb = c == CL33 ? 3 : (c == CL11 ? 1 : 6);       // 1xN: up to 6 columns (unused ones stay exp(0) = 1, never read)
        if (ra < na && rb < nb) { const size_t pi = (size_t)e * C.cap + sl; C.P[pi] = expf(-C.P[pi]); }
    }
    // old edge beliefs = 1 (rotamer.cpp:1015-1032); also for slots without an in-range bead pair this step,
    // whose unit message then multiplies as an exact 1
    for (int i = tid; i < bp_start[NN] * 8; i += nt) C.inbox[i] = 1.f;
    __syncthreads();
    // fold edges to 1-state partners into the node probabilities (move_edge_prob_to_node2, rotamer.cpp:378-385)
    for (int g = tid; g < NN; g += nt) {
        const int n = nrot[g];
        if (n == 1) continue;
        for (int k = 0; k < adj_cnt[g]; ++k) {
            const int sl = adj_slot[g * R.adj_cap + k];
            if (!C.active[sl]) continue;
            for (int r = 0; r < n; ++r) prob[g * 6 + r] *= C.P[(size_t)r * C.cap + sl];
        }
    }
    __syncthreads();
    for (int i = tid; i < NN * 6; i += nt) { nb0[i] = prob[i]; nb1[i] = prob[i]; }   // old node belief = prob (rotamer.cpp:1009-1013)
    __syncthreads();

    float* nb_old = nb0; float* nb_cur = nb1;
    int iter = 0;
    float maxdev = 1e10f;
    const int gl = tid % BP_GROUP, n_grp = nt / BP_GROUP;
    // sweep -1 is calculate_new_beliefs(0.f, true): only its messages survive and the "old" node belief becomes
    // prob / max(prob) (rotamer.cpp:1034 with the swap at 995-1001)
    if (trace) tr_pro = wall_clock64();
    for (int sweep = -1;; ++sweep) {
        long long tr_a = 0, tr_b = 0;
        if (trace) tr_a = wall_clock64();
        // ---- edge phase: every residue pair rewrites its two messages in place from the old node beliefs
        bp_edge_range<3, 3>(C, cls[CL33], cls[CL33 + 1], nb_old, tid, nt);
        bp_edge_range<3, 6>(C, cls[CL36], cls[CL36 + 1], nb_old, tid, nt);
        bp_edge_range<6, 6>(C, cls[CL66], cls[CL66 + 1], nb_old, tid, nt);
        __syncthreads();
        if (trace) { tr_b = wall_clock64(); tr_edge += tr_b - tr_a; }
        // ---- node phase: BP_GROUP lanes per node stream the node's inbox, multiply, and combine by shuffles
        float dev = 0.f;
        for (int g0 = 0; g0 < NN; g0 += n_grp) {
            const int g = g0 + tid / BP_GROUP;
            const bool live = g < NN && nrot[g] > 1;
            const int n = live ? nrot[g] : 0;
            float bb[6] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            if (live && sweep >= 0) {
                const int k1 = bp_start[g + 1];
                int parity = 0;
                for (int k = bp_start[g] + gl; k < k1; k += BP_GROUP) {
                    const float4 m0 = *(const float4*)(C.inbox + (size_t)k * 8);
                    const float2 m1 = *(const float2*)(C.inbox + (size_t)k * 8 + 4);
                    bb[0] *= m0.x; bb[1] *= m0.y; bb[2] *= m0.z;
                    if (n == 6) { bb[3] *= m0.w; bb[4] *= m1.x; bb[5] *= m1.y; }
                    if ((++parity & 1) == 0) {          // keep the running product O(1) (rotamer.cpp:489-493 re-normalises too)
                        float mx = fmaxf(fmaxf(bb[0], bb[1]), bb[2]);
                        if (n == 6) mx = fmaxf(fmaxf(mx, bb[3]), fmaxf(bb[4], bb[5]));
                        const float rm = fast_rcp(mx);
#pragma unroll
                        for (int r = 0; r < 6; ++r) bb[r] *= rm;
                    }
                }
            }
#pragma unroll
            for (int off = BP_GROUP / 2; off > 0; off >>= 1) {
                float mx = 0.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) { bb[r] *= __shfl_xor(bb[r], off, UP_WAVE); mx = fmaxf(mx, r < n ? bb[r] : 0.f); }
                const float rm = mx > 0.f ? fast_rcp(mx) : 1.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) bb[r] *= rm;
            }
            if (live) {
                // b = prob * product, then standardize (rotamer.cpp:258-273); lane gl finishes states gl and gl+4
                float v[6], mx = 0.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) { v[r] = r < n ? prob[g * 6 + r] * bb[r] : 0.f; mx = fmaxf(mx, v[r]); }
                const float rm = rcp(mx);
                const float damp = sweep < 0 ? 0.f : R.damping;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    if (r < n && (r % BP_GROUP) == gl) {
                        const float o = nb_old[g * 6 + r];
                        const float nv = damp != 0.f ? (1.f - damp) * rm * v[r] + damp * o : rm * v[r];
                        nb_cur[g * 6 + r] = nv;
                        dev = fmaxf(nv - o, dev);                      // signed, rotamer.cpp:275-281
                    }
                }
            }
        }
        __syncthreads();
        if (trace) tr_node += wall_clock64() - tr_b;
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
    if (trace) tr_loop = wall_clock64();

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
    en += bp_marginal_range<3, 3>(C, cls[CL33], cls[CL33 + 1], nb_cur, tid, nt, want_energy);
    en += bp_marginal_range<3, 6>(C, cls[CL36], cls[CL36 + 1], nb_cur, tid, nt, want_energy);
    en += bp_marginal_range<6, 6>(C, cls[CL66], cls[CL66 + 1], nb_cur, tid, nt, want_energy);
    if (want_energy) {
        for (int sl = cls[CL11] + tid; sl < cls[CL11 + 1]; sl += nt)   // 1-1 edges (rotamer.cpp:861)
            if (C.active[sl]) en += -logf(C.P[sl]);
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
    __syncthreads();
    // leave the accumulators clean for the next force evaluation
    for (int i = tid; i < n_slot * 36; i += nt) C.P[(size_t)(i / n_slot) * C.cap + (i % n_slot)] = 0.f;
    int* active_w = R.slot_active + (size_t)s * R.slot_cap;
    int* active_last = R.slot_active_last + (size_t)s * R.slot_cap;
    for (int i = tid; i < n_slot; i += nt) { active_last[i] = active_w[i]; active_w[i] = 0; }
    if (trace) {
        long long* T = R.bp_trace + (size_t)s * 16;
        T[0] = tr_pro - tr_t0; T[1] = tr_loop - tr_pro; T[2] = wall_clock64() - tr_loop; T[3] = tr_edge; T[4] = tr_node;
        T[5] = iter; T[6] = n_slot; T[7] = bp_start[NN];
        for (int c = 0; c <= N_CLASS; ++c) T[8 + c] = cls[c];
    }
}

extern "C" int upk_rotamer_bp(const upk_launch_t* L, const upk_rotamer_t* R, int want_energy) {
    const size_t lds = ((size_t)R->n_node * 20 + 64) * sizeof(float);
    if (lds > 155 * 1024) return 9004;
    hipLaunchKernelGGL(k_rotamer_bp, dim3(1, L->n_system), dim3(BP_BLOCK), lds, ST(L), *R, want_energy);
    return launch_status();
}
