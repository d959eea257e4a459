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
//     slot in the upper bits of its list word, so the pair kernels need no table lookup;
//   * slot matrices are structure-of-arrays ([36][slot_cap]); BP messages live in an "inbox" grouped by receiving
//     node so the node update streams them;
//   * belief propagation is ONE persistent workgroup per system with node beliefs in LDS; the bead-pair kernels
//     stage the spline table and every bead (coordinates + packed metadata) of the system in LDS.
#include "device_math.h"
#include "../../include/upside_hip_kernels.h"
#include "igraph_device.h"
#include "pair2_device.h"

using namespace up;

#define ST(L) ((hipStream_t)(L)->stream)
#define BP_BLOCK 1024
#ifndef UPK_LAUNCH_STATUS_DEFINED
#define UPK_LAUNCH_STATUS_DEFINED
static inline int launch_status() { return (int)hipGetLastError(); }
#endif
#define C_OUT(c, s)  ((c).out  + (size_t)(s) * (c).n_elem * (c).stride)
#define C_SENS(c, s) ((c).sens + (size_t)(s) * (c).n_elem * (c).stride)
// Slot matrices (pair energies / exp(-E), pair marginals): entry (i, j) of slot sl at [i][sl][j] -- the 6 entries of a matrix
// row are contiguous (one 24-byte record), records of consecutive slots are adjacent.  The pair passes, whose neighbouring
// lanes hold the rotamer states j = 0..5 of one partner residue, then touch one cache line where an entry-major layout
// touched six; the solve, whose neighbouring lanes hold consecutive slots, still reads contiguous memory.
#ifdef PIDX_SLOT_MAJOR      // (experiment, round 3: one 144-byte record per slot -- belief propagation 8.1 instead of 6.4 ms, the energy pass 1.5 instead
                            //  of 1.27: lanes walk consecutive SLOTS, which the [row][slot] planes keep contiguous)
#define PIDX6(cap, sl, i, j) ((((size_t)(sl)) * 6 + (i)) * 6 + (j))
#else
#define PIDX6(cap, sl, i, j) ((((size_t)(i)) * (cap) + (sl)) * 6 + (j))
#endif
#define PIDX(R, sl, e) PIDX6((R).slot_cap, sl, (e) / 6, (e) % 6)

// slot classes in storage order
enum { CL33 = 0, CL36 = 1, CL66 = 2, CL11 = 3, CL1X = 4, N_CLASS = 5 };
__device__ __forceinline__ int slot_class(int na, int nb) {   // na <= nb
    if (na == 1) return nb == 1 ? CL11 : CL1X;
    if (na == 3) return nb == 3 ? CL33 : CL36;
    return CL66;
}

// ------------------------------------------------------------------------------------------------
// rebuild step 0: clear the node x node mark table of the flagged systems (before the list build marks it)
__device__ __forceinline__ void d_rotamer_clear_slots(const upk_rotamer_t& R, const BX B, float* lds_unused) {
    const int* fl = UPK_FLAG_LIST(R.G);
    const int n_flagged = fl[0];
    const int n16 = R.G.mark_stride / 16;
    for (int fi = B.by; fi < n_flagged; fi += B.gy) {
        uint4* m = (uint4*)(R.mark + (size_t)fl[1 + fi] * R.G.mark_stride);
        for (int i = B.bx * blockDim.x + threadIdx.x; i < n16; i += B.gx * blockDim.x) m[i] = make_uint4(0, 0, 0, 0);
    }
}
__global__ void k_rotamer_clear_slots(upk_rotamer_t R)  { d_rotamer_clear_slots(R, BX_REAL, nullptr); }
extern "C" int upk_rotamer_clear_slots(const upk_launch_t* L, const upk_rotamer_t* R) {
    // (upk_pairlist_check clears the table of a system it flags; this launcher is for callers that mark a table without that test)
    const int n16 = R->G.mark_stride / 16;
    int blocks = (n16 + 1023) / 1024; if (blocks > 64) blocks = 64;
    if (batch_add(L, BK_CLEAR_SLOTS, blocks, UPK_FLAG_GRID(L->n_system), 0, R, sizeof(*R))) return 0;
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_rotamer_clear_slots, dim3(blocks, UPK_FLAG_GRID(L->n_system)), dim3(1024), 0, ST(L), *R);
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
#ifndef SLOT_WAVES
#define SLOT_WAVES 8
#endif
__device__ __forceinline__ void d_rotamer_build_slots(const upk_rotamer_t& R, const BX B, float* bits_lds) {
    unsigned long long* bits = (unsigned long long*)bits_lds;            // [NN][W]   // (8 waves per SIMD = 64 VGPRs: two workgroups per CU)
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
    // B.gx workgroups share a system: each packs the marks and runs the scans (a few microseconds), and numbers the pairs of every
    // B.gx-th node -- the numbering is a stream of scattered 4-byte stores (eight per pair) that one CU issues one address at a time
    const int K = B.gx, kk = B.bx;
    for (int fi = B.by; fi < n_flagged; fi += B.gy) {
        const int s = fl[1 + fi];
        const unsigned char* mark = R.mark + (size_t)s * G.mark_stride;
        int* slot_of = R.slot_of + (size_t)s * NN * NN;
        __syncthreads();                                     // LDS of the previous system is no longer read
        // ---- pack the marks: a row is W * 64 bytes (G.mark_ld), so 16 bytes = 16 bits of ONE row word: every lane loads 16-byte
        // pieces (all of its loads in flight at once) and stores 16 bits each.  (Round 3 read bytes, one 64-lane ballot per word and
        // eight loads in flight per wavefront: 39 of the 97 us of a 300-node system.)
        {
            const uint4* m16 = (const uint4*)mark;
            unsigned short* bits16 = (unsigned short*)bits;
            const int n16 = NN * W * 4;
            for (int i0 = tid; i0 < n16; i0 += 4 * (int)blockDim.x) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + u * (int)blockDim.x; v[u] = i < n16 ? m16[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * (int)blockDim.x;
                    auto nib = [](unsigned x) { return (x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u); };   // bytes 0 / 1 -> 4 bits
                    if (i < n16) bits16[i] = (unsigned short)(nib(v[u].x) | (nib(v[u].y) << 4) | (nib(v[u].z) << 8) | (nib(v[u].w) << 12));
                }
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
            dbp = na == 1 ? 0 : dbp * (na == 6 ? 2 : 1);     // inbox size in 16-byte quads: 1 per message to a 3-state node, 2 to a 6-state node
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
            if (kk == 0) {
                int* cs = R.class_start + (size_t)s * (N_CLASS + 1);
                for (int c = 0; c <= N_CLASS; ++c) cs[c] = cls_lds[c] < R.slot_cap ? cls_lds[c] : R.slot_cap;
                R.n_slot[s] = acc < R.slot_cap ? acc : R.slot_cap;
                if (acc > R.slot_cap) *G.error_flag = 2;
            }
        }
        __syncthreads();
        int* bp_start = R.bp_start + (size_t)s * (NN + 1);
        if (kk == 0) for (int g = tid; g <= NN; g += blockDim.x) bp_start[g] = bp_s[g];
        // the same inbox as ROWS (one per message; a 6-state row is two quads): rows of the 3-state nodes in [0, rows3), rows of
        // the 6-state nodes from R6 = rows3 rounded up to 32 on -- so that every 32-row word of a per-solve activity mask holds
        // rows of ONE width (k_rotamer_bp<.., COMPACT>)
        const int rows3 = bp_s[e3 < NN ? e3 : NN], R6 = (rows3 + 31) & ~31;
        auto row_of = [&](int g) { return g < e3 ? bp_s[g] : R6 + ((bp_s[g] - rows3) >> 1); };   // first row of node g (g = NN: end)
        if (R.row_start && kk == 0) {
            int* row_start = R.row_start + (size_t)s * (NN + 2);
            for (int g = tid; g <= NN; g += blockDim.x) row_start[g] = g < e1 ? 0 : row_of(g);
            if (tid == 0) row_start[NN + 1] = R6;
        }
        if (a < NN && kk == 0) {
            const int d = na > 1 ? deg1[a] : 0;
            R.adj_cnt[(size_t)s * NN + a] = d < R.adj_cap ? d : R.adj_cap;
            if (d > R.adj_cap) *G.error_flag = 3;
        }
        // ---- assign: the pairs (a, b > a) of node a are numbered in ascending b; everything a pair needs follows from
        // popcounts of the two rows (messages TO node g sit at inbox quad bp_start[g] + rank among g's multi-state
        // partners x (1 quad = 4 floats for a 3-state g, 2 for a 6-state g); folded 1-state partners of g are listed
        // in ascending id).  One WAVEFRONT per (row a, word c) item, one lane per partner bit: a pair's numbers are popcounts, so
        // the partners of a word are independent (round 3 walked them one lane per node: a chain of up to 60 trips of scattered stores)
        {
            int* slot_a = R.slot_a + (size_t)s * R.slot_cap;
            int* slot_b = R.slot_b + (size_t)s * R.slot_cap;
            int* slot_off = R.slot_off + (size_t)s * R.slot_cap * 2;
            int* slot_row = R.slot_row ? R.slot_row + (size_t)s * R.slot_cap * 2 : nullptr;
            const unsigned long long lane_below = (1ull << lane) - 1ull;
            for (int item = kk * n_wave + wave; item < NN * W; item += K * n_wave) {
                const int ra = item / W, c = item - ra * W;                       // (wave-uniform)
                if (c < ra / 64) continue;
                const unsigned long long w = bits[ra * W + c] & ~below(ra + 1, c);
                if (!w) continue;
                const int nra = nrot[ra];
                const int rcend = nra == 1 ? e1 : (nra == 3 ? e3 : NN);          // end of the row node's own class
                const int cl_lo = slot_class(nra, nra), cl_hi = nra == 1 ? CL1X : (nra == 3 ? CL36 : CL66);
                const int first = nra == 1 ? 0 : (nra == 3 ? e1 : e3);           // first row of that class
                const int base_lo = first < NN ? row_lo[first] : 0, base_hi = first < NN ? row_hi[first] : 0;
                // partners of the row in the words before this one: above it in its own class / in a higher class; all of them below word c
                int n_lo = 0, n_hi = 0, n_all = 0;
                for (int c2 = 0; c2 < c; ++c2) {
                    const unsigned long long w2 = bits[ra * W + c2];
                    const unsigned long long up2 = w2 & ~below(ra + 1, c2);
                    n_lo += __popcll(up2 & below(rcend, c2)); n_hi += __popcll(up2 & ~below(rcend, c2)); n_all += __popcll(w2);
                }
                if (!((w >> lane) & 1ull)) continue;
                const int pb = c * 64 + lane;
                const bool lo = pb < rcend;
                const unsigned long long same = lo ? (w & below(rcend, c)) : (w & ~below(rcend, c));
                const int sl = (lo ? cls_lds[cl_lo] + row_lo[ra] - base_lo + n_lo : cls_lds[cl_hi] + row_hi[ra] - base_hi + n_hi) + __popcll(same & lane_below);
                const int npb = nrot[pb];
                if (sl < R.slot_cap) {
                    slot_a[sl] = ra; slot_b[sl] = pb; slot_of[(size_t)ra * NN + pb] = sl; slot_of[(size_t)pb * NN + ra] = sl;
                    if (nra == 1 && npb > 1) {
                        const int pos = rank_below(pb, ra);             // 1-state partners of b below a
                        if (pos < R.adj_cap) R.adj_slot[((size_t)s * NN + pb) * R.adj_cap + pos] = sl;
                    } else if (nra > 1) {
                        const int k_multi = n_all + __popcll(bits[ra * W + c] & lane_below) - deg1[ra];   // multi-state partners of a below b
                        const int kb = rank_below(pb, ra) - deg1[pb];
                        slot_off[sl * 2] = (bp_s[ra] + k_multi * (nra == 6 ? 2 : 1)) * 4;
                        slot_off[sl * 2 + 1] = (bp_s[pb] + kb * (npb == 6 ? 2 : 1)) * 4;
                        if (slot_row) { slot_row[sl * 2] = row_of(ra) + k_multi; slot_row[sl * 2 + 1] = row_of(pb) + kb; }
                    }
                } else { slot_of[(size_t)ra * NN + pb] = -1; slot_of[(size_t)pb * NN + ra] = -1; }
            }
        }
    }
}
__global__ void __launch_bounds__(BP_BLOCK, SLOT_WAVES) k_rotamer_build_slots(upk_rotamer_t R)  {
    extern __shared__ __attribute__((aligned(16))) float lds_dyn_[];
    d_rotamer_build_slots(R, BX_REAL, lds_dyn_);
}
extern "C" int upk_rotamer_build_slots(const upk_launch_t* L, const upk_rotamer_t* R) {
    if (R->n_node > 1024) return 9003;
    const size_t lds = (size_t)R->n_node * ((R->n_node + 63) / 64) * 8;
    // (tried in round 3: stamping the slots into the list words inside this kernel from the popcounts of the bit matrix, without
    //  the node x node table -- 1.33 ms per step against 0.53 + 0.59 for the two kernels: one workgroup per system walks its 66 k
    //  list words through a chain of dependent loads, the separate kernel spreads them over hundreds of workgroups)
    const int wgs = 1024;   // workgroups looping over the flagged systems
    // a small system's slot stamping (upk_rotamer_nbr_slots) rides behind the numbering in the same workgroup: one launch less on the upkeep chain
    if (R->G.n1 <= 512 && batch_add(L, BK_SLOTS_BOTH, 1, L->n_system < wgs ? L->n_system : wgs, lds, R, sizeof(*R))) { batch_of(L)->skip_nbr_slots = true; return 0; }
    // workgroups per system (see the kernel): several while the device has CUs to spare, one when systems fill it
    static int split = -1;    // UPSIDE_HIP_SLOT_SPLIT (experiments / tests)
    if (split < 0) { const char* e = getenv("UPSIDE_HIP_SLOT_SPLIT"); split = e ? atoi(e) : 0; if (split < 0 || split > 16) split = 0; }
    const int K = split ? split : ((L->n_system <= 256 && R->n_node > 64) ? 4 : 1);
    if (batch_add(L, BK_BUILD_SLOTS, K, L->n_system < wgs ? L->n_system : wgs, lds, R, sizeof(*R))) return 0;
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_rotamer_build_slots, dim3(K, L->n_system < wgs ? L->n_system : wgs), dim3(BP_BLOCK), lds, ST(L), *R);
    return launch_status();
}

// rebuild step 3: every cached bead pair remembers its slot, in the bits of its list word above the bead index
// (one 4-byte word per cached pair instead of a second array; the pair passes read nothing else)
__device__ __forceinline__ void d_rotamer_nbr_slots(const upk_rotamer_t& R, const BX B, float* lds_unused) {
    const upk_igraph_t& G = R.G;
    const int* fl = UPK_FLAG_LIST(G);
    const int n_flagged = fl[0];
    const int NN = R.n_node;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_wave = blockDim.x >> 6;
    for (int fi = B.by; fi < n_flagged; fi += B.gy) {
        const int s = fl[1 + fi];
        const int* slot_of = R.slot_of + (size_t)s * NN * NN;
        // (a word is a chain of three dependent loads -- word, node of the partner, slot of the node pair -- and a row is rarely longer
        //  than one trip: four rows per wavefront at a time, so that every lane has four chains in flight: 0.55 -> 0.31 ms per step at 4096 systems)
        constexpr int RB = 4;
        for (int row0 = (B.bx * n_wave + wave) * RB; row0 < G.n1; row0 += B.gx * n_wave * RB) {
            int cnt[RB], a[RB], cmax = 0;
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int row = row0 + u < G.n1 ? row0 + u : row0;
                cnt[u] = row0 + u < G.n1 ? G.cnt1[(size_t)s * G.n1 + row] : 0;
                a[u] = R.bead_node[row];
                cmax = cnt[u] > cmax ? cnt[u] : cmax;
            }
            for (int k = lane; k < cmax; k += 64) {
                int j[RB], nb[RB], sl[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) j[u] = k < cnt[u] ? G.nbr1[((size_t)s * G.n1 + row0 + u) * G.cap1 + k] : 0;     // freshly built: a bare bead index
#pragma unroll
                for (int u = 0; u < RB; ++u) nb[u] = R.bead_node[j[u]];
#pragma unroll
                for (int u = 0; u < RB; ++u) sl[u] = slot_of[a[u] * NN + nb[u]];
#pragma unroll
                for (int u = 0; u < RB; ++u)
                    if (k < cnt[u]) G.nbr1[((size_t)s * G.n1 + row0 + u) * G.cap1 + k] = j[u] | ((sl[u] < 0 ? UPK_ROT_SLOT_NONE : sl[u]) << UPK_ROT_J_BITS);
            }
        }
    }
}
__global__ void k_rotamer_nbr_slots(upk_rotamer_t R)  { d_rotamer_nbr_slots(R, BX_REAL, nullptr); }
extern "C" int upk_rotamer_nbr_slots(const upk_launch_t* L, const upk_rotamer_t* R) {
    if (batch_of(L) && batch_of(L)->skip_nbr_slots) { batch_of(L)->skip_nbr_slots = false; return 0; }      // (done by the BK_SLOTS_BOTH item just queued)
    if (batch_add(L, BK_NBR_SLOTS, (R->G.n1 + 63) / 64, UPK_FLAG_GRID(L->n_system), 0, R, sizeof(*R))) return 0;      // (16 wavefronts x 4 rows)
    UPK_FLUSH(L);
    int blocks = (R->G.n1 + 15) / 16;          // 4 wavefronts x 4 rows
    hipLaunchKernelGGL(k_rotamer_nbr_slots, dim3(blocks, UPK_FLAG_GRID(L->n_system)), dim3(256), 0, ST(L), *R);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// 1-body energies -> node probabilities (rotamer.cpp:811-826, 239-256)
// one node's 1-body energies -> probabilities; pr[6] out
__device__ __forceinline__ void rotamer_node_prob_one(const upk_rotamer_t& R, int s, int g, float* pr_out) {
    const int* __restrict__ nb_start = R.node_bead_start; const int* __restrict__ nb_list = R.node_bead_list;
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
    for (int r = 0; r < 6; ++r) { const float v = r < n_rot ? expf(off - e[r]) : 0.f; pr[r] = v; pr_out[r] = v; }
    R.node_off[(size_t)s * R.n_node + g] = off;
}
__global__ void k_rotamer_node_prob(upk_rotamer_t R) {
    const int s = blockIdx.y;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0 && R.bp_bar) R.bp_bar[s] = 0;             // cluster barrier counter of the solve that follows
    if (g >= R.n_node) return;
    float pr[6];
    rotamer_node_prob_one(R, s, g, pr);
}
extern "C" int upk_rotamer_node_prob(const upk_launch_t* L, const upk_rotamer_t* R) {
    if (R->node_prob_in_solve) return 0;     // (the one-workgroup solve computes them in its prologue)
    UPK_FLUSH(L);
    hipLaunchKernelGGL(k_rotamer_node_prob, dim3((R->n_node + 255) / 256, L->n_system), dim3(256), 0, ST(L), *R);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// bead-pair passes over this step's hit lists (igraph_device.h): table + all beads of the system in LDS.  Bead row:
// [0,6) pos+dir, [6] type | rot<<8 | nrot<<12, [7] node id (raw int bits).  A hit-list word is
// partner bead | residue-pair slot << UPK_ROT_J_BITS; the lists hold each pair once (partner above the row).
// The pair table is staged as its upper triangle (stage_table_sym): 34 instead of 64 KB for the 20 x 20 x 40 table, which is
// what lets the gradient pass keep 48 bytes of exact accumulators per bead next to it.
// STAGED = false (systems whose beads do not fit LDS next to the table): the rows are read from a packed global copy
// written by k_rotamer_pack_beads and the gradient accumulates in global memory.
struct RotLds { float* tab; const float* rows; unsigned long long* acc; int* range; unsigned short* ord; int* counter; };
template <bool STAGED, bool POLY = false, bool SENTINEL = false>
__device__ __forceinline__ RotLds rot_stage(const upk_rotamer_t& R, float* lds, int s, int tab_floats, bool want_acc) {
    const upk_igraph_t& G = R.G;
    RotLds r;
    r.tab = lds;
    float* p = lds + ((tab_floats + 3) & ~3);
    stage_table(r.tab, POLY ? R.param_tri_poly : R.param_tri, tab_floats);
    r.acc = nullptr;
    if (STAGED) {
        if (want_acc) { r.acc = (unsigned long long*)p; p += G.n1 * 12; for (int t = threadIdx.x; t < G.n1 * 6; t += blockDim.x) r.acc[t] = 0ull; }
        r.rows = p; p += (G.n1 + (SENTINEL ? 1 : 0)) * 8;
        if (SENTINEL)     // the packed passes: two 16-byte planes + the sentinel row (type 0, state 0 of 1, node 0)
            stage_rows_planes((float*)r.rows, G.node1, s, G.loc1, G.n1, 6, R.bead_node, R.bead_meta, nullptr, 0, __int_as_float(1 << 12), __int_as_float(0));
        else stage_rows((float*)r.rows, G.node1, s, G.loc1, G.n1, 6, R.bead_node, R.bead_meta, nullptr, 0);
    } else r.rows = R.bead_pack + (size_t)s * G.n1 * 8;
    r.range = (int*)p; r.ord = (unsigned short*)(r.range + G.n1); r.counter = r.range + PG_WALK_LDS_WORDS(G.n1);
    stage_ranges(r.range, r.ord, G.hcnt1 + (size_t)s * G.n1, nullptr, G.ord1 + (size_t)s * G.n1, G.n1);
    if (threadIdx.x == 0) *r.counter = 0;
    __syncthreads();
    return r;
}
// parameter row of the bead pair (row type tr, partner type to) in the triangle table, and where the two beads' angular
// coefficients start in it
template <bool POLY = false>
__device__ __forceinline__ const float* rot_param_row(const upk_rotamer_t& R, const float* tab, int tr, int to, int& off_row, int& off_oth) {
    const bool sw = tr > to;
    const int ka = POLY ? 4 * (R.G.n_knot_angular - 3) : R.G.n_knot_angular;     // length of one angular block in the row
    off_row = sw ? ka : 0; off_oth = sw ? 0 : ka;
    return tab + tri_row(sw ? to : tr, sw ? tr : to, R.G.n_type1) * (POLY ? R.n_poly : R.G.n_param);
}
__global__ void k_rotamer_pack_beads(upk_rotamer_t R) {
    const int s = blockIdx.y, n = R.G.n1;
    const float* base = R.G.node1.out + (size_t)s * R.G.node1.n_elem * R.G.node1.stride;
    float* out = R.bead_pack + (size_t)s * n * 8;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n * 8; t += gridDim.x * blockDim.x) {
        const int i = t >> 3, c = t & 7;
        out[t] = c < 6 ? base[(size_t)R.G.loc1[i] * R.G.node1.stride + c]
                       : __int_as_float(c == 6 ? R.bead_meta[i] : R.bead_node[i]);
    }
}

// bead-pair energies into the slot matrices (interaction_graph.h:470-503 + rotamer.cpp:832-846)
template <bool POLY>
struct RotEnergyOp {
    const upk_rotamer_t& R; const QuadShape Q; const RotLds& L;
    float* P; int* active;
    float xr[8];
    __device__ __forceinline__ RotEnergyOp(const upk_rotamer_t& R_, const RotLds& L_, int s)
        : R(R_), Q(quad_shape(R_.G)), L(L_), P(R_.P + (size_t)s * R_.slot_cap * 36), active(R_.slot_active + (size_t)s * R_.slot_cap) {}
    __device__ __forceinline__ void begin(int row) { load_row8(xr, L.rows + row * 8); }
    __device__ __forceinline__ void body(int, int w, bool live) {
        const int j = w & ((1 << UPK_ROT_J_BITS) - 1), sl = (int)((unsigned)w >> UPK_ROT_J_BITS);
        float xo[8];
        load_row8(xo, L.rows + j * 8);
        const int mr = __float_as_int(xr[6]), a = __float_as_int(xr[7]);
        const int mo = __float_as_int(xo[6]), b = __float_as_int(xo[7]);
        int o1, o2;
        const float* p = rot_param_row<POLY>(R, L.tab, mr & 0xFF, mo & 0xFF, o1, o2);   // row < partner: types [type(i1)][type(i2)], i1 < i2
        const float E = quadspline_pair<0, POLY>(Q, p, xr, xo, nullptr, nullptr, nullptr, o1, o2);
        if (!live || sl == UPK_ROT_SLOT_NONE) return;         // (no slot: only after a capacity overflow, the error flag is set)
        const int ra = (mr >> 8) & 0xF, rb = (mo >> 8) & 0xF;
        float* pe = P + PIDX6(R.slot_cap, sl, a < b ? ra : rb, a < b ? rb : ra);
        // With one bead per rotamer state (every shipped side-chain library) an entry has a single writer and a plain
        // store does; device-scope float atomics on scattered lines cost 0.56 ms of this kernel's 1.37 at 1024 systems.
        if (R.one_bead_per_state) *pe = R.p_prob ? expf(-E) : E;
        else atomicAdd(pe, E);
        active[sl] = 1;
    }
    __device__ __forceinline__ void flush(int) {}
};
template <bool STAGED, bool POLY>
__global__ void __launch_bounds__(1024) PG_KERNEL_ATTR k_rotamer_pair_energy(upk_rotamer_t R, int tab_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const upk_igraph_t& G = R.G;
    const RotLds L = rot_stage<STAGED, POLY>(R, lds, s, tab_floats, false);
    RotEnergyOp<POLY> op(R, L, s);
    group_batch_loop(op, G.n1, L.ord, L.range, G.hit1 + (size_t)s * G.n1 * G.cap1, G.cap1, L.counter, blockIdx.x, gridDim.x);
}

// 1 = table + beads (+ accumulators) staged in LDS, 0 = beads read from the packed global copy, -1 = not even the table fits
// ---- packed passes (pair2_device.h): two partners per lane, 4-lane row groups ------------------------------------------
// hit-list word -> partner bead, slot; the bead rows carry type | rot << 8 | nrot << 12 in [6] and the node id in [7]
struct RotPairMeta { int j, sl, mo, b; };
__device__ __forceinline__ RotPairMeta rot_pair_meta(int w, const float* xo) {
    RotPairMeta m;
    m.j = w & ((1 << UPK_ROT_J_BITS) - 1); m.sl = (int)((unsigned)w >> UPK_ROT_J_BITS);
    m.mo = __float_as_int(xo[6]); m.b = __float_as_int(xo[7]);
    return m;
}
template <bool POLY>
struct RotGradOp2 {
    const upk_rotamer_t& R; const QuadShape Q; const RotLds& L;
    const float* marg; const float* nbm;
    v2 x1[6], acc[6]; int mr, a;
    __device__ __forceinline__ RotGradOp2(const upk_rotamer_t& R_, const RotLds& L_, int s)
        : R(R_), Q(quad_shape(R_.G)), L(L_), marg(R_.marg + (size_t)s * R_.slot_cap * 36), nbm(R_.nb_cur + (size_t)s * R_.n_node * 6) {}
    __device__ __forceinline__ void begin(int row) {
        float xr[8]; load_row8_planes(xr, L.rows, R.G.n1 + 1, row);
#pragma unroll
        for (int c = 0; c < 6; ++c) { x1[c] = bc2(xr[c]); acc[c] = bc2(0.f); }
        mr = __float_as_int(xr[6]); a = __float_as_int(xr[7]);
    }
    // pair sensitivity = pair marginal of the two rotamer states (the node marginal when one side has a single state,
    // rotamer.cpp:956-966): ONE unconditional load from a selected address, issued before the functor
    __device__ __forceinline__ float sens_of(const RotPairMeta& m, bool live) const {
        const int ra = (mr >> 8) & 0xF, na = (mr >> 12) & 0xF, rb = (m.mo >> 8) & 0xF, nb = (m.mo >> 12) & 0xF;
        const bool both1 = na == 1 && nb == 1, multi = na > 1 && nb > 1, no_slot = multi && m.sl == UPK_ROT_SLOT_NONE;
        // (selects on integers, no divergent address code: the two tables differ in base and offset only)
        const bool lo = a < m.b;
        const int off_m = (int)PIDX6(R.slot_cap, no_slot ? 0 : m.sl, lo ? ra : rb, lo ? rb : ra);
        const int off_n = na == 1 ? m.b * 6 + rb : a * 6 + ra;
        const uintptr_t base = multi ? (uintptr_t)marg : (uintptr_t)nbm;
        const float pv = ((const float*)base)[multi ? off_m : off_n];
        return !live ? 0.f : (both1 ? 1.f : (no_slot ? 0.f : pv));
    }
    __device__ __forceinline__ void body(int, int wA, int wB, bool liveA, bool liveB) {
        float xa[8], xb[8];
        load_row8_planes(xa, L.rows, R.G.n1 + 1, wA & ((1 << UPK_ROT_J_BITS) - 1));
        load_row8_planes(xb, L.rows, R.G.n1 + 1, wB & ((1 << UPK_ROT_J_BITS) - 1));
        const RotPairMeta mA = rot_pair_meta(wA, xa), mB = rot_pair_meta(wB, xb);
        const v2 ps = mk2(sens_of(mA, liveA), sens_of(mB, liveB));
        v2 x2[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) x2[c] = mk2(xa[c], xb[c]);
        int o1A, o2A, o1B, o2B;
        const float* pA = rot_param_row<POLY>(R, L.tab, mr & 0xFF, mA.mo & 0xFF, o1A, o2A);
        const float* pB = rot_param_row<POLY>(R, L.tab, mr & 0xFF, mB.mo & 0xFF, o1B, o2B);
        v2 dd[3], g1[3], g2[3];
        quadspline_pair2<true, POLY>(Q, pA, pB, x1, x2, dd, g1, g2, o1A, o2A, o1B, o2B);
        v2 od[6];                                           // the partners' shares
        const v2 pss = ps * bc2(P2_FIX_SCALE);              // (the partners' shares leave in fixed point: scaled here, two per instruction)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            od[c] = pss * dd[c]; od[3 + c] = pss * g2[c];
            acc[c] = fma2(-ps, dd[c], acc[c]); acc[3 + c] = fma2(ps, g1[c], acc[3 + c]);
        }
        // accumulators as six planes [component][bead]: the consecutive partners of a row group fall into distinct banks
        const int n1 = R.G.n1;
        if (liveA) {
#pragma unroll
            for (int c = 0; c < 6; ++c) lds_add_fixed22_scaled(L.acc + c * n1 + mA.j, od[c].x);
        }
        if (liveB) {
#pragma unroll
            for (int c = 0; c < 6; ++c) lds_add_fixed22_scaled(L.acc + c * n1 + mB.j, od[c].y);
        }
    }
    __device__ __forceinline__ void flush(int row) {
        float t[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) t[c] = group_sum4(acc[c].x + acc[c].y);
        if ((threadIdx.x & (P2_LANES - 1)) != 0) return;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (!(fabsf(t[c]) < 1e30f)) *R.G.error_flag = 8;     // NaN / overflow would vanish in the integer conversion: report the step
            lds_add_fixed22_wide(L.acc + c * R.G.n1 + row, t[c]);
        }
    }
};
template <bool POLY>
__global__ void __launch_bounds__(1024) PG_KERNEL_ATTR k_rotamer_grad2(upk_rotamer_t R, int tab_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const upk_igraph_t& G = R.G;
    const RotLds L = rot_stage<true, POLY, true>(R, lds, s, tab_floats, true);
    {
        RotGradOp2<POLY> op(R, L, s);
        group2_batch_loop(op, G.n1, L.ord, L.range, G.hit1 + (size_t)s * G.n1 * G.cap1, G.cap1, L.counter, blockIdx.x, gridDim.x, G.n1);
    }
    __syncthreads();
    float* sens = C_SENS(G.node1, s);
    unsigned long long* gacc = R.grad_acc + (size_t)s * G.n1 * 6;
    const bool alone = gridDim.x == 1;     // the system's only workgroup: its accumulators are the totals
    if (alone && (G.node1.stride & 3) == 0) {
        // (element by element: the bead's row index, its six sums, one 16-byte read-modify-write pair of its sens row -- the first form walked
        //  the 6 n accumulators one by one, seven trips per lane, each a chain of two dependent global loads)
        for (int i = threadIdx.x; i < G.n1; i += blockDim.x) {
            float add[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) add[c] = from_fixed22(L.acc[c * G.n1 + i]);
            float4* row = (float4*)(sens + (size_t)G.loc1[i] * G.node1.stride);
            float4 r0 = row[0], r1 = row[1];
            r0.x += add[0]; r0.y += add[1]; r0.z += add[2]; r0.w += add[3]; r1.x += add[4]; r1.y += add[5];
            row[0] = r0; row[1] = r1;
        }
    } else
    for (int t = threadIdx.x; t < G.n1 * 6; t += blockDim.x) {
        const int i = t / 6, c = t - i * 6;
        const unsigned long long a = L.acc[c * G.n1 + i];
        if (!a) continue;
        if (alone) sens[(size_t)G.loc1[i] * G.node1.stride + c] += from_fixed22(a);
        else atomicAdd(gacc + t, a);       // several workgroups share the system: exact partial sums, k_rotamer_grad_finish converts
    }
    if (alone) {   // the node marginal of the bead's rotamer state goes to the 1-body parents (rotamer.cpp:968-984)
        const float* nbm = R.nb_cur + (size_t)s * R.n_node * 6;
        for (int i = threadIdx.x; i < G.n1; i += blockDim.x) {
            const float mg = nbm[R.bead_node[i] * 6 + ((R.bead_meta[i] >> 8) & 0xF)];
            const int loc = G.loc1[i];
            // (the parents' values first, then the stores: the arrays may alias as far as the compiler knows, which made every parent a
            //  load -> store chain of its own)
            constexpr int KB = 4;
            for (int k0 = 0; k0 < R.n_prob; k0 += KB) {
                float* q[KB]; float v[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int k = k0 + u < R.n_prob ? k0 + u : k0;
                    q[u] = R.prob_sens[k] + (size_t)s * R.prob_sys_stride[k] + (size_t)loc * R.prob_stride[k];
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) v[u] = *q[u];
#pragma unroll
                for (int u = 0; u < KB; ++u) if (k0 + u < R.n_prob) *q[u] = v[u] + mg;
            }
        }
    }
}

static bool rot_pair2_enabled() {          // UPSIDE_HIP_PAIR2=0: the scalar passes (one partner per lane) -- A/B and tests
    static int v = -1;
    if (v < 0) { const char* e = getenv("UPSIDE_HIP_PAIR2"); v = (e && !atoi(e)) ? 0 : 1; }
    return v != 0;
}
static int rot_geometry(const upk_launch_t* L, const upk_rotamer_t* R, bool want_acc, int& tab_floats, size_t& lds_bytes, dim3& grid, dim3& block,
                        bool poly = false, bool pair2 = false) {
    const int nt = R->G.n_type1;
    tab_floats = (nt * (nt + 1) / 2) * (poly ? R->n_poly : R->G.n_param);
    const size_t fixed = ((size_t)((tab_floats + 3) & ~3) + PG_WALK_LDS_WORDS(R->G.n1) + 4) * sizeof(float);
    static int force_unstaged = -1;   // UPSIDE_HIP_ROT_UNSTAGED=1 exercises the large-system path
    if (force_unstaged < 0) { const char* e = getenv("UPSIDE_HIP_ROT_UNSTAGED"); force_unstaged = (e && atoi(e)) ? 1 : 0; }
    int staged = 1;
    lds_bytes = fixed + (size_t)R->G.n1 * (want_acc ? 20 : 8) * sizeof(float) + (pair2 ? 32 : 0);
    if (lds_bytes > 158 * 1024 || force_unstaged || R->bead_pack) { staged = 0; lds_bytes = fixed; }   // (a system whose gradient pass needs the packed copy uses it in both passes)
    if (lds_bytes > 158 * 1024 || (!staged && !R->bead_pack)) return -1;
    int bps, threads;
    if (pair2) pair2_geometry(L->n_system, R->G.n1, bps, threads);
    else pair_geometry(L->n_system, R->G.n1, bps, threads);
    grid = dim3(bps, L->n_system); block = dim3(threads);
    return staged;
}
extern "C" int upk_rotamer_pair_energy(const upk_launch_t* L, const upk_rotamer_t* R) {
    if (!list_words_match(&R->G)) return 9010;
    UPK_FLUSH(L);
    int tab_floats; size_t lds; dim3 grid, block;
    static int no_poly = -1;      // UPSIDE_HIP_ROT_POLY=0 keeps the energy pass on the spline-coefficient table (A/B and the large-table path)
    if (no_poly < 0) { const char* e = getenv("UPSIDE_HIP_ROT_POLY"); no_poly = (e && !atoi(e)) ? 1 : 0; }
    // (the packed two-partners-per-lane form of THIS pass was built in round 3 and removed in round 6: the pass is bound by its scattered
    //  pair-matrix stores, and 16 rows x 4 partners per store instruction touch 1.7x the cache lines of 8 rows x 8 partners: 1.45 against 1.26 ms)
    if (R->param_tri_poly && !no_poly && rot_geometry(L, R, false, tab_floats, lds, grid, block, true) == 1) {   // polynomial table + beads fit LDS
        hipLaunchKernelGGL((k_rotamer_pair_energy<true, true>), grid, block, lds, ST(L), *R, tab_floats);
        return launch_status();
    }
    const int staged = rot_geometry(L, R, false, tab_floats, lds, grid, block);
    if (staged < 0) return 9005;   // interaction table larger than LDS
    if (staged) hipLaunchKernelGGL((k_rotamer_pair_energy<true, false>), grid, block, lds, ST(L), *R, tab_floats);
    else {
        hipLaunchKernelGGL(k_rotamer_pack_beads, dim3((R->G.n1 * 8 + 255) / 256, L->n_system), dim3(256), 0, ST(L), *R);   // also serves upk_rotamer_grad
        hipLaunchKernelGGL((k_rotamer_pair_energy<false, false>), grid, block, lds, ST(L), *R, tab_floats);
    }
    return launch_status();
}

// derivative push (rotamer.cpp:956-985 + interaction_graph.h:525-555): ONE visit per in-range bead pair yields both beads'
// gradients, weighted by the pair sensitivity = pair marginal of the two rotamer states (the node marginal when one side has
// a single state).  The row bead's share accumulates in registers over its hits; the partner's goes through 64-bit integer
// LDS atomics as exact fixed point (to_fixed32), so the per-bead totals do not depend on the order in which pairs arrive.
template <bool STAGED>
struct RotGradOp {
    const upk_rotamer_t& R; const QuadShape Q; const RotLds& L;
    const float* marg; const float* nbm; unsigned long long* gacc;
    float xr[8], acc[6];
    __device__ __forceinline__ RotGradOp(const upk_rotamer_t& R_, const RotLds& L_, int s)
        : R(R_), Q(quad_shape(R_.G)), L(L_), marg(R_.marg + (size_t)s * R_.slot_cap * 36), nbm(R_.nb_cur + (size_t)s * R_.n_node * 6),
          gacc(R_.grad_acc + (size_t)s * R_.G.n1 * 6) {}
    __device__ __forceinline__ void add(int bead, int c, float v) const {
        if (STAGED) lds_add_fixed(L.acc + bead * 6 + c, v);
        else atomicAdd(gacc + bead * 6 + c, to_fixed32(v));
    }
    __device__ __forceinline__ void begin(int row) {
        load_row8(xr, L.rows + row * 8);
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] = 0.f;
    }
    __device__ __forceinline__ void body(int, int w, bool live) {
        const int j = w & ((1 << UPK_ROT_J_BITS) - 1), sl = (int)((unsigned)w >> UPK_ROT_J_BITS);
        float xo[8];
        load_row8(xo, L.rows + j * 8);
        const int mr = __float_as_int(xr[6]), a = __float_as_int(xr[7]);
        const int mo = __float_as_int(xo[6]), b = __float_as_int(xo[7]);
        const int ra = (mr >> 8) & 0xF, na = (mr >> 12) & 0xF, rb = (mo >> 8) & 0xF, nb = (mo >> 12) & 0xF;
        // the pair sensitivity is a gather from global memory, issued before the functor: ONE unconditional load from a
        // selected address (node marginal of the multi-state side, or the pair marginal)
        const bool both1 = na == 1 && nb == 1, no_slot = na > 1 && nb > 1 && sl == UPK_ROT_SLOT_NONE;
        const float* pp = na == 1 ? nbm + b * 6 + rb : (nb == 1 ? nbm + a * 6 + ra : marg + PIDX6(R.slot_cap, no_slot ? 0 : sl, a < b ? ra : rb, a < b ? rb : ra));
        const float pv = *pp;
        const float ps = both1 ? 1.f : (no_slot ? 0.f : pv);
        int o1, o2;
        const float* p = rot_param_row(R, L.tab, mr & 0xFF, mo & 0xFF, o1, o2);
        float dd[3], g1[3], g2[3];
        quadspline_pair<3>(Q, p, xr, xo, dd, g1, g2, o1, o2);
#pragma unroll
        for (int c = 0; c < 3; ++c) { acc[c] = live ? fmaf(-ps, dd[c], acc[c]) : acc[c]; acc[3 + c] = live ? fmaf(ps, g1[c], acc[3 + c]) : acc[3 + c]; }
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { add(j, c, ps * dd[c]); add(j, 3 + c, ps * g2[c]); }
        }
    }
    __device__ __forceinline__ void flush(int row) {
        float t[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) t[c] = group_sum(acc[c]);
        if ((threadIdx.x & (PG_LANES - 1)) != 0) return;
#pragma unroll
        for (int c = 0; c < 6; ++c) add(row, c, t[c]);
    }
};
template <bool STAGED>
__global__ void __launch_bounds__(1024) PG_KERNEL_ATTR k_rotamer_grad(upk_rotamer_t R, int tab_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y;
    const upk_igraph_t& G = R.G;
    const RotLds L = rot_stage<STAGED>(R, lds, s, tab_floats, true);
    {
        RotGradOp<STAGED> op(R, L, s);
        group_batch_loop(op, G.n1, L.ord, L.range, G.hit1 + (size_t)s * G.n1 * G.cap1, G.cap1, L.counter, blockIdx.x, gridDim.x);
    }
    if (!STAGED) return;                   // accumulated in global memory: k_rotamer_grad_finish converts
    __syncthreads();
    float* sens = C_SENS(G.node1, s);
    unsigned long long* gacc = R.grad_acc + (size_t)s * G.n1 * 6;
    const bool alone = gridDim.x == 1;     // the system's only workgroup: its accumulators are the totals
    for (int t = threadIdx.x; t < G.n1 * 6; t += blockDim.x) {
        const unsigned long long a = L.acc[t];
        if (!a) continue;
        if (alone) { const int i = t / 6, c = t - i * 6; sens[(size_t)G.loc1[i] * G.node1.stride + c] += from_fixed32(a); }
        else atomicAdd(gacc + t, a);       // several workgroups share the system: exact partial sums, k_rotamer_grad_finish converts
    }
    if (alone) {   // the node marginal of the bead's rotamer state goes to the 1-body parents (rotamer.cpp:968-984)
        const float* nbm = R.nb_cur + (size_t)s * R.n_node * 6;
        for (int i = threadIdx.x; i < G.n1; i += blockDim.x) {
            const float mg = nbm[R.bead_node[i] * 6 + ((R.bead_meta[i] >> 8) & 0xF)];
            const int loc = G.loc1[i];
            for (int k = 0; k < R.n_prob; ++k) R.prob_sens[k][(size_t)s * R.prob_sys_stride[k] + (size_t)loc * R.prob_stride[k]] += mg;
        }
    }
}
// global accumulators -> sens (and cleared for the next evaluation); the 1-body marginal push of the systems that took this path
__global__ void k_rotamer_grad_finish(upk_rotamer_t R, double unit) {   // unit: value of one accumulator count (2^-32 or 2^-22)
    const int s = blockIdx.y;
    const upk_igraph_t& G = R.G;
    float* sens = C_SENS(G.node1, s);
    unsigned long long* gacc = R.grad_acc + (size_t)s * G.n1 * 6;
    const float* nbm = R.nb_cur + (size_t)s * R.n_node * 6;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < G.n1 * 6; t += gridDim.x * blockDim.x) {
        const int i = t / 6, c = t - i * 6;
        const unsigned long long a = gacc[t];
        const int loc = G.loc1[i];
        if (a) { sens[(size_t)loc * G.node1.stride + c] += (float)((double)(long long)a * unit); gacc[t] = 0ull; }
        if (c == 0) {
            const float mg = nbm[R.bead_node[i] * 6 + ((R.bead_meta[i] >> 8) & 0xF)];
            for (int k = 0; k < R.n_prob; ++k) R.prob_sens[k][(size_t)s * R.prob_sys_stride[k] + (size_t)loc * R.prob_stride[k]] += mg;
        }
    }
}

extern "C" int upk_rotamer_grad(const upk_launch_t* L, const upk_rotamer_t* R) {
    if (!list_words_match(&R->G)) return 9010;
    UPK_FLUSH(L);
    int tab_floats; size_t lds; dim3 grid, block;
    const double unit32 = 1.0 / 4294967296.0, unit22 = 1.0 / (double)(1 << P2_FIX_BITS);
    if (rot_pair2_enabled()) {                     // packed passes: polynomial table if it fits beside the accumulators, else spline coefficients
        for (int poly = 1; poly >= 0; --poly) {
            if (poly && !R->param_tri_poly) continue;
            if (rot_geometry(L, R, true, tab_floats, lds, grid, block, poly != 0, true) != 1) continue;
            if (poly) hipLaunchKernelGGL(k_rotamer_grad2<true>, grid, block, lds, ST(L), *R, tab_floats);
            else hipLaunchKernelGGL(k_rotamer_grad2<false>, grid, block, lds, ST(L), *R, tab_floats);
            if (grid.x > 1) hipLaunchKernelGGL(k_rotamer_grad_finish, dim3((R->G.n1 * 6 + 255) / 256, L->n_system), dim3(256), 0, ST(L), *R, unit22);
            return launch_status();
        }
    }
    const int staged = rot_geometry(L, R, true, tab_floats, lds, grid, block);
    if (staged < 0) return 9005;
    if (staged) hipLaunchKernelGGL(k_rotamer_grad<true>, grid, block, lds, ST(L), *R, tab_floats);
    else hipLaunchKernelGGL(k_rotamer_grad<false>, grid, block, lds, ST(L), *R, tab_floats);   // beads packed by upk_rotamer_pair_energy this step
    if (!staged || grid.x > 1)
        hipLaunchKernelGGL(k_rotamer_grad_finish, dim3((R->G.n1 * 6 + 255) / 256, L->n_system), dim3(256), 0, ST(L), *R, unit32);
    return launch_status();
}

// parameter derivative of one system (rotamer.cpp:1064-1066 -> interaction_graph.h:404-416): every in-range bead pair
// (i1 < i2, types [type(i1)][type(i2)]) weighted by the pair sensitivity of rotamer.cpp:956-966 -- the same weight
// k_rotamer_grad uses.  Reads the beads, the converged beliefs and pair marginals of the last solve.
__global__ void k_rotamer_param_deriv(upk_rotamer_t R, int s, float* __restrict__ table) {
    const upk_igraph_t& G = R.G;
    const int lane = threadIdx.x & 63;
    const float cut2 = G.cutoff * G.cutoff;
    const float* base = G.node1.out + (size_t)s * G.node1.n_elem * G.node1.stride;
    const float* marg = R.marg + (size_t)s * R.slot_cap * 36;
    const float* nbm = R.nb_cur + (size_t)s * R.n_node * 6;
    const QuadShape Q = quad_shape(G);
    for (int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < G.n1; row += gridDim.x * (blockDim.x >> 6)) {
        const int* nbr = G.nbr1 + ((size_t)s * G.n1 + row) * G.cap1;
        const int cnt = G.cnt1[(size_t)s * G.n1 + row];
        float xr[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) xr[c] = base[(size_t)G.loc1[row] * G.node1.stride + c];
        const int mr = R.bead_meta[row], a = R.bead_node[row];
        const int ra = (mr >> 8) & 0xF, na = (mr >> 12) & 0xF;
        for (int k = lane; k < cnt; k += 64) {
            const int w = nbr[k];
            const int j = w & ((1 << UPK_ROT_J_BITS) - 1), sl = (int)((unsigned)w >> UPK_ROT_J_BITS);   // (the list holds partners above the row only)
            float xo[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) xo[c] = base[(size_t)G.loc1[j] * G.node1.stride + c];
            if (!(dist2_exact(xr[0], xr[1], xr[2], xo[0], xo[1], xo[2]) < cut2)) continue;
            const int mo = R.bead_meta[j], b = R.bead_node[j];
            const int rb = (mo >> 8) & 0xF, nb = (mo >> 12) & 0xF;
            float ps;
            if (na == 1 && nb == 1) ps = 1.f;
            else if (na == 1) ps = nbm[b * 6 + rb];
            else if (nb == 1) ps = nbm[a * 6 + ra];
            else ps = sl == UPK_ROT_SLOT_NONE ? 0.f : marg[PIDX6(R.slot_cap, sl, a < b ? ra : rb, a < b ? rb : ra)];
            // the reference's edge is (i1 < i2) in the configuration's own bead order: types [type(i1)][type(i2)], x1 = bead i1
            const bool row_first = !R.bead_orig || R.bead_orig[row] < R.bead_orig[j];
            const size_t prow = (size_t)(row_first ? (mr & 0xFF) * G.n_type2 + (mo & 0xFF) : (mo & 0xFF) * G.n_type2 + (mr & 0xFF)) * G.n_param;
            if (row_first) quadspline_param_accum(Q, G.param + prow, xr, xo, ps, table + prow);
            else quadspline_param_accum(Q, G.param + prow, xo, xr, ps, table + prow);
        }
    }
}
extern "C" int upk_rotamer_param_deriv(const upk_launch_t* L, const upk_rotamer_t* R, int system, float* table) {
    UPK_FLUSH(L);
    if (system < 0 || system >= L->n_system) return 9101;
    hipLaunchKernelGGL(k_rotamer_param_deriv, dim3((R->G.n1 + 3) / 4), dim3(256), 0, ST(L), *R, system, table);
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

// write-through / agent-scope helpers of the cluster solve (protocol: see k_rotamer_bp_cluster)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    const uintptr_t p = (uintptr_t)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// 16- / 8-byte sc1 loads: served by L2 past this CU's L1, so that what another workgroup published with sc1 stores is seen without an
// L1 invalidate (MI355X_MICROARCH.md, inter-workgroup visibility: "sc1 loads may replace the acquire only when the producer stored sc1")
__device__ __forceinline__ float4 ld_wt16(__amdgpu_buffer_rsrc_t r, int float_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, 16);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 ld_wt8(__amdgpu_buffer_rsrc_t r, int float_off) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, float_off * 4, 0, 16);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ void st_wt16(__amdgpu_buffer_rsrc_t r, int float_off, float a, float b, float c, float d) {   // 16-byte sc1 store
    u32x4 v; v.x = __float_as_uint(a); v.y = __float_as_uint(b); v.z = __float_as_uint(c); v.w = __float_as_uint(d);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, float_off * 4, 0, 16);
}

// zero the accumulator entries a slot class uses, for slots [lo, hi)
template <int NA, int NB>
__device__ __forceinline__ void clear_class(float* P, int cap, int lo, int hi, int tid, int nt) {
    const int n = hi - lo;
    for (int i = tid; i < n * NA * NB; i += nt) {
        const int e = i / n, l = i - e * n;
        P[PIDX6(cap, lo + l, e / NB, e % NB)] = 0.f;
    }
}
// end of a solve: zero the accumulator entries of the slots that were written this step and hand their flags on
template <int NA, int NB>
__device__ __forceinline__ void retire_class(float* P, int cap, int lo, int hi, int* __restrict__ active_w, int* __restrict__ active_last, int tid, int nt, float rest) {
    for (int sl = lo + tid; sl < hi; sl += nt) {
        const int act = active_w[sl];
        active_last[sl] = act; active_w[sl] = 0;
        if (act) {
#pragma unroll
            for (int e = 0; e < NA * NB; ++e) P[PIDX6(cap, sl, e / NB, e % NB)] = rest;
        }
    }
}
__device__ __forceinline__ void clear_all_classes(float* P, int cap, const int* cls, int part, int n_part, int tid, int nt) {
    auto sub = [&](int c, int& lo, int& hi) { const int b = cls[c], n = cls[c + 1] - b; lo = b + (int)((long)n * part / n_part); hi = b + (int)((long)n * (part + 1) / n_part); };
    int lo, hi;
    sub(CL33, lo, hi); clear_class<3, 3>(P, cap, lo, hi, tid, nt);
    sub(CL36, lo, hi); clear_class<3, 6>(P, cap, lo, hi, tid, nt);
    sub(CL66, lo, hi); clear_class<6, 6>(P, cap, lo, hi, tid, nt);
    sub(CL11, lo, hi); clear_class<1, 1>(P, cap, lo, hi, tid, nt);
    sub(CL1X, lo, hi); clear_class<1, 6>(P, cap, lo, hi, tid, nt);
}

struct BpCtx {
    const int *slot_a, *slot_b, *active, *slot_off;
    const int* slot_row = nullptr;
    float *P, *inbox, *marg;
    int cap;
    const int4* rec = nullptr;     // one-workgroup solve: the active slots of each class, packed (see bp_pack_active)
    // the first lds_floats floats of the inbox (messages to the 3-state nodes come first) may live in LDS instead of
    // global memory: the one-workgroup solve re-reads and rewrites them every sweep
    float* inbox_lds = nullptr; int lds_floats = 0;
    // (a pointer selected between two address spaces: FLAT accesses.  Round 3 instantiated the rest of the solve a second time for an inbox
    //  that fits LDS as a whole -- plain ds_read / ds_write there --, the branch taken once per solve: 6.01 against 5.98 ms, no gain)
    static constexpr int W3 = 4;       // floats per message row to a 3-state node
    __device__ __forceinline__ float* msg(int off) const { return off < lds_floats ? inbox_lds + off : inbox + off; }
};
// ALL: the whole inbox of this solve sits in LDS (decided per solve, once the dense layout is known): message rows are then plain LDS
// accesses (ds_read / ds_write) instead of the flat ones a two-address-space pointer costs -- measured with every row in LDS,
// 150 residues: edge phase 4.99 -> 3.72 us, node phase 4.40 -> 3.70 us per sweep (one system 2.40 -> 2.62 k steps/s, 512 systems
// 277 -> 296 k); 300 residues / 7 A x 512: 261 -> 272 k.  The solve below is instantiated for both.
// W3_: 3 = rows to 3-state nodes hold 3 floats instead of 4 (two LDS instructions per row instead of one b128, 11 KB less for the
// benchmark protein): taken only when that is what makes the inbox fit (then ALL is set as well).
template <bool ALL, int W3_ = 4> struct BpCtxT : BpCtx {
    static constexpr int W3 = W3_;
    __device__ __forceinline__ float* msg(int off) const { if (ALL) return inbox_lds + off; return off < lds_floats ? inbox_lds + off : inbox + off; }
};

// One message row, wide: a row to a 3-state node is 4 floats at a 16-byte boundary (one b128 access, the 4th word unused), a
// row to a 6-state node 6 or 8 floats at an 8-byte boundary (three b64 accesses) -- in both inbox layouts.  (Dword accesses
// cost an LDS instruction each: the dense inbox of 3 / 6 dwords per row kept the LDS index unit busy 52 % of the solve.)
template <int N, int W3 = 4> __device__ __forceinline__ void bp_load_row(const float* p, float* v) {
    if (N == 3 && W3 == 3) { v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; }       // (12-byte rows: dword aligned only)
    else if (N == 3) { const float4 a = *(const float4*)p; v[0] = a.x; v[1] = a.y; v[2] = a.z; }
    else { const float2 a = ((const float2*)p)[0], b = ((const float2*)p)[1], c = ((const float2*)p)[2]; v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; }
}
template <int N, int W3 = 4> __device__ __forceinline__ void bp_store_row(float* p, const float* v) {
    if (N == 3 && W3 == 3) { p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; }
    else if (N == 3) *(float4*)p = make_float4(v[0], v[1], v[2], 1.f);
    else { ((float2*)p)[0] = make_float2(v[0], v[1]); ((float2*)p)[1] = make_float2(v[2], v[3]); ((float2*)p)[2] = make_float2(v[4], v[5]); }
}
// edge phase over one class range: new messages from the old beliefs (update_beliefs, rotamer.cpp:468-499 and the
// L1 normalisation of 506-521), rewritten in place.  1-ulp hardware reciprocals: the reference itself uses the
// 12-bit rcpps here (Float4.h:199-212).
// a node's belief row in LDS: NS floats per node (8: one b128 for three states, + one b64 for six; 6: dword reads)
template <int N, int NS> __device__ __forceinline__ void bp_load_nb(const float* p, float* v) {
    if (NS == 8) {
        const float4 a = *(const float4*)p; v[0] = a.x; v[1] = a.y; v[2] = a.z;
        if (N == 6) { const float2 b = *(const float2*)(p + 4); v[3] = a.w; v[4] = b.x; v[5] = b.y; }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = p[i];
    }
}
template <int NA, int NB, bool WT, int NS = 6, typename CTX = BpCtx>   // WT: messages leave through 16-byte write-through stores (cluster solve)
__device__ __forceinline__ void bp_edge_slot(const CTX& C, int oa, int ob, int a, int b, const float (&P)[NA * NB], const float* __restrict__ nb_old,
                                             __amdgpu_buffer_rsrc_t inbox_w) {
    float* pa = C.msg(oa);
    float* pb = C.msg(ob);
    float ma[NA], mb[NB];
    bp_load_row<NA, CTX::W3>(pa, ma); bp_load_row<NB, CTX::W3>(pb, mb);
    float va[NA], vb[NB];
    bp_load_nb<NA, NS>(nb_old + a * NS, va); bp_load_nb<NB, NS>(nb_old + b * NS, vb);
#pragma unroll
    for (int i = 0; i < NA; ++i) va[i] *= fast_rcp(1e-10f + ma[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) vb[j] *= fast_rcp(1e-10f + mb[j]);
    float ta[NA], tb[NB], sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) { float t = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) t += P[i * NB + j] * vb[j];
        ta[i] = t; sa += t; }
#pragma unroll
    for (int j = 0; j < NB; ++j) { float t = 0.f;
#pragma unroll
        for (int i = 0; i < NA; ++i) t += va[i] * P[i * NB + j];
        tb[j] = t; sb += t; }
    // (tried: both products on explicit v_pk_fma_f32 pairs -- the compiler's own pairing already issues as many packed
    //  operations, no change in time, and the re-associated row sums cost the bit-identity between the solve variants)
    const float ra = fast_rcp(sa), rb = fast_rcp(sb);
    if (WT) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ta[i] *= ra;
#pragma unroll
        for (int j = 0; j < NB; ++j) tb[j] *= rb;
        st_wt16(inbox_w, oa, ta[0], ta[1], ta[2], NA == 6 ? ta[NA - 3] : 1.f);
        if (NA == 6) st_wt16(inbox_w, oa + 4, ta[NA - 2], ta[NA - 1], 1.f, 1.f);
        st_wt16(inbox_w, ob, tb[0], tb[1], tb[2], NB == 6 ? tb[NB - 3] : 1.f);
        if (NB == 6) st_wt16(inbox_w, ob + 4, tb[NB - 2], tb[NB - 1], 1.f, 1.f);
    } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) ta[i] *= ra;
#pragma unroll
        for (int j = 0; j < NB; ++j) tb[j] *= rb;
        bp_store_row<NA, CTX::W3>(pa, ta); bp_store_row<NB, CTX::W3>(pb, tb);
    }
}
// the NA x NB entries of one slot, row-major, from the [36][cap] table
template <int NA, int NB>
__device__ __forceinline__ void bp_load_matrix(const BpCtx& C, int sl, float (&P)[NA * NB]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {      // one 24-byte record per matrix row: 8-byte loads
        const float2* r = (const float2*)(C.P + PIDX6(C.cap, sl, i, 0));
        const float2 a = r[0]; P[i * NB] = a.x; P[i * NB + 1] = a.y;
        if (NB == 6) { const float2 b = r[1], c = r[2]; P[i * NB + 2] = b.x; P[i * NB + 3] = b.y; P[i * NB + 4] = c.x; P[i * NB + 5] = c.y; }
        else P[i * NB + 2] = ((const float*)r)[2];
    }
}
template <int NA, int NB, bool WT, int NS = 6, typename CTX = BpCtx>
__device__ __forceinline__ void bp_edge_range_impl(const CTX& C, int lo, int hi, const float* __restrict__ nb_old, int tid, int nt,
                                                   __amdgpu_buffer_rsrc_t inbox_w) {
    // (measured and rejected: fetching the next slot's flag and message offsets one trip ahead.  In the 6x6 instance it
    // spills 58 VGPRs (sweep 34.5 -> 47.7 us at 1024 systems); in the 3x3 / 3x6 instances alone it costs 1 % of the
    // benchmark -- under load the sweep is limited by memory throughput, not by the two-trip chain)
    for (int sl = lo + tid; sl < hi; sl += nt) {
        const int act = C.active[sl], oa = C.slot_off[sl * 2], ob = C.slot_off[sl * 2 + 1];
        if (!act) continue;
        float P[NA * NB];
        bp_load_matrix<NA, NB>(C, sl, P);
        bp_edge_slot<NA, NB, WT, NS>(C, oa, ob, C.slot_a[sl], C.slot_b[sl], P, nb_old, inbox_w);
    }
}
// Active slots of the multi-state classes, packed once per solve: the sweeps then read one 16-byte record per ACTIVE slot
// instead of 20 bytes of flags, offsets and node ids per CACHED slot (a quarter of the cached residue pairs has no bead
// pair in range on a given step), and trips and pinned registers are spent on active slots only.  Order-preserving (the
// free-energy sum keeps its summation order from run to run): 64-slot chunks are counted by ballot, one barrier, then
// every chunk finds its base as the sum of the counts before it in its class.  chunk_cnt: LDS scratch, one int per chunk.
template <typename OffFn>      // off(sl, side): float offset of the slot's message row to node a (side 0) / node b (side 1)
__device__ __forceinline__ void bp_pack_active(const BpCtx& C, int4* __restrict__ rec, const int* cls, int* n_act, int* chunk_cnt, int tid, int nt, OffFn off) {
    const int lane = tid & 63, wave = tid >> 6, n_wave = nt >> 6;
    const int nc0 = (cls[CL33 + 1] - cls[CL33] + 63) >> 6, nc1 = (cls[CL36 + 1] - cls[CL36] + 63) >> 6, nc2 = (cls[CL66 + 1] - cls[CL66] + 63) >> 6;
    for (int pass = 0; pass < 2; ++pass) {
        for (int ch = wave; ch < nc0 + nc1 + nc2; ch += n_wave) {
            const int c = ch < nc0 ? CL33 : (ch < nc0 + nc1 ? CL36 : CL66);
            const int first = c == CL33 ? 0 : (c == CL36 ? nc0 : nc0 + nc1);          // first chunk of the class
            const int sl = cls[c] + (ch - first) * 64 + lane;
            const bool act = sl < cls[c + 1] && C.active[sl] != 0;
            const unsigned long long b = __ballot(act);
            if (pass == 0) { if (lane == 0) chunk_cnt[ch] = __popcll(b); continue; }
            int before = 0;
            for (int k = first + lane; k < ch; k += 64) before += chunk_cnt[k];
            before = __builtin_amdgcn_readfirstlane((int)wave_sum((float)before));      // counts are < 2^24: exact in fp32
            if (act) rec[cls[c] + before + __popcll(b & ((1ull << lane) - 1ull))] =
                make_int4(off(sl, 0), off(sl, 1), C.slot_a[sl] | (C.slot_b[sl] << 16), sl);
            if (lane == 0 && ch == first + (c == CL33 ? nc0 : (c == CL36 ? nc1 : nc2)) - 1) n_act[c] = before + __popcll(b);
        }
        __syncthreads();
    }
}
template <int NA, int NB, int NS = 6, typename CTX = BpCtx>
__device__ __forceinline__ void bp_edge_packed(const CTX& C, int first, int end, const float* __restrict__ nb_old, int tid, int nt,
                                               __amdgpu_buffer_rsrc_t rs) {
#ifndef BP_REC_AHEAD
#define BP_REC_AHEAD 1
#endif
#ifndef BP_MATRIX_AHEAD
#define BP_MATRIX_AHEAD 1
#endif
    // A trip is a chain of dependent round trips: record -> matrix + messages -> stores.  At two wavefronts per SIMD nothing
    // else hides them, so the NEXT trip's record (16 bytes) is fetched while this one is processed (solve 8.3 -> 7.9 ms at 4096
    // systems), and in the 3x3 / 3x6 instances, whose registers are not the kernel's peak, the next trip's matrix as well
    // (which needs the record after that one step earlier still).
    if (BP_MATRIX_AHEAD && NA * NB <= 18) {
        int idx = first + tid;
        if (idx >= end) return;
        int4 r0 = C.rec[idx], r1 = r0;
        if (idx + nt < end) r1 = C.rec[idx + nt];
        float P0[NA * NB];
        bp_load_matrix<NA, NB>(C, r0.w, P0);
        for (; idx < end; idx += nt) {
            int4 r2 = r1;
            if (idx + 2 * nt < end) r2 = C.rec[idx + 2 * nt];
            float P1[NA * NB];
            bp_load_matrix<NA, NB>(C, r1.w, P1);                  // (the last trip re-reads its own matrix: harmless)
            bp_edge_slot<NA, NB, false, NS>(C, r0.x, r0.y, r0.z & 0xffff, r0.z >> 16, P0, nb_old, rs);
            r0 = r1; r1 = r2;
#pragma unroll
            for (int e = 0; e < NA * NB; ++e) P0[e] = P1[e];
        }
        return;
    }
    int idx = first + tid;
    int4 r = idx < end ? C.rec[idx] : make_int4(0, 0, 0, 0);
    for (; idx < end; idx += nt) {
        int4 rn = r;
        if (BP_REC_AHEAD && idx + nt < end) rn = C.rec[idx + nt];
        float P[NA * NB];
        bp_load_matrix<NA, NB>(C, r.w, P);
        bp_edge_slot<NA, NB, false, NS>(C, r.x, r.y, r.z & 0xffff, r.z >> 16, P, nb_old, rs);
        if (BP_REC_AHEAD) r = rn; else if (idx + nt < end) r = C.rec[idx + nt];
    }
}
template <int NA, int NB, int NS = 6, typename CTX = BpCtx>
__device__ __forceinline__ void bp_edge_range(const CTX& C, int lo, int hi, const float* __restrict__ nb_old, int tid, int nt) {
    bp_edge_range_impl<NA, NB, false, NS>(C, lo, hi, nb_old, tid, nt, make_rsrc(C.inbox, 0u));
}
// energies -> probabilities, in place, for slots [lo, hi) of one class (rotamer.cpp:835)
// One slot per lane and trip: all NA*NB loads of the slot are issued before the first store, so a trip costs one
// memory round trip instead of NA*NB dependent ones (the in-place update otherwise serialises load -> store -> load).
template <int NA, int NB>
__device__ __forceinline__ void exp_class(float* P, int cap, int lo, int hi, int tid, int nt, const int* __restrict__ active = nullptr) {
    for (int sl = lo + tid; sl < hi; sl += nt) {
        if (active && !active[sl]) continue;      // untouched accumulators stay 0: nobody reads the matrix of an inactive slot
        float v[NA * NB];
#pragma unroll
        for (int e = 0; e < NA * NB; ++e) v[e] = P[PIDX6(cap, sl, e / NB, e % NB)];
#pragma unroll
        for (int e = 0; e < NA * NB; ++e) P[PIDX6(cap, sl, e / NB, e % NB)] = expf(-v[e]);
    }
}

// pair marginals and (optionally) their Bethe free-energy terms (rotamer.cpp:405-451)
template <int NA, int NB, int NS = 6, typename CTX = BpCtx>
__device__ __forceinline__ float bp_marginal_slot(const CTX& C, int sl, int oa, int ob, int a, int b, const float (&P)[NA * NB],
                                                  const float* __restrict__ nbm, bool want_energy) {
    float en = 0.f;
    float ma[NA], mb[NB];
    bp_load_row<NA, CTX::W3>(C.msg(oa), ma); bp_load_row<NB, CTX::W3>(C.msg(ob), mb);
    // the unnormalised marginals are formed twice (sum, then store) rather than kept: 36 fewer live registers next
    // to the resident matrices, and the products round identically both times
    float bc1[NA], bc2[NB], sum = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i) bc1[i] = nbm[a * NS + i] * rcp(1e-10f + ma[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) bc2[j] = nbm[b * NS + j] * rcp(1e-10f + mb[j]);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) sum += P[i * NB + j] * bc1[i] * bc2[j];
    const float rs = rcp(sum);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float pm = (P[i * NB + j] * bc1[i] * bc2[j]) * rs;
            C.marg[PIDX6(C.cap, sl, i, j)] = pm;
            if (want_energy) en += pm * logf((1e-10f + pm) * rcp(1e-10f + P[i * NB + j] * nbm[a * NS + i] * nbm[b * NS + j]));
        }
    return en;
}
template <int NA, int NB, int NS = 6, typename CTX = BpCtx>
__device__ __forceinline__ float bp_marginal_range(const CTX& C, int lo, int hi, const float* __restrict__ nbm, int tid, int nt,
                                                   bool want_energy) {
    float en = 0.f;
    for (int sl = lo + tid; sl < hi; sl += nt) {
        if (!C.active[sl]) continue;
        float P[NA * NB];
        bp_load_matrix<NA, NB>(C, sl, P);
        en += bp_marginal_slot<NA, NB, NS>(C, sl, C.slot_off[sl * 2], C.slot_off[sl * 2 + 1], C.slot_a[sl], C.slot_b[sl], P, nbm, want_energy);
    }
    return en;
}

template <int NA, int NB, int NS = 6, typename CTX = BpCtx>
__device__ __forceinline__ float bp_marginal_packed(const CTX& C, int first, int end, const float* __restrict__ nbm, int tid, int nt, bool want_energy) {
    float en = 0.f;
    int idx = first + tid;                                         // (the next trip's record is fetched one trip ahead, as in bp_edge_packed)
    int4 r = idx < end ? C.rec[idx] : make_int4(0, 0, 0, 0);
    for (; idx < end; idx += nt) {
        int4 rn = r;
        if (idx + nt < end) rn = C.rec[idx + nt];
        float P[NA * NB];
        bp_load_matrix<NA, NB>(C, r.w, P);
        en += bp_marginal_slot<NA, NB, NS>(C, r.w, r.x, r.y, r.z & 0xffff, r.z >> 16, P, nbm, want_energy);
        r = rn;
    }
    return en;
}
// end of a solve, packed classes: the matrices of the ACTIVE slots back to their resting value, straight from the records (no
// flag loads in front of the stores); the flags themselves are handed on by retire_flags
template <int NA, int NB>
__device__ __forceinline__ void retire_packed(const BpCtx& C, int first, int end, int tid, int nt, float rest) {
    for (int idx = first + tid; idx < end; idx += nt) {
        const int sl = C.rec[idx].w;
#pragma unroll
        for (int e = 0; e < NA * NB; ++e) C.P[PIDX6(C.cap, sl, e / NB, e % NB)] = rest;
    }
}
__device__ __forceinline__ void retire_flags(int lo, int hi, int* __restrict__ active_w, int* __restrict__ active_last, int tid, int nt) {
    for (int sl = lo + tid; sl < hi; sl += nt) { active_last[sl] = active_w[sl]; active_w[sl] = 0; }
}

// the running message product of a node (states 0..n-1 of bb) scaled by a power of two so that its largest entry lies in [0.5, 1): exact
__device__ __forceinline__ void bp_rescale_pow2(float (&bb)[6], int n) {
    float mx = fmaxf(fmaxf(bb[0], bb[1]), bb[2]);
    if (n == 6) mx = fmaxf(fmaxf(mx, bb[3]), fmaxf(bb[4], bb[5]));
    const int e = -__builtin_amdgcn_frexp_expf(mx);          // (0 for an all-zero product)
#pragma unroll
    for (int r = 0; r < 6; ++r) bb[r] = __builtin_amdgcn_ldexpf(bb[r], e);
}
#define BP_GROUP 4   // lanes cooperating on one node in the node phase (upper bound: the combine is the butterfly of a quad)
// ... on a 6-state / a 3-state node.  The 140 3-state nodes of the benchmark protein need two rounds at four lanes each (128 + 12) and
// one at two; measured at 4096 systems, solve in ms by (6-state, 3-state) lanes: (4,4) 6.36, (4,2) 6.14, (4,1) 6.66, (2,2) 7.28, (2,1) 6.92
#ifndef BP_GROUP6
#define BP_GROUP6 4
#endif
#ifndef BP_EDGE_REVERSE
#define BP_EDGE_REVERSE 1
#endif
#ifndef BP_GROUP3
#define BP_GROUP3 2
#endif
#ifndef BP_NODE_STRIDE
#define BP_NODE_STRIDE 6      // (8 = one b128 + one b64 per node row: measured equal, and 6 leaves 7 KB more of the LDS to the inbox)
#endif
#define BP_NODE_ARRAYS 2      // node arrays of BP_NODE_STRIDE floats in the one-workgroup solve's LDS: probabilities, beliefs
#ifndef BP_NODE_ROWS_512
#define BP_NODE_ROWS_512 4
#endif
#ifndef BP_WIDE_NODE_GROUPS
#define BP_WIDE_NODE_GROUPS 1
#endif

// Pair matrices pinned in registers for the whole solve.  The edge phase is bandwidth bound (at 1024 systems every CU
// streams ~0.6 MB per sweep, half of it exp(-E) matrices that never change during the solve), and the register file
// of a CU (512 KB) is three times its LDS: a workgroup of BP_BLOCK / 2 lanes may use 256 VGPRs per lane, enough to keep the
// matrices of its first K trips through a class, their message offsets and node ids for all sweeps and the marginals.
template <int NA, int NB, int K>
struct BpResident {
    float P[K > 0 ? K : 1][NA * NB];
    int oa[K > 0 ? K : 1], ob[K > 0 ? K : 1], ab[K > 0 ? K : 1], sl[K > 0 ? K : 1];     // ab = a | b << 16; -1: no slot in this trip
    __device__ __forceinline__ void load(const BpCtx& C, int first, int end, int tid, int nt) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int idx = first + tid + k * nt;
            ab[k] = -1; oa[k] = 0; ob[k] = 0; sl[k] = 0;
#pragma unroll
            for (int e = 0; e < NA * NB; ++e) P[k][e] = 0.f;
            if (idx < end) {
                const int4 r = C.rec[idx];
                bp_load_matrix<NA, NB>(C, r.w, P[k]);
                oa[k] = r.x; ob[k] = r.y; ab[k] = r.z; sl[k] = r.w;
            }
        }
    }
    template <int NS, typename CTX> __device__ __forceinline__ void edge(const CTX& C, const float* __restrict__ nb_old, __amdgpu_buffer_rsrc_t rs) const {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (ab[k] >= 0) bp_edge_slot<NA, NB, false, NS>(C, oa[k], ob[k], ab[k] & 0xffff, ab[k] >> 16, P[k], nb_old, rs);
    }
    template <int NS, typename CTX> __device__ __forceinline__ float marginal(const CTX& C, const float* __restrict__ nbm, bool want_energy) const {
        float en = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (ab[k] >= 0) en += bp_marginal_slot<NA, NB, NS>(C, sl[k], oa[k], ob[k], ab[k] & 0xffff, ab[k] >> 16, P[k], nbm, want_energy);
        return en;
    }
};

// The per-solve DENSE layout of the message inbox (COMPACT solves): which rows carry a message in this solve, where each active row sits,
// and the packed records of the active slots.  In: bp_start = first ROW of every node (LDS), cls, the activity flags.  Out: bp_start =
// first FLOAT of every node, n_act, C.rec / the records in R.bp_rec, and the sizes returned.  Runs inside the solve (small batches) or, from
// 512 systems on, as k_rotamer_bp_layout in front of it: the layout is a chain of small dependent steps (flags -> row pairs -> bit
// matrix -> two block scans -> records) that a workgroup owning a whole CU walks no faster than one of four sharing it -- 22 us of a
// 303 us solve at one workgroup per CU, sixteen rounds per launch at 4096 systems.
struct BpLayout { int inbox_floats, inbox_floats3, w3; };
template <int BLOCK>
__device__ __forceinline__ BpLayout bp_dense_layout(const upk_rotamer_t& R, const int s, BpCtx& C, const int NN, const int* cls, int* n_act, int* bp_start,
                                                    float* scratch, float* nb0, const int lds_msg_floats, const int tid, const int nt) {
    constexpr int NS = BP_NODE_STRIDE;
    int inbox_floats, inbox_floats3, w3 = 4;
    // Which rows carry a message in this solve: an activity bit per cached row (LDS, in the region the messages will take),
    // a prefix sum over its 32-row words (every word holds rows of one width: build_slots pads the 3-state block to 32), and
    // the dense position of row r is  wbase[r / 32] + width * popcount(bits of the word below r).
    const int R6 = R.row_start[(size_t)s * (NN + 2) + NN + 1], n_rows = bp_start[NN], n_words = (n_rows + 31) >> 5;
    unsigned* rmask = (unsigned*)C.inbox_lds;      // [n_words]
    int* wbase = (int*)(rmask + n_words);          // [n_words + 1]
    for (int i = tid; i < n_words; i += nt) rmask[i] = 0u;
    __syncthreads();
    // The slots are visited in the 64-slot chunks of the packing (chunk ch belongs to wavefront ch mod n_wave): ONE pass loads the
    // activity flags -- four chunks' flags in flight per lane, then their row pairs --, counts each chunk, marks the rows and keeps
    // the flags as a bit per visit; the records are written from those bits once the layout is known.  (The first form walked the
    // slots three times, every visit a dependent global round trip: 36 -> 21 us of the solve's 430; all visits in flight at once,
    // kept in registers for the record pass, measured no better: 24 us.)
    constexpr int PK_MAXIT = BLOCK <= 256 ? 32 : 16, PK_UNR = 4;      // (visits per wavefront kept as bits of one word: at most 32)
    const int pk_lane = tid & 63, pk_wave = tid >> 6, pk_nwave = nt >> 6;
    const int nc0 = (cls[CL33 + 1] - cls[CL33] + 63) >> 6, nc1 = (cls[CL36 + 1] - cls[CL36] + 63) >> 6, nc2 = (cls[CL66 + 1] - cls[CL66] + 63) >> 6;
    const int n_chunk = nc0 + nc1 + nc2, n_it = (n_chunk + pk_nwave - 1) / pk_nwave;
    const bool fast_pack = n_it <= PK_MAXIT && 2 * n_chunk <= NN * NS;
    int* chunk_cnt = (int*)nb0; int* chunk_base = chunk_cnt + n_chunk;     // (nb0 / nb1 are filled after the fold below)
    auto chunk_of = [&](int it, int& ch, int& c, int& first, int& sl) -> bool {       // visit `it` of this wavefront; false past the last chunk
        ch = pk_wave + it * pk_nwave;
        c = ch < nc0 ? CL33 : (ch < nc0 + nc1 ? CL36 : CL66);
        first = c == CL33 ? 0 : (c == CL36 ? nc0 : nc0 + nc1);
        sl = cls[c] + (ch - first) * 64 + pk_lane;
        return ch < n_chunk;
    };
    unsigned actbits = 0u;
    if (fast_pack) {
#pragma unroll
        for (int it0 = 0; it0 < PK_MAXIT; it0 += PK_UNR) {
            if (it0 >= n_it) break;
            int ch[PK_UNR], sl[PK_UNR], fl[PK_UNR]; bool ok[PK_UNR]; int2 rr[PK_UNR];
#pragma unroll
            for (int u = 0; u < PK_UNR; ++u) { int c, first; ok[u] = chunk_of(it0 + u, ch[u], c, first, sl[u]); ok[u] = ok[u] && sl[u] < cls[c + 1]; }
#pragma unroll
            for (int u = 0; u < PK_UNR; ++u) fl[u] = ok[u] ? C.active[sl[u]] : 0;
#pragma unroll
            for (int u = 0; u < PK_UNR; ++u) rr[u] = fl[u] ? ((const int2*)C.slot_row)[sl[u]] : make_int2(0, 0);
#pragma unroll
            for (int u = 0; u < PK_UNR; ++u) {
                const unsigned long long b = __ballot(fl[u] != 0);
                if (pk_lane == 0 && ch[u] < n_chunk) chunk_cnt[ch[u]] = __popcll(b);
                if (fl[u]) {
                    actbits |= 1u << (it0 + u);
                    atomicOr(&rmask[rr[u].x >> 5], 1u << (rr[u].x & 31)); atomicOr(&rmask[rr[u].y >> 5], 1u << (rr[u].y & 31));
                }
            }
        }
    } else {
        for (int sl = cls[CL33] + tid; sl < cls[CL66 + 1]; sl += nt)
            if (C.active[sl]) {
                const int ra = C.slot_row[sl * 2], rb = C.slot_row[sl * 2 + 1];
                atomicOr(&rmask[ra >> 5], 1u << (ra & 31)); atomicOr(&rmask[rb >> 5], 1u << (rb & 31));
            }
    }
    __syncthreads();
    if (fast_pack && tid < n_chunk) {          // records of the class in front of chunk tid
        const int first = tid < nc0 ? 0 : (tid < nc0 + nc1 ? nc0 : nc0 + nc1);
        int before = 0;
        for (int k = first; k < tid; ++k) before += chunk_cnt[k];
        chunk_base[tid] = before;
    }
    {
        // active rows to 3-state nodes (T3) and to 6-state nodes (T6): two scans over row counts, then the row width of the 3-state
        // block is chosen -- 4 floats if the inbox then fits the LDS (or does not fit either way), else 3 if that makes it fit --
        // and the 6-state block starts at an even float (8-byte accesses)
        const int wpl = (n_words + nt - 1) / nt, w0 = tid * wpl;       // words per lane, consecutive
        int c3 = 0, c6 = 0;
        for (int k = 0; k < wpl; ++k) { const int w = w0 + k; if (w < n_words) { const int c = __popc(rmask[w]); if (w * 32 < R6) c3 += c; else c6 += c; } }
        int T3, T6;
        int x3 = block_excl_scan(c3, (int*)scratch, &T3);
        int x6 = block_excl_scan(c6, (int*)scratch, &T6);
        w3 = (4 * T3 + 6 * T6 > lds_msg_floats && 3 * T3 + (T3 & 1) + 6 * T6 <= lds_msg_floats) ? 3 : 4;
        const int base6 = w3 * T3 + (w3 == 3 ? (T3 & 1) : 0);
        for (int k = 0; k < wpl; ++k) { const int w = w0 + k; if (w < n_words) {
            const int c = __popc(rmask[w]);
            if (w * 32 < R6) { wbase[w] = w3 * x3; x3 += c; } else { wbase[w] = base6 + 6 * x6; x6 += c; } } }
        inbox_floats = base6 + 6 * T6; inbox_floats3 = base6;
        if (tid == 0) wbase[n_words] = inbox_floats;
    }
    __syncthreads();
    auto dense = [&](int r) { const int w = r >> 5; return w >= n_words ? wbase[n_words] : wbase[w] + (w * 32 < R6 ? w3 : 6) * __popc(rmask[w] & ((1u << (r & 31)) - 1u)); };
    int my_start[(1024 + BLOCK - 1) / BLOCK + 1];          // first message float of the nodes this lane copies (NN <= 1024)
#pragma unroll
    for (int k = 0; k < (int)(sizeof(my_start) / sizeof(int)); ++k) { const int g = tid + k * nt; my_start[k] = g <= NN ? dense(bp_start[g]) : 0; }
    int4* rec = (int4*)R.bp_rec + (size_t)s * R.slot_cap;
    C.rec = rec;
    if (fast_pack) {
#pragma unroll
        for (int it0 = 0; it0 < PK_MAXIT; it0 += PK_UNR) {
            if (it0 >= n_it) break;
            int ch[PK_UNR], cc[PK_UNR], sl[PK_UNR], sa[PK_UNR], sb[PK_UNR]; bool act[PK_UNR]; int2 rr[PK_UNR];
#pragma unroll
            for (int u = 0; u < PK_UNR; ++u) { int first; chunk_of(it0 + u, ch[u], cc[u], first, sl[u]); act[u] = (actbits >> (it0 + u)) & 1u; }
#pragma unroll
            for (int u = 0; u < PK_UNR; ++u) {
                rr[u] = act[u] ? ((const int2*)C.slot_row)[sl[u]] : make_int2(0, 0);
                sa[u] = act[u] ? C.slot_a[sl[u]] : 0; sb[u] = act[u] ? C.slot_b[sl[u]] : 0;
            }
#pragma unroll
            for (int u = 0; u < PK_UNR; ++u) {
                if (ch[u] >= n_chunk) continue;                              // (wave-uniform)
                const unsigned long long b = __ballot(act[u]);
                const int base = chunk_base[ch[u]];
                if (act[u]) rec[cls[cc[u]] + base + __popcll(b & ((1ull << pk_lane) - 1ull))] = make_int4(dense(rr[u].x), dense(rr[u].y), sa[u] | (sb[u] << 16), sl[u]);
                const int last = cc[u] == CL33 ? nc0 - 1 : (cc[u] == CL36 ? nc0 + nc1 - 1 : n_chunk - 1);
                if (pk_lane == 0 && ch[u] == last) n_act[cc[u]] = base + __popcll(b);
            }
        }
        __syncthreads();
    } else
        bp_pack_active(C, rec, cls, n_act, (int*)nb0, tid, nt, [&](int sl, int side) { return dense(C.slot_row[sl * 2 + side]); });   // (ends with a barrier)
#pragma unroll
    for (int k = 0; k < (int)(sizeof(my_start) / sizeof(int)); ++k) { const int g = tid + k * nt; if (g <= NN) bp_start[g] = my_start[k]; }   // rows -> floats
    return BpLayout{inbox_floats, inbox_floats3, w3};
}
// [S][bp_layout_stride(n_node)] ints (upk_rotamer_t::bp_layout): first float of every node [n_node + 1], active slots per class [3], inbox
// floats, floats of the rows to 3-state nodes, row width of those rows; from n_node + 8 on the folded node probabilities [n_node][6]
__host__ __device__ static inline int bp_layout_stride(int n_node) { return (UPK_BP_LAYOUT_PER_NODE * n_node + UPK_BP_LAYOUT_EXTRA); }
// node probabilities with the edges to 1-state partners folded in (move_edge_prob_to_node2, rotamer.cpp:378-385), four partners per trip:
// slot ids, then flags, then the matrix rows, each as one batch of loads; the same product order wherever it runs
__device__ __forceinline__ void bp_fold_node(const upk_rotamer_t& R, const BpCtx& C, const int* adj_cnt, const int* adj_slot, int g, int n, float (&pr)[6]) {
    const int cnt = adj_cnt[g];
    for (int k0 = 0; k0 < cnt; k0 += 4) {
        int sl[4], act[4]; float row[4][6];
#pragma unroll
        for (int u = 0; u < 4; ++u) sl[u] = adj_slot[g * R.adj_cap + (k0 + u < cnt ? k0 + u : k0)];
#pragma unroll
        for (int u = 0; u < 4; ++u) act[u] = k0 + u < cnt ? C.active[sl[u]] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 6; ++r) row[u][r] = (r < 3 || n == 6) ? C.P[PIDX6(C.cap, sl[u], 0, r)] : 1.f;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (act[u])
#pragma unroll
                for (int r = 0; r < 6; ++r) if (r < n) pr[r] *= row[u][r];
    }
}
// scratch_words: LDS ints behind C.inbox_lds for the row masks and word bases (two per 32 cached rows + 1).  The launcher sizes it for a
// quarter of the slot capacity (the capacity allows 96 residue pairs per node, a protein has ~30): a system with more rows than that sets
// word n_node + 7 of its record and its solve lays the inbox out itself.
#ifndef BPL_THREADS
#define BPL_THREADS 512
#endif
__global__ void __launch_bounds__(BPL_THREADS) k_rotamer_bp_layout(upk_rotamer_t R, int lds_msg_floats, int* __restrict__ out, int scratch_words) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, NN = R.n_node;
    constexpr int NS = BP_NODE_STRIDE;
    float* nb0 = lds;                                        // [NN * NS] scratch of the packing pass
    float* scratch = lds + NN * NS;                          // [32]
    int* bp_start = (int*)(scratch + 32);                    // [NN + 1]
    int* cls = bp_start + NN + 1;                            // [N_CLASS + 1]
    int* n_act = cls + N_CLASS + 1;                          // [3]
    BpCtx C;
    C.cap = R.slot_cap;
    C.slot_a = R.slot_a + (size_t)s * R.slot_cap; C.slot_b = R.slot_b + (size_t)s * R.slot_cap;
    C.active = R.slot_active + (size_t)s * R.slot_cap; C.slot_off = R.slot_off + (size_t)s * R.slot_cap * 2;
    C.slot_row = R.slot_row + (size_t)s * R.slot_cap * 2;
    C.P = R.P + (size_t)s * R.slot_cap * 36; C.inbox = nullptr; C.marg = nullptr;
    C.inbox_lds = lds + (((int)((float*)(n_act + 4) - lds) + 3) & ~3);      // row masks and word bases of the layout
    for (int i = tid; i <= NN; i += nt) bp_start[i] = R.row_start[(size_t)s * (NN + 2) + i];
    if (tid <= N_CLASS) cls[tid] = R.class_start[(size_t)s * (N_CLASS + 1) + tid];
    if (tid < 3) n_act[tid] = 0;
    __syncthreads();
    int* o = out + (size_t)s * bp_layout_stride(NN);
    if (2 * ((bp_start[NN] + 31) >> 5) + 1 > scratch_words) { if (tid == 0) o[NN + 7] = 1; return; }      // (uniform: left to the solve)
    if (tid == 0) o[NN + 7] = 0;
    const BpLayout ly = bp_dense_layout<BPL_THREADS>(R, s, C, NN, cls, n_act, bp_start, scratch, nb0, lds_msg_floats, tid, nt);
    __syncthreads();
    for (int i = tid; i <= NN; i += nt) o[i] = bp_start[i];
    if (tid < 3) o[NN + 1 + tid] = n_act[tid];
    if (tid == 0) { o[NN + 4] = ly.inbox_floats; o[NN + 5] = ly.inbox_floats3; o[NN + 6] = ly.w3; }
    // the fold of the solve's prologue (a chain of dependent loads per node: slot ids -> flags -> matrix rows)
    if (!R.node_prob_in_solve) {
        const int* adj_cnt = R.adj_cnt + (size_t)s * NN;
        const int* adj_slot = R.adj_slot + (size_t)s * NN * R.adj_cap;
        float* fo = (float*)(o + NN + 8);
        for (int g = tid; g < NN; g += nt) {
            const int n = R.node_nrot[g];
            float pr[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) pr[r] = R.node_prob[((size_t)s * NN + g) * 6 + r];
            if (n != 1) bp_fold_node(R, C, adj_cnt, adj_slot, g, n, pr);
#pragma unroll
            for (int r = 0; r < 6; ++r) fo[g * 6 + r] = pr[r];
        }
    }
}

// COMPACT: the message inbox is laid out per solve for the slots ACTIVE in this evaluation only, 4 / 6 floats per row instead of
// the cached layout's 4 / 8 for every cached residue pair (a quarter of which has no bead pair in range on a given step): 151 KB
// instead of 244 KB for the 300-residue benchmark protein, so that most of it stays in the LDS for all sweeps.
template <int BLOCK, int K66, int K36, int K33, bool COMPACT = false>
__global__ void __launch_bounds__(BLOCK) k_rotamer_bp(upk_rotamer_t R, int want_energy, int only_fallback, int lds_msg_floats) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int s = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    if (only_fallback) {
        // this launch follows every cluster launch in the stream: it resets the cluster's barrier counter AND its hand-over word for the
        // next solve (the cluster kernel itself only ever SETS the word: a late workgroup can then never erase its partners' verdict)
        const int handed_over = R.bp_fallback[s];
        __syncthreads();
        if (threadIdx.x == 0) { if (R.bp_bar) R.bp_bar[s] = 0; R.bp_fallback[s] = 0; }
        if (!handed_over) return;      // solved by the cluster kernel
    }
    const int NN = R.n_node;
    float* prob = lds;                 // [NN][6]  node probabilities with the 1-state partners folded in
    constexpr int NS = BP_NODE_STRIDE;   // floats per node in the LDS belief arrays: 8, so that a node's states are one b128 (+ one b64) access
    // ONE belief array: the edge phase reads any node's belief, the node phase reads and rewrites a node's belief with the lanes that
    // own the node, and a workgroup barrier separates the phases -- the reference's old / current pair (rotamer.cpp:1040-1044) updated in
    // place.  (Round 5: the second array's 7 KB go to the message inbox.)
    float* nb0 = lds + NN * NS;        // [NN][NS]
    float* nb1 = nb0;
    float* scratch = lds + NN * BP_NODE_ARRAYS * NS;   // [32]: [0, 17) the reductions and scans, [24, 32) a message row of ones (node phase)
    const float* ones = scratch + 24;
    if (tid < 8) scratch[24 + tid] = 1.f;
    int* nrot = (int*)(lds + NN * BP_NODE_ARRAYS * NS + 32);  // [NN]   state counts
    int* bp_start = nrot + NN;                   // [NN+1] inbox CSR
    int* cls = bp_start + NN + 1;                // [N_CLASS+1]
    int* n_act = cls + N_CLASS + 1;              // [3] active slots of the 3x3 / 3x6 / 6x6 classes
    const int n_slot = R.n_slot[s];
    long long tr_t0 = 0, tr_edge = 0, tr_node = 0, tr_pro = 0, tr_loop = 0, tr_n1 = 0, tr_n2 = 0, tr_n3 = 0, tr_nm = 0;
    const bool trace = R.bp_trace != nullptr && tid == 0;
    if (trace) tr_t0 = wall_clock64();
    long long* TX = R.bp_trace ? R.bp_trace + (size_t)s * 32 + 16 : nullptr;     // sub-phase stamps since the kernel's start (diagnostics)
#define BP_STAMP(k) do { if (trace) TX[k] = wall_clock64() - tr_t0; } while (0)
    BpCtx C;
    C.cap = R.slot_cap;
    C.slot_a = R.slot_a + (size_t)s * R.slot_cap;
    C.slot_b = R.slot_b + (size_t)s * R.slot_cap;
    C.active = R.slot_active + (size_t)s * R.slot_cap;
    C.slot_off = R.slot_off + (size_t)s * R.slot_cap * 2;
    C.P = R.P + (size_t)s * R.slot_cap * 36;
    C.inbox = R.msg_cur + (size_t)s * R.slot_cap * 16;
    C.marg = R.marg + (size_t)s * R.slot_cap * 36;
    // [lds_msg_floats], 16-byte aligned; kept as an OFFSET from the kernel's LDS symbol (no integer round trip), so that the address
    // space stays visible to the compiler
    C.inbox_lds = lds + ((((int)((float*)(cls + N_CLASS + 5) - lds)) + 3) & ~3);
    const int* adj_cnt = R.adj_cnt + (size_t)s * NN;
    const int* adj_slot = R.adj_slot + (size_t)s * NN * R.adj_cap;
    for (int i = tid; i < NN; i += nt) nrot[i] = R.node_nrot[i];
    if (COMPACT) { C.slot_row = R.slot_row + (size_t)s * R.slot_cap * 2; for (int i = tid; i <= NN; i += nt) bp_start[i] = R.row_start[(size_t)s * (NN + 2) + i]; }   // (rows for now)
    else for (int i = tid; i <= NN; i += nt) bp_start[i] = R.bp_start[(size_t)s * (NN + 1) + i];
    if (tid <= N_CLASS) cls[tid] = R.class_start[(size_t)s * (N_CLASS + 1) + tid];
    if (tid < 3) n_act[tid] = 0;
    if (R.node_prob_in_solve) {      // the 1-body pass of upk_rotamer_node_prob, one lane per node, at the head of the solve (a launch less)
        for (int g = tid; g < NN; g += nt) { float pr[6]; rotamer_node_prob_one(R, s, g, pr); for (int r = 0; r < 6; ++r) prob[g * NS + r] = pr[r]; }
    } else
    for (int i = tid; i < NN * 6; i += nt) prob[(i / 6) * NS + i % 6] = R.node_prob[(size_t)s * NN * 6 + i];
    __syncthreads();

    // energies -> probabilities for the entries each class uses (rotamer.cpp:835); 1xN rows carry up to 6 columns
    // (unused ones stay exp(0) = 1, never read)
    if (!R.p_prob) {                           // (p_prob: the pair-energy kernel stored exp(-E) already)
        exp_class<3, 3>(C.P, C.cap, cls[CL33], cls[CL33 + 1], tid, nt, C.active);
        exp_class<3, 6>(C.P, C.cap, cls[CL36], cls[CL36 + 1], tid, nt, C.active);
        exp_class<6, 6>(C.P, C.cap, cls[CL66], cls[CL66 + 1], tid, nt, C.active);
        exp_class<1, 1>(C.P, C.cap, cls[CL11], cls[CL11 + 1], tid, nt, C.active);
        exp_class<1, 6>(C.P, C.cap, cls[CL1X], cls[CL1X + 1], tid, nt, C.active);
    }
    BP_STAMP(0);
    // (the streaming variant serves small, latency-bound batches: packing costs it more than the sweeps get back)
    constexpr bool PACK = K66 + K36 + K33 > 0 || COMPACT;      // (the compact inbox is laid out by the packing pass)
    int inbox_floats, inbox_floats3;       // all message floats of this solve, and those of the rows to 3-state nodes (they come first)
    int w3 = 4;                            // floats per row to a 3-state node (dense layout: 4, or 3 when only that makes the inbox fit the LDS)
    if (COMPACT) {
        const int* LY = R.bp_layout ? R.bp_layout + (size_t)s * bp_layout_stride(NN) : nullptr;
        if (LY && LY[NN + 7] == 0) {          // laid out by k_rotamer_bp_layout in front of this launch
            __syncthreads();        // (bp_start still holds the rows loaded above: nobody reads them)
            for (int i = tid; i <= NN; i += nt) bp_start[i] = LY[i];
            if (tid < 3) n_act[tid] = LY[NN + 1 + tid];
            inbox_floats = LY[NN + 4]; inbox_floats3 = LY[NN + 5]; w3 = LY[NN + 6];
            C.rec = (int4*)R.bp_rec + (size_t)s * R.slot_cap;
        } else {
            const BpLayout ly = bp_dense_layout<BLOCK>(R, s, C, NN, cls, n_act, bp_start, scratch, nb0, lds_msg_floats, tid, nt);
            inbox_floats = ly.inbox_floats; inbox_floats3 = ly.inbox_floats3; w3 = ly.w3;
        }
    } else {
        if (PACK) {
            int4* rec = (int4*)R.bp_rec + (size_t)s * R.slot_cap;
            C.rec = rec;
            bp_pack_active(C, rec, cls, n_act, (int*)nb0, tid, nt, [&](int sl, int side) { return C.slot_off[sl * 2 + side]; });     // (nb0 / nb1 are filled after the fold below)
        }
        inbox_floats3 = bp_start[R.n_node1 + R.n_node3] * 4; inbox_floats = bp_start[NN] * 4;
    }
    // old edge beliefs = 1 (rotamer.cpp:1015-1032); also for slots without an in-range bead pair this step,
    // whose unit message then multiplies as an exact 1
    // the head of the inbox stays in LDS as far as it reaches: the rows to the 3-state nodes come first, then the rows to the
    // 6-state nodes (the boundary never cuts a row)
    {
        constexpr int W6 = COMPACT ? 6 : 8;
        int n = lds_msg_floats < inbox_floats ? lds_msg_floats : inbox_floats;
        // (n == inbox_floats3 keeps the padding float behind an odd number of 3-float rows: with no row to a 6-state node the truncation
        //  to whole rows would otherwise leave lds_floats one short of the inbox and send 3-float rows into the 4-float solve)
        if (n >= inbox_floats3) n = inbox_floats3 + ((n - inbox_floats3) / W6) * W6; else n = (n / w3) * w3;
        C.lds_floats = n;
        if (COMPACT && w3 == 3 && n < inbox_floats && tid == 0) *R.G.error_flag = 9;   // (cannot happen: 3-float rows are chosen only when the inbox then fits LDS)
    }
    __syncthreads();       // (bp_start holds float offsets now; the scratch of the compact layout is dead)
    BP_STAMP(1);
    // the rest of the solve, instantiated twice: for an inbox that sits in LDS as a whole (BpCtxT<true>) and for one with a tail in
    // global memory; which one runs is decided here, once per solve (uniform over the workgroup)
    const BpCtx& C_base = C;
    auto solve = [&](auto all_lds_tag) __attribute__((always_inline)) {      // tag: 0 = inbox with a tail in global memory, 1 = all in LDS, 2 = all in LDS with 3-float rows
    constexpr int SOLVE_KIND = decltype(all_lds_tag)::value;
    BpCtxT<SOLVE_KIND != 0, SOLVE_KIND == 2 ? 3 : 4> C; static_cast<BpCtx&>(C) = C_base;
    constexpr int CW3 = SOLVE_KIND == 2 ? 3 : 4;
    for (int i = tid; i < inbox_floats; i += nt) *C.msg(i) = 1.f;
    __syncthreads();
    BP_STAMP(2);
    // fold edges to 1-state partners into the node probabilities (move_edge_prob_to_node2, rotamer.cpp:378-385)
    // (four partners per trip: slot ids, then flags, then the rows, each as one batch of loads; same product order)
    const bool pre_folded = COMPACT && R.bp_layout && !R.node_prob_in_solve && R.bp_layout[(size_t)s * bp_layout_stride(NN) + NN + 7] == 0;      // (k_rotamer_bp_layout folded: its probabilities replace the ones loaded above)
    if (pre_folded) {
        const float* fo = (const float*)(R.bp_layout + (size_t)s * bp_layout_stride(NN) + NN + 8);
        for (int i = tid; i < NN * 6; i += nt) prob[(i / 6) * NS + i % 6] = fo[i];
    } else
    for (int g = tid; g < NN; g += nt) {
        const int n = nrot[g];
        if (n == 1) continue;
        float pr[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) pr[r] = prob[g * NS + r];
        bp_fold_node(R, C, adj_cnt, adj_slot, g, n, pr);
#pragma unroll
        for (int r = 0; r < 6; ++r) if (r < n) prob[g * NS + r] = pr[r];
    }
    __syncthreads();
    for (int i = tid; i < NN * NS; i += nt) nb0[i] = (i % NS) < 6 ? prob[i] : 0.f;   // old node belief = prob (rotamer.cpp:1009-1013)
    __syncthreads();
    BP_STAMP(3);
    BpResident<3, 3, K33> r33; BpResident<3, 6, K36> r36; BpResident<6, 6, K66> r66;
    const int e33 = cls[CL33] + n_act[CL33], e36 = cls[CL36] + n_act[CL36], e66 = cls[CL66] + n_act[CL66];   // ends of the packed records
    r33.load(C, cls[CL33], e33, tid, nt); r36.load(C, cls[CL36], e36, tid, nt); r66.load(C, cls[CL66], e66, tid, nt);
    BP_STAMP(4);
    const __amdgpu_buffer_rsrc_t inbox_rs = make_rsrc(C.inbox, 0u);

    float* nb_old = nb0; float* nb_cur = nb1;
    int iter = 0;
    float maxdev = 1e10f;
    // sweep -1 is calculate_new_beliefs(0.f, true): only its messages survive and the "old" node belief becomes
    // prob / max(prob) (rotamer.cpp:1034 with the swap at 995-1001)
    if (trace) tr_pro = wall_clock64();
    for (int sweep = -1;; ++sweep) {
        long long tr_a = 0, tr_b = 0;
        if (trace) tr_a = wall_clock64();
        // ---- edge phase: every residue pair rewrites its two messages in place from the old node beliefs
        // (measured and rejected: dealing 64-slot chunks of all classes to the wavefronts round-robin, heavy classes first, to
        // even out the trip counts -- 1 % slower: the phase is limited by bytes, not by trips)
        r33.template edge<NS>(C, nb_old, inbox_rs); r36.template edge<NS>(C, nb_old, inbox_rs); r66.template edge<NS>(C, nb_old, inbox_rs);
        if (PACK) {
            // (no barrier separates the classes: a wavefront's edge phase is the sum of its trips through all of them.  A class of
            //  n slots gives its first n mod nt lanes one slot more; dealt from lane 0 in every class the extra trips pile up on the
            //  first wavefronts -- 8 trips there, 5 on the last one for the benchmark protein.  The streamed 3x3 and 3x6 classes are
            //  dealt from the LAST lane instead: 7 at most.  Which lane serves a slot changes nothing in its arithmetic.)
            const int tid_r = BP_EDGE_REVERSE ? nt - 1 - tid : tid;
            bp_edge_packed<3, 3, NS>(C, cls[CL33] + K33 * nt, e33, nb_old, tid_r, nt, inbox_rs);
            bp_edge_packed<3, 6, NS>(C, cls[CL36] + K36 * nt, e36, nb_old, tid_r, nt, inbox_rs);
            bp_edge_packed<6, 6, NS>(C, cls[CL66] + K66 * nt, e66, nb_old, tid, nt, inbox_rs);
        } else {
            bp_edge_range<3, 3, NS>(C, cls[CL33], cls[CL33 + 1], nb_old, tid, nt);
            bp_edge_range<3, 6, NS>(C, cls[CL36], cls[CL36 + 1], nb_old, tid, nt);
            bp_edge_range<6, 6, NS>(C, cls[CL66], cls[CL66 + 1], nb_old, tid, nt);
        }
        __syncthreads();
        if (trace) { tr_b = wall_clock64(); tr_edge += tr_b - tr_a; }
        // ---- node phase: BP_GROUP lanes per node stream the node's inbox, multiply, and combine by shuffles
        float dev = 0.f;
        // Rounds of nt / (lanes per node) nodes; a round waits for its slowest load.  The inbox rows of the 6-state nodes are the ones that
        // spill to global memory, those of the 3-state nodes all sit in LDS: the 6-state nodes go first, in rounds of their own
        // (one round for the 128 of the benchmark protein), instead of being spread over every round; 1-state nodes have no
        // inbox and take no slot.  (Node order in the arrays: 1-state, 3-state, 6-state.)
        // (tried in round 3: the 3-state nodes in one round as a MIX -- 116 on four lanes, 24 on two: slower, the round waits for the
        //  two-lane nodes; ALL of them on two lanes is what won)
        const int e1n = R.n_node1, e3n = R.n_node1 + R.n_node3;
        // one node: lanes glx = 0 .. grp-1 of its group (`live`: the group has a node)
        auto node_update = [&](const int grp, const int glx, const int g, const bool live, const int wd) {
            const int n = live ? nrot[g] : 0;
            float bb[6] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            if (live && sweep >= 0) {
                const int q = COMPACT ? (n == 6 ? 6 : CW3) : (n == 6 ? 2 : 1), base = bp_start[g], deg = (bp_start[g + 1] - base) / q;
                // ROWS rows per trip are fetched before the first multiply.  A lane past the end of its node's rows reads the row of ones
                // kept behind the scan scratch (an exact factor 1: no per-component selects, no bounds branch in the multiply loop), and the
                // running product is rescaled ONCE per trip by a power of two (v_frexp_exp / v_ldexp: exact, so where the rescaling happens
                // changes no bit of the result -- round 4 divided by the maximum every second row, 15 of its 27 instructions per two rows)
                constexpr int ROWS = BP_NODE_ROWS_512;
                for (int k0 = glx; k0 < deg; k0 += ROWS * grp) {
                    float4 m0[ROWS]; float2 m1[ROWS];
#pragma unroll
                    for (int u = 0; u < ROWS; ++u) {
                        const int k = k0 + u * grp;
                        if (COMPACT) {       // dense rows: 4 (or 3) floats to a 3-state node, 6 floats (8-byte aligned) to a 6-state node
                            const float* m = k < deg ? C.msg(base + k * q) : ones;
                            if (n == 6) { const float2 a = ((const float2*)m)[0], b = ((const float2*)m)[1]; m0[u] = make_float4(a.x, a.y, b.x, b.y); m1[u] = ((const float2*)m)[2]; }
                            else if (CW3 == 3) { m0[u] = make_float4(m[0], m[1], m[2], 1.f); m1[u] = make_float2(1.f, 1.f); }
                            else { m0[u] = *(const float4*)m; m1[u] = make_float2(1.f, 1.f); }
                        } else {
                            const float* m = k < deg ? C.msg((base + k * q) * 4) : ones;
                            m0[u] = *(const float4*)m; m1[u] = make_float2(1.f, 1.f);
                            if (n == 6) m1[u] = *(const float2*)(m + 4);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < ROWS; ++u) {
                        bb[0] *= m0[u].x; bb[1] *= m0[u].y; bb[2] *= m0[u].z;
                        if (n == 6) { bb[3] *= m0[u].w; bb[4] *= m1[u].x; bb[5] *= m1[u].y; }
                    }
                    bp_rescale_pow2(bb, n);
                }
            }
            static_assert(BP_GROUP == 4, "the combine below is the lane^2, lane^1 butterfly of a quad");
            if (trace) { const long long t = wall_clock64(); tr_n1 += t - tr_nm; tr_nm = t; }      // (diagnostics: rows in, per-lane products done)
#if BP_WIDE_NODE_GROUPS
            // groups of 16 / 8 lanes (small systems, one round): two more combine stages in front of the quad's, L <-> 15-L and
            // L <-> 7-L of the DPP row (every lane of a group ends with the same bits: each stage multiplies a symmetric pair)
            if (wd >= 4) {       // (`wd` is uniform over the workgroup: no lane sits out a DPP step; which lanes multiply is a per-lane select)
#pragma unroll
                for (int r = 0; r < 6; ++r) { const float o = dpp_mov<UP_DPP_ROW_MIRROR>(bb[r]); bb[r] *= grp >= 16 ? o : 1.f; }
                bp_rescale_pow2(bb, n);
            }
            if (wd >= 2) {
#pragma unroll
                for (int r = 0; r < 6; ++r) { const float o = dpp_mov<UP_DPP_HALF_MIRROR>(bb[r]); bb[r] *= grp >= 8 ? o : 1.f; }
                bp_rescale_pow2(bb, n);
            }
#endif
            {
#pragma unroll
                for (int r = 0; r < 6; ++r) { const float o = dpp_mov<UP_DPP_XOR2>(bb[r]); bb[r] *= grp >= 4 ? o : 1.f; }   // (a pair has no lane^2 partner)
                bp_rescale_pow2(bb, n);
            }
            {
#pragma unroll
                for (int r = 0; r < 6; ++r) { const float o = dpp_mov<UP_DPP_XOR1>(bb[r]); bb[r] *= grp >= 2 ? o : 1.f; }
                bp_rescale_pow2(bb, n);
            }
            if (trace) { const long long t = wall_clock64(); tr_n2 += t - tr_nm; tr_nm = t; }      // (combine done)
            if (live) {
                // b = prob * product, then standardize (rotamer.cpp:258-273); lane glx of the node finishes states glx, glx + grp, ...
                float v[6], mx = 0.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) { v[r] = r < n ? prob[g * NS + r] * bb[r] : 0.f; mx = fmaxf(mx, v[r]); }
                const float rm = rcp(mx);
                const float damp = sweep < 0 ? 0.f : R.damping;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    if (r < n && (r & (grp - 1)) == glx) {
                        const float o = nb_old[g * NS + r];
                        const float nv = damp != 0.f ? (1.f - damp) * rm * v[r] + damp * o : rm * v[r];
                        nb_cur[g * NS + r] = nv;
                        dev = fmaxf(nv - o, dev);                      // signed, rotamer.cpp:275-281
                    }
                }
            }
        };
        // A small system has lanes for all its nodes at once -- four per 6-state node, then two per 3-state node --: ONE round, one
        // chain of dependent LDS accesses per sweep instead of two (the arithmetic of a node does not depend on which lanes hold it).
        if (trace) tr_nm = wall_clock64();
        const int lanes6 = (NN - e3n) * BP_GROUP6, lanes3 = (e3n - e1n) * BP_GROUP3;
#if BP_WIDE_NODE_GROUPS
        // ... and where the lanes are there, four or two times as many per node: a node's inbox is then ONE batch of row loads per lane
        // instead of two to four dependent ones (a 56-residue protein has ~50 multi-state nodes for 512 lanes)
        const int wide = (lanes6 + lanes3) * 4 <= nt ? 4 : ((lanes6 + lanes3) * 2 <= nt ? 2 : 1);
#else
        constexpr int wide = 1;
#endif
        if (lanes6 + lanes3 <= nt) {
            const bool six = tid < lanes6 * wide;
            const int grp = (six ? BP_GROUP6 : BP_GROUP3) * wide, t = six ? tid : tid - lanes6 * wide;
            const int g = (six ? e3n : e1n) + t / grp;
            node_update(grp, t & (grp - 1), g, six || t < lanes3 * wide, wide);
        } else
        for (int part = 0; part < 2; ++part)
        for (int g0 = part == 0 ? e3n : e1n, g_hi = part == 0 ? NN : e3n; g0 < g_hi; g0 += part == 0 ? nt / BP_GROUP6 : nt / BP_GROUP3) {
            const int grp = part == 0 ? BP_GROUP6 : BP_GROUP3;
            const int g = g0 + tid / grp;
            node_update(grp, tid & (grp - 1), g, g < g_hi, 1);
        }
        if (trace) { const long long t = wall_clock64(); tr_n3 += t - tr_nm; tr_nm = t; }          // (beliefs written; what is left of tr_node is the barrier)
        __syncthreads();
        if (trace) tr_node += wall_clock64() - tr_b;
        if (sweep >= 0) {
            ++iter;
            if (iter % R.chunk == 0) {
                maxdev = block_max(dev, scratch);
                if (!(maxdev > R.tol && iter < R.max_iter)) break;     // rotamer.cpp:1038
            }
        }
        // (rotamer.cpp:1040-1044 swaps its old / current arrays here: one array, updated in place by the node phase)
    }
    if (tid == 0) { R.iters[s] = iter; if (iter >= R.max_iter - R.chunk - 1) R.n_bad[s] += 1; }
    if (trace) tr_loop = wall_clock64();

    // ---- marginals (rotamer.cpp:1053-1059)
    for (int g = tid; g < NN; g += nt) {
        const int n = nrot[g];
        float sum = 0.f;
        if (n == 1) { nb_cur[g * NS] = 1.f; for (int r = 1; r < 6; ++r) nb_cur[g * NS + r] = 0.f; continue; }
        for (int r = 0; r < n; ++r) sum += nb_cur[g * NS + r];
        const float rs = rcp(sum);
        for (int r = 0; r < n; ++r) nb_cur[g * NS + r] *= rs;
    }
    __syncthreads();
    float en = 0.f;
    // (measured and rejected in round 3: retiring a slot's matrix right behind its marginals instead of in a pass of its own -- the second
    //  store stream in the marginal loops costs more than the pass saves, epilogue 75 -> 100 us; fetching the next slot's matrix one trip
    //  ahead in the marginal loops -- no change: the epilogue is bound by its scattered 12- and 24-byte stores)
    en += r33.template marginal<NS>(C, nb_cur, want_energy);
    en += r36.template marginal<NS>(C, nb_cur, want_energy);
    en += r66.template marginal<NS>(C, nb_cur, want_energy);
    BP_STAMP(5);
    if (PACK) {
        en += bp_marginal_packed<3, 3, NS>(C, cls[CL33] + K33 * nt, e33, nb_cur, tid, nt, want_energy);
        en += bp_marginal_packed<3, 6, NS>(C, cls[CL36] + K36 * nt, e36, nb_cur, tid, nt, want_energy);
        en += bp_marginal_packed<6, 6, NS>(C, cls[CL66] + K66 * nt, e66, nb_cur, tid, nt, want_energy);
    } else {
        en += bp_marginal_range<3, 3, NS>(C, cls[CL33], cls[CL33 + 1], nb_cur, tid, nt, want_energy);
        en += bp_marginal_range<3, 6, NS>(C, cls[CL36], cls[CL36 + 1], nb_cur, tid, nt, want_energy);
        en += bp_marginal_range<6, 6, NS>(C, cls[CL66], cls[CL66 + 1], nb_cur, tid, nt, want_energy);
    }
    BP_STAMP(6);
    if (want_energy) {
        for (int sl = cls[CL11] + tid; sl < cls[CL11 + 1]; sl += nt)   // 1-1 edges (rotamer.cpp:861)
            if (C.active[sl]) en += -logf(C.P[PIDX6(C.cap, sl, 0, 0)]);
        for (int g = tid; g < NN; g += nt) {   // node_free_energy, rotamer.cpp:292-302
            const int n = nrot[g];
            float e = R.node_off[(size_t)s * NN + g];
            for (int r = 0; r < n; ++r) { const float b = nb_cur[g * NS + r]; e += b * logf((1e-10f + b) * rcp(1e-10f + prob[g * NS + r])); }
            en += e;
        }
        const float tot = block_sum(en, scratch);
        if (tid == 0) R.energy[s] = tot;
    }
    for (int i = tid; i < NN * 6; i += nt) R.nb_cur[(size_t)s * NN * 6 + i] = nb_cur[(i / 6) * NS + i % 6];
    __syncthreads();
    BP_STAMP(7);
    // leave the accumulators clean for the next force evaluation
    // (only the slots written this step: the others were left at 0 by the prologue); the flags move to active_last
    // (Round 6, measured and not taken: this tail as a launch of its own behind the solve, as the layout is one in front of it -- solve +
    //  tail 4.68 -> 5.20 ms at 4096 systems: inside the solve the 1.6 GB of stores overlap with the sweeps of the workgroups on other
    //  CUs; alone they are a store-bound kernel of their own.)
    int* active_w = R.slot_active + (size_t)s * R.slot_cap;
    int* active_last = R.slot_active_last + (size_t)s * R.slot_cap;
    const float rest = R.p_prob ? 1.f : 0.f;
    if (PACK) {     // the records name the active slots of the multi-state classes: no flag round trip in front of the stores
        retire_packed<3, 3>(C, cls[CL33], e33, tid, nt, rest);
        retire_packed<3, 6>(C, cls[CL36], e36, tid, nt, rest);
        retire_packed<6, 6>(C, cls[CL66], e66, tid, nt, rest);
        retire_flags(cls[CL33], cls[CL66 + 1], active_w, active_last, tid, nt);      // (the three classes are adjacent: CL33 < CL36 < CL66)
    } else {
        retire_class<3, 3>(C.P, C.cap, cls[CL33], cls[CL33 + 1], active_w, active_last, tid, nt, rest);
        retire_class<3, 6>(C.P, C.cap, cls[CL36], cls[CL36 + 1], active_w, active_last, tid, nt, rest);
        retire_class<6, 6>(C.P, C.cap, cls[CL66], cls[CL66 + 1], active_w, active_last, tid, nt, rest);
    }
    retire_class<1, 1>(C.P, C.cap, cls[CL11], cls[CL11 + 1], active_w, active_last, tid, nt, rest);
    retire_class<1, 6>(C.P, C.cap, cls[CL1X], cls[CL1X + 1], active_w, active_last, tid, nt, rest);
    for (int i = cls[N_CLASS] + tid; i < n_slot; i += nt) { active_last[i] = active_w[i]; active_w[i] = 0; }   // (no slot lies outside the classes)
    if (trace) {
        long long* T = R.bp_trace + (size_t)s * 32;
        T[0] = tr_pro - tr_t0; T[1] = tr_loop - tr_pro; T[2] = wall_clock64() - tr_loop; T[3] = tr_edge; T[4] = tr_node;
        T[24] = tr_n1; T[25] = tr_n2; T[26] = tr_n3;
        T[5] = iter; T[6] = n_slot; T[7] = COMPACT ? C.lds_floats : bp_start[NN];
        for (int c = 0; c <= N_CLASS; ++c) T[8 + c] = cls[c];
        T[27] = SOLVE_KIND;
    }
    };      // solve
    if (COMPACT && C.lds_floats >= inbox_floats) { if (w3 == 3) solve(std::integral_constant<int, 2>{}); else solve(std::integral_constant<int, 1>{}); }
    else solve(std::integral_constant<int, 0>{});
}

// ------------------------------------------------------------------------------------------------
// Belief propagation with the pair matrices ON CHIP: a cluster of C workgroups serves one system.  Workgroup c
// owns a contiguous share of every multi-state slot class (its exp(-E) matrices live in LDS for the whole
// solve, its two messages per slot in registers) and a contiguous share of the multi-state nodes.  Per sweep the
// cluster exchanges only the messages (slot owner -> node owner) and the node beliefs (node owner -> everyone)
// through device-scope loads/stores, separated by two counter barriers.  HBM/L2 traffic per sweep drops from
// P + 3 x messages to 2 x messages, and a system gets C x 1024 lanes.
//
// Cross-workgroup protocol (placement independent, cdna_hip_programming.md guideline 16): payload is written with
// 16-byte write-through (sc1) buffer stores, every storing wave drains them, __syncthreads, ONE lane bumps an
// agent-scope counter and polls it relaxed, ONE agent-scope acquire drops the CU's stale L1 lines, __syncthreads,
// then plain loads.  All C workgroups of a cluster must be resident at once: the launcher sends at most
// (CUs / C) systems per launch, one workgroup per CU, and every spin is bounded.
struct BpcShared {            // cluster-visible state of one system
    const float* inbox;       // message rows [rows][8] (plain loads after an acquire)
    __amdgpu_buffer_rsrc_t inbox_w, nbx_w;   // write-through views of the inbox and of nbx
    const float* nbx;         // [2][NN][8] node beliefs, double buffered by sweep
    float* dev;               // [2][16] per-workgroup max deviation, double buffered
    float* en_part;           // [16] energy partial sums
    int* bar;                 // barrier counter (zeroed by upk_rotamer_node_prob)
};
// (Round 4, measured and not taken: everything a workgroup reads from its partners after a barrier read with sc1 loads, so that the barrier
//  needs no acquire fence -- one 300-residue system, 6 workgroups: 520 us per step against 500 with the fence.)
// Returns false when a partner never arrived (the cluster's workgroups were not all resident -- other kernels held CUs -- or a
// partner has given up already).  `fallback` non-null: the system is handed to the one-workgroup solve that follows the cluster
// launch in the stream (nothing the solve consumes has been modified yet: callers return at once); null (the last barrier, behind
// the epilogue): reported as error 7.
__device__ __forceinline__ bool cluster_barrier(int* bar, int& phase, int C, int* error_flag, int* fallback, int spin_limit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave: its write-through stores have landed
    __syncthreads();
    ++phase;
    int gave_up = 0;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = phase * C;
        int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit) {            // report / hand over instead of hanging the device
                gave_up = 1;
                if (fallback) __hip_atomic_store(fallback, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *error_flag = 7;
                break;
            }
            if (fallback && !(spins & 255) && __hip_atomic_load(fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { gave_up = 1; break; }   // a partner gave up
        }
        // the counter alone can let a late workgroup through (partners that have left did arrive at this barrier before they gave up at a later one)
        if (fallback && !gave_up && __hip_atomic_load(fallback, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) gave_up = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");         // one buffer_inv for the whole CU
    }
    return __syncthreads_or(gave_up) == 0;
}

template <int NA, int NB> struct BpcSlot { int a, b, offa, offb, sl; bool live; float ma[NA], mb[NB]; };

template <int NA, int NB>
__device__ __forceinline__ void bpc_init(BpcSlot<NA, NB>& st, const upk_rotamer_t& R, int s, int lo, int n_own, __amdgpu_buffer_rsrc_t inbox) {
    const int i = threadIdx.x;
    st.sl = i < n_own ? lo + i : -1; st.live = false; st.a = st.b = 0; st.offa = st.offb = 0;
#pragma unroll
    for (int k = 0; k < NA; ++k) st.ma[k] = 1.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) st.mb[k] = 1.f;
    if (st.sl < 0) return;
    const size_t so = (size_t)s * R.slot_cap + st.sl;
    st.live = R.slot_active[so] != 0;
    st.a = R.slot_a[so]; st.b = R.slot_b[so];
    st.offa = R.slot_off[so * 2]; st.offb = R.slot_off[so * 2 + 1];
    // old edge beliefs = 1 (rotamer.cpp:1015-1032), also for slots without an in-range bead pair this step
    st_wt16(inbox, st.offa, 1.f, 1.f, 1.f, 1.f);
    if (NA == 6) st_wt16(inbox, st.offa + 4, 1.f, 1.f, 1.f, 1.f);
    st_wt16(inbox, st.offb, 1.f, 1.f, 1.f, 1.f);
    if (NB == 6) st_wt16(inbox, st.offb + 4, 1.f, 1.f, 1.f, 1.f);
}
// stage exp(-E) of the own slots of one class: Pl[(i*NB+j)*n_own + local]
template <int NA, int NB>
__device__ __forceinline__ void bpc_stage(float* Pl, const upk_rotamer_t& R, int s, int lo, int n_own) {
    const float* P = R.P + (size_t)s * R.slot_cap * 36;
    for (int t = threadIdx.x; t < n_own * NA * NB; t += blockDim.x) {
        const int e = t / n_own, l = t - e * n_own, i = e / NB, j = e - i * NB;
        Pl[t] = expf(-P[PIDX6(R.slot_cap, lo + l, i, j)]);
    }
}
// update_beliefs for one slot (rotamer.cpp:468-499, 506-521); messages stay in registers, copies go to the inbox
template <int NA, int NB>
__device__ __forceinline__ void bpc_edge(BpcSlot<NA, NB>& st, const float* Pl, int n_own, const float* nb_old, __amdgpu_buffer_rsrc_t inbox) {
    if (!st.live) return;
    const int l = threadIdx.x;
    float P[NA][NB], va[NA], vb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) P[i][j] = Pl[(i * NB + j) * n_own + l];
#pragma unroll
    for (int i = 0; i < NA; ++i) va[i] = nb_old[st.a * 6 + i] * fast_rcp(1e-10f + st.ma[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) vb[j] = nb_old[st.b * 6 + j] * fast_rcp(1e-10f + st.mb[j]);
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
    for (int i = 0; i < NA; ++i) st.ma[i] = ta[i] * ra;
#pragma unroll
    for (int j = 0; j < NB; ++j) st.mb[j] = tb[j] * rb;
    st_wt16(inbox, st.offa, st.ma[0], st.ma[1], st.ma[2], NA == 6 ? st.ma[NA - 3] : 1.f);
    if (NA == 6) st_wt16(inbox, st.offa + 4, st.ma[NA - 2], st.ma[NA - 1], 1.f, 1.f);
    st_wt16(inbox, st.offb, st.mb[0], st.mb[1], st.mb[2], NB == 6 ? st.mb[NB - 3] : 1.f);
    if (NB == 6) st_wt16(inbox, st.offb + 4, st.mb[NB - 2], st.mb[NB - 1], 1.f, 1.f);
}
// pair marginal of one slot and its Bethe term (rotamer.cpp:405-451)
template <int NA, int NB>
__device__ __forceinline__ float bpc_marginal(const BpcSlot<NA, NB>& st, const float* Pl, int n_own, const float* nbm, float* marg, int cap,
                                              bool want_energy) {
    if (!st.live) return 0.f;
    const int l = threadIdx.x;
    float en = 0.f, P[NA][NB], bc1[NA], bc2[NB], sum = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) P[i][j] = Pl[(i * NB + j) * n_own + l];
#pragma unroll
    for (int i = 0; i < NA; ++i) bc1[i] = nbm[st.a * 6 + i] * rcp(1e-10f + st.ma[i]);
#pragma unroll
    for (int j = 0; j < NB; ++j) bc2[j] = nbm[st.b * 6 + j] * rcp(1e-10f + st.mb[j]);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) sum += P[i][j] * bc1[i] * bc2[j];
    const float rs = rcp(sum);
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float pm = (P[i][j] * bc1[i] * bc2[j]) * rs;      // recomputed: no second 36-register matrix
            marg[PIDX6(cap, st.sl, i, j)] = pm;
            if (want_energy) en += pm * logf((1e-10f + pm) * rcp(1e-10f + P[i][j] * nbm[st.a * 6 + i] * nbm[st.b * 6 + j]));
        }
    return en;
}

#define BPC_GROUP 16  // lanes cooperating on one node

#define BPC_BLOCK 512   // 8 waves: 256 VGPRs per lane keep the three slot states + a 6x6 matrix out of scratch
// RESIDENT = true: the form described above (512 lanes, needs C large enough for the matrices to fit LDS).
// RESIDENT = false ("split" solve, 1024 lanes): the same exchange protocol, but the matrices stay in global memory and
// a lane loops over its share of the slots like the one-workgroup kernel does -- any C works; used for mid-size
// batches that leave CUs idle under the one-workgroup solve.
template <bool RESIDENT>
__global__ void __launch_bounds__(BPC_BLOCK) k_rotamer_bp_cluster(upk_rotamer_t R, int want_energy, int C, int sys0, int n_sys, int p_cap) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x >= n_sys) return;
    const int s = sys0 + blockIdx.x, c = blockIdx.y, tid = threadIdx.x, nt = blockDim.x;
    const int NN = R.n_node, e1 = R.n_node1;
    // ---- ownership
    const int* cls = R.class_start + (size_t)s * (N_CLASS + 1);
    int lo[3], n_own[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int b = cls[k], n = cls[k + 1] - b;
        lo[k] = b + (int)((long)n * c / C); n_own[k] = b + (int)((long)n * (c + 1) / C) - lo[k];
    }
    const int n_multi = NN - e1;
    const int g_lo = e1 + (int)((long)n_multi * c / C), g_hi = e1 + (int)((long)n_multi * (c + 1) / C);
    const int need = n_own[0] * 9 + n_own[1] * 18 + n_own[2] * 36;
    // every workgroup of the cluster must reach the same verdict: test all shares, not only the own one
    bool fits = true;
    for (int cc = 0; RESIDENT && cc < C; ++cc) {
        int nd = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int b = cls[k], n = cls[k + 1] - b;
            const int m = (int)((long)n * (cc + 1) / C) - (int)((long)n * cc / C);
            if (m > BPC_BLOCK) fits = false;
            nd += m * (k == 0 ? 9 : (k == 1 ? 18 : 36));
        }
        if (nd > p_cap) fits = false;
    }
    if (!fits) { if (c == 0 && tid == 0) R.bp_fallback[s] = 1; return; }   // the single-workgroup kernel takes this system
    // a barrier that gives up writes 1 here; the word is 0 on entry (cleared by the one-workgroup launch that follows every cluster
    // launch), is only ever set by this kernel, and every barrier also reads it: a workgroup that arrives after its partners have
    // given up leaves at its first barrier instead of running phases on data nobody produced
    int* fb = R.bp_fallback + s;
    const int spin_limit = R.bp_test_abort ? (1 << 12) : (1 << 22);
    if (R.bp_test_abort && c == C - 1) return;          // tests: a partner that never arrives

    float* nb = lds;                          // [NN][6] latest node beliefs (all nodes)
    float* prob = lds + NN * 6;               // [NN][6] own nodes only are valid: probabilities with 1-state partners folded
    float* scratch = lds + NN * 12;           // [32]
    int* nrot = (int*)(lds + NN * 12 + 32);   // [NN]
    int* bp_start = nrot + NN;                // [NN+1]
    float* Pl0 = (float*)(bp_start + NN + 1 + 3);
    Pl0 = lds + ((((int)(Pl0 - lds)) + 3) & ~3);      // 16-byte aligned, as an OFFSET from the LDS symbol: an integer round trip of the pointer turns every read of the resident matrices into a flat access
    float* Pl1 = Pl0 + n_own[0] * 9;
    float* Pl2 = Pl1 + n_own[1] * 18;
    (void)need;
    BpcShared X;
    X.inbox = R.msg_cur + (size_t)s * R.slot_cap * 16;
    X.inbox_w = make_rsrc(X.inbox, (unsigned)R.slot_cap * 64u);
    X.nbx = R.bp_nbx + (size_t)s * NN * 16;
    X.nbx_w = make_rsrc(X.nbx, (unsigned)NN * 64u);   // 2 halves x NN rows x 32 bytes
    X.dev = R.bp_dev + (size_t)s * 2 * 16;
    X.en_part = R.bp_en_part + (size_t)s * 16;
    X.bar = R.bp_bar + s;
    int phase = 0;

    for (int i = tid; i < NN; i += nt) nrot[i] = R.node_nrot[i];
    for (int i = tid; i <= NN; i += nt) bp_start[i] = R.bp_start[(size_t)s * (NN + 1) + i];
    BpcSlot<3, 3> s33; BpcSlot<3, 6> s36; BpcSlot<6, 6> s66;
    BpCtx Cx;                                  // split solve: slot data in global memory
    Cx.cap = R.slot_cap;
    Cx.slot_a = R.slot_a + (size_t)s * R.slot_cap; Cx.slot_b = R.slot_b + (size_t)s * R.slot_cap;
    Cx.active = R.slot_active + (size_t)s * R.slot_cap; Cx.slot_off = R.slot_off + (size_t)s * R.slot_cap * 2;
    Cx.P = R.P + (size_t)s * R.slot_cap * 36; Cx.inbox = R.msg_cur + (size_t)s * R.slot_cap * 16;
    Cx.marg = R.marg + (size_t)s * R.slot_cap * 36;
    if (RESIDENT) {
        bpc_stage<3, 3>(Pl0, R, s, lo[0], n_own[0]);
        bpc_stage<3, 6>(Pl1, R, s, lo[1], n_own[1]);
        bpc_stage<6, 6>(Pl2, R, s, lo[2], n_own[2]);
        bpc_init(s33, R, s, lo[0], n_own[0], X.inbox_w);
        bpc_init(s36, R, s, lo[1], n_own[1], X.inbox_w);
        bpc_init(s66, R, s, lo[2], n_own[2], X.inbox_w);
    } else {
        // exp(-E) in place for the own multi-state slots only: nobody else reads them (the 1-state classes stay
        // energies: the fold below and the 1-1 energy term take exp / the energy themselves)
        exp_class<3, 3>(Cx.P, Cx.cap, lo[0], lo[0] + n_own[0], tid, nt);
        exp_class<3, 6>(Cx.P, Cx.cap, lo[1], lo[1] + n_own[1], tid, nt);
        exp_class<6, 6>(Cx.P, Cx.cap, lo[2], lo[2] + n_own[2], tid, nt);
        for (int k = 0; k < 3; ++k)            // old edge beliefs = 1 (rotamer.cpp:1015-1032), both rows of every own slot
            for (int sl = lo[k] + tid; sl < lo[k] + n_own[k]; sl += nt) {
                const int oa = Cx.slot_off[sl * 2], ob = Cx.slot_off[sl * 2 + 1];
                st_wt16(X.inbox_w, oa, 1.f, 1.f, 1.f, 1.f); st_wt16(X.inbox_w, ob, 1.f, 1.f, 1.f, 1.f);
                if (k == 2) st_wt16(X.inbox_w, oa + 4, 1.f, 1.f, 1.f, 1.f);
                if (k >= 1) st_wt16(X.inbox_w, ob + 4, 1.f, 1.f, 1.f, 1.f);
            }
    }
    // fold the edges to 1-state partners into the probabilities of the own nodes (rotamer.cpp:378-385)
    if (R.node_prob_in_solve && c == 0)      // (the 1-state nodes' 1-body terms: read back by this workgroup's energy sum)
        for (int g = tid; g < e1; g += nt) { float p1[6]; rotamer_node_prob_one(R, s, g, p1); }
    {
        const float* P = R.P + (size_t)s * R.slot_cap * 36;
        const int* active = R.slot_active + (size_t)s * R.slot_cap;
        const int* adj_cnt = R.adj_cnt + (size_t)s * NN;
        const int* adj_slot = R.adj_slot + (size_t)s * NN * R.adj_cap;
        for (int g = g_lo + tid; g < g_hi; g += nt) {
            const int n = R.node_nrot[g], na = adj_cnt[g];
            float p[6];
            if (R.node_prob_in_solve) rotamer_node_prob_one(R, s, g, p);     // (1-body energies -> probabilities of the own nodes: no launch of their own)
            else {
#pragma unroll
                for (int r = 0; r < 6; ++r) p[r] = R.node_prob[((size_t)s * NN + g) * 6 + r];
            }
            for (int k = 0; k < na; ++k) {
                const int sl = adj_slot[g * R.adj_cap + k];
                if (!active[sl]) continue;
#pragma unroll
                for (int r = 0; r < 6; ++r) if (r < n) p[r] *= expf(-P[PIDX6(R.slot_cap, sl, 0, r)]);
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) prob[g * 6 + r] = p[r];
            st_wt16(X.nbx_w, g * 8, p[0], p[1], p[2], p[3]);          // old node belief = prob (rotamer.cpp:1009-1013)
            st_wt16(X.nbx_w, g * 8 + 4, p[4], p[5], 0.f, 0.f);
        }
    }
    if (!cluster_barrier(X.bar, phase, C, R.G.error_flag, fb, spin_limit)) return;      // (handed to the one-workgroup solve)
    auto reload = [&](int half) {   // beliefs of every multi-state node from the exchange buffer into LDS
        for (int g = tid; g < NN; g += nt) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f); float2 b = make_float2(0.f, 0.f);
            if (g >= e1) { const float* src = X.nbx + ((size_t)half * NN + g) * 8; a = *(const float4*)src; b = *(const float2*)(src + 4); }
            nb[g * 6] = a.x; nb[g * 6 + 1] = a.y; nb[g * 6 + 2] = a.z; nb[g * 6 + 3] = a.w; nb[g * 6 + 4] = b.x; nb[g * 6 + 5] = b.y;
        }
        __syncthreads();
    };
    reload(0);

    int iter = 0, cur = 0;                    // nbx half holding the beliefs in `nb`
    const int gl = tid % BPC_GROUP, n_grp = nt / BPC_GROUP;
    for (int sweep = -1;; ++sweep) {
        // ---- edge phase
        if (RESIDENT) {
            bpc_edge(s33, Pl0, n_own[0], nb, X.inbox_w);
            bpc_edge(s36, Pl1, n_own[1], nb, X.inbox_w);
            bpc_edge(s66, Pl2, n_own[2], nb, X.inbox_w);
        } else {
            bp_edge_range_impl<3, 3, true>(Cx, lo[0], lo[0] + n_own[0], nb, tid, nt, X.inbox_w);
            bp_edge_range_impl<3, 6, true>(Cx, lo[1], lo[1] + n_own[1], nb, tid, nt, X.inbox_w);
            bp_edge_range_impl<6, 6, true>(Cx, lo[2], lo[2] + n_own[2], nb, tid, nt, X.inbox_w);
        }
        if (!cluster_barrier(X.bar, phase, C, R.G.error_flag, fb, spin_limit)) return;      // (handed to the one-workgroup solve)
        // ---- node phase over the own nodes
        float dev = 0.f;
        const int new_row0 = (cur ^ 1) * NN;
        for (int g0 = g_lo; g0 < g_hi; g0 += n_grp) {
            const int g = g0 + tid / BPC_GROUP;
            const bool live = g < g_hi;
            const int n = live ? nrot[g] : 0;
            float bb[6] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            if (live && sweep >= 0) {
                const int q = n == 6 ? 2 : 1, base = bp_start[g], deg = (bp_start[g + 1] - base) / q;
                for (int kb = gl; kb < deg; kb += BPC_GROUP * 4) {   // 4 independent row loads in flight
                    float4 lo4[4]; float2 hi2[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = kb + u * BPC_GROUP;
                        lo4[u] = make_float4(1.f, 1.f, 1.f, 1.f); hi2[u] = make_float2(1.f, 1.f);
                        if (k < deg) {
                            const float* m = X.inbox + (size_t)(base + k * q) * 4;
                            lo4[u] = *(const float4*)m;
                            if (n == 6) hi2[u] = *(const float2*)(m + 4);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        bb[0] *= lo4[u].x; bb[1] *= lo4[u].y; bb[2] *= lo4[u].z;
                        if (n == 6) { bb[3] *= lo4[u].w; bb[4] *= hi2[u].x; bb[5] *= hi2[u].y; }
                    }
                    // keep the running product O(1) (rotamer.cpp:489-493 re-normalises too)
                    float mx = fmaxf(fmaxf(bb[0], bb[1]), bb[2]);
                    if (n == 6) mx = fmaxf(fmaxf(mx, bb[3]), fmaxf(bb[4], bb[5]));
                    const float rm = fast_rcp(mx);
#pragma unroll
                    for (int r = 0; r < 6; ++r) bb[r] *= rm;
                }
            }
#pragma unroll
            for (int off = BPC_GROUP / 2; off > 0; off >>= 1) {
                float mx = 0.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) { bb[r] *= __shfl_xor(bb[r], off, UP_WAVE); mx = fmaxf(mx, r < n ? bb[r] : 0.f); }
                const float rm = mx > 0.f ? fast_rcp(mx) : 1.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) bb[r] *= rm;
            }
            if (live && gl == 0) {
                // b = prob * product, then standardize (rotamer.cpp:258-273) and damp
                float v[6], nv[6], mx = 0.f;
#pragma unroll
                for (int r = 0; r < 6; ++r) { v[r] = r < n ? prob[g * 6 + r] * bb[r] : 0.f; mx = fmaxf(mx, v[r]); }
                const float rm = rcp(mx);
                const float damp = sweep < 0 ? 0.f : R.damping;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    nv[r] = 0.f;
                    if (r < n) {
                        const float o = nb[g * 6 + r];
                        nv[r] = damp != 0.f ? (1.f - damp) * rm * v[r] + damp * o : rm * v[r];
                        dev = fmaxf(nv[r] - o, dev);                  // signed, rotamer.cpp:275-281
                    }
                }
                st_wt16(X.nbx_w, (new_row0 + g) * 8, nv[0], nv[1], nv[2], nv[3]);
                st_wt16(X.nbx_w, (new_row0 + g) * 8 + 4, nv[4], nv[5], 0.f, 0.f);
            }
        }
        const bool check = sweep >= 0 && ((iter + 1) % R.chunk == 0);
        if (check) {
            const float wg_dev = block_max(dev, scratch);
            if (tid == 0) st_agent(X.dev + (cur ^ 1) * 16 + c, wg_dev);
        }
        if (!cluster_barrier(X.bar, phase, C, R.G.error_flag, fb, spin_limit)) return;      // (handed to the one-workgroup solve)
        cur ^= 1;
        reload(cur);
        if (sweep >= 0) {
            ++iter;
            if (check) {
                float maxdev = 0.f;
                for (int cc = 0; cc < C; ++cc) maxdev = fmaxf(maxdev, ld_agent(X.dev + cur * 16 + cc));
                if (!(maxdev > R.tol && iter < R.max_iter)) break;     // rotamer.cpp:1038
            }
        }
    }
    if (c == 0 && tid == 0) { R.iters[s] = iter; if (iter >= R.max_iter - R.chunk - 1) R.n_bad[s] += 1; }

    // ---- marginals (rotamer.cpp:1053-1059): own nodes normalise, everyone reloads
    float* out_nb = R.nb_cur + (size_t)s * NN * 6;
    for (int g = g_lo + tid; g < g_hi; g += nt) {
        const int n = nrot[g];
        float sum = 0.f;
        for (int r = 0; r < n; ++r) sum += nb[g * 6 + r];
        const float rs = rcp(sum);
        for (int r = 0; r < 6; ++r) st_agent(out_nb + g * 6 + r, r < n ? nb[g * 6 + r] * rs : 0.f);
    }
    if (c == 0) for (int g = tid; g < e1; g += nt) { st_agent(out_nb + g * 6, 1.f); for (int r = 1; r < 6; ++r) st_agent(out_nb + g * 6 + r, 0.f); }
    if (!cluster_barrier(X.bar, phase, C, R.G.error_flag, fb, spin_limit)) return;      // (handed to the one-workgroup solve)
    for (int i = tid; i < NN * 6; i += nt) nb[i] = ld_agent(out_nb + i);
    __syncthreads();
    float* marg = R.marg + (size_t)s * R.slot_cap * 36;
    float en = 0.f;
    if (RESIDENT) {
        en += bpc_marginal(s33, Pl0, n_own[0], nb, marg, R.slot_cap, want_energy);
        en += bpc_marginal(s36, Pl1, n_own[1], nb, marg, R.slot_cap, want_energy);
        en += bpc_marginal(s66, Pl2, n_own[2], nb, marg, R.slot_cap, want_energy);
    } else {
        en += bp_marginal_range<3, 3>(Cx, lo[0], lo[0] + n_own[0], nb, tid, nt, want_energy);
        en += bp_marginal_range<3, 6>(Cx, lo[1], lo[1] + n_own[1], nb, tid, nt, want_energy);
        en += bp_marginal_range<6, 6>(Cx, lo[2], lo[2] + n_own[2], nb, tid, nt, want_energy);
    }
    // ---- leave the accumulators clean for the next force evaluation (every class, split over the cluster)
    float* P = R.P + (size_t)s * R.slot_cap * 36;
    int* active_w = R.slot_active + (size_t)s * R.slot_cap;
    int* active_last = R.slot_active_last + (size_t)s * R.slot_cap;
    const int n11 = cls[CL11 + 1] - cls[CL11], lo11 = cls[CL11] + (int)((long)n11 * c / C), hi11 = cls[CL11] + (int)((long)n11 * (c + 1) / C);   // as clear_all_classes splits it
    if (want_energy) {
        for (int sl = lo11 + tid; sl < hi11; sl += nt)                     // 1-1 edges (rotamer.cpp:861): -log(exp(-E))
            if (active_w[sl]) en += P[PIDX6(R.slot_cap, sl, 0, 0)];
        for (int g = g_lo + tid; g < g_hi; g += nt) {                      // node_free_energy, rotamer.cpp:292-302
            const int n = nrot[g];
            float e = R.node_off[(size_t)s * NN + g];
            for (int r = 0; r < n; ++r) { const float b = nb[g * 6 + r]; e += b * logf((1e-10f + b) * rcp(1e-10f + prob[g * 6 + r])); }
            en += e;
        }
        if (c == 0) for (int g = tid; g < e1; g += nt) {                   // 1-state nodes: b = prob = 1
            en += R.node_off[(size_t)s * NN + g] + 1.f * logf((1e-10f + 1.f) * rcp(1e-10f + R.node_prob[((size_t)s * NN + g) * 6]));
        }
        const float tot = block_sum(en, scratch);
        if (tid == 0) st_agent(X.en_part + c, tot);
    }
    __syncthreads();   // all reads of P / active by this workgroup are done
    clear_all_classes(P, R.slot_cap, cls, c, C, tid, nt);
    for (int k = 0; k < N_CLASS; ++k) {   // retire the activity flags over the SAME per-class shares this workgroup read them on
        const int b = cls[k], n = cls[k + 1] - b;
        const int lo_k = b + (int)((long)n * c / C), hi_k = b + (int)((long)n * (c + 1) / C);
        for (int i = lo_k + tid; i < hi_k; i += nt) { active_last[i] = active_w[i]; active_w[i] = 0; }
    }
    if (want_energy) {
        (void)cluster_barrier(X.bar, phase, C, R.G.error_flag, nullptr, spin_limit);
        if (c == 0 && tid == 0) { float t = 0.f; for (int cc = 0; cc < C; ++cc) t += ld_agent(X.en_part + cc); R.energy[s] = t; }
    }
}

static int device_cu_count() {
    static int n = 0;
    if (!n) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 64; }
    return n;
}
extern "C" int upk_device_cu_count() { return device_cu_count(); }
extern "C" int upk_rotamer_bp_cluster_threads() { return BPC_BLOCK; }   // slots of one class a workgroup can own
extern "C" int upk_rotamer_bp_cluster_capacity(const upk_rotamer_t* R) {   // floats of pair matrices one workgroup can hold
    const int fixed = R->n_node * 14 + 48;
    return (int)(156 * 1024 / sizeof(float)) - fixed;
}
// one-workgroup solve: BP_BLOCK lanes streaming every matrix, or BP_BLOCK / 2 lanes with the first trips of each class pinned in registers
static void bp_launch(const upk_launch_t* L, const upk_rotamer_t* R, int want_energy, int only_fallback, size_t lds, int lds_msg_floats) {
    // TWO compiled solves.  (i) 512 lanes x 256 registers, the dense per-solve inbox, one slot of every class pinned in registers per lane:
    // every batch size (round 4 measured it against two pinned 6x6 slots, a 1024-lane streaming form and larger pinned sets across proteins
    // and batch sizes; the losers are gone: `git log` has them).  (ii) 1024 lanes streaming every matrix over the CACHED inbox layout: the
    // hand-over solve behind a cluster launch, systems whose layout scratch does not fit, UPSIDE_HIP_BP_COMPACT=0 (tests).
    const dim3 grid(1, L->n_system);
    static int compact = -1;  // UPSIDE_HIP_BP_COMPACT=0: the cached inbox layout (tests)
    if (compact < 0) { const char* e = getenv("UPSIDE_HIP_BP_COMPACT"); compact = (e && !atoi(e)) ? 0 : 1; }
    // (the layout pass borrows the LDS inbox for an activity bit per cached row and a prefix per 32 rows: at most two rows per slot)
    const size_t layout_scratch = (((size_t)2 * R->slot_cap + 64) / 32 * 2 + 2) * sizeof(int);
    const bool dense = compact && R->slot_row && R->row_start && (size_t)lds_msg_floats * sizeof(float) >= layout_scratch;
    if (!only_fallback && dense) {
        upk_rotamer_t Rl = *R;
        // the dense layout as a launch of its own in front of the solve (R->bp_layout allocated by the host node: from 512 systems on),
        // when its scratch fits a quarter of a CU's LDS; else inside the solve
        static int hand_back = -1;      // UPSIDE_HIP_BP_LAYOUT=2 (tests): no scratch at all, so that every system is handed back to its solve
        if (hand_back < 0) { const char* e = getenv("UPSIDE_HIP_BP_LAYOUT"); hand_back = (e && atoi(e) == 2) ? 1 : 0; }
        const int scratch_words = hand_back ? 1 : (int)((((size_t)2 * (R->slot_cap / 4 + 32) + 64) / 32) * 2 + 2);
        const size_t layout_lds = ((size_t)R->n_node * (BP_NODE_STRIDE + 1) + 64 + N_CLASS + 16) * sizeof(float) + (size_t)scratch_words * sizeof(int) + 64;
        if (R->bp_layout && layout_lds <= 40 * 1024)
            hipLaunchKernelGGL(k_rotamer_bp_layout, dim3(L->n_system), dim3(BPL_THREADS), layout_lds, ST(L), *R, lds_msg_floats, R->bp_layout, scratch_words);
        else Rl.bp_layout = nullptr;
        hipLaunchKernelGGL((k_rotamer_bp<BP_BLOCK / 2, 1, 1, 1, true>), grid, dim3(BP_BLOCK / 2), lds, ST(L), Rl, want_energy, only_fallback, lds_msg_floats);
    } else
        hipLaunchKernelGGL((k_rotamer_bp<BP_BLOCK, 0, 0, 0>), grid, dim3(BP_BLOCK), lds, ST(L), *R, want_energy, only_fallback, lds_msg_floats);
}
extern "C" int upk_rotamer_bp(const upk_launch_t* L, const upk_rotamer_t* R, int want_energy) {
    UPK_FLUSH(L);
    const size_t lds_base = ((size_t)R->n_node * (BP_NODE_ARRAYS * BP_NODE_STRIDE + 2) + 64 + 8) * sizeof(float);
    if (lds_base > 155 * 1024) return 9004;
    // LDS left over holds the messages to the 3-state nodes (at most all of the inbox: 16 floats per slot)
    static int lds_msg_kb = -1;   // UPSIDE_HIP_BP_LDS_MSG_KB (experiments): 0 keeps every message in global memory
    if (lds_msg_kb < 0) { const char* e = getenv("UPSIDE_HIP_BP_LDS_MSG_KB"); lds_msg_kb = e ? atoi(e) : 160; }
    size_t msg_bytes = (size_t)lds_msg_kb * 1024;
    // LDS of the one-workgroup solve, beliefs + inbox: all 160 KB of the CU (the kernel has no static LDS)
    if (lds_base + msg_bytes > (size_t)160 * 1024) msg_bytes = lds_base >= (size_t)160 * 1024 ? 0 : (size_t)160 * 1024 - lds_base;
    if (msg_bytes > (size_t)R->slot_cap * 64) msg_bytes = (size_t)R->slot_cap * 64;
    msg_bytes &= ~(size_t)15;
    const int lds_msg_floats = (int)(msg_bytes / sizeof(float));
    const size_t lds = lds_base + msg_bytes;
    const int C = R->bp_C;
    if (C > 1) {
        // clusters of C co-resident workgroups, one per CU: at most CUs / C systems per launch (multiples of 8 keep a
        // cluster on one XCD under round-robin workgroup dispatch, so its exchange stays in that XCD's L2)
        int chunk = device_cu_count() / C;
        if (chunk >= 8) chunk &= ~7;
        if (chunk >= 1) {
            const int p_cap = upk_rotamer_bp_cluster_capacity(R);
            const size_t split_lds = ((size_t)R->n_node * 14 + 64) * sizeof(float);
            for (int s0 = 0; s0 < L->n_system; s0 += chunk) {
                const int n = L->n_system - s0 < chunk ? L->n_system - s0 : chunk;
                // grid.x rounded up to a multiple of 8 (the surplus workgroups leave at once): workgroup b is dispatched to XCD b mod 8, so the C
                // workgroups of a system -- linear ids s + c * grid.x -- then share an XCD and their exchange stays in its L2, also when fewer
                // than 8 systems are solved (placement for speed only: the protocol is placement independent).  One 300-residue system, 6
                // workgroups: solve 211 -> 200 us.  (Round 6, measured on that placement and not taken: plain stores + sc1 loads and no acquire
                // fence, which is valid only while the cluster shares an L2: 180 us, 2 235 -> 2 287 steps/s -- 0.9 us per barrier; the rest of a
                // 10 us sweep is the counter round trips and the phases' own dependent loads.)
                const int gx = (n + 7) & ~7;
                if (R->bp_resident) hipLaunchKernelGGL(k_rotamer_bp_cluster<true>, dim3(gx, C), dim3(BPC_BLOCK), 156 * 1024, ST(L), *R, want_energy, C, s0, n, p_cap);
                else hipLaunchKernelGGL(k_rotamer_bp_cluster<false>, dim3(gx, C), dim3(BPC_BLOCK), split_lds, ST(L), *R, want_energy, C, s0, n, p_cap);
            }
            bp_launch(L, R, want_energy, 1, lds_base, 0);   // rare path: no LDS inbox, so that the (normally empty) launch does not wait for a whole CU's LDS
            return launch_status();
        }
    }
    bp_launch(L, R, want_energy, 0, lds, lds_msg_floats);
    return launch_status();
}
