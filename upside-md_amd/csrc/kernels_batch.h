// Merged launches: several INDEPENDENT kernels of a force pass -- the list upkeep of the five interaction graphs, the forward pair
// passes of graphs that do not depend on each other, their backward passes -- run as workgroup ranges of ONE launch.
//
// Why: a small batch is a chain of dependent launches (4.5-5 us of host time per eager launch on this platform, ~3 us on the device
// between dependent kernels, tools/ubench/launch_chain.hip); kernels of different graphs need not wait for each other, and side by
// side in one launch they also fill each other's tails.  A large batch saves the drain / fill between its many short kernels.
//
// How: between upk_batch_begin and upk_batch_end the launchers of kernels_igraph.hip / kernels_pair.hip / kernels_rotamer.hip that
// have a batch form do not launch: they append (kind, grid, LDS bytes, arguments) to the CHAIN the caller named with
// upk_batch_chain -- one chain per graph.  Launches of one chain keep their order; the k-th launches of all chains form STAGE k,
// and upk_batch_end runs stage after stage, each as one launch of k_batch_*: a workgroup finds its entry from its index, takes the
// entry's arguments from the kernel-argument block (nothing is staged through device memory) and runs the kernel's body with the
// block coordinates it would have had on its own.  Every merged launch uses 1024-lane workgroups: the bodies take blockDim.x as
// their size (more wavefronts than a body asked for only share its rows or elements among more wavefronts).
// A launcher without a batch form that is called inside a batch first runs what the batch holds (upk_batch_run), then launches.
#pragma once
#include <vector>

enum { BK_CHECK = 1, BK_BUILD_ROT, BK_BUILD_COV, BK_BUILD_ENV, BK_BUILD_HB, BK_REFINE, BK_REFINE_SYM, BK_REFINE_SHORT, BK_ORDER,
       BK_CLEAR_SLOTS, BK_BUILD_SLOTS, BK_NBR_SLOTS, BK_SLOTS_BOTH,
       BK_ROWS_HB_FWD, BK_ROWS_HB_BWD, BK_ROWS_ENV_FWD, BK_COV_ROWS2, BK_COV_ROWS2_POLY, BK_ENV_BWD, BK_COV_BWD2, BK_COV_BWD2_POLY, BK_BWD_FINISH };
static inline bool bk_is_pair(int kind) { return kind >= BK_ROWS_HB_FWD; }

#define BATCH_MAX_ENTRIES 8
#define BATCH_BLOB 3584
struct BatchEntryHdr { int kind, gx, gy, wg_end, off, i0, i1, pad; double d0; };     // wg_end: end of the entry's workgroup range; i0, i1, d0: scalar arguments
struct BatchArgs { int n; int pad[3]; BatchEntryHdr e[BATCH_MAX_ENTRIES]; unsigned char blob[BATCH_BLOB]; };
static_assert(sizeof(BatchArgs) <= 4096, "the merged launch takes its arguments as one kernel-argument block");

namespace {
struct BatchItem { int kind, gx, gy, i0, i1; double d0; size_t lds; std::vector<unsigned char> args; };
struct BatchState {
    bool open = false; int chain = 0; bool skip_nbr_slots = false;
    std::vector<char> chain_fused;                  // chain c has submitted fused per-element ops that are still queued (program order: they precede its next item)
    std::vector<std::vector<BatchItem>> chains;     // chains[c] = the launches of chain c in order
    long n_merged = 0, n_items = 0;
};
inline BatchState* batch_of(const upk_launch_t* L) { return (BatchState*)L->batch; }
}  // namespace

extern "C" int upk_batch_run(const upk_launch_t* L);
// append a launch to the current chain; false: no batch is open (the caller launches by itself)
static bool batch_add(const upk_launch_t* L, int kind, int gx, int gy, size_t lds, const void* a0, size_t n0, const void* a1 = nullptr, size_t n1 = 0,
                      int i0 = 0, int i1 = 0, double d0 = 0.) {
    BatchState* s = batch_of(L);
    if (!s || !s->open) return false;
    // program order inside a chain: fused per-element ops THIS chain has queued since the batch opened go out before this item joins
    // (upk_fuse_flush runs the batch's earlier items first); ops queued by other chains are independent of it and stay queued
    if ((int)s->chain_fused.size() > s->chain && s->chain_fused[s->chain]) {
        if (upk_fuse_flush(L)) return false;
        std::fill(s->chain_fused.begin(), s->chain_fused.end(), 0);
    }
    if (gx < 1 || gy < 1) return true;
    if ((int)s->chains.size() <= s->chain) s->chains.resize(s->chain + 1);
    BatchItem it; it.kind = kind; it.gx = gx; it.gy = gy; it.i0 = i0; it.i1 = i1; it.d0 = d0; it.lds = lds;
    const size_t o1 = (n0 + 15) & ~(size_t)15;
    it.args.assign(n1 ? o1 + n1 : n0, 0);
    memcpy(it.args.data(), a0, n0);
    if (n1) memcpy(it.args.data() + o1, a1, n1);
    s->chains[s->chain].push_back(std::move(it));
    return true;
}
// a fused per-element op was queued (kernels_basic.hip: fuse_submit_raw): it follows the current chain's items so far -- which is kept by
// running the batch in front of every flush of the fused queue (upk_fuse_flush) -- and precedes the chain's next item (batch_add)
extern "C" void upk_batch_fused_submitted(const upk_launch_t* L) {
    BatchState* s = batch_of(L);
    if (!s || !s->open) return;
    if ((int)s->chain_fused.size() <= s->chain) s->chain_fused.resize(s->chain + 1, 0);
    s->chain_fused[s->chain] = 1;
}
// a launcher without a batch form: what the batch holds runs first
#define UPK_BATCH_BREAK(L) do { const int r_ = upk_batch_run(L); if (r_) return r_; } while (0)
