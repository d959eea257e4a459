// Replica exchange across GPUs (the reference's ReplicaExchange::attempt_swaps, /root/reference/src/main.cpp:227-275, for a
// temperature ladder spread over one process per GPU): RCCL over xGMI, no host staging.
//
// Global system g lives on rank g / n_system (contiguous temperature blocks, so only the pairs that straddle a block
// boundary cross GPUs).  One swap set, all on the engine's stream:
//   1. (first set of an attempt) force pass, total potentials summed on the device, ncclAllGather of ONE fp32 per replica;
//   2. every rank runs the identical Metropolis kernel on the identical gathered arrays with the shared counter RNG
//      (k_replica_decide) -- no verdict is communicated; accepted pairs trade their gathered energies, so the later sets of
//      the attempt reuse them (a temperature exchange of one Hamiltonian permutes the energies);
//   3. every pair that straddles two ranks sends its coordinates to the partner rank and receives the partner's into a
//      staging row (one grouped ncclSend/ncclRecv of 3*n_atom floats per pair), whatever the verdict -- a couple of KB per
//      rank and set buys a path with no host round trip; k_replica_apply then swaps accepted on-rank pairs in place and
//      copies the staging row in for accepted cross-rank pairs.  Momenta and temperatures stay with the slot (main.cpp:244-247).
// librccl is loaded on first use (dlopen), so processes that never exchange across GPUs do not pay for it.
#include "../../include/upside_engine_c.h"
#include "engine.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace std;

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};
Rccl& rccl() {
    static Rccl r;
    if (r.lib) return r;
    // UPSIDE_HIP_COMM_LIB names another library with the same nine entry points (tests: tests/plugin/libshmccl.so runs
    // two ranks on one GPU, which RCCL cannot).  A TEST seam: honoured only together with UPSIDE_HIP_TESTING=1, so that a stray
    // variable in a production environment cannot make the library load arbitrary code in RCCL's place.
    const char* over = getenv("UPSIDE_HIP_COMM_LIB");
    const char* testing = getenv("UPSIDE_HIP_TESTING");
    if (over && !(testing && atoi(testing) == 1)) {
        static bool warned = false;
        if (!warned) { warned = true; fprintf(stderr, "upside_hip: UPSIDE_HIP_COMM_LIB is ignored without UPSIDE_HIP_TESTING=1 (using RCCL)\n"); }
        over = nullptr;
    }
    if (over) {
        r.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib) throw string("cannot load UPSIDE_HIP_COMM_LIB=") + over + ": " + dlerror();
    } else {
        // a librccl that is ALREADY mapped into the process wins (torch ships its own next to libtorch_hip.so and has loaded it by the
        // time bench.py gets here): two RCCL instances in one process would each run their own bootstrap and proxy threads
        for (const char* name : {"librccl.so.1", "librccl.so"}) { r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (r.lib) break; }
        if (!r.lib) {      // (RTLD_NOLOAD matches by soname; a copy loaded under a private path is found through the global scope)
            void* self = dlopen(nullptr, RTLD_NOW);
            if (self && dlsym(self, "ncclCommInitRank") && dlsym(self, "ncclGroupEnd")) r.lib = self;
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { if (r.lib) break; r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); }
    }
    if (!r.lib) throw string("cannot load librccl.so: ") + dlerror();
    auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) throw string("librccl.so lacks ") + n; return p; };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");        // optional
    return r;
}
void nccl_check(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw string("RCCL error in ") + what + ": " + rccl().GetErrorString(r);
}
}  // namespace

// host staging of one swap set's pair list and plan: pinned, owned by the communicator and recycled only after the copy that
// reads it has completed (an asynchronous copy from a pageable or stack buffer would either block the call or outlive its source)
struct PinnedSet {
    int* host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool pending = false;
    int* reserve(size_t n) {
        if (pending) { hip_check(hipEventSynchronize(done), "hipEventSynchronize"); pending = false; }
        if (n > cap) {
            if (host) (void)hipHostFree(host);
            hip_check(hipHostMalloc((void**)&host, n * sizeof(int), hipHostMallocDefault), "hipHostMalloc"); cap = n;
        }
        if (!done) hip_check(hipEventCreateWithFlags(&done, hipEventDisableTiming), "hipEventCreate");
        return host;
    }
    void sent(hipStream_t st) { hip_check(hipEventRecord(done, st), "hipEventRecord"); pending = true; }
    ~PinnedSet() { if (done) { if (pending) (void)hipEventSynchronize(done); (void)hipEventDestroy(done); } if (host) (void)hipHostFree(host); }
};
struct ReplicaComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    DevBuf<float> energy_local, energy_all, beta_all, staging;
    DevBuf<const float*> node_pot;
    DevBuf<int> pairs_dev, plan_dev, accepted_dev, draw_dev;
    PinnedSet stage[2];              // [0] pair list, [1] plan; double use per call is serialised by the events
    int n_node_pot = 0, pair_cap = 0, staging_rows = 0;
    bool broken = false;             // a collective failed part-way: the communicator is aborted, later calls are refused
    uint64_t attempt_round = ~0ull; uint64_t attempt_compute = 0;   // the attempt the gathered energies belong to
    ~ReplicaComm() { if (comm) (void)((broken && rccl().CommAbort) ? rccl().CommAbort(comm) : rccl().CommDestroy(comm)); }
};
static void comm_deleter(void* p) { delete (ReplicaComm*)p; }

#define API_TRY try {
#define API_CATCH } catch (const string& s) { upside_hip_set_last_error(s.c_str()); return 1; } catch (const std::exception& e) { upside_hip_set_last_error(e.what()); return 1; } \
    catch (...) { upside_hip_set_last_error("unknown error"); return 1; }
extern "C" void upside_hip_set_last_error(const char* msg);

extern "C" int upside_hip_comm_get_unique_id(char* id_out) {
    API_TRY
    static_assert(sizeof(ncclUniqueId) == UPSIDE_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return 0;
    API_CATCH
}

extern "C" int upside_hip_comm_init(DerivEngine* e, int rank, int world, const char* id_in, const float* temperature_global) {
    API_TRY
    if (!e || world < 1 || rank < 0 || rank >= world) throw string("invalid rank / world");
    const int S = e->ctx.n_system;
    unique_ptr<ReplicaComm> c(new ReplicaComm);
    c->rank = rank; c->world = world;
    ncclUniqueId id; memcpy(&id, id_in, sizeof(id));
    nccl_check(rccl().CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
    c->energy_local.alloc(S); c->energy_all.alloc((size_t)world * S); c->draw_dev.alloc(1);
    vector<float> beta((size_t)world * S);
    for (size_t g = 0; g < beta.size(); ++g) beta[g] = 1.f / temperature_global[g];
    c->beta_all.upload(beta);
    vector<const float*> ptrs;
    for (auto& n : e->nodes) if (n.computation->potential_term) ptrs.push_back(static_cast<PotentialNode*>(n.computation.get())->potential_dev.p);
    c->n_node_pot = (int)ptrs.size(); c->node_pot.upload(ptrs);
    if (e->comm) e->comm_free(e->comm);
    e->comm = c.release(); e->comm_free = comm_deleter;
    return 0;
    API_CATCH
}

// Do all ranks hold the same 64-bit value (the launcher's digest of /input/potential)?  A collective: every rank learns the
// answer, so a mismatch ends every rank together instead of leaving the others waiting in a later collective.
extern "C" int upside_hip_comm_agree(DerivEngine* e, unsigned long long value, int* first_differing_rank) {
    API_TRY
    if (!e || !e->comm) throw string("no communicator");
    ReplicaComm& c = *(ReplicaComm*)e->comm;
    e->ctx.flush();
    DevBuf<unsigned long long> mine, all;
    mine.upload(vector<unsigned long long>(1, value)); all.alloc((size_t)c.world);
    const ncclResult_t ag = rccl().AllGather(mine.p, all.p, sizeof(value), ncclChar, c.comm, e->ctx.stream);
    if (ag != ncclSuccess) { c.broken = true; nccl_check(ag, "ncclAllGather"); }
    vector<unsigned long long> got((size_t)c.world);
    hip_check(hipMemcpyAsync(got.data(), all.p, got.size() * sizeof(value), hipMemcpyDeviceToHost, e->ctx.stream), "hipMemcpyAsync");
    hip_check(hipStreamSynchronize(e->ctx.stream), "hipStreamSynchronize");
    int bad = -1;
    for (int r = c.world - 1; r >= 1; --r) if (got[r] != got[0]) bad = r;
    if (first_differing_rank) *first_differing_rank = bad;
    return 0;
    API_CATCH
}

extern "C" int upside_hip_comm_free(DerivEngine* e) {
    if (e && e->comm) { e->sync(); e->comm_free(e->comm); e->comm = nullptr; }
    return 0;
}

extern "C" int upside_hip_comm_replica_swap(DerivEngine* e, int n_pair, const int* pairs_global, uint32_t base_seed, uint64_t round,
                                            int first_set, int* accepted) {
    API_TRY
    if (!e || !e->comm) throw string("upside_hip_comm_init first");
    ReplicaComm& c = *(ReplicaComm*)e->comm;
    if (c.broken) throw string("the communicator was aborted after a failed collective");
    const int S = e->ctx.n_system, G = S * c.world, lo = c.rank * S;
    hipStream_t st = e->ctx.stream;
    const int n_row = e->pos->n_elem * e->pos->stride;
    // plan of this set for this rank (host arithmetic on the pair list only)
    vector<int> plan((size_t)n_pair * 3, 0);
    struct Cross { int local, peer, slot; };
    vector<Cross> cross;
    vector<char> used((size_t)G, 0);
    for (int p = 0; p < n_pair; ++p) {
        const int g1 = pairs_global[2 * p], g2 = pairs_global[2 * p + 1];
        if (g1 < 0 || g1 >= G || g2 < 0 || g2 >= G || g1 == g2) throw string("invalid system in swap set");
        if (used[g1] || used[g2]) throw string("Overlapping indices in swap set.");
        used[g1] = used[g2] = 1;
        const int r1 = g1 / S, r2 = g2 / S;
        if (r1 == c.rank && r2 == c.rank) { plan[3 * p] = 1; plan[3 * p + 1] = g1 - lo; plan[3 * p + 2] = g2 - lo; }
        else if (r1 == c.rank || r2 == c.rank) {
            const int mine = r1 == c.rank ? g1 : g2, peer = r1 == c.rank ? r2 : r1;
            plan[3 * p] = 2; plan[3 * p + 1] = mine - lo; plan[3 * p + 2] = (int)cross.size();
            cross.push_back(Cross{mine - lo, peer, (int)cross.size()});
        }
    }
    if (n_pair > c.pair_cap) { c.pair_cap = n_pair; c.pairs_dev.alloc((size_t)n_pair * 2); c.plan_dev.alloc((size_t)n_pair * 3); c.accepted_dev.alloc(n_pair); }
    if ((int)cross.size() > c.staging_rows) { c.staging_rows = (int)cross.size(); c.staging.alloc((size_t)c.staging_rows * n_row); }
    if (n_pair) {      // through the communicator's pinned buffers: truly asynchronous, and the sources outlive the copies
        int* hp = c.stage[0].reserve((size_t)n_pair * 2); memcpy(hp, pairs_global, (size_t)n_pair * 2 * sizeof(int));
        hip_check(hipMemcpyAsync(c.pairs_dev.p, hp, (size_t)n_pair * 2 * sizeof(int), hipMemcpyHostToDevice, st), "H2D");
        c.stage[0].sent(st);
        int* hq = c.stage[1].reserve(plan.size()); memcpy(hq, plan.data(), plan.size() * sizeof(int));
        hip_check(hipMemcpyAsync(c.plan_dev.p, hq, plan.size() * sizeof(int), hipMemcpyHostToDevice, st), "H2D");
        c.stage[1].sent(st);
    }
    if (first_set) {   // main.cpp:251-256: energies of every system, once per attempt
        e->compute(PotentialAndDerivMode);
        upk_check(upk_sum_potentials(&e->ctx.L, c.node_pot.p, c.n_node_pot, c.energy_local.p), "sum_potentials");
        { const ncclResult_t ag = rccl().AllGather(c.energy_local.p, c.energy_all.p, (size_t)S, ncclFloat, c.comm, st);
          if (ag != ncclSuccess) { c.broken = true; nccl_check(ag, "ncclAllGather"); } }
        hip_check(hipMemsetAsync(c.draw_dev.p, 0, sizeof(int), st), "memset");
        c.attempt_round = round; c.attempt_compute = e->n_compute;
    } else if (c.attempt_round != round || c.attempt_compute != e->n_compute)
        throw string("a later swap set needs the first set of the same attempt (same round, no evaluation in between)");
    if (n_pair) upk_check(upk_replica_decide(&e->ctx.L, c.energy_all.p, c.beta_all.p, n_pair, c.pairs_dev.p, base_seed, round, c.draw_dev.p, c.accepted_dev.p), "replica_decide");
    if (!cross.empty()) {   // coordinates of the straddling pairs, both directions, one group
        nccl_check(rccl().GroupStart(), "ncclGroupStart");
        try {
            for (auto& x : cross) {
                nccl_check(rccl().Send(e->pos->output.p + (size_t)x.local * n_row, (size_t)n_row, ncclFloat, x.peer, c.comm, st), "ncclSend");
                nccl_check(rccl().Recv(c.staging.p + (size_t)x.slot * n_row, (size_t)n_row, ncclFloat, x.peer, c.comm, st), "ncclRecv");
            }
        } catch (...) {      // a group left open would leave this rank's peers waiting: close it, give the communicator up
            (void)rccl().GroupEnd();
            c.broken = true;
            throw;
        }
        const ncclResult_t ge = rccl().GroupEnd();
        if (ge != ncclSuccess) { c.broken = true; nccl_check(ge, "ncclGroupEnd"); }
    }
    if (n_pair) upk_check(upk_replica_apply(&e->ctx.L, e->pos->coord(), n_pair, c.plan_dev.p, c.accepted_dev.p, c.staging.p), "replica_apply");
    if (accepted && n_pair) {   // the caller wants the verdicts (logging): the only synchronisation of the call
        hip_check(hipMemcpyAsync(accepted, c.accepted_dev.p, (size_t)n_pair * sizeof(int), hipMemcpyDeviceToHost, st), "D2H");
        e->sync();
    }
    return 0;
    API_CATCH
}
