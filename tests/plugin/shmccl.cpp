// TEST TRANSPORT, not a product component: a stand-in for the nine RCCL entry points that csrc/comm_rccl.cpp binds, so that the
// cross-rank branch of the replica exchange (straddling swap pairs, staging rows, grouped send/receive, the multi-rank loop of
// upside_main) can run with TWO PROCESSES ON ONE GPU -- RCCL itself needs one GPU per rank.  Selected with
// UPSIDE_HIP_COMM_LIB=<path to this library>.  Ranks meet in a POSIX shared-memory segment named by the "unique id";
// payloads are staged device -> shared host memory -> device with blocking copies after draining the caller's stream
// (a collective library may complete earlier than stream order demands, never later).
//
// Semantics kept from RCCL: ncclCommInitRank returns when every rank has joined; ncclAllGather is a collective of all ranks;
// ncclSend / ncclRecv between ncclGroupStart / ncclGroupEnd are matched per ordered (source, destination) pair in call order
// and do not block each other (all sends of the group are posted before its receives are awaited).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclHalf = 6,
               ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ShmComm* ncclComm_t;
}

namespace {
constexpr int MAX_RANKS = 4, RING = 16;
constexpr size_t SLOT_BYTES = 64 * 1024, GATHER_BYTES = 64 * 1024;
struct Mailbox {                       // one per ordered (source, destination) pair: a ring of RING messages
    std::atomic<uint64_t> written, consumed;
    size_t bytes[RING];
    char data[RING][SLOT_BYTES];
};
struct Segment {
    std::atomic<int> joined;
    std::atomic<uint64_t> barrier_count;                 // monotone: every rank adds 1 per barrier
    char gather[MAX_RANKS][GATHER_BYTES];
    Mailbox box[MAX_RANKS][MAX_RANKS];
};
size_t type_size(ncclDataType_t t) {
    switch (t) { case ncclInt8: case ncclUint8: return 1; case ncclFloat16: return 2; case ncclInt32: case ncclUint32: case ncclFloat32: return 4; default: return 8; }
}
bool spin_until(const std::function<bool()>& done, double seconds = 120.) {
    const auto t0 = std::chrono::steady_clock::now();
    for (long i = 0; !done(); ++i) {
        if (i > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
    }
    return true;
}
struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t stream; };
thread_local int group_depth = 0;
thread_local std::vector<std::pair<ShmComm*, Op>> group_ops;
}  // namespace

struct ShmComm {
    Segment* seg = nullptr; std::string name; int rank = 0, world = 1; uint64_t barriers_done = 0;
    bool barrier() {
        ++barriers_done;
        seg->barrier_count.fetch_add(1);
        const uint64_t target = barriers_done * (uint64_t)world;
        return spin_until([&] { return seg->barrier_count.load() >= target; });
    }
};

static ncclResult_t run_ops(std::vector<std::pair<ShmComm*, Op>>& ops) {
    // drain the streams the payloads were produced on, post every send, then await the receives in call order
    for (auto& co : ops) if (hipStreamSynchronize(co.second.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (auto& co : ops) {
        ShmComm& c = *co.first; const Op& o = co.second;
        if (!o.send) continue;
        if (o.bytes > SLOT_BYTES) return ncclInvalidArgument;
        Mailbox& m = c.seg->box[c.rank][o.peer];
        const uint64_t w = m.written.load();
        if (!spin_until([&] { return w - m.consumed.load() < (uint64_t)RING; })) return ncclSystemError;
        if (hipMemcpy(m.data[w % RING], o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        m.bytes[w % RING] = o.bytes;
        m.written.store(w + 1, std::memory_order_release);
    }
    for (auto& co : ops) {
        ShmComm& c = *co.first; const Op& o = co.second;
        if (o.send) continue;
        Mailbox& m = c.seg->box[o.peer][c.rank];
        const uint64_t r = m.consumed.load();
        if (!spin_until([&] { return m.written.load(std::memory_order_acquire) > r; })) return ncclSystemError;
        if (m.bytes[r % RING] != o.bytes) return ncclInvalidArgument;
        if (hipMemcpy(o.buf, m.data[r % RING], o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        m.consumed.store(r + 1, std::memory_order_release);
    }
    return ncclSuccess;
}

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/upside_shmccl_%ld_%lld", (long)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ShmComm* c = new ShmComm; c->rank = rank; c->world = nranks; c->name.assign(id.internal, strnlen(id.internal, sizeof(id.internal)));
    // every rank may create: O_CREAT without O_EXCL, ftruncate to the same size (a fresh segment is zero-filled = the initial state)
    const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void* p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->seg = (Segment*)p;
    c->seg->joined.fetch_add(1);
    if (!spin_until([&] { return c->seg->joined.load() >= nranks; })) { munmap(p, sizeof(Segment)); delete c; return ncclSystemError; }
    *comm = c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->seg) {
        const int left = c->seg->joined.fetch_sub(1) - 1;
        munmap(c->seg, sizeof(Segment));
        if (left == 0) shm_unlink(c->name.c_str());
    }
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t type, ncclComm_t c, hipStream_t stream) {
    const size_t bytes = sendcount * type_size(type);
    if (bytes > GATHER_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpy(c->seg->gather[c->rank], sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (!c->barrier()) return ncclSystemError;                       // every contribution is in place
    for (int r = 0; r < c->world && bytes; ++r)
        if (hipMemcpy((char*)recvbuff + (size_t)r * bytes, c->seg->gather[r], bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!c->barrier()) return ncclSystemError;                       // ... and read by everyone before the next collective overwrites it
    return ncclSuccess;
}
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    group_ops.push_back({c, Op{true, const_cast<void*>(sendbuff), count * type_size(type), peer, stream}});
    if (group_depth == 0) { auto ops = std::move(group_ops); group_ops.clear(); return run_ops(ops); }
    return ncclSuccess;
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    group_ops.push_back({c, Op{false, recvbuff, count * type_size(type), peer, stream}});
    if (group_depth == 0) { auto ops = std::move(group_ops); group_ops.clear(); return run_ops(ops); }
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { ++group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (group_depth <= 0) return ncclInvalidUsage;
    if (--group_depth > 0) return ncclSuccess;
    auto ops = std::move(group_ops); group_ops.clear();
    return run_ops(ops);
}
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error"; case ncclUnhandledCudaError: return "unhandled HIP error"; case ncclSystemError: return "system error (shared memory / peer timeout)";
        case ncclInvalidArgument: return "invalid argument"; case ncclInvalidUsage: return "invalid usage"; default: return "internal error";
    }
}
}
