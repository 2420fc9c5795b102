// Node-level communicator of the C ABI (include/clair_amd.h, "multi-GPU"): a thin binding of RCCL for the trivial
// work-split / gather of the sharded path.  Candidates shard with no data-path collective; what crosses
// xGMI is the 9.5 MB weight blob (once), the per-rank output rows (360 B per candidate, optional) and a few counters.
//
// librccl.so is loaded with dlopen on the first clair_comm_* call, so a single-GPU process never depends on it.
#include "../../include/clair_amd.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only; the entry points are resolved at run time

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

namespace {

thread_local std::string g_comm_error;

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;     // optional: tear-down that does not wait for peers
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r = [] {
        Rccl q;
        // CLAIR_AMD_RCCL_LIBRARY: another build of RCCL -- or the stand-in whose ncclCommInitRank never returns on one rank
        // (tests/test_comm_gpu.py: the deadline of clair_comm_create_timed against a real hang)
        const char *names[] = {getenv("CLAIR_AMD_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            q.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (q.handle) break;
        }
        if (!q.handle) { q.error = std::string("cannot load librccl.so: ") + dlerror(); return q; }
#define CLAIR_SYM(field, name)                                                            \
        q.field = (decltype(q.field))dlsym(q.handle, name);                               \
        if (!q.field && q.error.empty()) q.error = std::string("librccl.so lacks ") + name;
        CLAIR_SYM(GetUniqueId, "ncclGetUniqueId")
        CLAIR_SYM(CommInitRank, "ncclCommInitRank")
        CLAIR_SYM(CommDestroy, "ncclCommDestroy")
        CLAIR_SYM(GetErrorString, "ncclGetErrorString")
        CLAIR_SYM(Broadcast, "ncclBroadcast")
        CLAIR_SYM(AllReduce, "ncclAllReduce")
        CLAIR_SYM(AllGather, "ncclAllGather")
#undef CLAIR_SYM
        q.CommAbort = (decltype(q.CommAbort))dlsym(q.handle, "ncclCommAbort");
        return q;
    }();
    return r;
}

}  // namespace

struct clair_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    void *scratch = nullptr;   // device staging buffer for the host-pointer collectives
    size_t scratch_bytes = 0;
    std::string error;
};

namespace {

int cfail(clair_comm *c, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->error = buf; else g_comm_error = buf;
    return 1;
}

#define COMM_HIP(c, call)                                                                                    \
    do {                                                                                                     \
        hipError_t err__ = (call);                                                                           \
        if (err__ != hipSuccess) return cfail((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), __FILE__, __LINE__); \
    } while (0)
#define COMM_NCCL(c, call)                                                                                   \
    do {                                                                                                     \
        ncclResult_t res__ = (call);                                                                         \
        if (res__ != ncclSuccess) return cfail((c), "%s failed: %s (%s:%d)", #call, rccl().GetErrorString(res__), __FILE__, __LINE__); \
    } while (0)

std::string fmt(const char *f, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

// ncclCommInitRank and the communicator's stream; with `first_collective` also one all-reduce of a double (RCCL connects its
// transports on first use: a fresh communicator's first collective is where a broken xGMI / bootstrap set-up shows).  Errors go
// to `err` (this may run on a helper thread whose thread-local error text nobody reads).
int comm_init(int device, int rank, int world, const uint8_t *id, bool first_collective, clair_comm **out, std::string &err) {
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { err = fmt("rank %d / world %d out of range", rank, world); return 1; }
    int ndev = 0;
    hipError_t herr = hipGetDeviceCount(&ndev);
    if (herr != hipSuccess || ndev <= 0) { err = fmt("no HIP device available (hipGetDeviceCount: %s); RCCL needs one GPU per rank", hipGetErrorString(herr)); return 1; }
    if (device < 0 || device >= ndev) { err = fmt("device %d out of range [0,%d)", device, ndev); return 1; }
    if (!rccl().error.empty()) { err = rccl().error; return 1; }
    if ((herr = hipSetDevice(device)) != hipSuccess) { err = fmt("hipSetDevice(%d) failed: %s", device, hipGetErrorString(herr)); return 1; }
    clair_comm *c = new clair_comm();
    c->device = device; c->rank = rank; c->world = world;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t res = rccl().CommInitRank(&c->comm, world, u, rank);
    if (res != ncclSuccess) {
        err = fmt("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl().GetErrorString(res));
        delete c;
        return 1;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        err = "hipStreamCreate failed for the communicator stream";
        (void)rccl().CommDestroy(c->comm);
        delete c;
        return 1;
    }
    if (first_collective) {
        double one = 1.0;
        if (clair_comm_allreduce_f64(c, &one, 1, CLAIR_COMM_SUM) || one != (double)world) {
            err = fmt("first all-reduce on the new communicator failed on rank %d of %d: %s", rank, world,
                      c->error.empty() ? fmt("sum of ones = %g", one).c_str() : c->error.c_str());
            clair_comm_abort(c);
            return 1;
        }
    }
    *out = c;
    return 0;
}

int need_scratch(clair_comm *c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return 0;
    if (c->scratch) (void)hipFree(c->scratch);
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    COMM_HIP(c, hipMalloc(&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return 0;
}

}  // namespace

extern "C" {

const char *clair_comm_last_error(const clair_comm_t *c) { return c ? c->error.c_str() : g_comm_error.c_str(); }

int clair_comm_unique_id(uint8_t *id) {
    if (!id) return cfail(nullptr, "id is NULL");
    if (!rccl().error.empty()) return cfail(nullptr, "%s", rccl().error.c_str());
    static_assert(sizeof(ncclUniqueId) == CLAIR_COMM_ID_BYTES, "CLAIR_COMM_ID_BYTES must match ncclUniqueId");
    ncclUniqueId u;
    COMM_NCCL(nullptr, rccl().GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return 0;
}

int clair_comm_preflight(int device) {
    int ndev = 0;
    hipError_t err = hipGetDeviceCount(&ndev);
    if (err != hipSuccess || ndev <= 0)
        return cfail(nullptr, "no HIP device available (hipGetDeviceCount: %s); RCCL needs one GPU per rank", hipGetErrorString(err));
    if (device < 0 || device >= ndev) return cfail(nullptr, "device %d out of range [0,%d)", device, ndev);
    if (!rccl().error.empty()) return cfail(nullptr, "%s", rccl().error.c_str());
    COMM_HIP(nullptr, hipSetDevice(device));
    void *probe = nullptr;
    COMM_HIP(nullptr, hipMalloc(&probe, 256));
    COMM_HIP(nullptr, hipFree(probe));
    return 0;
}

int clair_comm_create(int device, int rank, int world, const uint8_t *id, clair_comm_t **out) {
    if (!out) return cfail(nullptr, "out is NULL");
    *out = nullptr;
    if (!id) return cfail(nullptr, "id is NULL");
    std::string err;
    clair_comm *c = nullptr;
    if (comm_init(device, rank, world, id, /*first_collective=*/false, &c, err)) return cfail(nullptr, "%s", err.c_str());
    *out = c;
    return 0;
}

// The same with a deadline.  ncclCommInitRank is collective and blocking: a peer that never joins, a bootstrap interface that
// swallows packets or a driver that never completes the xGMI set-up leave it waiting for ever, and the first collective on a fresh
// communicator (transports are connected lazily) can hang the same way.  Both run on a helper thread; the caller waits for it up
// to `timeout_ms`.  On expiry the caller gets CLAIR_COMM_TIMED_OUT and goes on WITHOUT this communicator (clair_amd/shard.py: every
// rank hears of it over the bootstrap sockets and all of them use the socket transport); the helper thread is left to itself, and
// if RCCL ever returns there the communicator is aborted on the spot (ncclCommAbort) and freed.
int clair_comm_create_timed(int device, int rank, int world, const uint8_t *id, int timeout_ms, clair_comm_t **out) {
    if (!out) return cfail(nullptr, "out is NULL");
    *out = nullptr;
    if (!id) return cfail(nullptr, "id is NULL");
    if (timeout_ms <= 0) return cfail(nullptr, "timeout_ms must be positive");
    struct Pending {
        std::mutex m;
        std::condition_variable cv;
        bool done = false, abandoned = false;
        int rc = 1;
        std::string err;
        clair_comm *c = nullptr;
        uint8_t id[CLAIR_COMM_ID_BYTES];
    };
    auto st = std::make_shared<Pending>();
    memcpy(st->id, id, CLAIR_COMM_ID_BYTES);
    std::thread([st, device, rank, world] {
        std::string err;
        clair_comm *c = nullptr;
        const int rc = comm_init(device, rank, world, st->id, /*first_collective=*/true, &c, err);
        std::unique_lock<std::mutex> lk(st->m);
        if (st->abandoned) {          // nobody is waiting any more: this communicator must not outlive the decision to do without it
            lk.unlock();
            if (c) clair_comm_abort(c);
            return;
        }
        st->rc = rc; st->err = err; st->c = c; st->done = true;
        st->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(st->m);
    if (!st->cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return st->done; })) {
        st->abandoned = true;
        cfail(nullptr, "ncclCommInitRank + first all-reduce of rank %d of %d did not return within %.1f s: abandoned on its helper thread "
                       "(aborted there if it ever returns)", rank, world, timeout_ms / 1e3);
        return CLAIR_COMM_TIMED_OUT;
    }
    if (st->rc) return cfail(nullptr, "%s", st->err.c_str());
    *out = st->c;
    return 0;
}

void clair_comm_destroy(clair_comm_t *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Tear-down of a communicator whose peers may never have come up (one rank's ncclCommInitRank failed while this one's succeeded):
// ncclCommDestroy is collective in effect -- it flushes outstanding work and may wait for a peer that is not there -- ncclCommAbort is not.
void clair_comm_abort(clair_comm_t *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)(rccl().CommAbort ? rccl().CommAbort(c->comm) : rccl().CommDestroy(c->comm));
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int clair_comm_allreduce_f64(clair_comm_t *c, double *values, int count, int op) {
    if (!c) return cfail(nullptr, "communicator is NULL");
    if (!values || count < 1) return cfail(c, "bad arguments to clair_comm_allreduce_f64");
    if (op < CLAIR_COMM_SUM || op > CLAIR_COMM_MIN) return cfail(c, "unknown reduction %d", op);
    COMM_HIP(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)count * sizeof(double);
    if (need_scratch(c, bytes)) return 1;
    COMM_HIP(c, hipMemcpyAsync(c->scratch, values, bytes, hipMemcpyHostToDevice, c->stream));
    const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
    COMM_NCCL(c, rccl().AllReduce(c->scratch, c->scratch, (size_t)count, ncclFloat64, ops[op], c->comm, c->stream));
    COMM_HIP(c, hipMemcpyAsync(values, c->scratch, bytes, hipMemcpyDeviceToHost, c->stream));
    COMM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int clair_comm_barrier(clair_comm_t *c) {
    double one = 1.0;
    return clair_comm_allreduce_f64(c, &one, 1, CLAIR_COMM_SUM);
}

int clair_comm_broadcast(clair_comm_t *c, void *host, int64_t bytes, int root) {
    if (!c) return cfail(nullptr, "communicator is NULL");
    if (!host || bytes < 1 || root < 0 || root >= c->world) return cfail(c, "bad arguments to clair_comm_broadcast");
    COMM_HIP(c, hipSetDevice(c->device));
    if (need_scratch(c, (size_t)bytes)) return 1;
    if (c->rank == root) COMM_HIP(c, hipMemcpyAsync(c->scratch, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    COMM_NCCL(c, rccl().Broadcast(c->scratch, c->scratch, (size_t)bytes, ncclUint8, root, c->comm, c->stream));
    if (c->rank != root) COMM_HIP(c, hipMemcpyAsync(host, c->scratch, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    COMM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int clair_comm_allgather(clair_comm_t *c, const void *send_host, void *recv_host, int64_t bytes_per_rank) {
    if (!c) return cfail(nullptr, "communicator is NULL");
    if (!send_host || !recv_host || bytes_per_rank < 1) return cfail(c, "bad arguments to clair_comm_allgather");
    COMM_HIP(c, hipSetDevice(c->device));
    const size_t b = (size_t)bytes_per_rank;
    if (need_scratch(c, b * (c->world + 1))) return 1;
    char *recv = (char *)c->scratch, *send = recv + b * c->world;
    COMM_HIP(c, hipMemcpyAsync(send, send_host, b, hipMemcpyHostToDevice, c->stream));
    COMM_NCCL(c, rccl().AllGather(send, recv, b, ncclUint8, c->comm, c->stream));
    COMM_HIP(c, hipMemcpyAsync(recv_host, recv, b * c->world, hipMemcpyDeviceToHost, c->stream));
    COMM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int clair_comm_allgather_device(clair_comm_t *c, const void *send_dev, void *recv_dev, int64_t bytes_per_rank) {
    if (!c) return cfail(nullptr, "communicator is NULL");
    if (!send_dev || !recv_dev || bytes_per_rank < 1) return cfail(c, "bad arguments to clair_comm_allgather_device");
    COMM_HIP(c, hipSetDevice(c->device));
    COMM_NCCL(c, rccl().AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclUint8, c->comm, c->stream));
    COMM_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
