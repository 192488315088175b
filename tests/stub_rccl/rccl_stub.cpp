// TEST INFRASTRUCTURE, never shipped and never linked into libalpine_hip.so: a shared-memory stand-in for the RCCL entry
// points the library calls (ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll, ncclCommCount, ncclCommUserRank, ncclAllReduce,
// ncclCommDestroy, ncclGetErrorString),
// preloaded (LD_PRELOAD) into the rank processes of tests/test_gpu_comm_stub.py.
//
// Why: RCCL refuses two ranks on one device and the GPU boxes of the test pool have ONE GPU, so the library's native
// multi-rank loop (alpine_run / alpine_iter / alpine_batch_step / alpine_epoch_loss with a communicator attached) could
// otherwise only ever run at world size 1, where every all-reduce is the identity.  With this stand-in two processes
// share cuda:0 and the C loop's sequencing (which slot is reduced when, ranks with an empty batch, the per-group exchange
// of the block-coordinate branch) is exercised with real two-rank sums.  It says nothing about RCCL or xGMI themselves.
//
// Semantics kept from RCCL: in-place or out-of-place float sum all-reduce, ordered with the work already enqueued on the
// given stream (here: by synchronising it), identical result on every rank (slots are added in rank order).
//
// The HIP runtime is looked up at first use (dlsym) so that preloading this file does not pull a second libamdhip64 into
// a process whose torch brings its own.
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

constexpr size_t SLOT_FLOATS = (size_t)8 << 20;      // 32 MiB per rank (the largest reduce block of the tests is ~5 MB)
constexpr int WAIT_SECONDS = 120;

struct Shared {
    std::atomic<int> attached;
    std::atomic<int> arrived;
    std::atomic<int> generation;
    char pad[128 - 3 * sizeof(std::atomic<int>)];
    float data[1];
};

using memcpy_fn = int (*)(void*, const void*, size_t, int);
using sync_fn = int (*)(void*);
memcpy_fn hip_memcpy = nullptr;
sync_fn hip_stream_sync = nullptr;

bool bind_hip()
{
    // several ranks may live in ONE process (ncclCommInitAll: one host thread per rank): look the runtime up once, into locals, and
    // publish the pair only when both are found -- a thread must never see a half-bound or a transiently NULL pointer
    static std::once_flag once;
    std::call_once(once, [] {
        memcpy_fn m = nullptr; sync_fn s = nullptr;
        // global scope first (a plain C host links the runtime directly); a Python process has it in a local scope (pulled in
        // by an extension module), so ask for the already-loaded object by its soname next; load it only as the last resort
        void* scopes[3] = {RTLD_DEFAULT, nullptr, nullptr};
        for (int i = 0; i < 3 && !(m && s); ++i) {
            if (i == 1) scopes[i] = dlopen("libamdhip64.so.7", RTLD_NOLOAD | RTLD_LAZY);
            if (i == 2) scopes[i] = dlopen("libamdhip64.so.7", RTLD_LAZY);
            if (i > 0 && !scopes[i]) continue;
            m = (memcpy_fn)dlsym(scopes[i], "hipMemcpy");
            s = (sync_fn)dlsym(scopes[i], "hipStreamSynchronize");
        }
        if (m && s) { hip_memcpy = m; hip_stream_sync = s; }
    });
    return hip_memcpy && hip_stream_sync;
}

}   // namespace

struct ncclComm {
    Shared* sh = nullptr;
    size_t bytes = 0;
    int nranks = 1, rank = 0;
    char name[64] = {0};
    std::vector<float> sum;

    float* slot(int r) const { return sh->data + (size_t)r * SLOT_FLOATS; }

    bool barrier()
    {
        const int gen = sh->generation.load();
        if (sh->arrived.fetch_add(1) + 1 == nranks) {
            sh->arrived.store(0);
            sh->generation.fetch_add(1);
            return true;
        }
        const auto t0 = std::chrono::steady_clock::now();
        while (sh->generation.load() == gen) {
            sched_yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(WAIT_SECONDS)) return false;
        }
        return true;
    }
};

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    static std::atomic<int> counter{0};
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "/alpine_stub_%d_%d_%lld", (int)getpid(), counter.fetch_add(1),
                  (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm;
    c->nranks = nranks; c->rank = rank;
    std::memcpy(c->name, id.internal, sizeof c->name - 1);
    c->bytes = sizeof(Shared) + sizeof(float) * SLOT_FLOATS * (size_t)nranks;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (Shared*)p;                                   // a fresh segment is zero-filled: the counters start at 0
    c->sh->attached.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->sh->attached.load() < nranks) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(WAIT_SECONDS)) { munmap(p, c->bytes); delete c; return ncclSystemError; }
    }
    *out = c;
    return ncclSuccess;
}

// one process, ndev ranks (the library's single-process drop-in: one host thread per rank afterwards): the same segment, every
// rank attached at once
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* /*devlist*/)
{
    if (!comms || ndev < 1) return ncclInvalidArgument;
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    const size_t bytes = sizeof(Shared) + sizeof(float) * SLOT_FLOATS * (size_t)ndev;
    for (int r = 0; r < ndev; ++r) {
        ncclComm* c = new ncclComm;
        c->nranks = ndev; c->rank = r; c->bytes = bytes;
        std::memcpy(c->name, id.internal, sizeof c->name - 1);
        const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { delete c; return ncclSystemError; }
        c->sh = (Shared*)p;
        c->sh->attached.fetch_add(1);
        comms[r] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count)
{
    if (!c || !count) return ncclInvalidArgument;
    *count = c->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t c, int* rank)
{
    if (!c || !rank) return ncclInvalidArgument;
    *rank = c->rank;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    const bool last = c->sh->attached.fetch_sub(1) == 1;
    munmap(c->sh, c->bytes);
    if (last) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t c, hipStream_t stream)
{
    if (!c || type != ncclFloat || op != ncclSum || count > SLOT_FLOATS) return ncclInvalidArgument;
    if (!bind_hip()) return ncclUnhandledCudaError;
    if (hip_stream_sync((void*)stream) != 0) return ncclUnhandledCudaError;
    if (hip_memcpy(c->slot(c->rank), send, sizeof(float) * count, 2 /* hipMemcpyDeviceToHost */) != 0) return ncclUnhandledCudaError;
    if (!c->barrier()) return ncclSystemError;
    c->sum.assign(count, 0.0f);
    for (int r = 0; r < c->nranks; ++r) {
        const float* s = c->slot(r);
        for (size_t i = 0; i < count; ++i) c->sum[i] += s[i];
    }
    if (!c->barrier()) return ncclSystemError;          // nobody overwrites a slot before everybody has read it
    if (hip_memcpy(recv, c->sum.data(), sizeof(float) * count, 1 /* hipMemcpyHostToDevice */) != 0) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclInvalidArgument: return "stub: invalid argument (float sum of at most 8 Mi elements only)";
        case ncclSystemError: return "stub: shared-memory segment or a peer did not arrive";
        case ncclUnhandledCudaError: return "stub: HIP runtime call failed or was not found";
        default: return "stub: error";
    }
}

}   // extern "C"
