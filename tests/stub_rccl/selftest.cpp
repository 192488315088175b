// Check of the RCCL stand-in itself (tests/stub_rccl/rccl_stub.cpp) without a GPU: R forked processes sum host buffers
// through it; the two HIP calls it looks up are provided here as plain memcpy / no-op.  argv[1] = ranks, argv[2] = rounds.
#include <rccl/rccl.h>

#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

int main(int argc, char** argv)
{
    const int R = argc > 1 ? std::atoi(argv[1]) : 3, rounds = argc > 2 ? std::atoi(argv[2]) : 100, n = 4096;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
    for (int r = 0; r < R; ++r) {
        if (fork() != 0) continue;
        ncclComm_t c;
        if (ncclCommInitRank(&c, R, id, r) != ncclSuccess) _exit(3);
        std::vector<float> v(n), out(n);
        for (int it = 0; it < rounds; ++it) {
            for (int i = 0; i < n; ++i) v[i] = (float)(r + 1) * (float)(i % 97 + it);
            // alternate in-place and out-of-place, whole buffer and a slice
            const int off = (it % 3 == 2) ? 100 : 0, cnt = (it % 3 == 2) ? 1000 : n;
            float* dst = (it & 1) ? out.data() : v.data();
            if (ncclAllReduce(v.data() + off, dst + off, cnt, ncclFloat, ncclSum, c, nullptr) != ncclSuccess) _exit(4);
            const float tri = (float)(R * (R + 1) / 2);
            for (int i = off; i < off + cnt; ++i)
                if (dst[i] != tri * (float)(i % 97 + it)) { std::fprintf(stderr, "rank %d round %d element %d: %g\n", r, it, i, dst[i]); _exit(5); }
        }
        if (ncclCommDestroy(c) != ncclSuccess) _exit(6);
        _exit(0);
    }
    int bad = 0;
    for (int r = 0; r < R; ++r) { int st = 0; wait(&st); bad |= st; }
    std::printf("ranks %d rounds %d: %s\n", R, rounds, bad ? "FAILED" : "ok");
    return bad != 0;
}
