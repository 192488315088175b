// Standalone harness around the one-pass wide sweep (stream_gemm_x3w2_kernel, 128 < K <= 224): the library's kernel and its panel packing on
// cfg3's matrix (20 096 x 200 064), count-like (one-plane form) and full-significand data, both sweep orientations, teams of 8 as the library
// launches it.  Timing-only ablations are compiled in with -DX3W2_ABLATE=<bits> (1: no wait for the panel DMA, 2: no per-stage barrier,
// 4: no panel DMA after the first stage -- wrong results, valid memory): one binary per ablation, run back to back in one gpurun call.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DX3W2_ABLATE=1] -o tools/x3w2_bench tools/x3w2_bench.hip && tools/x3w2_bench
#include "../alpine_amd/csrc/kernels_x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace alpine;

__global__ void fill_kernel(float* __restrict__ x, size_t n, int mode, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761ull) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        float v;
        if (mode == 0) { const unsigned r = h & 255u; v = r < 150u ? 0.f : (float)((h >> 8) & 7u); }
        else if (mode == 1) v = (float)(h >> 8) * (1.0f / 16777216.0f) * 3.7f;
        else v = (float)(h >> 8) * (1.0f / 16777216.0f);
        x[i] = v;
    }
}

template <int M16A>
static void shape(int64_t G, int64_t N, float* X, float* P, u32x4* Pk, float* pieces, size_t piece_floats, int reps, int gw)
{
    constexpr int KPA = 16 * M16A;
    printf("#### one-pass wide sweep, %d components staged (%d tiles), teams of %d, ablation bits %d; X %lld x %lld\n", KPA, M16A, gw,
#ifdef X3W2_ABLATE
           X3W2_ABLATE,
#else
           0,
#endif
           (long long)G, (long long)N);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int data = 0; data < 2; ++data) {
        hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, X, (size_t)G * N, data, 12345u);
        for (int orient = 0; orient < 2; ++orient) {
            const int64_t F = orient == 0 ? N : G, R = orient == 0 ? G : N;
            hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, P, (size_t)2 * R * 128, 2, 77u);
            const int64_t plane_stride = (R / 8) * KPA;
            hipLaunchKernelGGL(pack_panel3_wide_kernel, dim3(2048), dim3(256), 0, 0, P, P + R * 128, (int)R, KPA, Pk, plane_stride);
            for (int nw = 4; nw <= (data == 0 && M16A <= 10 ? 8 : 4); nw += 4) {
                const SweepGeom g = sg_make_geom(F, R, 256 / gw, 0, 64 * nw * gw, 0, gw);
                if ((size_t)2 * g.nwg * g.maxp * g.bf * 128 > piece_floats) { printf("pieces buffer too small\n"); return; }
                float* p0 = pieces; float* p1 = pieces + (size_t)g.nwg * g.maxp * g.bf * 128;
                std::vector<float> ms;
                for (int r = -3; r < reps; ++r) {
                    (void)hipEventRecord(e0);
                    if (data == 0 && nw == 8) {
                        if constexpr (M16A <= 10) hipLaunchKernelGGL((stream_gemm_x3w2_kernel<M16A, true, 8>), dim3(sg_grid(g)), dim3(512), 0, 0, X, Pk, plane_stride, p0, p1, F, g, (int*)nullptr);
                    }
                    else if (data == 0) hipLaunchKernelGGL((stream_gemm_x3w2_kernel<M16A, true>), dim3(sg_grid(g)), dim3(256), 0, 0, X, Pk, plane_stride, p0, p1, F, g, (int*)nullptr);
                    else hipLaunchKernelGGL((stream_gemm_x3w2_kernel<M16A, false>), dim3(sg_grid(g)), dim3(256), 0, 0, X, Pk, plane_stride, p0, p1, F, g, (int*)nullptr);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float t; (void)hipEventElapsedTime(&t, e0, e1);
                    if (r >= 0) ms.push_back(t);
                }
                const hipError_t e = hipGetLastError();
                if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); exit(1); }
                std::sort(ms.begin(), ms.end());
                const double med = ms[ms.size() / 2];
                printf("%-16s %-5s %d waves  grid %4d (spans %4d x %d pieces)  min %.3f med %.3f ms  %.2f TB/s  %.3f of 8 TB/s\n", data == 0 ? "count-like" : "full-significand",
                       orient == 0 ? "W^TX" : "XH^T", nw, sg_grid(g), g.nwg, g.maxp, ms[0], med, (double)F * R * 4.0 / (med * 1e-3) / 1e12, (double)F * R * 4.0 / (med * 1e-3) / 8e12);
            }
        }
    }
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 9;
    const int gw = argc > 2 ? atoi(argv[2]) : 8;
    const int64_t G = 20096, N = 200064;
    float *X, *P, *pieces; u32x4* Pk;
    (void)hipMalloc(&X, sizeof(float) * G * N);
    (void)hipMalloc(&P, sizeof(float) * 2 * N * 128);
    (void)hipMalloc(&Pk, sizeof(u32x4) * 3 * (N / 8) * 256);
    const size_t piece_floats = (size_t)2 * 600 * 3 * 2048 * 128 / 4;
    (void)hipMalloc(&pieces, sizeof(float) * piece_floats);
    shape<10>(G, N, X, P, Pk, pieces, piece_floats, reps, gw);
    return 0;
}
