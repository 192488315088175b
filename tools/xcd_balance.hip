// Diagnostic: do the XCDs of an MI355X stream from HBM at the same rate, and does giving the slower ones less work pay?
// 256 workgroups (one per CU, forced by the LDS footprint) each sum a contiguous span of a 16 GB buffer with 16-byte
// non-temporal loads.  Unit u = (blockIdx / 8) * 8 + XCC_ID (blockIdx % 8 only LABELS an XCD; the id comes from the hardware
// register); the span of unit u is (1 + d) or (1 - d) times the mean for even / odd XCC_ID.  Prints the kernel time and the
// per-XCD workgroup lifetimes (s_memrealtime, 100 MHz) for a range of d.  Not part of the library.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ x, const long long* __restrict__ bounds, unsigned long long* stamp, float* out)
{
    extern __shared__ float pad[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    const int u = (int)(blockIdx.x >> 3) * 8 + (int)xcc;
    const long long b0 = bounds[u], b1 = bounds[u + 1];
    const f32x4* p = x + b0;
    const long long per = b1 - b0;
    f32x4 acc = {0, 0, 0, 0};
    for (long long i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        f32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = __builtin_nontemporal_load(p + i + k * 256);
#pragma unroll
        for (int k = 0; k < U; ++k) acc += v[k];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) { out[blockIdx.x] = acc[0]; pad[threadIdx.x] = acc[1]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stamp[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
        stamp[3 * blockIdx.x + 1] = xcc;
        stamp[3 * blockIdx.x + 2] = (unsigned long long)u;
    }
}

// the sweeps' access pattern: the buffer is a matrix of `ld` floats per row; unit u covers `rows` consecutive rows of a 1024-column
// tile; a wave instruction loads 2 rows (8 apart) x 512 B, 16 loads per lane in flight x 2 stages (kernels_x3.hpp)
__global__ __launch_bounds__(256) void rd_tiles(const float* __restrict__ x, long long ld, long long R, long long L, long long total,
                                                const long long* __restrict__ bounds, unsigned long long* stamp, float* out)
{
    extern __shared__ float pad[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    const int u = (int)(blockIdx.x >> 3) * 8 + (int)xcc;
    long long pos = bounds[u];
    const long long pos_end = bounds[u + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    f32x4 acc = {0, 0, 0, 0};
    while (pos < pos_end) {
        const long long ft = pos / R, r0 = pos - ft * R;
        const long long r1 = r0 + (pos_end - pos) < R ? r0 + (pos_end - pos) : R;
        pos += r1 - r0;
        const float* base = x + (r0 + 8 * h) * ld + ft * 1024 + wave * 256 + 4 * c;
        for (long long r = 0; r + 32 <= r1 - r0; r += 32) {
            f32x4 v[2][2][8];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        v[p][hf][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (r + 16 * p + e) * ld + 128 * hf));
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc += v[p][hf][e];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) { out[blockIdx.x] = acc[0]; pad[threadIdx.x] = acc[1]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        stamp[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
        stamp[3 * blockIdx.x + 1] = xcc;
        stamp[3 * blockIdx.x + 2] = (unsigned long long)u;
    }
}

static void report(const char* tag, double d, float best, double bytes, const std::vector<unsigned long long>& st, int grid)
{
    std::vector<double> ev, od; std::vector<int> seen(grid, 0); int dup = 0;
    for (int w = 0; w < grid; ++w) {
        ((st[3 * w + 1] & 1) ? od : ev).push_back(st[3 * w] / 100.0);
        if (seen[st[3 * w + 2]]++) ++dup;
    }
    std::sort(ev.begin(), ev.end()); std::sort(od.begin(), od.end());
    printf("%s d=%.3f: %.3f ms  %.2f TB/s | even XCDs: lifetime median %.1f us (max %.1f) | odd: median %.1f us (max %.1f) | duplicate units %d\n",
           tag, d, best, bytes / (best * 1e-3) / 1e12, ev[ev.size() / 2], ev.back(), od[od.size() / 2], od.back(), dup);
}

static void tiles(const float* x, long long ld, long long R, int nft, long long* bounds, unsigned long long* stamp, float* out, const char* tag)
{
    const int grid = 256;
    const long long total = (long long)nft * R;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)rd_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (double d : {0.0, 0.02, 0.03, 0.04, 0.0}) {
        std::vector<long long> b(grid + 1, 0);
        double acc = 0;
        for (int u = 0; u < grid; ++u) {
            acc += ((u & 1) ? 1.0 - d : 1.0 + d) / grid;
            b[u + 1] = std::min<long long>(total, (long long)(acc * (double)total) / 64 * 64);
        }
        b[grid] = total;
        hipMemcpy(bounds, b.data(), sizeof(long long) * (grid + 1), hipMemcpyHostToDevice);
        float best = 1e9;
        std::vector<unsigned long long> st(3 * grid);
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(rd_tiles, dim3(grid), dim3(256), 100 * 1024, 0, x, ld, R, 0LL, total, bounds, stamp, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; (void)hipMemcpy(st.data(), stamp, sizeof(unsigned long long) * 3 * grid, hipMemcpyDeviceToHost); }
        }
        report(tag, d, best, (double)total * 4096.0, st, grid);
    }
}

int main()
{
    const size_t bytes = (size_t)16 << 30;
    const int grid = 256;
    f32x4* x; float* out; long long* bounds; unsigned long long* stamp;
    hipMalloc(&x, bytes); hipMalloc(&out, 1 << 20); hipMalloc(&bounds, sizeof(long long) * (grid + 1)); hipMalloc(&stamp, sizeof(unsigned long long) * 3 * grid);
    hipMemset(x, 0, bytes);
    const long long n4 = (long long)(bytes / 16);
    hipFuncSetAttribute((const void*)rd<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (double d : {0.0, 0.01, 0.02, 0.03, 0.04, 0.05, 0.0}) {
        std::vector<long long> b(grid + 1, 0);
        double acc = 0;
        for (int u = 0; u < grid; ++u) {
            acc += ((u & 1) ? 1.0 - d : 1.0 + d) / grid;
            b[u + 1] = std::min<long long>(n4, (long long)(acc * (double)n4) / 4096 * 4096);
        }
        b[grid] = n4;
        hipMemcpy(bounds, b.data(), sizeof(long long) * (grid + 1), hipMemcpyHostToDevice);
        float best = 1e9;
        std::vector<unsigned long long> st(3 * grid);
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((rd<16>), dim3(grid), dim3(256), 100 * 1024, 0, x, bounds, stamp, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) { best = ms; hipMemcpy(st.data(), stamp, sizeof(unsigned long long) * 3 * grid, hipMemcpyDeviceToHost); }
        }
        std::vector<double> ev, od; std::vector<int> seen(grid, 0); int dup = 0;
        for (int w = 0; w < grid; ++w) {
            ((st[3 * w + 1] & 1) ? od : ev).push_back(st[3 * w] / 100.0);
            if (seen[st[3 * w + 2]]++) ++dup;
        }
        std::sort(ev.begin(), ev.end()); std::sort(od.begin(), od.end());
        printf("d=%.2f: %.3f ms  %.2f TB/s | even XCDs: %zu wgs, lifetime median %.1f us (max %.1f) | odd: %zu wgs, median %.1f us (max %.1f) | duplicate units %d\n",
               d, best, (double)bytes / (best * 1e-3) / 1e12, ev.size(), ev.empty() ? 0 : ev[ev.size() / 2], ev.empty() ? 0 : ev.back(),
               od.size(), od.empty() ? 0 : od[od.size() / 2], od.empty() ? 0 : od.back(), dup);
    }
    // sweep A of cfg3: rows = 200 064 cells of 20 480 genes (20 tiles); sweep B: rows = 20 096 genes of 200 704 cells (196 tiles)
    tiles((const float*)x, 20480, 200064, 20, bounds, stamp, out, "tiles A (ld 80 KB)  ");
    tiles((const float*)x, 200704, 20096, 196, bounds, stamp, out, "tiles B (ld 784 KB) ");
    return 0;
}
