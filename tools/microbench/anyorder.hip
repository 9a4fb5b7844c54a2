// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950 (hip_ext.h says "not supported on GFX9xx")?
// k(a) spins 200 us, k(b) 20 us, launched back to back on one stream: with the barrier bit b starts after a ends.
// Also: the same with grids that fill the chip (256 workgroups of 1024 threads), to see the dispatch order.
// hipcc -O3 --offload-arch=gfx950 anyorder.hip -o anyorder && ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_spin(unsigned long long* out, int slot, unsigned long long ticks) {   // 100 MHz ticks
    if (slot == 4 && (blockIdx.x & 1)) ticks >>= 1;      // (slot 4: every other workgroup takes half the time -- is the freed half of the chip backfilled?)
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    do { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); __builtin_amdgcn_s_sleep(8); } while (t1 - t0 < ticks);
    if (threadIdx.x == 0) {
        atomicMin(&out[slot * 2], t0);
        atomicMax(&out[slot * 2 + 1], t1);
    }
}

static int run(int grid, int block, unsigned flags, const char* what) {
    unsigned long long* d;
    CHECK(hipMalloc(&d, 64));
    std::vector<unsigned long long> h = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
    CHECK(hipMemcpy(d, h.data(), 64, hipMemcpyHostToDevice));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipExtLaunchKernelGGL(k_spin, dim3(grid), dim3(block), 0, s, nullptr, nullptr, 0, d, 3, 1000ull);     // warm-up
    CHECK(hipStreamSynchronize(s));
    hipExtLaunchKernelGGL(k_spin, dim3(grid), dim3(block), 0, s, nullptr, nullptr, flags, d, 0, 20000ull);   // 200 us
    hipExtLaunchKernelGGL(k_spin, dim3(grid), dim3(block), 0, s, nullptr, nullptr, flags, d, 1, 2000ull);    // 20 us
    hipExtLaunchKernelGGL(k_spin, dim3(grid), dim3(block), 0, s, nullptr, nullptr, flags, d, 2, 2000ull);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(h.data(), d, 64, hipMemcpyDeviceToHost));
    const double a0 = 0, a1 = (h[1] - h[0]) * 0.01, b0 = ((double)h[2] - (double)h[0]) * 0.01, b1 = ((double)h[3] - (double)h[0]) * 0.01, c0 = ((double)h[4] - (double)h[0]) * 0.01;
    printf("%-44s grid %3d x %4d: a [%.1f, %.1f] us   b starts %.1f ends %.1f   c starts %.1f   -> %s\n", what, grid, block, a0, a1, b0, b1, c0,
           b0 < a1 ? "b OVERLAPS a" : "b after a");
    hipFree(d);
    hipStreamDestroy(s);
    return 0;
}

static int train(int grid, int block, unsigned flags, const char* what) {
    unsigned long long* d;
    CHECK(hipMalloc(&d, 64));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < 100; ++i) hipExtLaunchKernelGGL(k_spin, dim3(grid), dim3(block), 0, s, nullptr, nullptr, flags, d, 3, 4000ull);   // 40 us per workgroup
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s grid %3d x %4d: 100 launches of 40 us workgroups: %.2f us per launch\n", what, grid, block, ms * 10.0);
    hipFree(d); hipStreamDestroy(s);
    return 0;
}

static int uneven(unsigned flags, const char* what) {
    unsigned long long* d;
    CHECK(hipMalloc(&d, 64));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < 100; ++i) hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(1024), 0, s, nullptr, nullptr, flags, d, 4, 4000ull);   // 40 / 20 us workgroups
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s 256 x 1024, workgroups of 40 and 20 us alternating: %.2f us per launch (backfilled: ~30 + boundary; not: ~40 + boundary)\n", what, ms * 10.0);
    hipFree(d); hipStreamDestroy(s);
    return 0;
}

int main() {
    if (uneven(0, "default launches")) return 1;
    if (uneven(hipExtAnyOrderLaunch, "hipExtAnyOrderLaunch")) return 1;
    if (train(256, 1024, 0, "default launches")) return 1;
    if (train(256, 1024, hipExtAnyOrderLaunch, "hipExtAnyOrderLaunch")) return 1;
    if (train(256, 1024, 0, "default launches")) return 1;
    if (train(256, 1024, hipExtAnyOrderLaunch, "hipExtAnyOrderLaunch")) return 1;
    if (run(1, 64, 0, "default launches")) return 1;
    if (run(1, 64, hipExtAnyOrderLaunch, "hipExtAnyOrderLaunch")) return 1;
    if (run(128, 1024, 0, "default launches, half the chip each")) return 1;
    if (run(256, 1024, 0, "default launches, the whole chip each")) return 1;
    if (run(128, 1024, hipExtAnyOrderLaunch, "hipExtAnyOrderLaunch, half the chip each")) return 1;
    if (run(256, 1024, hipExtAnyOrderLaunch, "hipExtAnyOrderLaunch, the whole chip each")) return 1;
    return 0;
}
