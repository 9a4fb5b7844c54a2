// What does a device-side gate in front of every launch of a two-queue chain cost?  Launch g goes to stream g & 1; every workgroup
// adds 1 to a counter when it starts; before launch g + 1 its stream waits (hipStreamWaitValue64, no CU held) until the counter says
// that every workgroup of launch g has started.  Spin kernels of the rollout's geometry (256 workgroups x 1024 threads, ~30 us).
// hipcc -O3 --offload-arch=gfx950 waitvalue.hip -o waitvalue && ./waitvalue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(1024) void k_work(unsigned long long* started, unsigned long long ticks) {
    if (threadIdx.x == 0 && started) __hip_atomic_fetch_add(started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    unsigned n = 0;
    do { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); } while (t - t0 < ticks + (blockIdx.x & 15) * 60 && ++n < (1u << 22));
}
int main() {
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    unsigned long long* sig = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 64, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { CK(hipMalloc((void**)&sig, 64)); }
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    const int G = 256, L = 400;
    for (int mode = 0; mode < 3; ++mode) {       // 0: no gate, 1: wait-value gate, 2: gate + counter increments only
        CK(hipMemset(sig, 0, 64)); CK(hipDeviceSynchronize());
        for (int rep = 0; rep < 3; ++rep) {
            unsigned long long base = 0;
            CK(hipMemset(sig, 0, 64)); CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int g = 0; g < L; ++g) {
                if (mode == 1 && g > 0) {
                    hipError_t w = hipStreamWaitValue64(s[g & 1], sig, base, hipStreamWaitValueGte, ~0ull);
                    if (w != hipSuccess) { printf("hipStreamWaitValue64: %s\n", hipGetErrorString(w)); return 1; }
                }
                hipLaunchKernelGGL(k_work, dim3(G), dim3(1024), 0, s[g & 1], mode ? sig : nullptr, 2400ull);
                base += G;
            }
            CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("mode %d (%s) rep %d: %.2f us per launch\n", mode, mode == 0 ? "two queues, no gate" : (mode == 1 ? "wait-value gate before every launch" : "counter only"), rep, us / L);
        }
    }
    return 0;
}
