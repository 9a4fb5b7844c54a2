// Back-to-back launch cost of a kernel that does nothing, in the geometry of the CU-wide rollout kernel (256 workgroups of
// 1024 threads, 36 KiB of LDS) and of the 256-thread one: the floor under the fixed cost of a rollout launch.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { char pad[256]; };
template <int LDS>
__global__ void k_empty(Big p, float* out) {
    __shared__ char sm[LDS];
    if (threadIdx.x == 0 && p.pad[0] == 77) { sm[0] = 1; out[blockIdx.x] = sm[0]; }
}
template <int LDS>
void run(const char* name, int grid, int block) {
    float* d; (void)hipMalloc(&d, 4096 * 4);
    Big p{}; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty<LDS>, dim3(grid), dim3(block), 0, 0, p, d);
    (void)hipDeviceSynchronize();
    const int n = 2000;
    (void)hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty<LDS>, dim3(grid), dim3(block), 0, 0, p, d);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %6.2f us per launch (back to back)\n", name, ms * 1e3 / n);
    (void)hipFree(d);
}
int main() {
    run<36096>("256 x 1024 threads, 36 KiB LDS", 256, 1024);
    run<9024>("1024 x 256 threads, 9 KiB LDS", 1024, 256);
    run<16>("1 x 64 threads", 1, 64);
    return 0;
}
