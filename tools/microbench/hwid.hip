// Where do the waves of a workgroup land?  Prints (XCC, SE, CU, SIMD, wave slot) of every wave of a few 1024-thread and
// 256-thread workgroups (HW_ID / XCC_ID registers, gfx950).   hipcc -O3 --offload-arch=gfx950 hwid.hip -o hwid && ./hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * waves + wave) * 2] = hw; out[(blockIdx.x * waves + wave) * 2 + 1] = xcc; }
    // stay resident for a while so that all workgroups of the grid coexist
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);
}
int main() {
    for (int block : {1024, 256}) {
        const int grid = block == 1024 ? 256 : 1024, waves = block / 64;
        unsigned* d; hipMalloc(&d, grid * waves * 8);
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d);
        std::vector<unsigned> h(grid * waves * 2);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        printf("block=%d grid=%d\n", block, grid);
        int simd_eq_wmod4 = 0, total = 0;
        std::vector<int> per_cu(8 * 64, 0);
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < waves; ++w) {
                const unsigned hw = h[(b * waves + w) * 2], xcc = h[(b * waves + w) * 2 + 1] & 15;
                const int slot = hw & 15, simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                if (b < 3 || b == grid - 1) printf("  wg %4d wave %2d: xcc %u se %d sh %d cu %2d simd %d slot %d\n", b, w, xcc, se, sh, cu, simd, slot);
                simd_eq_wmod4 += simd == (w & 3); ++total;
                per_cu[xcc * 64 + se * 16 + cu] += 1;
            }
        int used = 0, mx = 0;
        for (int v : per_cu) { used += v > 0; mx = v > mx ? v : mx; }
        printf("  simd == wave %% 4 for %d of %d waves; distinct (xcc,se,cu) used %d, max waves on one %d\n", simd_eq_wmod4, total, used, mx);
        hipFree(d);
    }
    return 0;
}
