// Instruction-issue microbenchmark for gfx950 (what shaped the pair loops and the "who runs when" devices of DESIGN.md 4).
//
// What one wave -- and 2, 4, 8 waves sharing a SIMD -- sustains, in SHADER CYCLES PER WAVE-INSTRUCTION measured inside the
// kernel: every wave brackets its loop with s_memtime (the shader-clock counter; s_memrealtime, the constant 100 MHz
// counter, next to it gives the clock the loop ran at) and reports its HW_ID, so the host checks the placement it assumes
// -- exactly W waves on every SIMD that has any -- instead of trusting the dispatcher (the round-2
// version assumed a flat 2.4 GHz clock and one wave per SIMD for 256 blocks, and its "16 waves per SIMD" row was 8 in two
// rounds).  Printed per instruction kind and W: cycles per instruction of ONE wave (median over waves), and the SIMD's
// aggregate = that / W (cycles of SIMD time per wave-instruction when W waves interleave).
//
// hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITERS = 1024;
constexpr int REP = 8;          // every asm block below is issued REP times per loop iteration: the loop's own 3 scalar instructions + taken branch drown
typedef float f2 __attribute__((ext_vector_type(2)));

struct Rec { unsigned long long cyc, real; unsigned hw, xcc; };

template <int KIND>
__global__ __launch_bounds__(1024) void k(Rec* out, float* sink, int n_iter, float seed) {
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = 1.0001f, c = 1e-4f;
    __shared__ float4 tile[64];
    if (threadIdx.x < 64) tile[threadIdx.x] = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
      for (int rep = 0; rep < REP; ++rep) {
        if constexpr (KIND == 0) {          // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 1) {   // 8 DEPENDENT v_fma_f32 (one chain)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(b), "v"(c));
        } else if constexpr (KIND == 2) {   // 8 independent v_pk_fma_f32 (two f32 lanes each)
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {b, b}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if constexpr (KIND == 3) {   // 8 v_rsq_f32 (transcendental pipe)
            asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 4) {   // 8 independent DPP adds (row_shr:1)
            asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 5) {   // a DEPENDENT DPP chain (the shape of a wave reduction): 8 x (s_nop 1 + dpp add)
            asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         : "+v"(a0));
        } else if constexpr (KIND == 6) {   // 4 x (s_nop 1 + v_permlane32_swap) on independent register pairs
            asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1\n s_nop 1\n v_permlane32_swap_b32 %2, %3\n s_nop 1\n v_permlane32_swap_b32 %4, %5\n s_nop 1\n v_permlane32_swap_b32 %6, %7\n"
                         "s_nop 1\n v_permlane32_swap_b32 %0, %1\n s_nop 1\n v_permlane32_swap_b32 %2, %3\n s_nop 1\n v_permlane32_swap_b32 %4, %5\n s_nop 1\n v_permlane32_swap_b32 %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 7) {   // 8 ds_read_b128 broadcasts (uniform address), each result consumed by 1 add
            float4 t0_ = tile[(it * 8 + rep + 0) & 63], t1_ = tile[(it * 8 + 1) & 63], t2_ = tile[(it * 8 + 2) & 63], t3_ = tile[(it * 8 + 3) & 63];
            float4 t4_ = tile[(it * 8 + 4) & 63], t5_ = tile[(it * 8 + 5) & 63], t6_ = tile[(it * 8 + 6) & 63], t7_ = tile[(it * 8 + 7) & 63];
            a0 += t0_.x; a1 += t1_.y; a2 += t2_.z; a3 += t3_.w; a4 += t4_.x; a5 += t5_.y; a6 += t6_.z; a7 += t7_.w;
        } else if constexpr (KIND == 8) {   // 8 v_readlane_b32 -> SGPR, each consumed by a v_add with that SGPR
            const int l = (it + rep) & 63;
            float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a0), l));
            float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a1), l));
            float s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a2), l));
            float s3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a3), l));
            a0 += s1; a1 += s2; a2 += s3; a3 += s0;
        } else if constexpr (KIND == 9) {   // 8 independent s_add_u32 (the scalar ALU)
            unsigned s0 = it, s1 = it + 1, s2 = it + 2, s3 = it + 3;
            asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)::"scc");
            a0 += (float)(s0 ^ s1 ^ s2 ^ s3) * 1e-30f;
        } else if constexpr (KIND == 10) {  // 4 x (v_fma_f32 ; s_add_u32): can a wave's scalar instruction ride along with its vector one?
            unsigned s0 = it;
            asm volatile("v_fma_f32 %0, %0, %5, %6\n s_add_u32 %4, %4, 3\n v_fma_f32 %1, %1, %5, %6\n s_add_u32 %4, %4, 3\n v_fma_f32 %2, %2, %5, %6\n s_add_u32 %4, %4, 3\n v_fma_f32 %3, %3, %5, %6\n s_add_u32 %4, %4, 3\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0) : "v"(b), "v"(c) : "scc");
            a4 += (float)s0 * 1e-30f;
        }
      }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Rec{t1 - t0, r1 - r0, hw, xcc & 15u};
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
int run(const char* name, int insts_per_iter, int W, Rec* d_rec, float* d_sink) {
    // W waves per SIMD: 256 workgroups (one per CU) of 256 W threads for W <= 4; two 1024-thread workgroups per CU for W = 8
    const int block = W <= 4 ? 256 * W : 1024, grid = W <= 4 ? 256 : 512, waves = grid * (block / 64);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(block), 0, 0, d_rec, d_sink, 64, 1.0f);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(block), 0, 0, d_rec, d_sink, ITERS, 1.0f);
    CHECK(hipDeviceSynchronize());
    std::vector<Rec> h(waves);
    CHECK(hipMemcpy(h.data(), d_rec, sizeof(Rec) * waves, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per_simd;
    std::vector<double> cpi, mhz;
    for (const Rec& r : h) {
        const unsigned simd = (r.hw >> 4) & 3, cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7;
        per_simd[(((r.xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd] += 1;
        cpi.push_back((double)r.cyc / ((double)ITERS * REP * insts_per_iter));
        mhz.push_back((double)r.cyc / (double)r.real * 100.0);
    }
    int lo = 1 << 30, hi = 0;
    for (auto& kv : per_simd) { lo = std::min(lo, kv.second); hi = std::max(hi, kv.second); }
    std::sort(cpi.begin(), cpi.end()); std::sort(mhz.begin(), mhz.end());
    const double med = cpi[cpi.size() / 2];
    printf("%-44s W=%d  %6.2f cycles per instruction of one wave (min %5.2f, max %5.2f)  = %5.2f SIMD-cycles per wave-instruction; clock %4.0f MHz; "
           "SIMDs used %4zu, waves per SIMD %d..%d%s\n", name, W, med, cpi.front(), cpi.back(), med / W, mhz[mhz.size() / 2], per_simd.size(), lo, hi,
           (lo == W && hi == W) ? "" : "  ** PLACEMENT IS NOT WHAT THE ROW ASSUMES **");
    return 0;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    Rec* d_rec; float* d_sink;
    CHECK(hipMalloc(&d_rec, sizeof(Rec) * 512 * 16));
    CHECK(hipMalloc(&d_sink, sizeof(float) * 512 * 1024));
    for (int W : {1, 2, 4, 8}) {
        run<0>("v_fma_f32, 8 independent", 8, W, d_rec, d_sink);
        run<1>("v_fma_f32, dependent chain", 8, W, d_rec, d_sink);
        run<2>("v_pk_fma_f32, independent", 8, W, d_rec, d_sink);
        run<3>("v_rsq_f32, independent", 8, W, d_rec, d_sink);
        run<4>("v_add_f32_dpp row_shr, independent", 8, W, d_rec, d_sink);
        run<5>("s_nop 1 + v_add_f32_dpp, dependent chain (per pair)", 8, W, d_rec, d_sink);
        run<6>("s_nop 1 + v_permlane32_swap_b32 (per pair)", 8, W, d_rec, d_sink);
        run<7>("ds_read_b128 broadcast + v_add (per pair)", 8, W, d_rec, d_sink);
        run<8>("v_readlane_b32 + v_add with the SGPR (per pair)", 4, W, d_rec, d_sink);
        run<9>("s_add_u32, independent", 8, W, d_rec, d_sink);
        run<10>("v_fma_f32 ; s_add_u32 alternating (per pair)", 4, W, d_rec, d_sink);
    }
    CHECK(hipFree(d_rec)); CHECK(hipFree(d_sink));
    return 0;
}
