// Instruction-rate microbenchmark for gfx950: decides the shape of the all-pairs loop.
// hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITERS = 32768;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int n_iter, float seed) {
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = 1.0001f, c = 1e-4f;
    __shared__ float4 tile[64];
    if (threadIdx.x < 64) tile[threadIdx.x] = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
    __syncthreads();
    for (int it = 0; it < n_iter; ++it) {
        if constexpr (KIND == 0) {   // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 1) {   // 4 independent v_pk_fma_f32 (8 flops-pairs)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {b, b}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if constexpr (KIND == 2) {   // 8 v_mul_lo_u32
            unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3);
            unsigned u4 = __float_as_uint(a4), u5 = __float_as_uint(a5), u6 = __float_as_uint(a6), u7 = __float_as_uint(a7);
            asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(0x9E3779B9u));
            a0 = __uint_as_float(u0); a1 = __uint_as_float(u1); a2 = __uint_as_float(u2); a3 = __uint_as_float(u3);
            a4 = __uint_as_float(u4); a5 = __uint_as_float(u5); a6 = __uint_as_float(u6); a7 = __uint_as_float(u7);
        } else if constexpr (KIND == 3) {   // 8 x (v_cmp + v_cndmask)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                         "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n"
                         "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
        } else if constexpr (KIND == 4) {   // 8 ds_read_b128 broadcast (uniform address), results consumed by 1 add each
            float4 t0 = tile[(it * 8 + 0) & 63], t1 = tile[(it * 8 + 1) & 63], t2 = tile[(it * 8 + 2) & 63], t3 = tile[(it * 8 + 3) & 63];
            float4 t4 = tile[(it * 8 + 4) & 63], t5 = tile[(it * 8 + 5) & 63], t6 = tile[(it * 8 + 6) & 63], t7 = tile[(it * 8 + 7) & 63];
            a0 += t0.x + t0.w; a1 += t1.y + t1.z; a2 += t2.z + t2.x; a3 += t3.w + t3.y; a4 += t4.x + t4.z; a5 += t5.y + t5.w; a6 += t6.z + t6.y; a7 += t7.w + t7.x;
        } else if constexpr (KIND == 5) {   // 8 v_readlane_b32 -> sgpr, each consumed by a v_add with sgpr operand
            const int l = it & 63;
            float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a0), l));
            float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a1), l));
            float s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a2), l));
            float s3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a3), l));
            float s4 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a4), l));
            float s5 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a5), l));
            float s6 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a6), l));
            float s7 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a7), l));
            a0 += s1; a1 += s2; a2 += s3; a3 += s4; a4 += s5; a5 += s6; a6 += s7; a7 += s0;
        } else if constexpr (KIND == 6) {   // 8 v_fma_f32 with clamp modifier
            asm volatile("v_fma_f32 %0, %0, %8, %9 clamp\n v_fma_f32 %1, %1, %8, %9 clamp\n v_fma_f32 %2, %2, %8, %9 clamp\n v_fma_f32 %3, %3, %8, %9 clamp\n"
                         "v_fma_f32 %4, %4, %8, %9 clamp\n v_fma_f32 %5, %5, %8, %9 clamp\n v_fma_f32 %6, %6, %8, %9 clamp\n v_fma_f32 %7, %7, %8, %9 clamp\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (KIND == 7) {   // 8 v_rsq_f32
            asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == 8) {   // 8 DPP row_shr adds (wave reductions)
            a0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x111, 0xf, 0xf, false));
            a1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0x112, 0xf, 0xf, false));
            a2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0x114, 0xf, 0xf, false));
            a3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a3), 0x118, 0xf, 0xf, false));
            a4 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a4), 0x142, 0xf, 0xf, false));
            a5 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a5), 0x143, 0xf, 0xf, false));
            a6 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a6), 0x111, 0xf, 0xf, false));
            a7 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a7), 0x112, 0xf, 0xf, false));
        } else if constexpr (KIND == 10) {  // 8 DEPENDENT v_fma_f32 (one chain)
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(b), "v"(c));
        } else if constexpr (KIND == 11) {  // 2 chains x 4
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
        } else if constexpr (KIND == 9) {   // 8 ds_bpermute (shfl_xor)
            a0 += __shfl_xor(a0, 32); a1 += __shfl_xor(a1, 16); a2 += __shfl_xor(a2, 32); a3 += __shfl_xor(a3, 16);
            a4 += __shfl_xor(a4, 32); a5 += __shfl_xor(a5, 16); a6 += __shfl_xor(a6, 32); a7 += __shfl_xor(a7, 16);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
int run(const char* name, int ops_per_iter, int blocks, float* d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, ITERS, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)blocks * 4;
    const double wave_instr = waves * ITERS * ops_per_iter;
    // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
    const double cyc = ms * 1e-3 * 2.4e9 * 1024.0 / wave_instr;
    printf("%-34s blocks=%5d  %8.3f ms  %.2f SIMD-cycles per wave-instruction (at 2.4 GHz)\n", name, blocks, ms, cyc);
    return 0;
}

int main() {
    float* d;
    CHECK(hipMalloc(&d, sizeof(float) * 256 * 8192));
    for (int blocks : {256, 512, 1024, 2048, 4096}) {   // 1 / 2 / 4 / 8 / 16 waves per SIMD
        run<0>("v_fma_f32", 8, blocks, d);
        run<10>("v_fma_f32 dependent chain", 8, blocks, d);
        run<11>("v_fma_f32 2 chains", 8, blocks, d);
        run<1>("v_pk_fma_f32", 4, blocks, d);
        run<6>("v_fma_f32 clamp", 8, blocks, d);
        run<2>("v_mul_lo_u32", 8, blocks, d);
        run<3>("v_cmp+v_cndmask (pair)", 8, blocks, d);
        run<7>("v_rsq_f32", 8, blocks, d);
        run<4>("ds_read_b128 bcast (+2 add)", 8, blocks, d);
        run<5>("v_readlane (+1 add)", 8, blocks, d);
        run<8>("dpp add", 8, blocks, d);
        run<9>("ds_bpermute (+1 add)", 8, blocks, d);
    }
    CHECK(hipFree(d));
    return 0;
}
