// v_mfma_f32_4x4x1_16b_f32 as a row-sum engine for the neighbour sweep: the sweep's two heading FMAs per
// (row, column) pair, s += w * (ux, uy), are a contraction I[rows x cols] . u[cols x 2] -- what the matrix pipe does, and
// that pipe idles in this code.  4x4x1 = 16 independent blocks of a 4x1 by 1x4 outer product per instruction; lane l = 4 b + i
// supplies A[b][i] and B[b][i]; the result D is four registers, register r of lane 4 b + j = sum A[b][r] * B[b][j].  With
// A = the pair weight of the lane's own row and B = (ux, uy, -, -) of the column in every block, register r of lanes 4 b + 0 /
// 4 b + 1 accumulates the x / y heading sum of the row of lane 4 b + r.
//
// Part 1 checks that layout and that the accumulation is a plain f32 fused multiply-add per term (same bits as a v_fma chain in
// the same order).  Part 2 times what the sweep would issue per (row pair, column): six packed instructions (today) against
// four packed + two MFMA, with W waves per SIMD -- cycles per iteration from s_memtime inside the kernel.
//
// hipcc -O3 --offload-arch=gfx950 mfma_4x4.hip -o mfma_4x4 && ./mfma_4x4
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
using f4 = float __attribute__((ext_vector_type(4)));
using f2 = float __attribute__((ext_vector_type(2)));

__global__ void k_layout(const float* a, const float* b, int n, float* d, float* ref) {
    const int l = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < n; ++j) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j * 64 + l], b[j * 64 + l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
    // the same sums by v_fma chains: row = the lane itself, columns in the same order
    float sx = 0.f, sy = 0.f;
    for (int j = 0; j < n; ++j) {
        const float w = a[j * 64 + l], ux = b[j * 64 + (l & ~3)], uy = b[j * 64 + (l & ~3) + 1];
        sx = __builtin_fmaf(w, ux, sx);
        sy = __builtin_fmaf(w, uy, sy);
    }
    ref[l * 2] = sx;
    ref[l * 2 + 1] = sy;
}

struct Rec { unsigned long long cyc; unsigned hw; };

// KIND 0: six packed instructions per iteration and row pair (2 subtractions, 2 FMAs for the 0/1 weights, 2 FMAs for the sums);
// KIND 1: four packed + two MFMA (one per row); KIND 2: the two MFMA alone; KIND 3: the four packed alone.  ROWS row pairs per lane.
template <int KIND, int PAIRS>
__global__ __launch_bounds__(1024) void k_mix(Rec* out, float* sink, int n_iter, float seed) {
    f2 X[PAIRS], Y[PAIRS], ax[PAIRS], ay[PAIRS];
    f4 acc[2 * PAIRS];
    for (int p = 0; p < PAIRS; ++p) {
        X[p] = f2{threadIdx.x * 1e-3f + seed + p, threadIdx.x * 2e-3f + seed};
        Y[p] = f2{threadIdx.x * 3e-3f + seed, 1.f + p};
        ax[p] = ay[p] = f2{0.f, 0.f};
        acc[2 * p] = acc[2 * p + 1] = f4{0.f, 0.f, 0.f, 0.f};
    }
    f2 t01 = {0.25f + seed, 0.5f}, t23 = {0.6f, 0.8f}, r2 = {1e-2f, 1e-2f};
    const float bpat = (threadIdx.x & 3) == 0 ? 0.6f : 0.8f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int p = 0; p < PAIRS; ++p) {
                f2 dx, dy, a, w;
                if constexpr (KIND != 2) {
                    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(X[p]), "v"(t01));
                    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(Y[p]), "v"(t01));
                    asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(a) : "v"(dy), "v"(r2));
                    asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(w) : "v"(dx), "v"(a));
                } else {
                    w = X[p];
                }
                if constexpr (KIND == 0) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ax[p]) : "v"(w), "v"(t23));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(ay[p]) : "v"(w), "v"(t23));
                } else if constexpr (KIND == 1 || KIND == 2) {
                    const float w0 = w.x, w1 = w.y;
                    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[2 * p]) : "v"(w0), "v"(bpat));
                    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[2 * p + 1]) : "v"(w1), "v"(bpat));
                } else {
                    ax[p] += w;      // (keeps the weights alive)
                }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int p = 0; p < PAIRS; ++p) s += ax[p].x + ax[p].y + ay[p].x + ay[p].y + acc[2 * p][0] + acc[2 * p][1] + acc[2 * p + 1][0] + acc[2 * p + 1][1];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Rec{t1 - t0, hw};
    }
}

template <int KIND, int PAIRS>
static double run(int waves_per_simd, Rec* d_out, float* d_sink) {
    const int n_iter = 512, block = 256 * waves_per_simd, grid = 256;       // one workgroup per CU, waves_per_simd on every SIMD
    hipLaunchKernelGGL((k_mix<KIND, PAIRS>), dim3(grid), dim3(block), 0, 0, d_out, d_sink, n_iter, 0.125f);
    hipLaunchKernelGGL((k_mix<KIND, PAIRS>), dim3(grid), dim3(block), 0, 0, d_out, d_sink, n_iter, 0.125f);
    hipDeviceSynchronize();
    const int n = grid * block / 64;
    std::vector<Rec> h(n);
    hipMemcpy(h.data(), d_out, n * sizeof(Rec), hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (auto& r : h) c.push_back((double)r.cyc / (n_iter * 4.0 * PAIRS));
    std::sort(c.begin(), c.end());
    return c[c.size() / 2];                 // cycles per (row pair, column) of ONE wave
}

int main() {
    // ---- part 1: layout and arithmetic ----
    const int n = 37;
    std::vector<float> a(n * 64), b(n * 64), d(256), ref(128);
    for (int j = 0; j < n; ++j)
        for (int l = 0; l < 64; ++l) {
            a[j * 64 + l] = ((l * 7 + j * 13) % 5 == 0) ? 0.f : 1.f;                  // 0/1 weights
            b[j * 64 + l] = std::sin(0.37f * (j + 1) + 0.11f * (l & 3) + 0.05f * (l >> 2));   // (the sweep has the same B in every block; here it differs per block on purpose)
        }
    float *da, *db, *dd, *dr;
    CHECK(hipMalloc(&da, a.size() * 4)); CHECK(hipMalloc(&db, b.size() * 4)); CHECK(hipMalloc(&dd, 1024)); CHECK(hipMalloc(&dr, 512));
    CHECK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, n, dd, dr);
    CHECK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ref.data(), dr, 512, hipMemcpyDeviceToHost));
    int bad_bits = 0, bad_val = 0;
    for (int l = 0; l < 64; ++l) {
        const int blk = l & ~3, r = l & 3;
        const float sx = d[(blk + 0) * 4 + r], sy = d[(blk + 1) * 4 + r];           // register r of lanes 4b + 0 / 4b + 1
        if (std::memcmp(&sx, &ref[l * 2], 4) || std::memcmp(&sy, &ref[l * 2 + 1], 4)) ++bad_bits;
        if (std::fabs(sx - ref[l * 2]) > 1e-5f || std::fabs(sy - ref[l * 2 + 1]) > 1e-5f) ++bad_val;
    }
    printf("layout: row of lane 4b+r = register r of lanes 4b+0 (x) / 4b+1 (y): %s   (%d of 64 rows differ beyond 1e-5, %d differ in bits from the v_fma chain)\n",
           bad_val == 0 ? "confirmed" : "NOT confirmed", bad_val, bad_bits);
    // ---- part 2: throughput of the mixed stream ----
    Rec* d_out; float* d_sink;
    CHECK(hipMalloc(&d_out, 256 * 16 * sizeof(Rec))); CHECK(hipMalloc(&d_sink, 16));
    printf("cycles of one wave per (row pair, column); W waves per SIMD on every SIMD; SIMD time per (row pair, column) = that / W\n");
    printf("%-44s %8s %8s %8s\n", "per (row pair, column)", "W=1", "W=2", "W=4");
    for (int pairs : {1, 2}) {
        for (int kind = 0; kind < 4; ++kind) {
            double c[3];
            int wi = 0;
            for (int w : {1, 2, 4}) {
                if (pairs == 1) c[wi] = kind == 0 ? run<0, 1>(w, d_out, d_sink) : kind == 1 ? run<1, 1>(w, d_out, d_sink) : kind == 2 ? run<2, 1>(w, d_out, d_sink) : run<3, 1>(w, d_out, d_sink);
                else c[wi] = kind == 0 ? run<0, 2>(w, d_out, d_sink) : kind == 1 ? run<1, 2>(w, d_out, d_sink) : kind == 2 ? run<2, 2>(w, d_out, d_sink) : run<3, 2>(w, d_out, d_sink);
                ++wi;
            }
            const char* names[4] = {"6 packed (today)", "4 packed + 2 MFMA 4x4x1", "2 MFMA 4x4x1 alone", "4 packed alone"};
            printf("%-34s rows/lane %d %8.1f %8.1f %8.1f   (SIMD at W=4: %.1f)\n", names[kind], 2 * pairs, c[0], c[1], c[2], c[2] / 4);
        }
    }
    return 0;
}
