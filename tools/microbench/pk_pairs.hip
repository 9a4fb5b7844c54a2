// Packed-f32 form of the neighbour test: correctness of the VOP3P modifiers it relies on (op_sel broadcast, neg_lo/neg_hi,
// clamp on both halves, inf / NaN behaviour) and its issue rate against the scalar form, on gfx950.
// hipcc -O3 --offload-arch=gfx950 pk_pairs.hip -o pk_pairs && ./pk_pairs
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// scalar form (what the kernels use today): 5 VALU per pair
__device__ __forceinline__ void pair_scalar(float XI, float YI, float tx, float ty, float ux, float uy, float r2b, float& sx, float& sy) {
    const float DX = XI - tx, DY = YI - ty;
    const float a = fmaf(-DY, DY, r2b);
    float w;
    asm("v_fma_f32 %0, -%1, %1, %2 clamp" : "=v"(w) : "v"(DX), "v"(a));
    sx = fmaf(w, ux, sx);
    sy = fmaf(w, uy, sy);
}
// packed over two PEERS: P = (XI, YI); TX = (x_j, x_j+1), TY, UX, UY likewise; 6 VOP3P per 2 pairs
__device__ __forceinline__ void pair_packed(f2 P, f2 TX, f2 TY, f2 UX, f2 UY, f2 R2, f2& SX, f2& SY) {
    f2 DX, DY, A, W;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(DX) : "v"(P), "v"(TX));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(DY) : "v"(P), "v"(TY));
    asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(A) : "v"(DY), "v"(R2));
    asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(W) : "v"(DX), "v"(A));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(SX) : "v"(W), "v"(UX));
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(SY) : "v"(W), "v"(UY));
}
// packed over two ROWS against one uniform peer t = (x, y, ux, uy): XR = (X_r, X_r+1), YR likewise
__device__ __forceinline__ void pair_packed_rows(f2 XR, f2 YR, f4 t, f2 R2, f2& SX, f2& SY) {
    f2 DX, DY, A, W;
    const f2 txy = {t.x, t.y}, tuv = {t.z, t.w};
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(DX) : "v"(XR), "v"(txy));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(DY) : "v"(YR), "v"(txy));
    asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(A) : "v"(DY), "v"(R2));
    asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(W) : "v"(DX), "v"(A));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(SX) : "v"(W), "v"(tuv));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(SY) : "v"(W), "v"(tuv));
}

// ---- correctness: every lane tests its point against 64 peers in all three forms ----
__global__ void k_check(const f4* pts, int n, float r2b, float* out) {
    const int i = threadIdx.x;
    const f4 me = pts[i];
    float sx = 0, sy = 0;
    for (int j = 0; j < n; ++j) pair_scalar(me.x, me.y, pts[j].x, pts[j].y, pts[j].z, pts[j].w, r2b, sx, sy);
    f2 SX = {0, 0}, SY = {0, 0};
    const f2 P = {me.x, me.y}, R2 = {r2b, r2b};
    for (int j = 0; j < n; j += 2) {
        const f4 a = pts[j], b = pts[j + 1];
        pair_packed(P, f2{a.x, b.x}, f2{a.y, b.y}, f2{a.z, b.z}, f2{a.w, b.w}, R2, SX, SY);
    }
    // rows form: rows (i, i ^ 1)
    const f4 other = pts[i ^ 1];
    f2 RX = {0, 0}, RY = {0, 0};
    for (int j = 0; j < n; ++j) pair_packed_rows(f2{me.x, other.x}, f2{me.y, other.y}, pts[j], R2, RX, RY);
    out[i * 8 + 0] = sx; out[i * 8 + 1] = sy;
    out[i * 8 + 2] = SX.x; out[i * 8 + 3] = SX.y; out[i * 8 + 4] = SY.x; out[i * 8 + 5] = SY.y;
    out[i * 8 + 6] = RX.x; out[i * 8 + 7] = RY.x;
}

// ---- rate: 16 peers per LDS round trip like the C2 kernel, scalar vs packed ----
template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int n_iter, int n_cols) {
    __shared__ f4 tile[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    tile[wv][lane] = f4{lane * 0.01f * 0x1.0p40f, lane * 0.02f * 0x1.0p40f, 0.6f, 0.8f};
    __syncthreads();
    const float XI = lane * 0.013f * 0x1.0p40f, YI = lane * 0.017f * 0x1.0p40f;
    const float r2b = 0.01f * 0x1.0p80f;
    float sx = 0, sy = 0;
    f2 SX = {0, 0}, SY = {0, 0};
    const f2 P = {XI, YI}, R2 = {r2b, r2b};
    const f4* __restrict__ t = tile[wv];
    for (int it = 0; it < n_iter; ++it) {
        for (int j = 0; j + 16 <= n_cols; j += 16) {
            f4 e[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) e[k] = t[j + k];
            if constexpr (KIND == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) pair_scalar(XI, YI, e[k].x, e[k].y, e[k].z, e[k].w, r2b, sx, sy);
            } else {
                // tile read as SoA pairs: entry 2k = (x0, x1, y0, y1), entry 2k+1 = (ux0, ux1, uy0, uy1)
#pragma unroll
                for (int k = 0; k < 16; k += 2)
                    pair_packed(P, f2{e[k].x, e[k].y}, f2{e[k].z, e[k].w}, f2{e[k + 1].x, e[k + 1].y}, f2{e[k + 1].z, e[k + 1].w}, R2, SX, SY);
            }
        }
        asm volatile("" : "+v"(sx), "+v"(sy), "+v"(SX), "+v"(SY));
    }
    out[blockIdx.x * 256 + threadIdx.x] = sx + sy + SX.x + SX.y + SY.x + SY.y;
}

template <int KIND>
int rate(const char* name, int blocks, float* d) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4096, n_cols = 64;
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16, n_cols);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, n_cols);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double pairs_per_simd = (double)blocks * 4 * iters * n_cols / 1024.0;
    printf("%-10s blocks=%5d  %8.3f ms  %.2f SIMD-cycles per wave-pair (at 2.4 GHz)\n", name, blocks, ms, ms * 1e-3 * 2.4e9 / pairs_per_simd);
    return 0;
}

int main() {
    const int n = 64;
    std::vector<f4> pts(n);
    srand(1);
    const float S = 0x1.0p40f;
    for (int i = 0; i < n; ++i) {
        const float x = (rand() % 2000 - 1000) * 1e-3f * 0.3f, y = (rand() % 2000 - 1000) * 1e-3f * 0.3f;
        pts[i] = f4{x * S, y * S, cosf(i * 0.7f), sinf(i * 0.7f)};
    }
    pts[5] = pts[4];                                           // coincident points
    pts[10] = f4{INFINITY, 0.3f * S, 0.0f, 0.0f};             // padding entries
    pts[11] = f4{INFINITY, -0.2f * S, 0.0f, 0.0f};
    pts[20].x = pts[21].x + 0.1f * S; pts[20].y = pts[21].y;  // (near-)tie on the radius
    f4* dp; float* dout;
    CHECK(hipMalloc(&dp, n * sizeof(f4))); CHECK(hipMalloc(&dout, n * 8 * sizeof(float)));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) pts[30].z = NAN;                        // NaN heading: poisons every row
        CHECK(hipMemcpy(dp, pts.data(), n * sizeof(f4), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dp, n, 0.01f * 0x1.0p80f, dout);
        std::vector<float> o(n * 8);
        CHECK(hipMemcpy(o.data(), dout, n * 8 * sizeof(float), hipMemcpyDeviceToHost));
        double worst_p = 0, worst_r = 0; int nan_mismatch = 0, nonzero = 0;
        for (int i = 0; i < n; ++i) {
            const float sx = o[i * 8], sy = o[i * 8 + 1];
            const float px = o[i * 8 + 2] + o[i * 8 + 3], py = o[i * 8 + 4] + o[i * 8 + 5];
            const float rx = o[i * 8 + 6], ry = o[i * 8 + 7];
            if (std::isnan(sx) != std::isnan(px) || std::isnan(sy) != std::isnan(py) || std::isnan(sx) != std::isnan(rx) || std::isnan(sy) != std::isnan(ry)) nan_mismatch++;
            if (!std::isnan(sx)) { worst_p = fmax(worst_p, fmax(fabs(sx - px), fabs(sy - py))); worst_r = fmax(worst_r, fmax(fabs(sx - rx), fabs(sy - ry))); }
            if (sx != 0 || sy != 0) nonzero++;
        }
        printf("check pass %d: rows with neighbours %d, max |scalar - packed peers| %.3g, max |scalar - packed rows| %.3g, NaN mismatches %d  %s\n",
               pass, nonzero, worst_p, worst_r, nan_mismatch, (worst_p < 1e-5 && worst_r == 0 && nan_mismatch == 0) ? "OK" : "FAIL");
    }
    float* d;
    CHECK(hipMalloc(&d, sizeof(float) * 256 * 8192));
    for (int blocks : {256, 512, 1024, 2048}) {
        rate<0>("scalar", blocks, d);
        rate<1>("packed", blocks, d);
    }
    return 0;
}
