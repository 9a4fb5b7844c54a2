// Is a 64-byte segment that ONE store instruction of one wave writes whole (four adjacent lanes x 16 bytes, `global_store_dwordx4 ... sc1`)
// ever seen half-written by a `global_load_dwordx4 ... sc1` of four adjacent lanes of a wave on another XCD?  Uncached device memory
// (hipDeviceMallocUncached), as the chain's exchange records.  Writers (even workgroups) count k = 1, 2, ... into all 16 words of
// their segments; readers (odd workgroups) load segments and check that the 16 words agree.  Also for whole 128-byte lines (8 lanes).
// A tag inside every segment would let a chained launch validate a record's data with the load that fetched it.
// hipcc -O3 --offload-arch=gfx950 seg_atomicity.hip -o seg_atomicity && ./seg_atomicity
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_dev(void* p, u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u4 load_dev(const void* p) { u4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }

// buf: S segments of 64 bytes; pairs of workgroups (2 j, 2 j + 1) share the segments [j * 16 * waves, ...): wave w of the pair owns 16 of them
__global__ __launch_bounds__(256) void k_hammer(char* buf, unsigned iters, unsigned long long* torn64, unsigned long long* torn128, unsigned long long* reads,
                                               unsigned* xcc_of_wg) {
    const int pair = blockIdx.x >> 1, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* mine = buf + ((size_t)(pair * 4 + wave) * 16) * 64;          // 16 segments = 1 KiB per wave: one store instruction covers them
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc_of_wg[blockIdx.x] = id & 15u;
    }
    if ((blockIdx.x & 1) == 0) {
        // (writes until every reader workgroup is done, bounded: the readers must meet writes in flight all the time)
        unsigned long long* done = reads + 1;
        for (unsigned k = 1; k <= 64u * iters; ++k) {
            store_dev(mine + lane * 16, u4{k, k, k, k});
            if ((k & 1023u) == 0 && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)(gridDim.x >> 1)) break;
        }
    } else {
        unsigned long long t64 = 0, t128 = 0, n = 0;
        for (unsigned k = 0; k < iters; ++k) {
            const u4 v = load_dev(mine + lane * 16);
            const bool own = v.x == v.y && v.y == v.z && v.z == v.w;
            // the lane's quad (64 bytes) and its octet (128 bytes): every word must carry the same count
            const unsigned q0 = __shfl(v.x, lane & ~3), o0 = __shfl(v.x, lane & ~7);
            const bool ok64 = own && v.x == q0, ok128 = own && v.x == o0;
            t64 += __popcll(__ballot(!ok64)) ? 1 : 0;
            t128 += __popcll(__ballot(!ok128)) ? 1 : 0;
            n += 1;
        }
        if (lane == 0) { atomicAdd(torn64, t64); atomicAdd(torn128, t128); atomicAdd(reads, n); }
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(reads + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
int main() {
    const int pairs = 512, wgs = 2 * pairs;
    char* buf = nullptr;
    const size_t bytes = (size_t)pairs * 4 * 16 * 64;
    CK(hipExtMallocWithFlags((void**)&buf, bytes, hipDeviceMallocUncached));
    CK(hipMemset(buf, 0, bytes));
    unsigned long long* cnt = nullptr; unsigned* xcc = nullptr;
    CK(hipMalloc((void**)&cnt, 32)); CK(hipMalloc((void**)&xcc, wgs * 4));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(cnt, 0, 32)); CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_hammer, dim3(wgs), dim3(256), 0, 0, buf, 100000u, cnt, cnt + 1, cnt + 2, xcc);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[3]; CK(hipMemcpy(h, cnt, 24, hipMemcpyDeviceToHost));
        unsigned hx[2 * 512]; CK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
        int cross = 0; for (int p = 0; p < pairs; ++p) cross += hx[2 * p] != hx[2 * p + 1];
        printf("rep %d: %.1f ms, %llu wave-reads of 16 segments (%d of %d writer / reader pairs on different XCDs): wave-reads with a torn 64-byte segment %llu, with a torn 128-byte line %llu\n",
               rep, ms, h[2], cross, pairs, h[0], h[1]);
    }
    return 0;
}
