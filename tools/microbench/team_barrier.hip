// Cost of one "publish -> team barrier -> gather" round among the K workgroups that would share one env (N = 1024 split
// over K CUs): every workgroup stores its 128 x 16 B segment to global memory, arrives at the team's counter (release,
// agent scope), spins until all K have arrived (acquire), then loads all K segments.  Teams are laid out either on ONE XCD
// (workgroup ids congruent mod 8: the dispatcher deals workgroups to the 8 XCDs round-robin) or spread over all XCDs.
// Every round is checked: a segment must carry the round's stamp.
// Modes: 0 acquire/release fences + non-temporal loads, 1 acquire/release fences, 2 device-scope (sc1) stores / loads, relaxed
// counter, 3 the same through the XCD's own L2 (plain stores, sc0 loads).
// hipcc -O3 --offload-arch=gfx950 team_barrier.hip -o team_barrier && ./team_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int SEG = 128;   // entries per workgroup

template <int K, int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void k_team(f4* buf /*[2][teams][K*SEG]*/, unsigned* counters, int teams, int rounds, int same_xcd,
                                              unsigned* errors, unsigned long long* cycles, int skew = 0, int gap = 0) {
    __shared__ f4 tile[K * SEG];
    int team, k;
    if (same_xcd) {            // b = j * 8 + xcd; team t lives on xcd t % 8, its k-th member is j = (t / 8) * K + k
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        team = (j / K) * 8 + xcd;
        k = j % K;
    } else {
        team = blockIdx.x / K;
        k = blockIdx.x % K;
    }
    if (team >= teams) return;
    unsigned* ctr = counters + team * 32;   // one counter per 128-byte line
    unsigned bad = 0;
    unsigned long long t0 = 0, t_pub = 0, t_wait = 0, t_gather = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int g = 0; g < gap; g += 64) __builtin_amdgcn_s_sleep(1);
        for (int g = 0; g < ((k + r) % K) * skew; g += 64) __builtin_amdgcn_s_sleep(1);   // imbalance between the members, rotating
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        f4* seg = buf + ((size_t)(r & 1) * teams + team) * (K * SEG);
        if (threadIdx.x < SEG) {
            const f4 val = f4{(float)r, (float)k, (float)threadIdx.x, 1.0f};
            f4* dst = &seg[k * SEG + threadIdx.x];
            if constexpr (MODE == 2) {   // device-scope write-through store, then wait for its acknowledgement
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(val) : "memory");
            } else if constexpr (MODE == 3) {   // XCD-local: a plain store lands in the XCD's L2 (the L1 is write-through); wait for its acknowledgement
                asm volatile("global_store_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(dst), "v"(val) : "memory");
            } else {
                *dst = val;
            }
        }
        __syncthreads();     // all stores of the workgroup issued
        unsigned long long t1;
        if (threadIdx.x == 0) {
            if constexpr (MODE == 2) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if constexpr (MODE == 3) asm volatile("global_atomic_add %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(ctr), "v"(1u) : "memory");   // in the XCD's L2
            else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            t_pub += t1 - t0;
            const unsigned target = (unsigned)(r + 1) * K;
            int spins = 0;     // bounded: a team that is not co-resident must not hang the GPU
            if constexpr (MODE == 2) {
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            } else if constexpr (MODE == 3) {          // sc0: past this CU's L1, served by the XCD's L2
                unsigned v = 0;
                do {
                    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ctr) : "memory");
                } while (v < target && ++spins < (1 << 14));
                if (spins >= (1 << 14)) spins = 1 << 20;
            } else {
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            }
            if (spins >= (1 << 20)) { atomicAdd(errors, 1000000u); rounds = r; }   // (give up: the other members will time out too)
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            t_wait += t0 - t1;
        }
        __syncthreads();
        if constexpr (MODE == 0) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (the other threads of the workgroup)
            for (int e = threadIdx.x; e < K * SEG; e += BLOCK) tile[e] = __builtin_nontemporal_load(&seg[e]);
        } else if constexpr (MODE == 1) {              // thread 0's acquire invalidated this CU's L1 (and the L2's non-coherent lines)
            for (int e = threadIdx.x; e < K * SEG; e += BLOCK) tile[e] = seg[e];
        } else if constexpr (MODE == 3) {              // XCD-scope loads
            for (int e = threadIdx.x; e < K * SEG; e += BLOCK) {
                f4 v;
                asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&seg[e]) : "memory");
                tile[e] = v;
            }
        } else {                                       // device-scope loads: no cache invalidation at all
            for (int e = threadIdx.x; e < K * SEG; e += BLOCK) {
                f4 v;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&seg[e]) : "memory");
                tile[e] = v;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            t_gather += t1 - t0;
        }
        for (int e = threadIdx.x; e < K * SEG; e += BLOCK) {
            const f4 v = tile[e];
            if (v.x != (float)r || v.y != (float)(e / SEG) || v.z != (float)(e % SEG)) ++bad;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
    if (threadIdx.x == 0) {
        atomicAdd(&cycles[0], t_pub);
        atomicAdd(&cycles[1], t_wait);
        atomicAdd(&cycles[2], t_gather);
    }
}

template <int K, int BLOCK, int MODE>
int run(int teams, int same_xcd, int rounds = 2000, int skew = 0, int gap = 0) {
    f4* buf; unsigned* ctr; unsigned* err; unsigned long long* cyc;
    CHECK(hipMalloc(&buf, sizeof(f4) * 2 * teams * K * SEG));
    CHECK(hipMalloc(&ctr, sizeof(unsigned) * teams * 32));
    CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&cyc, 24));
    for (int pass = 0; pass < 2; ++pass) {
        CHECK(hipMemset(ctr, 0, sizeof(unsigned) * teams * 32)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(cyc, 0, 24));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_team<K, BLOCK, MODE>), dim3(teams * K), dim3(BLOCK), 0, 0, buf, ctr, teams, rounds, same_xcd, err, cyc, skew, gap);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h_err; unsigned long long h_cyc[3];
        CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h_cyc, cyc, 24, hipMemcpyDeviceToHost));
        const double n = (double)teams * K * rounds;
        if (pass == 1)
            printf("mode %d K=%d block=%4d teams=%3d %-9s skew %4d gap %4d: %6.2f us per round | cycles per round: publish %5.0f, wait %5.0f, gather %5.0f | errors %u\n", MODE, K, BLOCK, teams,
                   same_xcd ? "one XCD" : "spread", skew, gap, ms * 1e3 / rounds, h_cyc[0] / n, h_cyc[1] / n, h_cyc[2] / n, h_err);
    }
    CHECK(hipFree(buf)); CHECK(hipFree(ctr)); CHECK(hipFree(err)); CHECK(hipFree(cyc));
    return 0;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    run<8, 256, 0>(32, 1);
    run<8, 256, 1>(32, 1); run<8, 256, 1>(32, 0);
    run<8, 256, 2>(32, 1); run<8, 256, 2>(32, 0);
    run<8, 1024, 1>(32, 1); run<8, 1024, 1>(32, 0);
    run<8, 1024, 2>(32, 1); run<8, 1024, 2>(32, 0);
    run<4, 1024, 2>(32, 1); run<2, 1024, 2>(32, 1);
    run<8, 1024, 2>(32, 1, 2000, 0, 2048); run<8, 1024, 2>(32, 1, 2000, 128, 2048); run<8, 1024, 2>(32, 1, 2000, 512, 2048);   // with a step's worth of work between the rounds and imbalance
    // mode 3: everything through the XCD's own L2 (valid only for teams on ONE XCD: the "spread" line must show errors)
    run<8, 1024, 3>(32, 1); run<8, 1024, 3>(32, 0, 20);
    run<4, 1024, 3>(32, 1); run<2, 1024, 3>(32, 1);
    return 0;
}
