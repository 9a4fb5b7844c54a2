// The team exchange WITHOUT a counter: every published word validates itself.  K workgroups share an env; per round every
// workgroup stores its 128 x 16 B entries (device scope, no wait for the acknowledgement) into slot set r % 3 and then polls
// the K x 128 entries of the round (thread t: entry t) until none of the four words is the sentinel any more.  After the
// gather of round r every writer knows that all members have read round r - 1 (they published r after reading it), so it
// resets its round r - 1 entries to the sentinel; that store is acknowledged before the writer publishes r + 1 (one round
// later -- the wait is free), hence visible before anybody can poll the set again at round r + 2.  Three slot sets make that
// ordering hold; with two the reset would race with the next poll.
// Compare tools/microbench/team_barrier.hip mode 2 (store, wait, counter, spin, load): 1.5 us per round.
// POLL = 0: one load per poll, wait, check.  POLL = 1: two loads in flight, half a round trip apart.
// GAP: cycles of "step" between the gather and the next publish (s_sleep), as in the real kernel.
// hipcc -O3 --offload-arch=gfx950 team_sentinel.hip -o team_sentinel && ./team_sentinel
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int SEG = 128;
constexpr unsigned SENT = 0xffffffffu;
constexpr int LIMIT = 1 << 14;

__device__ __forceinline__ void load_dev(u4& v, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void store_dev(void* p, u4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ bool fresh(u4 v) { return v.x != SENT && v.y != SENT && v.z != SENT && v.w != SENT; }

template <int K, int POLL>
__global__ __launch_bounds__(1024) void k_team(u4* buf /*[3][teams][K*SEG]*/, int teams, int rounds, int gap, int d0, int ds, int skew, unsigned* errors, unsigned long long* cycles) {
    __shared__ u4 tile[K * SEG];
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int team = (j / K) * 8 + xcd, k = j % K;
    if (team >= teams) return;
    unsigned bad = 0;
    unsigned long long t0 = 0, t1 = 0, t_gather = 0;
    bool lost = false;
    for (int r = 0; r < rounds; ++r) {
        u4* set = buf + ((size_t)(r % 3) * teams + team) * (K * SEG);
        for (int g = 0; g < ((k + r) % K) * skew; g += 64) __builtin_amdgcn_s_sleep(1);   // imbalance between the members, rotating
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the reset of this set's predecessor (issued a round ago) is acknowledged
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if (threadIdx.x < SEG) store_dev(&set[k * SEG + threadIdx.x], u4{(unsigned)r, (unsigned)k, threadIdx.x, 1u});
        const u4* src = &set[threadIdx.x];
        u4 v;
        if (!lost) {
            if constexpr (POLL == 0) {
                int tries = 0;
                for (int g = 0; g < d0; ++g) __builtin_amdgcn_s_sleep(1);
                for (;;) {
                    load_dev(v, src);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory");
                    if (__builtin_amdgcn_ballot_w64(!fresh(v)) == 0ull) break;
                    if (++tries >= LIMIT) { lost = true; break; }
                    for (int g = 0; g < ds; ++g) __builtin_amdgcn_s_sleep(1);
                }
            } else {
                u4 a, b;
                int tries = 0;
                load_dev(a, src);
                __builtin_amdgcn_s_sleep(6);
                load_dev(b, src);
                for (;;) {
                    asm volatile("s_waitcnt vmcnt(1)" : "+v"(a), "+v"(b)::"memory");
                    if (__builtin_amdgcn_ballot_w64(!fresh(a)) == 0ull) { v = a; break; }
                    load_dev(a, src);
                    asm volatile("s_waitcnt vmcnt(1)" : "+v"(a), "+v"(b)::"memory");
                    if (__builtin_amdgcn_ballot_w64(!fresh(b)) == 0ull) { v = b; break; }
                    load_dev(b, src);
                    if (++tries >= LIMIT) { lost = true; v = a; break; }
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(v)::"memory");   // the load still in flight owns its registers until it lands
            }
        }
        tile[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
            t_gather += t1 - t0;
        }
        // everybody has published round r, i.e. has finished reading round r - 1: its entries go back to the sentinel
        if (r >= 1 && threadIdx.x < SEG) store_dev(buf + ((size_t)((r - 1) % 3) * teams + team) * (K * SEG) + k * SEG + threadIdx.x, u4{SENT, SENT, SENT, SENT});
        const u4 c = tile[(threadIdx.x * 7 + 3) % (K * SEG)];
        const int e = (threadIdx.x * 7 + 3) % (K * SEG);
        if (!lost && (c.x != (unsigned)r || c.y != (unsigned)(e / SEG) || c.z != (unsigned)(e % SEG))) ++bad;
        for (int g = 0; g < gap; g += 64) __builtin_amdgcn_s_sleep(1);   // ~64 cycles each
        __syncthreads();
    }
    if (lost) atomicAdd(errors, 1000000u);
    if (bad) atomicAdd(errors, bad);
    if (threadIdx.x == 0) atomicAdd(&cycles[0], t_gather);
}

template <int K, int POLL>
int run(int teams, int gap, int d0 = 0, int ds = 0, int skew = 0, int rounds = 2000) {
    u4* buf; unsigned* err; unsigned long long* cyc;
    const size_t bytes = sizeof(u4) * 3 * teams * K * SEG;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&err, 4)); CHECK(hipMalloc(&cyc, 8));
    for (int pass = 0; pass < 2; ++pass) {
        CHECK(hipMemset(buf, 0xff, bytes)); CHECK(hipMemset(err, 0, 4)); CHECK(hipMemset(cyc, 0, 8));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_team<K, POLL>), dim3(teams * K), dim3(1024), 0, 0, buf, teams, rounds, gap, d0, ds, skew, err, cyc);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h_err; unsigned long long h_cyc;
        CHECK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&h_cyc, cyc, 8, hipMemcpyDeviceToHost));
        if (pass == 1)
            printf("sentinel K=%d poll=%d teams=%3d gap=%5d d0=%2d ds=%2d skew=%4d: %6.2f us per round | publish+gather %5.0f cycles per round | errors %u\n", K, POLL, teams, gap, d0, ds, skew,
                   ms * 1e3 / rounds, (double)h_cyc / ((double)teams * K * rounds), h_err);
    }
    CHECK(hipFree(buf)); CHECK(hipFree(err)); CHECK(hipFree(cyc));
    return 0;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (int d0 : {0, 4, 8, 12, 16}) run<8, 0>(32, 2048, d0, 0, 0);
    for (int ds : {2, 4, 8}) run<8, 0>(32, 2048, 8, ds, 0);
    for (int skew : {128, 512}) { run<8, 0>(32, 2048, 0, 0, skew); run<8, 0>(32, 2048, 8, 2, skew); run<8, 0>(32, 2048, 8, 4, skew); run<8, 1>(32, 2048, 0, 0, skew); }
    return 0;
}
