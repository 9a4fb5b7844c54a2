// Device code of libevac, part 2: the kernel FAMILIES -- how the lanes that own one env's pedestrians are grouped,
// how they exchange what the step needs from each other, and how the only O(N^2) part, the Vicsek neighbour sum
// (area.py:104-119 of the reference), is evaluated.  step_env (evac_device.h) is written once against this interface:
//
//   F::Smem, F::Ctx                 LDS block of a workgroup; "who am I" (env, slot, lane, pedestrian i, owner lane)
//   F::kEnvUniform                  per-env values (counts, flags) are wave-uniform (false only for Sub)
//   F::neighbour_sum(...)           heading sum over the moving pedestrians within the radius, per FOLLOWER/VISCEK lane
//   F::reduce<GUARD>(...)           three float sums and the set bits of up to eight ballots over the env's lanes
//   F::exit_publish / exit_fetch    the gravity exit term evaluated by the env's first idle lane
//
//   Sub<G>     G = 16 / 32 lanes of a wave per env (N <= 16 / 32): 4 / 2 envs per wave, no barrier at all
//   Wave<WPE>  WPE waves per env, all pairs: WPE = 1 for N <= 64 (4 envs per 256-thread workgroup, no workgroup
//              barrier), one workgroup per env for WPE = 2..16
//   Cells<WPE> one workgroup per env (N > 64) with a 16 x 16 cell list: peers are binned with LDS atomics, rows are
//              re-dealt to lanes in cell order, each row scans only the 3 x 3 cells around it
#pragma once

#include <type_traits>

#include "evac_common.h"

namespace evac {

// ------------------------------------------------------------------------------------------------
// Wave<WPE>: all pairs, wave-uniform broadcast reads of a compacted LDS tile
// ------------------------------------------------------------------------------------------------

// BLOCK1 (WPE == 1 only): threads per workgroup.  256 = four one-wave envs per workgroup (the default); 1024 = the
// CU-WIDE workgroup of the rollout kernel for batches that fill the chip (16 envs, one workgroup per CU, four waves per
// SIMD): the waves that share a SIMD are then known (wave w runs on SIMD w % 4), so the host can deal the envs to SIMDs by
// load (k_schedule) and the waves of a SIMD can keep pace with each other through LDS (rollout_body, kPace).
template <int WPE_, int BLOCK1_ = 256>
struct Wave {
    static constexpr int WPE = WPE_;
    static constexpr bool kEnvUniform = true;
    static constexpr int kThreadsPerEnv = WPE * kWave;
    // WPE == 1: several one-wave envs share a workgroup (no workgroup barrier is ever used there);
    // WPE >= 2: exactly one env per workgroup, so that __syncthreads() is a per-env barrier
    //           -- except in the CU-wide form (BLOCK1 = 1024: 16 / WPE envs per workgroup), where the waves of ONE env meet
    //           at a barrier of their own in LDS (sync())
    static constexpr int kBlock = (WPE == 1 || BLOCK1_ == 1024) ? (BLOCK1_ > kThreadsPerEnv ? BLOCK1_ : kThreadsPerEnv) : kThreadsPerEnv;
    static constexpr int kEnvsPerBlock = kBlock / kThreadsPerEnv;
    static constexpr bool kPace = kBlock == 1024 && kEnvsPerBlock > 1;      // CU-wide: the SIMD-mates are known, the waves keep pace (rollout_body)
    static constexpr bool kEnvBarrier = WPE > 1 && kEnvsPerBlock > 1;
    static constexpr bool kHelpers = false, kExitLane = true, kPipelined = false;
    static constexpr const char* kName = WPE == 1 ? (kPace ? "1 wave/env, all pairs, CU-wide workgroups" : "1 wave/env, all pairs") : (WPE == 2 ? "2 waves/env, all pairs" : (WPE == 4 ? (kPace ? "4 waves/env, all pairs, CU-wide workgroups" : "4 waves/env, all pairs") : (WPE == 8 ? "8 waves/env, all pairs" : "16 waves/env, all pairs")));

    // WPE > 1: two tiles used alternately, so that writing step t+1's tile needs no barrier against the waves still
    // reading step t's (the two barriers of step t+1 lie between a tile's last read and its next write)
    static constexpr int kTiles = WPE == 1 ? 1 : 2;
    // WPE > 1: ROW BLOCKING.  A wave-uniform ds_read_b128 costs 4 LDS cycles however many lanes share the address, and
    // with 16 waves per CU each reading every peer the LDS pipe, not the VALU, bounds the all-pairs loop (N = 256:
    // 16 x 256 x 4 = 16.4 k cycles per step against ~9 k of VALU).  So a wave tests each peer it fetches against the
    // rows of kRows waves (its group: kRows pedestrians per lane) and fetches only 1/kRows of the peers; the kRows
    // partial sums of a row meet in LDS.  Same VALU work, 1/kRows of the LDS reads, one more barrier.
    static constexpr int kRows = WPE == 1 ? 1 : (WPE == 2 ? 2 : 4);
    static constexpr int kPairs = WPE * kWave / 2;     // WPE == 1: column pairs of the packed tile (neighbour_sum)

    struct Smem {
        f4 tile[kTiles][kEnvsPerBlock][WPE * kWave];   // (x, y, ux, uy) of every pedestrian
        float2 rowpos[kTiles][kEnvsPerBlock][WPE == 1 ? 1 : WPE * kWave];   // (X, Y) of every pedestrian by index (row blocking)
        float2 part[kRows][kEnvsPerBlock][WPE == 1 ? 1 : WPE * kWave];       // partial heading sums [column share][pedestrian]
        f4 redf[kEnvsPerBlock][WPE];               // per-wave partial sums (reduce)
        i4 redi[kEnvsPerBlock][WPE];               // per-wave partial counts, packed in pairs
        int cols[kEnvsPerBlock][WPE];              // moving pedestrians per wave (tile compaction)
        float exitg[kEnvsPerBlock][2];            // gravity exit term from the lane that computed it (WPE > 1)
        int poison[2][kEnvsPerBlock];             // WPE > 1: a moving pedestrian has a NaN heading (by step parity)
        int bar[kEnvsPerBlock];                   // kEnvBarrier: arrivals at the env's own barrier (WPE per generation)
        int persist_cmd[kEnvsPerBlock][6];        // persistent kernels, WPE > 1: the command the env's first wave read (rollout_body)
        // rollout outputs of up to kStageSteps steps, flushed with ONE 64-lane store (GRAV kernels)
        alignas(16) float stage[kEnvsPerBlock][kStageSteps][12];   // rows written as two 16-byte vectors + 1 word
        alignas(16) int progress[kPace ? 16 : 4];                  // kPace: step counter of every wave, [SIMD][wave of the SIMD]
        int pace_sink[kPace ? 16 : 1][kPace ? kWave : 1];           // kPace: where lanes 1..63 of a wave store when lane 0 publishes the counter
        int deal_hist[kPace ? kWave + 2 : 1];                      // kPace: histogram of workgroup 0's sort of the next launch's envs (rollout_body)
        int chain_ok[kEnvsPerBlock];                               // chained launches of multi-wave envs: did the env's first wave see its generation?
    };

    struct Ctx {
        using Family = Wave<WPE_, BLOCK1_>;
        Smem& sm;
        int env, slot, wave_in_env, lane, i;
        bool owner;
        // WPE > 1: which tile this step fills; and the moving-pedestrian counts of the NEXT step (how many in the env,
        // how many in the waves before this one), which the step's own reduction delivers for free
        int par = 0, next_cols = 0, next_base = 0, next_rows = 0, next_rbase = 0;   // (rows: the pedestrians whose row is needed)
        bool have_next = false;
#ifdef EVAC_STAMP
        StampState stamp;
#endif
        __device__ __forceinline__ explicit Ctx(Smem& s) : sm(s) {
            const int t = threadIdx.x;
            slot = t / kThreadsPerEnv;
            const int tin = t - slot * kThreadsPerEnv;
            wave_in_env = tin / kWave;
            lane = tin & (kWave - 1);
            i = tin;
            env = blockIdx.x * kEnvsPerBlock + slot;
            if constexpr (WPE == 1) {   // wave-uniform by construction: let the compiler keep it in SGPRs
                env = __builtin_amdgcn_readfirstlane(env);
                slot = __builtin_amdgcn_readfirstlane(slot);
            }
            owner = i == 0;
        }
    };

    // Sync the WPE waves of one env.  WPE == 1: a wave is in lock-step; only keep the compiler from
    // moving LDS accesses across the point.  WPE > 1: one env per workgroup, so a workgroup barrier.
    static __device__ __forceinline__ void sync() {
        static_assert(!kEnvBarrier, "several multi-wave envs per workgroup: use sync(ctx)");
        if constexpr (WPE == 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    }
    // The barrier of ONE env.  One env per workgroup: the workgroup barrier.  CU-wide form: an arrival counter per env in
    // LDS -- a wave's LDS operations execute in order, so its earlier writes are visible to whoever sees its arrival; the
    // waves of the other envs of the workgroup are not held up.
    template <class C>
    static __device__ __forceinline__ void sync(C& c) {
        if constexpr (!kEnvBarrier) {
            sync();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (c.lane == 0) {
                int* ctr = &c.sm.bar[c.slot];
                const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int target = (old & ~(WPE - 1)) + WPE;      // (WPE is a power of two, the counter only grows)
                int spins = 0;       // (bounded: a lost sibling must not hang the GPU; it cannot happen -- the env's waves run one code path)
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }

    static __device__ __forceinline__ void init(Ctx& c) {
        if constexpr (WPE > 1) {
            if (c.wave_in_env == 0 && c.lane < 2) c.sm.poison[c.lane][c.slot] = 0;
            if (c.wave_in_env == 0 && c.lane == 2) c.sm.bar[c.slot] = 0;
            __syncthreads();   // (the one workgroup-wide barrier of the CU-wide form: once per kernel, every live wave passes it)
        }
    }
    // the statuses changed outside step_env (autoreset): the counts carried over from the last reduction are void
    static __device__ __forceinline__ void invalidate(Ctx& c) { c.have_next = false; }
    static __device__ __forceinline__ bool aborted(Ctx&) { return false; }   // (team kernels only: a barrier timed out)

    // Reduce 3 floats and up to 8 predicates over all lanes of the env.  Result in every lane.
    // GUARD: a barrier in front, for callers whose previous reduction may still be read by another wave.
    // WPE > 1: every wave leaves one 32-byte record (three sums, eight counts packed in pairs -- a count is at most
    // 1024); after the barrier lane w of every wave reads record w and the records are folded with DPP row_shr steps
    // (a fixed tree: deterministic), 2 LDS reads per wave instead of 11 * WPE.
    template <bool GUARD, class C>
    static __device__ __forceinline__ void reduce(const Params&, C& c, Sums& s, const unsigned long long (&pred)[8]) {
        wave_sum3(s.f0, s.f1, s.f2);
#pragma unroll
        for (int k = 0; k < 8; ++k) s.i[k] = mask_count(pred[k]);
        if constexpr (WPE > 1) {
            auto& sm = c.sm;
            if constexpr (GUARD) C::Family::sync(c);   // previous users of the records are done
            if (c.lane == 0) {
                sm.redf[c.slot][c.wave_in_env] = f4{s.f0, s.f1, s.f2, 0.0f};
                sm.redi[c.slot][c.wave_in_env] = i4{s.i[0] | (s.i[1] << 16), s.i[2] | (s.i[3] << 16), s.i[4] | (s.i[5] << 16), s.i[6] | (s.i[7] << 16)};
            }
            C::Family::sync(c);
            const int w = c.lane < WPE ? c.lane : WPE - 1;
            f4 rf = sm.redf[c.slot][w];
            i4 ri = sm.redi[c.slot][w];
#define EVAC_RED_STEP(CTRL)                                                                                      \
    rf.x = dpp_add<CTRL, 0xf>(rf.x); rf.y = dpp_add<CTRL, 0xf>(rf.y); rf.z = dpp_add<CTRL, 0xf>(rf.z);           \
    ri.x = dpp_addi<CTRL, 0xf>(ri.x); ri.y = dpp_addi<CTRL, 0xf>(ri.y); ri.z = dpp_addi<CTRL, 0xf>(ri.z);        \
    ri.w = dpp_addi<CTRL, 0xf>(ri.w);
            EVAC_RED_STEP(0x111)
            if constexpr (WPE > 2) { EVAC_RED_STEP(0x112) }
            if constexpr (WPE > 4) { EVAC_RED_STEP(0x114) }
            if constexpr (WPE > 8) { EVAC_RED_STEP(0x118) }
#undef EVAC_RED_STEP
            s.f0 = readlane_f(rf.x, WPE - 1);
            s.f1 = readlane_f(rf.y, WPE - 1);
            s.f2 = readlane_f(rf.z, WPE - 1);
            if constexpr (!GUARD && std::is_same<typename C::Family, Wave>::value) {
                // the step's own reduction: count 3 is "moves at the next step"; after the DPP steps lane w holds the sum
                // over waves 0..w, i.e. the tile offsets of the next step's compaction -- no exchange, no barrier then
                const int wi = __builtin_amdgcn_readfirstlane(c.wave_in_env);      // (uniform: keeps the selects scalar)
                const int before = wi == 0 ? 0 : __builtin_amdgcn_readlane(ri.y, max(wi - 1, 0));
                c.next_base = before >> 16;
                c.next_cols = __builtin_amdgcn_readlane(ri.y, WPE - 1) >> 16;
                // count 5 is "needs its row at the next step": the offsets of the next step's ROW compaction, likewise
                const int rbefore = wi == 0 ? 0 : __builtin_amdgcn_readlane(ri.z, max(wi - 1, 0));
                c.next_rbase = rbefore >> 16;
                c.next_rows = __builtin_amdgcn_readlane(ri.z, WPE - 1) >> 16;
                c.have_next = true;
            }
            const int a = __builtin_amdgcn_readlane(ri.x, WPE - 1), b = __builtin_amdgcn_readlane(ri.y, WPE - 1);
            const int d = __builtin_amdgcn_readlane(ri.z, WPE - 1), g = __builtin_amdgcn_readlane(ri.w, WPE - 1);
            s.i[0] = a & 0xffff; s.i[1] = a >> 16;
            s.i[2] = b & 0xffff; s.i[3] = b >> 16;
            s.i[4] = d & 0xffff; s.i[5] = d >> 16;
            s.i[6] = g & 0xffff; s.i[7] = g >> 16;
        }
    }

    // The gravity exit term is evaluated by the env's first idle lane (i == N) with the pedestrians' instructions.
    template <class C>
    static __device__ __forceinline__ void exit_publish(C& c, bool exit_lane, float gx, float gy) {
        if constexpr (WPE > 1) {
            if (exit_lane) {
                c.sm.exitg[c.slot][0] = gx;
                c.sm.exitg[c.slot][1] = gy;
            }
        }
    }
    template <class C>
    static __device__ __forceinline__ void exit_fetch(C& c, float gx, float gy, int src_i, float& ex, float& ey) {
        if constexpr (WPE == 1) {
            ex = readlane_f(gx, src_i);
            ey = readlane_f(gy, src_i);
        } else {
            ex = c.sm.exitg[c.slot][0];       // written before reduce's barrier
            ey = c.sm.exitg[c.slot][1];
        }
    }

    // Vicsek neighbour sum, all pairs: area.py:99-119.  `sx, sy` = sum of the unit headings of the moving
    // pedestrians within the radius (the count n_intersections only rescales the mean heading, which arctan2
    // ignores; it is not needed).
    static __device__ __forceinline__ void neighbour_sum(const Params& p, Ctx& c, const Ped& q, bool efv, bool fv,
                                                         float ux, float uy, float& sx, float& sy) {
        auto& sm = c.sm;
        int par = 0;
        if constexpr (WPE == 1) {
            // ---- one wave per env.  `fv`: this lane's row is needed (step_env: needs_row).  NO row at all (no VISCEK pedestrian left
            // under enslaving_degree 1: 45 % of the envs at t = 1000, tools/moving_distribution.py): no tile, no loop -- only the
            // reference's NaN poisoning (area.py:118-119: a NaN heading among the moving pedestrians makes every row NaN), which the
            // all-pairs loop carries through w * NaN, is applied by hand: it reaches the followers whose rows are not evaluated.
            sx = 0.0f;
            sy = 0.0f;
            if (ballot(fv) == 0ull) {
                if ((ballot(ux != ux || uy != uy) & ballot(efv)) != 0ull) sx = sy = __builtin_nanf("");
                return;
            }
            sync(c);   // tile readers of the previous step are done (a fence: the wave is in lock-step)
        } else {
            par = c.par;
            c.par = par ^ 1;
        }
        // The tile holds the moving pedestrians first, compacted in ascending pedestrian order -- the columns
        // pos[efv] of the reference's distance matrix (area.py:99-106) -- then the others as padding with
        // weight 0 (X = +inf) and heading 0.  Every lane writes exactly one entry.  Under a
        // RandomAgent most pedestrians have escaped by mid-episode, so the all-pairs loop shrinks from N to
        // n_efv iterations.
        int n_cols, n_rows = 0, row_rank = 0;
        unsigned long long moving_mask;
        {
            const unsigned long long m = moving_mask = ballot(efv);
            int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            n_cols = __popcll(m);
            if constexpr (WPE > 1) {
                // the ROWS are compacted too (`fv`: this pedestrian's row is needed, step_env: needs_row): late in an episode
                // most moving pedestrians are followers whose rows are not evaluated, and a lane then carries 1 row instead of kRows
                const unsigned long long mr = ballot(fv);
                row_rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mr >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mr, 0u));
                n_rows = __popcll(mr);
                if (c.have_next) {          // uniform: delivered by the previous step's reduction
                    n_cols = c.next_cols;
                    before += c.next_base;
                    n_rows = c.next_rows;
                    row_rank += c.next_rbase;
                } else {                    // first step of a launch, or right after an autoreset
                    if (c.lane == 0) sm.cols[c.slot][c.wave_in_env] = n_cols | (n_rows << 16);
                    sync(c);
                    int tot = 0, base = 0;
#pragma unroll
                    for (int w2 = 0; w2 < WPE; ++w2) {
                        const int k = sm.cols[c.slot][w2];
                        base += (w2 < c.wave_in_env) ? k : 0;
                        tot += k;
                    }
                    n_cols = tot & 0xffff;
                    before += base & 0xffff;                                    // moving pedestrians before this one
                    n_rows = tot >> 16;
                    row_rank += base >> 16;
                }
            }
            const int tid = c.wave_in_env * kWave + c.lane;
            const int idx = efv ? before : n_cols + (tid - before);             // a bijection onto [0, WPE*64)
            if constexpr (WPE == 1) {
                // columns in PAIRS for the packed loop: pair p = idx / 2 holds (X, X', Y, Y') in tile[p] and (ux, ux', uy, uy') in
                // tile[32 + p].  All position chunks first, then all heading chunks: a lane's four dwords go to dword 4 p + (idx & 1)
                // + {0, 2} and 128 more, so the 32 lanes of a store group land on 16 chunks x 2 = all 32 write banks, two deep at
                // most -- free for ds_write_b32 (MI355X_MICROARCH.md, LDS).  Interleaved (tile[2p], tile[2p + 1]) the same stores
                // hit 8 banks four deep: 15.6 % of the kernel's LDS cycles were conflict cycles (profiles/r03_g_c2_driver_pmc_summary.txt).
                float* tf = (float*)sm.tile[par][c.slot] + (idx >> 1) * 4 + (idx & 1);
                tf[0] = efv ? q.x * kTileScale : __builtin_inff();
                tf[2] = q.y * kTileScale;
                tf[kPairs * 4 + 0] = efv ? ux : 0.0f;
                tf[kPairs * 4 + 2] = efv ? uy : 0.0f;
            } else {
                sm.tile[par][c.slot][idx] = f4{efv ? q.x * kTileScale : __builtin_inff(), q.y * kTileScale, efv ? ux : 0.0f, efv ? uy : 0.0f};
            }
        }
        if constexpr (WPE > 1) {
            if (fv) sm.rowpos[par][c.slot][row_rank] = make_float2(q.x * kTileScale, q.y * kTileScale);
            // The reference's NaN poisoning (a NaN heading makes every FOLLOWER / VISCEK row NaN, area.py:118-119) reaches the
            // evaluated rows through w * NaN; the followers whose rows are skipped get it through this flag.
            if ((ballot(ux != ux || uy != uy) & moving_mask) != 0ull && c.lane == 0) sm.poison[par][c.slot] = 1;   // (conjunction on the masks)
        }
        sync(c);   // tile complete
        bool poisoned = false;
        if constexpr (WPE > 1) {
            poisoned = sm.poison[par][c.slot] != 0;
            if (c.wave_in_env == 0 && c.lane == 0) sm.poison[par ^ 1][c.slot] = 0;   // the other parity's flag: last read before this barrier
        }
        EVAC_T(c, 2);   // tile write
        sx = 0.0f;
        sy = 0.0f;
        const f4* __restrict__ tile = sm.tile[par][c.slot];
        const float r2b = kRPed2Big;
        if constexpr (WPE == 1) {
            // MANY rows: all pairs, every lane a row (rows exist only for FOLLOWER/VISCEK pedestrians, area.py:104; the lanes
            // without one compute a sum nobody uses).  Branch-free batches: the B wave-uniform ds_read_b128 broadcasts are issued
            // back to back (LDS latency paid once per batch, no VALU slot), then 6 full-rate VALU ops per pair.
            const int n8 = __builtin_amdgcn_readfirstlane((n_cols + 3) & ~3);
            const float XI = q.x * kTileScale, YI = q.y * kTileScale;
            // peers per LDS round trip: 16 (3.31 vs 3.34 us at 8, 3.49 at 4)
            constexpr int B = 16;
            if constexpr (!(EVAC_ABLATE & 1)) {
                // two columns per packed instruction (pair2_accumulate): 3 vector instructions per column instead of 5
                const f2 P = f2{XI, YI}, r2b2 = f2{r2b, r2b};
                f2 sx2 = f2{0.0f, 0.0f}, sy2 = f2{0.0f, 0.0f};
                const f4* __restrict__ txy = tile;             // pair m: (X, X', Y, Y')
                const f4* __restrict__ tuv = tile + kPairs;    //         (ux, ux', uy, uy')
                int m = 0;                                     // pair index = column / 2
                const int mp = n8 >> 1;
                for (; m + B / 2 <= mp; m += B / 2) {          // full batches: 16 columns = 8 pairs = 16 tile reads
                    f4 a[B / 2], u[B / 2];
#pragma unroll
                    for (int k = 0; k < B / 2; ++k) { a[k] = txy[m + k]; u[k] = tuv[m + k]; }
#pragma unroll
                    for (int k = 0; k < B / 2; ++k) pair2_accumulate(P, a[k], u[k], r2b2, sx2, sy2);
                }
                // the remainder (4, 8 or 12 columns: n8 is a multiple of 4) in at most two batches, 8 + 4 -- two LDS round trips,
                // not one per group of 4 (the heaviest envs, 57..60 moving pedestrians, have 12 left: their wave ends the
                // launch; a 12-column batch -- one trip -- was measured in round 4: no gain, profiles/r04_f_c2_ab_tail12_reflect_skip_no_gain.txt)
                if (mp - m >= 4) {
                    f4 a[4], u[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { a[k] = txy[m + k]; u[k] = tuv[m + k]; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) pair2_accumulate(P, a[k], u[k], r2b2, sx2, sy2);
                    m += 4;
                }
                if (m < mp) {
                    f4 a[2], u[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) { a[k] = txy[m + k]; u[k] = tuv[m + k]; }
#pragma unroll
                    for (int k = 0; k < 2; ++k) pair2_accumulate(P, a[k], u[k], r2b2, sx2, sy2);
                }
                sx = hsum2(sx2);                   // even columns + odd columns
                sy = hsum2(sy2);
            }
        } else {
            // this wave: up to kRows row slices (64 compacted rows each) of its group of kRows waves, column share `share`
            // (all of this is wave-uniform: kept in scalar registers, unsigned so that the divisions are shifts)
            const unsigned wu = (unsigned)__builtin_amdgcn_readfirstlane(c.wave_in_env);
            const unsigned nc = (unsigned)__builtin_amdgcn_readfirstlane(n_cols), nr = (unsigned)__builtin_amdgcn_readfirstlane(n_rows);
            const unsigned share_u = wu % (unsigned)kRows, first_row = (wu - share_u) * (unsigned)kWave;
            const int share = (int)share_u, gbase = (int)first_row + c.lane;
            const int slices = nr > first_row ? (int)min((nr - first_row + (unsigned)kWave - 1u) / (unsigned)kWave, (unsigned)kRows) : 0;
            const unsigned groups = (nc + 3u) >> 2;                                // peers in groups of 4 (padding weighs 0)
            const unsigned per = (groups + (unsigned)kRows - 1u) / (unsigned)kRows;
            const int jbeg = (int)(share_u * per * 4u);
            const int jend = (int)(min((share_u + 1u) * per, groups) * 4u);
            if constexpr (!(EVAC_ABLATE & 1)) {
                // R row slices per lane (slots beyond n_rows hold stale positions: computed, never read)
                auto sweep = [&](auto r_tag) {
                    constexpr int R = decltype(r_tag)::value;
                    constexpr int R2 = R / 2;         // pairs of rows taken in packed arithmetic (pair_accumulate_rows2)
                    float X[R], Y[R], ax[R], ay[R];
                    f2 X2[R2 ? R2 : 1], Y2[R2 ? R2 : 1], ax2[R2 ? R2 : 1], ay2[R2 ? R2 : 1];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float2 rp = sm.rowpos[par][c.slot][gbase + r * kWave];
                        X[r] = rp.x; Y[r] = rp.y;
                        ax[r] = 0.0f; ay[r] = 0.0f;
                    }
#pragma unroll
                    for (int r = 0; r < R2; ++r) {
                        X2[r] = f2{X[2 * r], X[2 * r + 1]}; Y2[r] = f2{Y[2 * r], Y[2 * r + 1]};
                        ax2[r] = f2{0.0f, 0.0f}; ay2[r] = f2{0.0f, 0.0f};
                    }
                    const f2 r2b2 = f2{r2b, r2b};
                    auto column = [&](f4 t) {
#pragma unroll
                        for (int r = 0; r < R2; ++r) pair_accumulate_rows2(X2[r], Y2[r], t, r2b2, ax2[r], ay2[r]);
#pragma unroll
                        for (int r = 2 * R2; r < R; ++r) pair_accumulate(X[r], Y[r], t, r2b, ax[r], ay[r]);
                    };
                    constexpr int B = (R <= 2 && !(kEnvBarrier && R == 2)) ? 8 : 4;   // peers per LDS round trip (register budget)
                    int j = jbeg;
                    for (; j + B <= jend; j += B) {
                        f4 t[B];
#pragma unroll
                        for (int k = 0; k < B; ++k) t[k] = tile[j + k];
#pragma unroll
                        for (int k = 0; k < B; ++k) column(t[k]);
                    }
                    if constexpr (B == 8) {
                        if (j < jend) {
                            f4 t[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) t[k] = tile[j + k];
#pragma unroll
                            for (int k = 0; k < 4; ++k) column(t[k]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R2; ++r) { ax[2 * r] = ax2[r].x; ax[2 * r + 1] = ax2[r].y; ay[2 * r] = ay2[r].x; ay[2 * r + 1] = ay2[r].y; }
#pragma unroll
                    for (int r = 0; r < R; ++r) sm.part[share][c.slot][gbase + r * kWave] = make_float2(ax[r], ay[r]);
                };
                if (slices == 1) sweep(std::integral_constant<int, 1>{});
                else if (slices == 2) sweep(std::integral_constant<int, 2>{});
                else if constexpr (kRows == 4) {
                    if (slices == 3) sweep(std::integral_constant<int, 3>{});
                    else if (slices == 4) sweep(std::integral_constant<int, 4>{});
                }
            }
            sync(c);
            if (fv) {
#pragma unroll
                for (int r = 0; r < kRows; ++r) {      // fixed order: deterministic
                    const float2 pr = sm.part[r][c.slot][row_rank];
                    sx += pr.x;
                    sy += pr.y;
                }
            }
            if (poisoned) sx = sy = __builtin_nanf("");
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Cells<WPE>: one workgroup per env, 16 x 16 cell list.
//
// Every step:
//   1. each moving pedestrian (E | F | V) takes a ticket in its cell's counter (ds_add_rtn; the order of arrival is
//      arbitrary);                                                                         -- barrier
//   2. wave 0 turns the 256 counters into an exclusive prefix (start[]) and clears them;    -- barrier
//   3. each moving pedestrian writes (X, Y, heading) to tile[start[cell] + ticket] and who[slot] = (i, cell):
//      the tile now lists the moving pedestrians cell by cell (x-major);                    -- barrier
//   4. ROWS ARE RE-DEALT IN TILE ORDER: the thread with workgroup index s evaluates the row of the pedestrian in tile
//      slot s.  Its peers within the radius lie in the 3 x 3 cells around its own, i.e. in three runs of three
//      consecutive cells (c-17..c-15, c-1..c+1, c+15..c+17 in x-major cell numbers; where a run wraps around the end
//      of a cell column it only picks up a few extra, harmless candidates).  The lane sweeps the three slot ranges
//      in ascending order, kRowBatch entries per LDS round trip; reading past the end of a run is harmless (the extra
//      entries are real pedestrians that are tested like any other, or +inf padding behind the last one) as long as
//      no slot is visited twice -- the next run starts where the previous batch ended if that is later.
//      Lanes of a wave own neighbouring cells, so their trip counts are similar and waves behind the last moving
//      pedestrian do nothing.  The heading sums are INTEGERS (pair_accumulate_int): exact, hence independent of
//      the arbitrary ticket order, so results are reproducible bit for bit.
//      The row result goes to res[i] of the pedestrian it belongs to;                        -- barrier
//   5. every pedestrian reads its own res[i].
// A NaN heading (0/0, area.py:101) poisons every row in the reference (NaN * 0, area.py:118-119); here such a
// pedestrian bumps counter 256 and every row adds NaN when that count is non-zero.
// ------------------------------------------------------------------------------------------------
template <int WPE_>
struct Cells {
    static constexpr int WPE = WPE_;
    static_assert(WPE >= 2, "Cells is a workgroup-per-env family");
    static constexpr bool kEnvUniform = true;
    static constexpr int kThreadsPerEnv = WPE * kWave;
    static constexpr int kBlock = kThreadsPerEnv;
    static constexpr int kEnvsPerBlock = 1;
    static constexpr bool kPace = false, kHelpers = false, kExitLane = true, kPipelined = false, kEnvBarrier = false;
    static constexpr int kPad = 8;   // +inf entries behind the last moving pedestrian (>= entries per batch)
    static constexpr int kRowBatch = 8;   // tile entries per LDS round trip of a row
    static constexpr int kTransposedWork = 1024;   // needed rows x passes of 64 columns up to which the sweep is transposed (step 4')
    static constexpr const char* kName = WPE == 2 ? "2 waves/env, cell list" : (WPE == 4 ? "4 waves/env, cell list" : (WPE == 8 ? "8 waves/env, cell list" : "16 waves/env, cell list"));

    struct Smem {
        f4 tile[1][kThreadsPerEnv + kPad];
        alignas(16) int cnt[kCells + 4];      // tickets per cell; [256] = pedestrians with a NaN heading
        alignas(16) int start[kCells + 4];    // exclusive prefix of cnt; [256] = moving pedestrians, [257] = NaN headings
        int who[kThreadsPerEnv];              // tile slot -> pedestrian | cell << 16
        int rowlist[kThreadsPerEnv];          // the tile slots whose row is needed, in ticket order (cnt[kCells + 1] of them)
        i2 res[kThreadsPerEnv];               // pedestrian -> integer heading sums of its row
        // 16-wave workgroups: claim more than half of the CU's 160 KiB so that the dispatcher places ONE env per CU
        // (two 1024-thread workgroups on one CU halve each other's speed while other CUs idle: 29.8 vs 16.3 us per step
        // at 256 envs)
        char one_workgroup_per_cu[WPE == 16 ? 52 * 1024 : 16];
        f4 redf[1][WPE];
        i4 redi[1][WPE];
        float exitg[1][2];
        alignas(16) float stage[1][kStageSteps][12];
    };

    struct Ctx {
        using Family = Cells<WPE_>;
        Smem& sm;
        int env, slot, wave_in_env, lane, i;
        bool owner;
#ifdef EVAC_STAMP
        StampState stamp;
#endif
        __device__ __forceinline__ explicit Ctx(Smem& s) : sm(s) {
            const int t = threadIdx.x;
            slot = 0;
            wave_in_env = t / kWave;
            lane = t & (kWave - 1);
            i = t;
            env = blockIdx.x;
            owner = i == 0;
        }
    };

    static __device__ __forceinline__ void sync() { __syncthreads(); }
    template <class C>
    static __device__ __forceinline__ void sync(C&) { __syncthreads(); }

    static __device__ __forceinline__ void invalidate(Ctx&) {}
    static __device__ __forceinline__ bool aborted(Ctx&) { return false; }
    // Once per kernel, before the first step: the counters start at zero (afterwards the prefix wave clears them).
    static __device__ __forceinline__ void init(Ctx& c) {
        for (int k = c.i; k < kCells + 4; k += kThreadsPerEnv) c.sm.cnt[k] = 0;
        __syncthreads();
    }

    template <bool GUARD, class C>
    static __device__ __forceinline__ void reduce(const Params& p, C& c, Sums& s, const unsigned long long (&pred)[8]) {
        Wave<WPE>::template reduce<GUARD>(p, c, s, pred);
    }
    template <class C>
    static __device__ __forceinline__ void exit_publish(C& c, bool exit_lane, float gx, float gy) {
        Wave<WPE>::exit_publish(c, exit_lane, gx, gy);
    }
    template <class C>
    static __device__ __forceinline__ void exit_fetch(C& c, float gx, float gy, int src_i, float& ex, float& ey) {
        Wave<WPE>::exit_fetch(c, gx, gy, src_i, ex, ey);
    }

    static __device__ __forceinline__ int cell_of(const Params& p, float x, float y) {
        // monotone in x and in y; the float -> int conversion saturates and maps NaN to 0
        int cx = (int)((x + p.cell_ox) * p.cell_inv_hx), cy = (int)((y + p.cell_oy) * p.cell_inv_hy);
        cx = min(max(cx, 0), kCellsX - 1);
        cy = min(max(cy, 0), kCellsY - 1);
        return cx * kCellsY + cy;
    }

    static __device__ __forceinline__ void neighbour_sum(const Params& p, Ctx& c, const Ped& q, bool efv, bool fv,
                                                         float ux, float uy, float& sx, float& sy) {
        auto& sm = c.sm;
        // ---- 1. tickets ----
        const int cell = cell_of(p, q.x, q.y);
        int ticket = 0;
        if (efv) {
            ticket = atomicAdd(&sm.cnt[cell], 1);                                 // ds_add_rtn_u32
            if (ux != ux || uy != uy) atomicAdd(&sm.cnt[kCells], 1);              // rare: the reference's NaN poisoning
        }
        __syncthreads();
        // ---- 2. exclusive prefix over the 256 cells by wave 0 (4 cells per lane), counters cleared ----
        if (c.wave_in_env == 0) {
            const i4 v = *(const i4*)&sm.cnt[4 * c.lane];
            const int p1 = v.x, p2 = p1 + v.y, p3 = p2 + v.z, tot = p3 + v.w;
            const int incl = wave_inclusive_scan(tot);
            const int ex = incl - tot;
            *(i4*)&sm.start[4 * c.lane] = i4{ex, ex + p1, ex + p2, ex + p3};
            *(i4*)&sm.cnt[4 * c.lane] = i4{0, 0, 0, 0};
            if (c.lane == kWave - 1) {
                sm.start[kCells] = incl;                                           // moving pedestrians
                sm.start[kCells + 1] = sm.cnt[kCells];
                sm.cnt[kCells] = 0;
                sm.cnt[kCells + 1] = 0;                                            // tickets of the needed rows (step 3)
            }
        }
        __syncthreads();
        // ---- 3. the tile in cell order ----
        const int n_cols = sm.start[kCells];
        const int n_nan = sm.start[kCells + 1];
        if (efv) {
            const int s = sm.start[cell] + ticket;
            const float hs = p.head_scale;
            const int hx = (int)__builtin_rintf(ux * hs), hy = (int)__builtin_rintf(uy * hs);
            sm.tile[0][s] = f4{q.x * kTileScale, q.y * kTileScale, __builtin_bit_cast(float, hx), __builtin_bit_cast(float, hy)};
            sm.who[s] = c.i | (fv ? 0x8000 : 0) | (cell << 16);   // bit 15: this pedestrian's row is needed
            if (fv) sm.rowlist[atomicAdd(&sm.cnt[kCells + 1], 1)] = s;
        }
        if (c.i < kPad) sm.tile[0][n_cols + c.i] = f4{__builtin_inff(), 0.0f, 0.0f, 0.0f};
        __syncthreads();
        EVAC_T(c, 2);   // binning
        // ---- 4'. FEW needed rows (most of an episode under enslaving_degree 1: only the VISCEK pedestrians have one) against
        // a tile of any size: one active lane per wave walking its cells is the worst use of the machine.  The needed rows are
        // dealt to the waves instead (in ticket order: arbitrary, and irrelevant for integer sums) and the LANES hold the
        // columns, 64 tile entries per pass whatever cell they are in -- all pairs, the same neighbour sets.
        const int n_need = __builtin_amdgcn_readfirstlane(sm.cnt[kCells + 1]);
        const int n_cols_u = __builtin_amdgcn_readfirstlane(n_cols);
        const bool transposed = n_need * ((n_cols_u + kWave - 1) / kWave) <= kTransposedWork;
        if constexpr (!(EVAC_ABLATE & 1)) {
          if (transposed) {
            const f4* __restrict__ tile = sm.tile[0];
            const int wv = __builtin_amdgcn_readfirstlane(c.wave_in_env);
            for (int k = wv; k < n_need; k += WPE) {
                const int s = sm.rowlist[k];                  // (uniform addresses: broadcasts)
                const f4 me = tile[s];
                const int ped = sm.who[s] & 0x7fff;
                int ax = 0, ay = 0;
                for (int j0 = 0; j0 < n_cols_u; j0 += kWave)
                    pair_accumulate_int(me.x, me.y, tile[min(j0 + c.lane, n_cols_u)], kRPed2Big, ax, ay);   // entry n_cols: padding, weight 0
                wave_sum2_int_lane63(ax, ay);
                if (c.lane == kWave - 1) sm.res[ped] = i2{ax, ay};
            }
          } else {
        // ---- 4. rows in tile order ----
            const int s = c.i;
            const int wc = s < n_cols ? sm.who[s] : 0;
            if (wc & 0x8000) {    // a moving pedestrian whose row is needed (a wave whose slots hold followers only does nothing)
                const f4 me = sm.tile[0][s];
                const int rc = wc >> 16;
                int lo[3], hi[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int c0 = rc + (r - 1) * kCellsY - 1;
                    lo[r] = sm.start[min(max(c0, 0), kCells)];
                    hi[r] = sm.start[min(max(c0 + 3, 0), kCells)];
                }
                int ax = 0, ay = 0;
                int j = lo[0];
                const f4* __restrict__ tile = sm.tile[0];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    j = max(j, lo[r]);
                    while (j < hi[r]) {
                        f4 t[kRowBatch];
#pragma unroll
                        for (int k = 0; k < kRowBatch; ++k) t[k] = tile[j + k];
#pragma unroll
                        for (int k = 0; k < kRowBatch; ++k) pair_accumulate_int(me.x, me.y, t[k], kRPed2Big, ax, ay);
                        j += kRowBatch;
                    }
                }
                sm.res[wc & 0x7fff] = i2{ax, ay};
            }
          }
        }
        __syncthreads();
        // ---- 5. back to the owner of the pedestrian ----
        const i2 r = sm.res[c.i];
        sx = fv ? (float)r.x : 0.0f;       // lanes without a row (not moving) hold stale words: unused
        sy = fv ? (float)r.y : 0.0f;
        if (n_nan != 0) sx = sy = __builtin_nanf("");
    }
};

// ------------------------------------------------------------------------------------------------
// Sub<G>: G lanes of a wave per env (G = 32 for N <= 32, G = 16 for N <= 16), i.e. 2 or 4 independent envs
// per 64-lane wave.  The reference's default is number_of_pedestrians = 10 (src/env/env/config.py:11); with one
// wave per env 54 of 64 lanes would idle.  What is "wave-uniform" in Wave<1> is "group-uniform" here:
//   * votes / counts: the 64-bit ballot is masked to the group's lanes (popcount / mbcnt on the masked word);
//   * float sums: DPP row_shr 1/2/4/8 reduce each 16-lane row (= a whole G=16 group), one row_bcast:15 joins
//     the two rows of a G=32 group; the total lands in the group's LAST lane, which therefore owns the
//     per-env outputs (reward, observation, state write-back);
//   * values that Wave<1> fetches with v_readlane come through ds_bpermute from a per-group source lane.
// There is no workgroup barrier anywhere (a wave is in lock-step).
// ------------------------------------------------------------------------------------------------
template <int G_, int BLOCK_ = 256>
struct Sub {
    static constexpr int G = G_;
    static_assert(G == 16 || G == 32, "sub-wave groups are 16 or 32 lanes");
    static constexpr bool kEnvUniform = false, kHelpers = false, kExitLane = true, kPipelined = false, kEnvBarrier = false;
    static constexpr int kThreadsPerEnv = G;
    static constexpr int kEnvsPerWave = kWave / G;
    static constexpr int kBlock = BLOCK_;
    static constexpr int kEnvsPerBlock = (kBlock / kWave) * kEnvsPerWave;
    static constexpr unsigned long long kGroupBits = G == 32 ? 0xffffffffull : 0xffffull;
    static constexpr const char* kName = G == 16 ? "4 envs/wave, all pairs" : "2 envs/wave, all pairs";

    struct Smem {
        f4 tile[kEnvsPerBlock][G];   // per env: moving pedestrians first, then zero-weight padding
    };

    struct Ctx {
        Smem& sm;
        int env, slot, lane, sub, i, li;
        unsigned long long gmask;   // this group's lanes in a 64-bit ballot
        bool owner;                 // the group's last lane: the DPP sums are valid there
#ifdef EVAC_STAMP
        StampState stamp;
#endif
        __device__ __forceinline__ explicit Ctx(Smem& s) : sm(s) {
            const int t = threadIdx.x;
            lane = t & (kWave - 1);
            sub = lane / G;
            li = i = lane - sub * G;
            slot = (t / kWave) * kEnvsPerWave + sub;
            env = blockIdx.x * kEnvsPerBlock + slot;
            gmask = kGroupBits << (sub * G);
            owner = li == G - 1;
        }
    };

    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    static __device__ __forceinline__ int count(unsigned long long m, unsigned long long gmask) { return __popcll(m & gmask); }
    // number of set lanes of the group below this lane
    static __device__ __forceinline__ int rank(unsigned long long m, unsigned long long gmask) {
        const unsigned long long g = m & gmask;
        return __builtin_amdgcn_mbcnt_hi((unsigned)(g >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)g, 0u));
    }
    // value held by lane `src_i` of this lane's group
    static __device__ __forceinline__ float fetch(float v, int sub_base, int src_i) { return __shfl(v, sub_base + src_i, kWave); }

    // Sums valid in the group's last lane (three chains interleaved, see wave_sum3); counts in every lane of the group.
    static __device__ __forceinline__ void invalidate(Ctx&) {}
    static __device__ __forceinline__ bool aborted(Ctx&) { return false; }
    template <bool GUARD, class C>
    static __device__ __forceinline__ void reduce(const Params&, C& c, Sums& s, const unsigned long long (&pred)[8]) {
        float &a = s.f0, &b = s.f1, &cc = s.f2;
        {
            float& c = cc;
            EVAC_DPP3(0x111, 0xf)
            EVAC_DPP3(0x112, 0xf)
            EVAC_DPP3(0x114, 0xf)
            EVAC_DPP3(0x118, 0xf)
            if constexpr (G == 32) { EVAC_DPP3(0x142, 0xa) }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s.i[k] = count(pred[k], c.gmask);
    }
    template <class C>
    static __device__ __forceinline__ void exit_publish(C&, bool, float, float) {}
    template <class C>
    static __device__ __forceinline__ void exit_fetch(C& c, float gx, float gy, int src_i, float& ex, float& ey) {
        ex = fetch(gx, c.sub * G, src_i);
        ey = fetch(gy, c.sub * G, src_i);
    }

    static __device__ __forceinline__ void neighbour_sum(const Params& p, Ctx& c, const Ped& q, bool efv, bool fv,
                                                         float ux, float uy, float& sx, float& sy) {
        auto& sm = c.sm;
        sync();   // tile readers of the previous step are done
        const unsigned long long m_efv = ballot(efv);
        const int n_cols = count(m_efv, c.gmask);
        {
            const int before = rank(m_efv, c.gmask);
            const int idx = efv ? before : n_cols + (c.li - before);    // a bijection onto the group's G slots
            float* tf = (float*)sm.tile[c.slot] + (idx >> 1) * 8 + (idx & 1);     // columns in pairs, as in Wave<1>::neighbour_sum
            tf[0] = efv ? q.x * kTileScale : __builtin_inff();
            tf[2] = q.y * kTileScale;
            tf[4] = efv ? ux : 0.0f;
            tf[6] = efv ? uy : 0.0f;
        }
        sync();
        sx = 0.0f;
        sy = 0.0f;
        // the loop runs to the largest column count of the wave's groups (wave-uniform); the padding entries
        // of smaller groups weigh 0
        int nmax = 0;
#pragma unroll
        for (int g = 0; g < kEnvsPerWave; ++g) {
            const int k = __popcll((m_efv >> (g * G)) & kGroupBits);
            nmax = k > nmax ? k : nmax;
        }
        const bool any_row = ballot(fv) != 0ull;
        const int n4 = any_row ? ((nmax + 3) & ~3) : 0;
        if (!any_row) {   // no loop: the NaN poisoning it would carry (area.py:118-119), per group
            const unsigned long long m_nan = ballot(ux != ux || uy != uy) & m_efv;
            if ((m_nan & c.gmask) != 0ull) sx = sy = __builtin_nanf("");
        }
        const f4* __restrict__ tile = sm.tile[c.slot];           // per lane: its group's tile
        const float XI = q.x * kTileScale, YI = q.y * kTileScale;
        int j = 0;
        // (the same packed two-column form and the same even / odd partial sums as the one-wave-per-env loop: bit-identical dynamics)
        const f2 P = f2{XI, YI}, r2b2 = f2{kRPed2Big, kRPed2Big};
        f2 sx2 = f2{0.0f, 0.0f}, sy2 = f2{0.0f, 0.0f};
        for (; j + 8 <= n4; j += 8) {
            f4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = tile[j + k];
#pragma unroll
            for (int k = 0; k < 8; k += 2) pair2_accumulate(P, t[k], t[k + 1], r2b2, sx2, sy2);
        }
        for (; j < n4; j += 4) {
            f4 t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = tile[j + k];
#pragma unroll
            for (int k = 0; k < 4; k += 2) pair2_accumulate(P, t[k], t[k + 1], r2b2, sx2, sy2);
        }
        if (any_row) {
            sx = sx2.x + sx2.y;
            sy = sy2.x + sy2.y;
        }
    }
};

}  // namespace evac
