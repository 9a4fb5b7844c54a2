// Device code of libevac, part 4: the TEAM family -- one env of 513..1024 pedestrians spread over K workgroups on K
// compute units (K = 2, 4 or 8), for batches that would otherwise leave most of the chip idle (BASELINE config 5 runs 32
// envs of 1024 pedestrians per GPU: with one workgroup per env 224 of 256 CUs have nothing to do, and mid-episode, when the
// crowd has flocked into one corner of the room, the neighbour sum of ONE env is ~350 k true pairs -- tools/row_lengths.py).
//
// Member k of a team owns pedestrians [k P, (k+1) P), P = 1024 / K, in the lanes of its first P / 64 waves ("ped waves");
// all 16 waves of the workgroup share the member's part of the pair work.  Per step the members meet twice through global
// memory (device-scope write-through stores, device-scope loads, a counter per team; no cache flush or invalidation:
// tools/microbench/team_barrier.hip measures 1.5 us per publish -> barrier -> gather round of 16 KiB among 8 CUs):
//   1. neighbour sum: every member publishes its moving pedestrians (position x 2^40, integer heading), compacted inside
//      its segment, and their count; after the barrier every member gathers all segments into its own LDS tile -- the
//      columns of the distance matrix.  The rows are the member's own pedestrians that need one (step_env: needs_row),
//      compacted, two per lane and pass; each of the 16 waves takes 1/16 of the columns (wave-uniform ds_read_b128
//      broadcasts) and the 16 partial sums of a row meet in LDS.  Heading sums are INTEGERS (pair_accumulate_int), so the
//      result does not depend on how the pairs were split -- it is bit-identical to the cell-list kernel's (Cells<16>).
//   2. reduction: every ped wave publishes the same 32-byte record as a wave of Cells<16> does through LDS; after the barrier
//      every wave folds the 16 records with the same DPP tree, so rewards / observations / flags are bit-identical too and
//      every member takes the same decisions (autoreset, termination) without further talk.
// Everything else is the common step body (step_env) and rollout scaffolding (rollout_body); waves without pedestrians
// ("helper" waves) skip the per-pedestrian arithmetic.
//
// The members of a team must be resident together (they spin on the team's counter): the host launches the kernel
// cooperatively, with at most one workgroup per CU, and places a team on ONE XCD (workgroup ids congruent mod 8 share an
// XCD -- round-robin dispatch; nothing depends on it but the latency).  Spins are bounded: a team that lost a member sets
// a sticky abort flag and runs to the end without waiting (the results of that launch are void, err[0] tells the host).
#pragma once

#include "evac_device.h"

namespace evac {

__device__ __forceinline__ void store_dev(void* ptr, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory"); }
__device__ __forceinline__ void store_dev(void* ptr, i2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory"); }
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int K_>
struct Team {
    static constexpr int K = K_;
    static_assert(K == 2 || K == 4 || K == 8, "a team has 2, 4 or 8 members");
    static constexpr int WPE = 16;                       // waves per workgroup
    static constexpr bool kEnvUniform = true, kPace = false, kHelpers = true, kExitLane = false;
    static constexpr int kBlock = 1024, kThreadsPerEnv = 1024, kEnvsPerBlock = 1;
    static constexpr int P = 1024 / K;                   // pedestrians per member
    static constexpr int PW = P / kWave;                 // ped waves per member
    static constexpr int kPad = 8;
    static constexpr const char* kName = K == 8 ? "8 CUs/env, all pairs over the team's tile" : (K == 4 ? "4 CUs/env, all pairs over the team's tile" : "2 CUs/env, all pairs over the team's tile");

    struct Smem {
        f4 tile[1024 + kPad];                 // the team's moving pedestrians: (X, Y, heading x, heading y as integers)
        float2 rowpos[P];                     // this member's rows, compacted
        i2 part[WPE][P];                      // partial heading sums [column share][row slot]
        int cols[PW], rows[PW], nans[PW];     // per ped wave: moving pedestrians, needed rows, NaN headings
        int abort;                            // sticky: a barrier timed out
        alignas(16) float stage[1][kStageSteps][12];
    };

    struct Ctx {
        using Family = Team<K_>;
        Smem& sm;
        int env, slot, wave_in_env, lane, i, member, wave;
        bool owner, helper;
        unsigned round = 0;                   // barrier rounds of this launch so far
        int par_tile = 0, par_rec = 0;        // double buffering of the exchange areas
#ifdef EVAC_STAMP
        StampState stamp;
#endif
        __device__ __forceinline__ explicit Ctx(Smem& s) : sm(s) {
            // workgroup b = j * 8 + xcd: team (j / K) * 8 + xcd, member j % K -- the K members of a team share an XCD
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
            env = (j / K) * 8 + xcd;
            member = j % K;
            const int t = threadIdx.x;
            wave = __builtin_amdgcn_readfirstlane(t / kWave);
            lane = t & (kWave - 1);
            helper = wave >= PW;
            slot = 0;
            i = helper ? (1 << 20) : member * P + t;                  // helper lanes own nobody
            wave_in_env = helper ? (1 << 10) + wave : member * PW + wave;   // ped waves: the wave index Cells<16> would have
            owner = i == 0;
        }
    };

    static __device__ __forceinline__ void sync() { __syncthreads(); }
    static __device__ __forceinline__ void invalidate(Ctx&) {}
    static __device__ __forceinline__ void init(Ctx& c) {
        if (threadIdx.x == 0) c.sm.abort = 0;
        __syncthreads();
    }

    // publish -> barrier: every thread's device-scope stores are acknowledged, then the workgroup arrives at the team's
    // counter and waits until all K members have.
    static __device__ __forceinline__ void team_round(const Params& p, Ctx& c) {
        wait_vmem();
        __syncthreads();
        c.round += 1u;
        if (threadIdx.x == 0) {
            unsigned* ctr = p.team_ctr + (size_t)c.env * 32;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!c.sm.abort) {
                const unsigned target = c.round * (unsigned)K;
                int spins = 0;
                while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && ++spins < (1 << 21))
                    __builtin_amdgcn_s_sleep(1);
                if (spins >= (1 << 21)) {
                    c.sm.abort = 1;
                    __hip_atomic_store(p.team_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
    }

    template <bool GUARD, class C>
    static __device__ __forceinline__ void reduce(const Params& p, C& c, Sums& s, const unsigned long long (&pred)[8]) {
        wave_sum3(s.f0, s.f1, s.f2);
#pragma unroll
        for (int k = 0; k < 8; ++k) s.i[k] = mask_count(pred[k]);
        const int par = c.par_rec;
        c.par_rec = par ^ 1;
        f4* rec = (f4*)p.team_rec + ((size_t)par * p.n_envs + c.env) * (2 * WPE);
        if (!c.helper && c.lane == 0) {
            store_dev(rec + 2 * c.wave_in_env, f4{s.f0, s.f1, s.f2, 0.0f});
            const i4 ri = i4{s.i[0] | (s.i[1] << 16), s.i[2] | (s.i[3] << 16), s.i[4] | (s.i[5] << 16), s.i[6] | (s.i[7] << 16)};
            store_dev(rec + 2 * c.wave_in_env + 1, __builtin_bit_cast(f4, ri));
        }
        team_round(p, c);
        const int w = c.lane < WPE ? c.lane : WPE - 1;
        f4 rf, rb;
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(rf), "=&v"(rb) : "v"(rec + 2 * w) : "memory");
        i4 ri = __builtin_bit_cast(i4, rb);
        // the fold of Wave<16>::reduce, instruction for instruction: same tree, same rounding
#define EVAC_RED_STEP(CTRL)                                                                                      \
    rf.x = dpp_add<CTRL, 0xf>(rf.x); rf.y = dpp_add<CTRL, 0xf>(rf.y); rf.z = dpp_add<CTRL, 0xf>(rf.z);           \
    ri.x = dpp_addi<CTRL, 0xf>(ri.x); ri.y = dpp_addi<CTRL, 0xf>(ri.y); ri.z = dpp_addi<CTRL, 0xf>(ri.z);        \
    ri.w = dpp_addi<CTRL, 0xf>(ri.w);
        EVAC_RED_STEP(0x111)
        EVAC_RED_STEP(0x112)
        EVAC_RED_STEP(0x114)
        EVAC_RED_STEP(0x118)
#undef EVAC_RED_STEP
        s.f0 = readlane_f(rf.x, WPE - 1);
        s.f1 = readlane_f(rf.y, WPE - 1);
        s.f2 = readlane_f(rf.z, WPE - 1);
        const int a = __builtin_amdgcn_readlane(ri.x, WPE - 1), b = __builtin_amdgcn_readlane(ri.y, WPE - 1);
        const int d = __builtin_amdgcn_readlane(ri.z, WPE - 1), g = __builtin_amdgcn_readlane(ri.w, WPE - 1);
        s.i[0] = a & 0xffff; s.i[1] = a >> 16;
        s.i[2] = b & 0xffff; s.i[3] = b >> 16;
        s.i[4] = d & 0xffff; s.i[5] = d >> 16;
        s.i[6] = g & 0xffff; s.i[7] = g >> 16;
    }
    template <class C>
    static __device__ __forceinline__ void exit_publish(C&, bool, float, float) {}
    template <class C>
    static __device__ __forceinline__ void exit_fetch(C&, float, float, int, float& ex, float& ey) { ex = ey = 0.0f; }

    static __device__ __forceinline__ void neighbour_sum(const Params& p, Ctx& c, const Ped& q, bool efv, bool row,
                                                         float ux, float uy, float& sx, float& sy) {
        auto& sm = c.sm;
        // ---- 1. compaction inside the member: moving pedestrians (columns) and needed rows ----
        const unsigned long long m_col = ballot(efv), m_row = ballot(row);
        int col_rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m_col >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_col, 0u));
        int row_rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m_row >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_row, 0u));
        if (!c.helper && c.lane == 0) {
            sm.cols[c.wave] = __popcll(m_col);
            sm.rows[c.wave] = __popcll(m_row);
            sm.nans[c.wave] = __popcll(ballot(efv && (ux != ux || uy != uy)));
        }
        __syncthreads();
        int n_member = 0, n_rows = 0, n_nan_member = 0;
#pragma unroll
        for (int w2 = 0; w2 < PW; ++w2) {
            const int kc = sm.cols[w2], kr = sm.rows[w2];
            col_rank += (w2 < c.wave) ? kc : 0;
            row_rank += (w2 < c.wave) ? kr : 0;
            n_member += kc;
            n_rows += kr;
            n_nan_member += sm.nans[w2];
        }
        // ---- 2. publish the member's segment ----
        const int par = c.par_tile;
        c.par_tile = par ^ 1;
        f4* gtile = (f4*)p.team_tile + ((size_t)par * p.n_envs + c.env) * 1024;
        i2* gcnt = (i2*)p.team_cnt + ((size_t)par * p.n_envs + c.env) * 8;
        const float X = q.x * kTileScale, Y = q.y * kTileScale;
        if (efv) {
            const float hs = p.head_scale;
            const int hx = (int)__builtin_rintf(ux * hs), hy = (int)__builtin_rintf(uy * hs);
            store_dev(gtile + c.member * P + col_rank, f4{X, Y, __builtin_bit_cast(float, hx), __builtin_bit_cast(float, hy)});
        }
        if (threadIdx.x == 0) store_dev(gcnt + c.member, i2{n_member, n_nan_member});
        if (row) sm.rowpos[row_rank] = make_float2(X, Y);
        team_round(p, c);
        // ---- 3. gather the team's tile ----
        int n_cols = 0, n_nan = 0;
        {
            i4 cv[4] = {};                       // (count, NaN headings) of two members per vector; loads and their wait in ONE statement
            if constexpr (K == 8)
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                             "global_load_dwordx4 %2, %4, off offset:32 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:48 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[2]), "=&v"(cv[3]) : "v"(gcnt) : "memory");
            else if constexpr (K == 4)
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(cv[0]), "=&v"(cv[1]) : "v"(gcnt) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(cv[0]) : "v"(gcnt) : "memory");
            const int t = threadIdx.x, j = t / P, e = t - j * P;     // thread -> (member j, entry e of its segment)
            int off = 0, cnt_j = 0;
#pragma unroll
            for (int h = 0; h < K / 2; ++h) {
                const int c0 = cv[h].x, c1 = cv[h].z;
                off += (2 * h < j ? c0 : 0) + (2 * h + 1 < j ? c1 : 0);
                cnt_j = 2 * h == j ? c0 : (2 * h + 1 == j ? c1 : cnt_j);
                n_cols += c0 + c1;
                n_nan += cv[h].y + cv[h].w;
            }
            if (e < cnt_j) {
                f4 v;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(gtile + j * P + e) : "memory");
                sm.tile[off + e] = v;
            }
            if (t < kPad) sm.tile[n_cols + t] = f4{__builtin_inff(), 0.0f, 0.0f, 0.0f};
        }
        __syncthreads();
        EVAC_T(c, 2);   // compaction + exchange
        // ---- 4. the member's rows against the tile: two rows per lane and pass, 1/16 of the columns per wave ----
        if constexpr (!(EVAC_ABLATE & 1)) {
            if (n_rows > 0) {
                const int groups = (n_cols + 3) >> 2;
                const int per = (groups + WPE - 1) / WPE;
                const int jbeg = __builtin_amdgcn_readfirstlane(c.wave * per * 4);
                const int jend = __builtin_amdgcn_readfirstlane(min((c.wave + 1) * per, groups) * 4);
                const f4* __restrict__ tile = sm.tile;
                for (int r0 = 0; r0 < n_rows; r0 += 2 * kWave) {
                    const float2 ra = sm.rowpos[r0 + c.lane], rb = sm.rowpos[min(r0 + kWave + c.lane, P - 1)];
                    int ax0 = 0, ay0 = 0, ax1 = 0, ay1 = 0;
                    const bool two = r0 + kWave < n_rows;              // uniform: a second row per lane in this pass
                    if (two) {
                        for (int j = jbeg; j < jend; j += 4) {
                            f4 t[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) t[k] = tile[j + k];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                pair_accumulate_int(ra.x, ra.y, t[k], kRPed2Big, ax0, ay0);
                                pair_accumulate_int(rb.x, rb.y, t[k], kRPed2Big, ax1, ay1);
                            }
                        }
                    } else {
                        for (int j = jbeg; j < jend; j += 4) {
                            f4 t[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) t[k] = tile[j + k];
#pragma unroll
                            for (int k = 0; k < 4; ++k) pair_accumulate_int(ra.x, ra.y, t[k], kRPed2Big, ax0, ay0);
                        }
                    }
                    sm.part[c.wave][r0 + c.lane] = i2{ax0, ay0};
                    if (two) sm.part[c.wave][r0 + kWave + c.lane] = i2{ax1, ay1};
                }
            }
        }
        __syncthreads();
        // ---- 5. the 16 partial sums of a row (integers: any order) ----
        int tx = 0, ty = 0;
        if (row) {
#pragma unroll
            for (int w2 = 0; w2 < WPE; ++w2) {
                const i2 v = sm.part[w2][row_rank];
                tx += v.x;
                ty += v.y;
            }
        }
        sx = row ? (float)tx : 0.0f;
        sy = row ? (float)ty : 0.0f;
        if (n_nan != 0) sx = sy = __builtin_nanf("");
    }
};

}  // namespace evac
