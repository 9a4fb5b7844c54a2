// Device code of libevac: the fused evacuation-env step for gfx950 (CDNA4, wave64).
//
// One env is owned by WPE waves (WPE = 1 for 33 <= N <= 64, else 2/4/8/16 = one workgroup; N <= 32 shares a wave
// between envs, see evac_subwave.h); lane i owns pedestrian i in registers.  The only O(N^2) part -- the
// Vicsek neighbour average, area.py:104-119 of the reference -- reads the moving peers' (x, y, unit heading)
// from a compacted LDS tile with wave-uniform (broadcast) ds_read_b128.  Everything else is O(N) per-lane
// work plus wave reductions (v_cmp ballots + s_bcnt1 for the counts, DPP trees for the float sums).
// No MFMA: there is no dense contraction here (output width 2).
//
// Built with -ffp-contract=off: every fused multiply-add is written explicitly (fmaf), so what is fused is
// a decision of this file, not of the compiler.  Divisions and square roots use the 1-ulp hardware
// v_rcp / v_rsq / v_sqrt (see frcp / frsq / fsqrt); the parity bar is 1e-5 absolute.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/evac.h"

// Profiling-only phase ablation (tools/ablate.sh builds side libraries with -DEVAC_ABLATE=mask; the
// shipped library is always built with 0).  1: no pair loop, 2: no observation epilogue,
// 4: no Philox (constant action / noise), 8: no status/reward reductions, 16: no per-step stores.
#ifndef EVAC_ABLATE
#define EVAC_ABLATE 0
#endif

// Diagnostic build only (-DEVAC_STAMP, tools/stamps.sh): s_memtime stamps around the phases of a step,
// summed per phase over all waves into g_stamps.  No stamp executes in the shipped library.
#ifdef EVAC_STAMP
__device__ unsigned long long g_stamps[16];
#define EVAC_T(k)                                                                         \
    do {                                                                                  \
        unsigned long long now_;                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                                \
        stamp_acc[k] += now_ - stamp_last;                                                \
        stamp_last = now_;                                                                \
    } while (0)
#define EVAC_STAMP_ARGS , stamp_acc, stamp_last
#else
#define EVAC_T(k) do { } while (0)
#define EVAC_STAMP_ARGS
#endif

namespace evac {

constexpr int kViscek = 1, kFollower = 2, kExiting = 3, kEscaped = 4;   // statuses.py:16-27
constexpr float kExitX = 0.0f, kExitY = -1.0f;                           // area.py:39
// constants.py:35-38 (not configurable in the reference either).  Squared radii are rounded from the double
// product.
constexpr float kRLeader2 = (float)(0.2 * 0.2), kRPed2 = (float)(0.1 * 0.1), kRExit = 0.4f, kREscape = 0.01f;
constexpr float kTileScale = 0x1.0p40f;                 // tile coordinates are stored times 2^40 (exact)
constexpr float kRPed2Big = kRPed2 * 0x1.0p80f;        // r_ped^2 * 2^80, exact: the pair test in scaled units
// boolean options packed into Params::flags (one SGPR instead of seven)
constexpr uint32_t kFlagNewExitingReward = 1u, kFlagNewFollowersReward = 2u, kFlagTermOnWall = 4u, kFlagNanGuard = 8u,
                   kFlagClipAction = 16u;
constexpr int kWave = 64;
constexpr int kStageSteps = 7, kGravRow = 9;   // 7 steps x (6 obs + reward + terminated + truncated) = 63 words <= 64 lanes
// native 16-byte vector: loads/stores of it are single ds_read_b128 / ds_write_b128 (HIP's float4 is
// copied member-wise and re-merged only to 8-byte alignment, i.e. ds_read2_b64 at half the LDS rate)
using f4 = float __attribute__((ext_vector_type(4)));

// Philox stream ids (counter word 3)
constexpr uint32_t kStreamNoise = 0x4e4f4953u;   // 'NOIS'
constexpr uint32_t kStreamReset = 0x52455345u;   // 'RESE'
constexpr uint32_t kStreamAction = 0x41435449u;  // 'ACTI'

struct Params {
    int32_t n_envs, n_ped;
    float width, height, step_size, noise_coef, eps;
    float ens, one_minus_ens;
    float init_reward, intrinsic_coef;
    int32_t max_timesteps;
    uint32_t flags;                                 // kFlag*
    float inv_n, inv_200n;                          // 1/N, 1/(200 N)
    int32_t obs_pos, obs_stat, obs_box, obs_dim;
    float alpha, neg_alpha, grav_pow;               // grav_pow = alpha + 2
    int32_t grav_pow_int;                           // alpha+2 if it is an integer in [1,32], else 0
    int32_t small_noise;                            // sin/cos regime: 2 short Taylor, 1 long Taylor, 0 ocml sincosf (noise_sincos)
    uint32_t seed_lo, seed_hi, env_id_offset;
    // bound state
    float4* ped;
    uint8_t* status;
    float4* agent;
    int4* clock;
    float4* acc;
};

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; Random123).  Restated in oracle/philox.py and checked there
// against the Random123 known-answer vectors.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
#ifndef EVAC_NO_KEY_BARRIER
    // Keep the ten round keys from being hoisted out of the caller's loop as 20 live SGPRs (the step loop is
    // already over the scalar-register budget); recomputing them is 20 s_add per call.
    asm volatile("" : "+s"(k0), "+s"(k1));
#endif
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a v_mul_hi_u32 / v_mul_lo_u32 pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// 24-bit uniform in [0,1): exact in f32
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 0x1.0p-24f; }
// U[-1,1): exact in f32 (pedestrians.py:17-18 draws U(-1,1); random_agent.py:8-9 samples Box(-1,1))
__device__ __forceinline__ float usym(uint32_t x) { return 2.0f * u01(x) - 1.0f; }

// ------------------------------------------------------------------------------------------------
// wave-level helpers
// ------------------------------------------------------------------------------------------------
// DPP add step: v + (v moved by `ctrl`), lanes without a source (or in rows masked off) add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// Sums over the 64 lanes, results wave-uniform (SGPRs).  row_shr 1/2/4/8 leave each row's total in its lane 15;
// row_bcast:15 / row_bcast:31 fold the rows into lane 63 (the rocPRIM gfx9 scheme), ~2.5x cheaper than six
// ds_bpermute butterflies (tools/microbench/valu_rates.hip).
// Three sums at once, the three DPP chains interleaved step by step: a DPP source written by the previous
// VALU instruction costs wait states (the compiler pads a single chain with s_nop); with three independent
// chains in lock-step the hazard is covered by real work.
__device__ __forceinline__ void wave_sum3(float& a, float& b, float& c) {
#define EVAC_DPP3(CTRL, MASK) a = dpp_add<CTRL, MASK>(a); b = dpp_add<CTRL, MASK>(b); c = dpp_add<CTRL, MASK>(c);
    EVAC_DPP3(0x111, 0xf)
    EVAC_DPP3(0x112, 0xf)
    EVAC_DPP3(0x114, 0xf)
    EVAC_DPP3(0x118, 0xf)
    EVAC_DPP3(0x142, 0xa)
    EVAC_DPP3(0x143, 0xc)
#undef EVAC_DPP3
    a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), 63));
    b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 63));
    c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63));
}
__device__ __forceinline__ int wave_count(bool p) { return __popcll(__ballot(p)); }

// 1-ulp hardware reciprocal / rsqrt / sqrt (v_rcp_f32, v_rsq_f32, v_sqrt_f32) instead of the ~10
// instruction IEEE division / sqrt sequences: the parity bar is 1e-5, these are ~1e-7 relative.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// Neighbour weight 1.0 if |p_i - p_j|^2 < r^2 else 0.0 without a compare: with coordinates pre-scaled by
// S = 2^40 (exact), r^2 S^2 - DX^2 - DY^2 = (r^2 - d^2) * 2^80 is evaluated by two FMAs, the second saturating
// to [0,1] through the VOP3 clamp modifier.  Any non-zero difference of two f32 numbers near 0.01 is at least
// ~1e-9, times 2^80 it is far above 1, so the result is exactly 1 or 0; an exact tie gives 0 (strict <, as
// distances.py / area.py:107); NaN gives 0 (DX10 clamp); padding entries carry X = +inf -> -inf -> 0.
// Two roundings sit between the true r^2 - d^2 and its sign -- the same tie sensitivity (~1e-9 in d) as
// computing d^2 in f32 at all.  v_cmp + v_cndmask would cost ~3 slots (tools/microbench/valu_rates.hip).
__device__ __forceinline__ float neighbour_weight(float DX, float DY, float r2_big) {
    const float a = fmaf(-DY, DY, r2_big);
    float w;
    asm("v_fma_f32 %0, -%1, %1, %2 clamp" : "=v"(w) : "v"(DX), "v"(a));
    return w;
}

// x^k for a wave-uniform integer k in [1,32]: straight-line binary powering (no loop, no branches;
// the selects take a wave-uniform condition).  A few ulp.
__device__ __forceinline__ float powi(float x, int k) {
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    float r = (k & 1) ? x : 1.0f;
    r *= (k & 2) ? x2 : 1.0f;
    r *= (k & 4) ? x4 : 1.0f;
    if (k & 24) {   // rare: alpha >= 6
        r *= (k & 8) ? x8 : 1.0f;
        r *= (k & 16) ? x16 : 1.0f;
        if (k & 32) r *= x16 * x16;
    }
    return r;
}

// sin/cos of the angular noise eta in [-noise_coef/2, noise_coef/2] (wave-uniform regime choice):
//   |eta| <= 0.2   (noise_coef <= 0.4, the reference's default is 0.2): Taylor to x^5 / x^4, remainder < 3e-9
//   |eta| <= pi/4  : Taylor to x^9 / x^10, remainder < 2e-9 relative
//   otherwise      : ocml sincosf with full range reduction
__device__ __forceinline__ void noise_sincos(float a, int regime, float& s, float& c) {
    if (regime == 2) {
        const float z = a * a;
        float ps = fmaf(z, 8.3333333e-3f, -1.6666667e-1f);
        ps = ps * z;
        s = fmaf(ps, a, a);
        float pc = fmaf(z, 4.1666667e-2f, -0.5f);
        c = fmaf(pc, z, 1.0f);
    } else if (regime == 1) {
        const float z = a * a;
        float ps = fmaf(z, 2.7557319e-6f, -1.9841270e-4f);
        ps = fmaf(ps, z, 8.3333333e-3f);
        ps = fmaf(ps, z, -1.6666667e-1f);
        ps = ps * z;
        s = fmaf(ps, a, a);
        float pc = fmaf(z, -2.7557319e-7f, 2.4801587e-5f);
        pc = fmaf(pc, z, -1.3888889e-3f);
        pc = fmaf(pc, z, 4.1666667e-2f);
        pc = fmaf(pc, z, -0.5f);
        c = fmaf(pc, z, 1.0f);
    } else {
        sincosf(a, &s, &c);
    }
}

// ------------------------------------------------------------------------------------------------
// Per-lane / per-env register state
// ------------------------------------------------------------------------------------------------
struct Ped {
    float x, y, dx, dy;
    int st;   // status code; 0 on lanes beyond n_ped
};
struct Env {
    float ax, ay, adx, ady;           // leader position / direction        area.py:12-30
    int now, n_resets;                // Time.now, reset count               area.py:42-59
    uint32_t total;                   // steps since creation (Philox counter)
    float acc_ret, acc_intr, acc_stat;   // env.py:65-67
};
struct StepOut {
    float reward;
    bool terminated, truncated;
    int n_escaped, n_exiting, n_follower, n_viscek;
    float gx, gy, ex, ey;   // gravity observation of the post-step state (GRAV kernels): ped sums, exit term * n_followers
};

template <int WPE>
struct Geometry {
    static constexpr int kThreadsPerEnv = WPE * kWave;
#ifndef EVAC_BLOCK1
#define EVAC_BLOCK1 256
#endif
    // WPE == 1: several one-wave envs share a workgroup (no workgroup barrier is ever used there);
    // WPE >= 2: exactly one env per workgroup, so that __syncthreads() is a per-env barrier
    static constexpr int kBlock = WPE == 1 ? EVAC_BLOCK1 : kThreadsPerEnv;
    static constexpr int kEnvsPerBlock = kBlock / kThreadsPerEnv;
};

template <int WPE>
struct Smem {
    f4 tile[Geometry<WPE>::kEnvsPerBlock][WPE * kWave];   // (x, y, ux, uy) of every pedestrian
    float redf[Geometry<WPE>::kEnvsPerBlock][WPE][4];
    int cols[Geometry<WPE>::kEnvsPerBlock][WPE];              // moving pedestrians per wave (tile compaction)
    float exitg[Geometry<WPE>::kEnvsPerBlock][2];            // gravity exit term from the lane that computed it (WPE > 1)
    // rollout outputs of up to kStageSteps steps, flushed with ONE 64-lane store (GRAV kernels)
    alignas(16) float stage[Geometry<WPE>::kEnvsPerBlock][kStageSteps][12];   // rows written as two 16-byte vectors + 1 word
    int redi[Geometry<WPE>::kEnvsPerBlock][WPE][8];
};

// Sync the WPE waves of one env.  WPE == 1: a wave is in lock-step; only keep the compiler from
// moving LDS accesses across the point.  WPE > 1: one env per workgroup, so a workgroup barrier.
template <int WPE>
__device__ __forceinline__ void env_sync() {
    if constexpr (WPE == 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

struct Sums {
    float f0, f1, f2;
    int i[8];
};
// Reduce 3 floats and up to 8 predicates over all lanes of the env.  Result in every lane.
template <int WPE>
__device__ __forceinline__ void env_reduce(Smem<WPE>& sm, int slot, int wave_in_env, int lane, Sums& s,
                                           const bool (&pred)[8]) {
    wave_sum3(s.f0, s.f1, s.f2);
#pragma unroll
    for (int k = 0; k < 8; ++k) s.i[k] = wave_count(pred[k]);
    if constexpr (WPE > 1) {
        env_sync<WPE>();   // previous users of redf/redi are done
        if (lane == 0) {
            sm.redf[slot][wave_in_env][0] = s.f0;
            sm.redf[slot][wave_in_env][1] = s.f1;
            sm.redf[slot][wave_in_env][2] = s.f2;
#pragma unroll
            for (int k = 0; k < 8; ++k) sm.redi[slot][wave_in_env][k] = s.i[k];
        }
        env_sync<WPE>();
        s.f0 = s.f1 = s.f2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s.i[k] = 0;
        for (int w = 0; w < WPE; ++w) {   // fixed order: deterministic
            s.f0 += sm.redf[slot][w][0];
            s.f1 += sm.redf[slot][w][1];
            s.f2 += sm.redf[slot][w][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) s.i[k] += sm.redi[slot][w][k];
        }
    }
}

// statuses.py:29-48 -- pure function of the position, the leader position and the exit.
// `de` returns the distance to the exit (reused by the intrinsic reward, distances.py:51-56).
__device__ __forceinline__ int classify(const Params& p, float x, float y, float ax, float ay, float& de,
                                        float& lx, float& ly, float& dl2) {
    lx = x - ax;
    ly = y - ay;
    dl2 = lx * lx + ly * ly;
    const float ex = x - kExitX, ey = y - kExitY;
    de = fsqrt(ex * ex + ey * ey);
    int st = kViscek;
    if (dl2 < kRLeader2) st = kFollower;
    if (de < kRExit) st = kExiting;
    if (de < kREscape) st = kEscaped;
    return st;
}

// ------------------------------------------------------------------------------------------------
// Observation epilogue: env.py:98-104 through the wrapper chain of wrappers/config.py:46-93.
// `viscek_gx/gy`, `n_follower` come from the caller's reduction when positions == grav.
// ------------------------------------------------------------------------------------------------
// gravity_encoding.py:15-16,35-37:  -alpha / (|R| + eps)^(alpha+2) * R, with |R|^2 given
__device__ __forceinline__ void grav_term2(const Params& p, float rx, float ry, float r2, float& gx, float& gy) {
    const float nrm = fsqrt(r2) + p.eps;
    const float pw = p.grav_pow_int ? powi(nrm, p.grav_pow_int) : powf(nrm, p.grav_pow);
    const float c = p.neg_alpha * frcp(pw);
    gx = c * rx;
    gy = c * ry;
}
__device__ __forceinline__ void grav_term(const Params& p, float rx, float ry, float& gx, float& gy) {
    grav_term2(p, rx, ry, rx * rx + ry * ry, gx, gy);
}

// Positions / statuses observations (abs | rel) x (no | ohe | cat) x (Dict | Box): env.py:98-104, wrappers.py:8-96.
// Purely per-lane writes (lane i owns pedestrian row i; lane 0 also writes the agent and exit rows).
__device__ __forceinline__ void write_obs_generic(const Params& p, int i, bool active, const Ped& q, const Env& e,
                                                  float* __restrict__ obs) {
    const bool rel = p.obs_pos == EVAC_POS_REL;
    const float ihyp = 0.70710678118f;                                // wrappers.py:12-18: 1/sqrt(1+1)
    float px = q.x, py = q.y, ex = kExitX, ey = kExitY;
    if (rel) {                                                        // wrappers.py:20-27
        px = (q.x - e.ax) * ihyp;
        py = (q.y - e.ay) * ihyp;
        ex = (kExitX - e.ax) * ihyp;
        ey = (kExitY - e.ay) * ihyp;
    }
    const int code = 4 - q.st;                                        // wrappers.py:49
    if (p.obs_box) {                                                  // wrappers.py:77-96
        const int C = p.obs_stat == EVAC_STAT_OHE ? 6 : (p.obs_stat == EVAC_STAT_CAT ? 3 : 2);
        if (i == 0) {
            obs[0] = e.ax;
            obs[1] = e.ay;
            obs[C + 0] = ex;
            obs[C + 1] = ey;
            if (p.obs_stat == EVAC_STAT_OHE) {
                obs[2] = obs[3] = obs[4] = obs[5] = 0.0f;
                obs[C + 2] = 1.0f;
                obs[C + 3] = obs[C + 4] = obs[C + 5] = 0.0f;
            } else if (p.obs_stat == EVAC_STAT_CAT) {
                obs[2] = 0.0f;
                obs[C + 2] = 1.0f;
            }
        }
        if (active) {
            float* row = obs + (size_t)(i + 2) * C;
            row[0] = px;
            row[1] = py;
            if (p.obs_stat == EVAC_STAT_OHE) {
                row[2] = code == 0 ? 1.0f : 0.0f;
                row[3] = code == 1 ? 1.0f : 0.0f;
                row[4] = code == 2 ? 1.0f : 0.0f;
                row[5] = code == 3 ? 1.0f : 0.0f;
            } else if (p.obs_stat == EVAC_STAT_CAT) {
                row[2] = (float)code * 0.25f;
            }
        }
        return;
    }
    // Dict, flattened in gymnasium key order: agent, exit, pedestrians_positions, pedestrians_statuses
    const int N = p.n_ped;
    if (i == 0) {
        obs[0] = e.ax;
        obs[1] = e.ay;
        obs[2] = ex;
        obs[3] = ey;
    }
    if (active) {
        obs[4 + 2 * i] = px;
        obs[5 + 2 * i] = py;
        float* st = obs + 4 + 2 * N;
        if (p.obs_stat == EVAC_STAT_OHE) {                            // wrappers.py:50-54
            st[4 * i + 0] = code == 0 ? 1.0f : 0.0f;
            st[4 * i + 1] = code == 1 ? 1.0f : 0.0f;
            st[4 * i + 2] = code == 2 ? 1.0f : 0.0f;
            st[4 * i + 3] = code == 3 ? 1.0f : 0.0f;
        } else if (p.obs_stat == EVAC_STAT_CAT) {                     // wrappers.py:55-56
            st[i] = (float)code * 0.25f;
        }
    }
}

// Gravity observation of the CURRENT state by a full reduction: used by reset / observe and after an
// in-kernel autoreset (the per-step path gets the same numbers fused into step_env's reduction).
// o6 = [agent(2), grad_potential_exit(2), grad_potential_pedestrians(2)], wave-uniform.
template <int WPE>
__device__ __forceinline__ void grav_observation(const Params& p, Smem<WPE>& sm, int slot, int wave_in_env, int lane,
                                                 bool active, const Ped& q, const Env& e, float (&o6)[6]) {
    Sums s{};
    float gx = 0.0f, gy = 0.0f;
    const bool visc = active && q.st == kViscek;
    grav_term(p, e.ax - q.x, e.ay - q.y, gx, gy);                   // gravity_encoding.py:8-25
    s.f0 = visc ? gx : 0.0f;
    s.f1 = visc ? gy : 0.0f;
    s.f2 = 0.0f;
    const bool pred[8] = {active && q.st == kFollower, false, false, false, false, false, false, false};
    env_reduce<WPE>(sm, slot, wave_in_env, lane, s, pred);
    float ex, ey;
    grav_term(p, e.ax - kExitX, e.ay - kExitY, ex, ey);              // gravity_encoding.py:28-38
    const float nf = (float)s.i[0];
    o6[0] = e.ax; o6[1] = e.ay; o6[2] = ex * nf; o6[3] = ey * nf; o6[4] = s.f0; o6[5] = s.f1;
}

template <int WPE, bool GRAV>
__device__ __forceinline__ void write_obs(const Params& p, Smem<WPE>& sm, int slot, int wave_in_env, int lane,
                                          int i, bool active, const Ped& q, const Env& e, float* __restrict__ obs) {
    if constexpr (GRAV) {
        float o6[6];
        grav_observation<WPE>(p, sm, slot, wave_in_env, lane, active, q, e, o6);
        if (i == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) obs[k] = o6[k];
        }
        return;
    }
    write_obs_generic(p, i, active, q, e, obs);
}

// ------------------------------------------------------------------------------------------------
// EvacuationEnv.reset: env.py:129-137, pedestrians.py:16-27, area.py:27-30, 49-51.
// `draw` = the four U(-1,1) numbers of this pedestrian (pos.x, pos.y, dir.x, dir.y).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void reset_env(const Params& p, bool active, float4 draw, Ped& q, Env& e) {
    e.ax = e.ay = e.adx = e.ady = 0.0f;
    e.now = 0;
    e.n_resets += 1;
    e.acc_ret = e.acc_intr = e.acc_stat = 0.0f;
    q.x = draw.x;
    q.y = draw.y;
    const float inrm = frsq(draw.z * draw.z + draw.w * draw.w);        // pedestrians.py:29-31
    q.dx = draw.z * inrm;
    q.dy = draw.w * inrm;
    float de;
    float lx, ly, dl2;
    q.st = active ? classify(p, q.x, q.y, 0.0f, 0.0f, de, lx, ly, dl2) : 0;
}

__device__ __forceinline__ float4 philox_reset_draw(const Params& p, uint32_t env_gid, int i, int n_resets) {
    const uint4 r = philox4x32_10(make_uint4(env_gid, (uint32_t)i, (uint32_t)n_resets, kStreamReset), p.seed_lo, p.seed_hi);
    return make_float4(usym(r.x), usym(r.y), usym(r.z), usym(r.w));
}
__device__ __forceinline__ float philox_noise(const Params& p, uint32_t env_gid, int i, uint32_t total) {
    const uint4 r = philox4x32_10(make_uint4(env_gid, (uint32_t)i, total >> 2, kStreamNoise), p.seed_lo, p.seed_hi);
    const uint32_t sel = total & 3u;
    const uint32_t w = sel == 0 ? r.x : (sel == 1 ? r.y : (sel == 2 ? r.z : r.w));
    return (u01(w) - 0.5f) * p.noise_coef;                             // area.py:124: U(-c/2, c/2)
}
__device__ __forceinline__ float2 philox_action(const Params& p, uint32_t env_gid, uint32_t total) {
    const uint4 r = philox4x32_10(make_uint4(env_gid, 0u, total, kStreamAction), p.seed_lo, p.seed_hi);
    return make_float2(usym(r.x), usym(r.y));
}

// One (i, j) pair of the neighbour sum: 2 subtractions, 2 FMAs for the 0/1 weight, 2 FMAs (packed by the
// compiler) for the heading sum.  (XI, YI) and t.x, t.y are the 2^40-scaled coordinates.
__device__ __forceinline__ void pair_accumulate(float XI, float YI, f4 t, float r2b, float& sx, float& sy) {
    const float w = neighbour_weight(XI - t.x, YI - t.y, r2b);
    sx = fmaf(w, t.z, sx);
    sy = fmaf(w, t.w, sy);
}

// ------------------------------------------------------------------------------------------------
// One env step: env.py:141-171.  All lanes of the env call this together.
// ------------------------------------------------------------------------------------------------
// area.py:189-192: a /= |a| + eps ; agent.direction = step_size * a
__device__ __forceinline__ float2 agent_direction(const Params& p, float act_x, float act_y) {
    if (p.flags & kFlagClipAction) {                                  // gym.wrappers.ClipAction (rpo_agent.py:27), wave-uniform
        act_x = __builtin_amdgcn_fmed3f(act_x, -1.0f, 1.0f);
        act_y = __builtin_amdgcn_fmed3f(act_y, -1.0f, 1.0f);
    }
    const float inrm = frcp(fsqrt(act_x * act_x + act_y * act_y) + p.eps);   // area.py:190
    return make_float2(p.step_size * (act_x * inrm), p.step_size * (act_y * inrm));
}

template <int WPE, bool GRAV>
__device__ __forceinline__ void step_env(const Params& p, Smem<WPE>& sm, int slot, int wave_in_env, int lane, int i,
                                         bool active, Ped& q, Env& e, float2 adir, float noise, StepOut& out
#ifdef EVAC_STAMP
                                         , unsigned long long (&stamp_acc)[16], unsigned long long& stamp_last
#endif
) {
    // The body is branch-free: every lane runs every instruction and the results are merged with
    // selects.  Divergent `if` blocks cost s_and_saveexec / s_cbranch pairs and fence the scheduler;
    // with 4 waves per SIMD at C2 the per-wave instruction stream is what bounds the step.

    // ---- Time.step: area.py:53-59 ----
    e.now += 1;
    e.total += 1u;
    out.truncated = e.now >= p.max_timesteps;

    // ---- Area.agent_step: area.py:182-210 (wave-uniform, every lane computes the same values) ----
    e.adx = adir.x;                                                         // area.py:192
    e.ady = adir.y;
    const float tx = e.ax + e.adx, ty = e.ay + e.ady;                       // area.py:201
    const bool hit = fabsf(tx) > p.width || fabsf(ty) > p.height;           // area.py:203-206: < -W or > W
    e.ax = hit ? e.ax : tx;                                                 // area.py:195
    e.ay = hit ? e.ay : ty;
    const float r_agent = hit ? -5.0f : 0.0f;                               // area.py:198
    const bool term_agent = hit && (p.flags & kFlagTermOnWall) != 0;

    // ---- Area.pedestrians_step: area.py:76-180 ----
    const int old_st = q.st;
    const bool esc = q.st == kEscaped, exi = q.st == kExiting;
    q.x = esc ? kExitX : q.x;                                               // area.py:79-81
    q.y = esc ? kExitY : q.y;
    q.dx = esc ? 0.0f : q.dx;
    q.dy = esc ? 0.0f : q.dy;
    if (__ballot(exi) != 0ull) {                                            // area.py:84-90 (area.py:85 `if any(exiting)`)
        const float vx = kExitX - q.x, vy = kExitY - q.y;
        const float l2 = vx * vx + vy * vy;
        const float il = frsq(l2);
        const float ln = l2 * il;                                           // |v|
        const float sz = ln > p.step_size ? p.step_size : ln;
        const float k = il * sz;                                            // (v / |v|) * min(|v|, step)
        q.dx = exi ? vx * k : q.dx;
        q.dy = exi ? vy * k : q.dy;
    }
    // lanes beyond n_ped carry status 0, so status tests need no `active &&` (saves mask algebra on the SALU)
    const bool efv = (unsigned)(q.st - kViscek) < 3u;                       // area.py:99  (V | F | E) = codes 1..3
    const bool fv = (unsigned)(q.st - kViscek) < 2u;                        // area.py:104 (V | F) = codes 1..2
    const bool fol = q.st == kFollower;

    // unit headings of the moving pedestrians: area.py:100-101.  0 * rsq(0) = 0 * inf = NaN, as 0/0.
    // A NaN heading is written to the tile as it is: w * NaN = NaN even for w = 0, so it poisons every
    // pedestrian's sum -- exactly the reference's (intersection * u).sum() with NaN * 0 = NaN
    // (area.py:118-119).  nan_guard (non-reference) zeroes it instead.
    const float inrm = frsq(q.dx * q.dx + q.dy * q.dy);
    float ux = q.dx * inrm, uy = q.dy * inrm;
    if (p.flags & kFlagNanGuard) {          // wave-uniform
        ux = (ux != ux) ? 0.0f : ux;
        uy = (uy != uy) ? 0.0f : uy;
    }
    EVAC_T(1);   // leader + per-lane pre-pair work
    env_sync<WPE>();   // tile readers of the previous step are done
    // The tile holds the moving pedestrians first, compacted in ascending pedestrian order -- the columns
    // pos[efv] of the reference's distance matrix (area.py:99-106) -- then the others as padding with
    // weight 0 (X = +inf) and heading 0.  Every lane writes exactly one entry.  Under a
    // RandomAgent most pedestrians have escaped by mid-episode, so the all-pairs loop shrinks from N to
    // n_efv iterations.
    int n_cols;
    {
        const unsigned long long m = __ballot(efv);
        int before = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        n_cols = __popcll(m);
        if constexpr (WPE > 1) {
            if (lane == 0) sm.cols[slot][wave_in_env] = n_cols;
            __syncthreads();
            int tot = 0, base = 0;
#pragma unroll
            for (int w2 = 0; w2 < WPE; ++w2) {
                const int c = sm.cols[slot][w2];
                base += (w2 < wave_in_env) ? c : 0;
                tot += c;
            }
            n_cols = tot;
            before += base;                                                 // moving pedestrians before this one
        }
        const int tid = wave_in_env * kWave + lane;
        const int idx = efv ? before : n_cols + (tid - before);             // a bijection onto [0, WPE*64)
        sm.tile[slot][idx] = f4{efv ? q.x * kTileScale : __builtin_inff(), q.y * kTileScale, efv ? ux : 0.0f, efv ? uy : 0.0f};
    }
    env_sync<WPE>();   // tile complete

    // ---- all-pairs neighbour sum: area.py:105-119.  The count n_intersections only rescales the
    // mean heading, which arctan2 ignores; it is not needed.
    EVAC_T(2);   // tile write + poison vote
    // rows of the distance matrix exist only for FOLLOWER/VISCEK pedestrians (area.py:104)
    bool any_fv;
    if constexpr (WPE == 1) any_fv = __ballot(fv) != 0ull;
    else any_fv = true;   // (a workgroup-wide OR would cost a barrier; the loop is short when n_cols is)
    float sx = 0.0f, sy = 0.0f;
    {
        // Branch-free, 8 peers per batch: the 8 wave-uniform ds_read_b128 broadcasts are issued back to
        // back (LDS latency paid once per batch, no VALU slot), then 7 full-rate VALU ops per pair.
        const f4* __restrict__ tile = sm.tile[slot];
        const int n8 = __builtin_amdgcn_readfirstlane(any_fv ? ((n_cols + 3) & ~3) : 0);   // batches of 8 (+ a half batch); no rows -> no loop
        const float r2b = kRPed2Big;
        const float XI = q.x * kTileScale, YI = q.y * kTileScale;
        // peers per LDS round trip: 16 where registers allow (1-wave kernels: 3.31 vs 3.34 us at 8, 3.49 at 4),
        // 8 in the multi-wave kernels, 4 in the 1024-thread one whose workgroup size caps it at 128 VGPRs
        constexpr int B = WPE <= 2 ? 16 : (WPE == 16 ? 4 : 8);
        int j = 0;
        if constexpr (!(EVAC_ABLATE & 1)) {
            for (; j + B <= n8; j += B) {      // full batches
                f4 t[B];
#pragma unroll
                for (int k = 0; k < B; ++k) t[k] = tile[j + k];
#pragma unroll
                for (int k = 0; k < B; ++k) pair_accumulate(XI, YI, t[k], r2b, sx, sy);
            }
            for (; j < n8; j += 4) {           // remainder in groups of 4 (n8 is a multiple of 4)
                f4 t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = tile[j + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) pair_accumulate(XI, YI, t[k], r2b, sx, sy);
            }
        }
    }
    EVAC_T(3);   // all-pairs loop

    // ---- new heading = mean heading rotated by the noise: area.py:120-136.
    // cos/sin(arctan2(my,mx)+eta) = rotation of (mx,my)/|m| by eta; arctan2(0,0) = 0.
    {
        const bool zero_mean = sx == 0.0f && sy == 0.0f;
        const float il = frsq(sx * sx + sy * sy);
        const float cx = zero_mean ? 1.0f : sx * il;
        const float cy = zero_mean ? 0.0f : sy * il;
        float sn, cs;
        noise_sincos(noise, p.small_noise, sn, cs);
        const float ndx = (cx * cs - cy * sn) * p.step_size;
        const float ndy = (cy * cs + cx * sn) * p.step_size;
        q.dx = fv ? ndx : q.dx;                                             // area.py:136
        q.dy = fv ? ndy : q.dy;
        const float bdx = p.ens * e.adx + p.one_minus_ens * q.dx;           // area.py:139-142
        const float bdy = p.ens * e.ady + p.one_minus_ens * q.dy;
        q.dx = fol ? bdx : q.dx;
        q.dy = fol ? bdy : q.dy;
        q.x += efv ? q.dx : 0.0f;                                           // area.py:145
        q.y += efv ? q.dy : 0.0f;
    }
    {   // area.py:148-152.  med3 of a NaN is not NaN, but then miss = NaN - finite = NaN, so the position,
        // the `miss != 0` test and the flipped direction come out exactly as with np.clip.
        const float cx = __builtin_amdgcn_fmed3f(q.x, -p.width, p.width);
        const float cy = __builtin_amdgcn_fmed3f(q.y, -p.height, p.height);
        const float mx = q.x - cx, my = q.y - cy;
        q.x = fmaf(-2.0f, mx, q.x);                                         // 2*miss is exact: same rounding as pos - 2*miss
        q.y = fmaf(-2.0f, my, q.y);
        q.dx = (mx != 0.0f) ? -q.dx : q.dx;
        q.dy = (my != 0.0f) ? -q.dy : q.dy;
    }

    EVAC_T(4);   // heading, blend, move, reflect
    // ---- statuses, rewards, termination: area.py:155-178, statuses.py:29-48, reward.py:19-47 ----
    // The first idle lane (i == N, if the env does not fill its waves) stands on the exit: it evaluates the
    // gravity exit term (gravity_encoding.py:28-38) with the very same instructions as the pedestrians'
    // terms instead of a separate single-lane block.  Its classifier result is discarded (inactive).
    const bool exit_lane = GRAV && i == p.n_ped;
    const float px = exit_lane ? kExitX : q.x, py = exit_lane ? kExitY : q.y;
    float de, lx, ly, dl2;
    const int cls = classify(p, px, py, e.ax, e.ay, de, lx, ly, dl2);
    const int new_st = active ? cls : 0;
    q.st = new_st;
    Sums s{};
    s.f0 = active ? de : 0.0f;
    // gravity observation of the post-step state, fused into the same reduction (gravity_encoding.py:8-25);
    // R = agent - pos = -(pos - agent) reuses the classifier's offset and squared distance.
    float gx = 0.0f, gy = 0.0f;
    if constexpr (GRAV) {
        grav_term2(p, -lx, -ly, dl2, gx, gy);
        const bool visc = new_st == kViscek;
        s.f1 = visc ? gx : 0.0f;
        s.f2 = visc ? gy : 0.0f;
    }
    // per-step counts: the two reward transitions, escaped (termination) and followers (gravity exit term);
    // exiting / viscek counts are only part of the episode record and are taken at episode end.
    const bool pred[8] = {
        (old_st == kViscek || old_st == kFollower) && new_st == kExiting,    // reward.py:35-39
        old_st == kViscek && new_st == kFollower,                             // reward.py:43-46
        new_st == kEscaped, false, new_st == kFollower, false, false, false};
    float ex = 0.0f, ey = 0.0f;
    if constexpr (GRAV && WPE > 1) {
        if (exit_lane) {
            sm.exitg[slot][0] = gx;
            sm.exitg[slot][1] = gy;
        }
    }
    if constexpr (!(EVAC_ABLATE & 8)) env_reduce<WPE>(sm, slot, wave_in_env, lane, s, pred);
    if constexpr (GRAV) {
        if (p.n_ped < WPE * kWave) {          // wave-uniform
            if constexpr (WPE == 1) {
                ex = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gx), p.n_ped));
                ey = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gy), p.n_ped));
            } else {
                ex = sm.exitg[slot][0];       // written before env_reduce's barriers
                ey = sm.exitg[slot][1];
            }
        } else {                              // the env fills its waves: no idle lane
            grav_term(p, e.ax - kExitX, e.ay - kExitY, ex, ey);
        }
        const float nf = (float)s.i[4];
        out.ex = ex * nf;
        out.ey = ey * nf;
        out.gx = s.f1;
        out.gy = s.f2;
    }
    EVAC_T(5);   // classify + reductions
    out.n_escaped = s.i[2];
    out.n_follower = s.i[4];
    out.n_exiting = out.n_viscek = 0;   // filled by finish_counts() when the episode ends

    const float tf = 1.0f - (float)e.now * p.inv_200n;                      // reward.py:26
    float r_ped = p.init_reward;
    if ((p.flags & kFlagNewExitingReward) && s.i[0]) r_ped += (15.0f + 10.0f * tf) * (float)s.i[0];      // uniform branches:
    if ((p.flags & kFlagNewFollowersReward) && s.i[1]) r_ped += (10.0f + 5.0f * tf) * (float)s.i[1];     // usually no transition
    const float intrinsic = 0.0f - s.f0 * p.inv_n;                          // reward.py:19-21
    out.reward = r_agent + r_ped + p.intrinsic_coef * intrinsic;           // env.py:158
    out.terminated = term_agent || (s.i[2] == p.n_ped);                     // area.py:175-178, env.py:171
    e.acc_ret += out.reward;                                                // env.py:168-170
    e.acc_intr += intrinsic;
    e.acc_stat += r_agent + r_ped;
    EVAC_T(6);   // rewards, flags
}

// ------------------------------------------------------------------------------------------------
// state <-> HBM
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_env(const Params& p, int env, int i, bool active, Ped& q, Env& e) {
    const float4 a = p.agent[env];
    const int4 c = p.clock[env];
    const float4 k = p.acc[env];
    e.ax = a.x; e.ay = a.y; e.adx = a.z; e.ady = a.w;
    e.now = c.x; e.n_resets = c.y; e.total = (uint32_t)c.z;
    e.acc_ret = k.x; e.acc_intr = k.y; e.acc_stat = k.z;
    if (active) {
        const float4 v = p.ped[(size_t)env * p.n_ped + i];
        q.x = v.x; q.y = v.y; q.dx = v.z; q.dy = v.w;
        q.st = p.status[(size_t)env * p.n_ped + i];
    } else {
        q.x = q.y = q.dx = q.dy = 0.0f;
        q.st = 0;
    }
}
__device__ __forceinline__ void store_env(const Params& p, int env, int i, bool active, const Ped& q, const Env& e) {
    if (active) {
        p.ped[(size_t)env * p.n_ped + i] = make_float4(q.x, q.y, q.dx, q.dy);
        p.status[(size_t)env * p.n_ped + i] = (uint8_t)q.st;
    }
    if (i == 0) {
        p.agent[env] = make_float4(e.ax, e.ay, e.adx, e.ady);
        p.clock[env] = make_int4(e.now, e.n_resets, (int)e.total, 0);
        p.acc[env] = make_float4(e.acc_ret, e.acc_intr, e.acc_stat, 0.0f);
    }
}

// exiting / viscek counts of the final state for the episode record (env.py:120-123); called by all lanes
// of the env when an episode ends (rare), so the per-step reduction does not carry them.
template <int WPE>
__device__ __forceinline__ void finish_counts(Smem<WPE>& sm, int slot, int wave_in_env, int lane, const Ped& q, StepOut& o) {
    Sums s{};
    const bool pred[8] = {q.st == kExiting, q.st == kViscek, false, false, false, false, false, false};
    env_reduce<WPE>(sm, slot, wave_in_env, lane, s, pred);
    o.n_exiting = s.i[0];
    o.n_viscek = s.i[1];
}
__device__ __forceinline__ void write_stats(evac_episode_stats_t* dst, const Env& e, const StepOut& o) {
    dst->episode_reward = e.acc_ret;
    dst->episode_length = (float)e.now;
    dst->episode_intrinsic_reward = e.acc_intr;
    dst->episode_status_reward = e.acc_stat;
    dst->escaped_pedestrians = (float)o.n_escaped;
    dst->exiting_pedestrians = (float)o.n_exiting;
    dst->following_pedestrians = (float)o.n_follower;
    dst->viscek_pedestrians = (float)o.n_viscek;
}

// Which env / pedestrian does this thread own?
template <int WPE>
struct Who {
    int env, slot, wave_in_env, lane, i;
    __device__ __forceinline__ Who() {
        const int t = threadIdx.x;
        slot = t / Geometry<WPE>::kThreadsPerEnv;
        const int tin = t - slot * Geometry<WPE>::kThreadsPerEnv;
        wave_in_env = tin / kWave;
        lane = tin & (kWave - 1);
        i = tin;
        env = blockIdx.x * Geometry<WPE>::kEnvsPerBlock + slot;
        if constexpr (WPE == 1) {   // wave-uniform by construction: let the compiler keep it in SGPRs
            env = __builtin_amdgcn_readfirstlane(env);
            slot = __builtin_amdgcn_readfirstlane(slot);
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Kernels.  __launch_bounds__(block, 4): at least 4 waves per SIMD, i.e. at most 128 VGPRs -- the 4-wave kernel
// once grew to 135 and silently lost a quarter of its occupancy (C3: 7.7 -> 8.5 us per step).
// ------------------------------------------------------------------------------------------------
template <int WPE, bool GRAV>
__global__ __launch_bounds__(Geometry<WPE>::kBlock, 4) void k_step(
    Params p, const float2* __restrict__ actions, const float* __restrict__ noise_in, float* __restrict__ obs_out,
    float* __restrict__ reward_out, uint8_t* __restrict__ term_out, uint8_t* __restrict__ trunc_out, int autoreset,
    float* __restrict__ final_obs, evac_episode_stats_t* __restrict__ final_stats) {
    __shared__ Smem<WPE> sm;
    const Who<WPE> w;
    if (w.env >= p.n_envs) return;   // whole waves (WPE == 1) or whole workgroups: no barrier is skipped by a subset
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    const uint32_t gid = p.env_id_offset + (uint32_t)w.env;
    const float2 a = actions[w.env];
    float nz = 0.0f;
    if (active) nz = noise_in ? noise_in[(size_t)w.env * p.n_ped + w.i] : philox_noise(p, gid, w.i, e.total);
    StepOut o;
#ifdef EVAC_STAMP
    unsigned long long stamp_acc[16] = {};
    unsigned long long stamp_last = 0;
#endif
    step_env<WPE, GRAV>(p, sm, w.slot, w.wave_in_env, w.lane, w.i, active, q, e, agent_direction(p, a.x, a.y), nz, o EVAC_STAMP_ARGS);
    const bool done = o.terminated || o.truncated;
    float* obs = obs_out + (size_t)w.env * p.obs_dim;
    float o6[6] = {e.ax, e.ay, o.ex, o.ey, o.gx, o.gy};   // GRAV: the observation came out of step_env's reduction
    if (done && autoreset) {
        if (final_obs) {
            float* fo = final_obs + (size_t)w.env * p.obs_dim;
            if constexpr (GRAV) {
                if (w.i == 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) fo[k] = o6[k];
                }
            } else {
                write_obs<WPE, GRAV>(p, sm, w.slot, w.wave_in_env, w.lane, w.i, active, q, e, fo);
            }
        }
        if (final_stats) {
            finish_counts<WPE>(sm, w.slot, w.wave_in_env, w.lane, q, o);
            if (w.i == 0) write_stats(final_stats + w.env, e, o);
        }
        reset_env(p, active, philox_reset_draw(p, gid, w.i, e.n_resets), q, e);
        if constexpr (GRAV) grav_observation<WPE>(p, sm, w.slot, w.wave_in_env, w.lane, active, q, e, o6);
    }
    if constexpr (GRAV) {
        if (w.i == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) obs[k] = o6[k];
        }
    } else {
        write_obs<WPE, GRAV>(p, sm, w.slot, w.wave_in_env, w.lane, w.i, active, q, e, obs);
    }
    store_env(p, w.env, w.i, active, q, e);
    if (w.i == 0) {
        reward_out[w.env] = o.reward;
        term_out[w.env] = o.terminated ? 1 : 0;
        trunc_out[w.env] = o.truncated ? 1 : 0;
    }
}

// T steps per launch, state in registers (rpo_agent.py:180-203 rollout loop, RandomAgent or given actions).
// Output: ONE packed f32 slab [T][E][D+3] = [obs(D) | reward | terminated | truncated] -- a single message
// for the all-gather and a single coalesced store stream for the kernel.  GRAV kernels stage the 9 words
// of up to 7 steps in LDS and flush them with one 64-lane store (five single-lane stores per step cost a
// third of the step before: profiles/r01_e_*).
template <int WPE, bool GRAV, bool CAPTURE>
__device__ __forceinline__ void rollout_body(
    Smem<WPE>& sm, const Params& p, int n_steps, const float2* __restrict__ actions, float2* __restrict__ actions_out,
    float* __restrict__ slab_out, evac_episode_stats_t* __restrict__ final_stats, int capture_envs,
    float* __restrict__ capture) {
    const Who<WPE> w;
    if (w.env >= p.n_envs) return;
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    const uint32_t gid = p.env_id_offset + (uint32_t)w.env;
    const size_t E = (size_t)p.n_envs;
    const int row = p.obs_dim + 3;
    uint4 nzr = make_uint4(0, 0, 0, 0);
    bool have = false;
    // RandomAgent actions and the leader directions they give (area.py:189-192) are produced 64 steps at a
    // time, one step per LANE (a per-wave scalar Philox would cost ~100 SALU instructions every step),
    // and fetched per step with v_readlane.
    float2 lane_act = make_float2(0.f, 0.f), lane_adir = make_float2(0.f, 0.f);
    // flush mapping of the staged outputs: lane l carries word l % 9 of staged step l / 9
    const int fl_s = w.lane / kGravRow, fl_k = w.lane - fl_s * kGravRow;
    int staged = 0, stage_t0 = 0;
    // Retire the state loads HERE, or their first use inside the loop puts `s_waitcnt vmcnt(0)` -- which
    // also waits for the previous step's stores -- into every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
#ifdef EVAC_STAMP
    unsigned long long stamp_acc[16] = {};
    unsigned long long stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    for (int t = 0; t < n_steps; ++t) {
        const int slot64 = t & 63;
        if (slot64 == 0) {
            if (actions) {
                if (t + w.lane < n_steps) lane_act = actions[(size_t)(t + w.lane) * E + w.env];
            } else if constexpr (EVAC_ABLATE & 4) {
                lane_act = make_float2(0.3f, -0.7f);
            } else {
                lane_act = philox_action(p, gid, e.total + (uint32_t)w.lane);   // e.total grows by exactly 1 per step
            }
            lane_adir = agent_direction(p, lane_act.x, lane_act.y);
        }
        float2 a = make_float2(0.f, 0.f), adir;
        if constexpr (CAPTURE) {
            a.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lane_act.x), slot64));
            a.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lane_act.y), slot64));
        }
        adir.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lane_adir.x), slot64));
        adir.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lane_adir.y), slot64));
        if (CAPTURE && actions_out && w.i == 0) actions_out[(size_t)t * E + w.env] = a;   // diagnostic face only
        // one Philox call serves four consecutive steps of this pedestrian
        const uint32_t sel = e.total & 3u;
        if ((!have || sel == 0u) && !(EVAC_ABLATE & 4)) {
            nzr = philox4x32_10(make_uint4(gid, (uint32_t)w.i, e.total >> 2, kStreamNoise), p.seed_lo, p.seed_hi);
            have = true;
        }
        const uint32_t wsel = sel == 0 ? nzr.x : (sel == 1 ? nzr.y : (sel == 2 ? nzr.z : nzr.w));
        const float nz = (u01(wsel) - 0.5f) * p.noise_coef;
        StepOut o;
        EVAC_T(0);   // action fetch + noise Philox
        step_env<WPE, GRAV>(p, sm, w.slot, w.wave_in_env, w.lane, w.i, active, q, e, adir, nz, o EVAC_STAMP_ARGS);
        // trajectory capture for rendering (Pedestrians.save / Agent.save, pedestrians.py:33-35, area.py:32-33):
        // the post-step, pre-reset state of the first `capture_envs` envs; row N holds the leader.
        if (CAPTURE && capture && w.env < capture_envs) {   // wave-/workgroup-uniform; compiled out of the default kernel
            float* cp = capture + (((size_t)t * capture_envs + w.env) * (p.n_ped + 1)) * 3;
            if (active) {
                cp[3 * w.i + 0] = q.x;
                cp[3 * w.i + 1] = q.y;
                cp[3 * w.i + 2] = (float)q.st;
            }
            if (w.i == 0) {
                cp[3 * p.n_ped + 0] = e.ax;
                cp[3 * p.n_ped + 1] = e.ay;
                cp[3 * p.n_ped + 2] = 0.0f;
            }
        }
        float o6[6] = {e.ax, e.ay, o.ex, o.ey, o.gx, o.gy};
        if (o.terminated || o.truncated) {   // wave-/workgroup-uniform, rare
            if (final_stats) {
                finish_counts<WPE>(sm, w.slot, w.wave_in_env, w.lane, q, o);
                if (w.i == 0) write_stats(final_stats + (size_t)t * E + w.env, e, o);
            }
            reset_env(p, active, philox_reset_draw(p, gid, w.i, e.n_resets), q, e);
            if constexpr (GRAV) grav_observation<WPE>(p, sm, w.slot, w.wave_in_env, w.lane, active, q, e, o6);
        }
        float* rowp = slab_out + ((size_t)t * E + w.env) * row;
        const float f_term = o.terminated ? 1.0f : 0.0f, f_trunc = o.truncated ? 1.0f : 0.0f;
        if constexpr (GRAV) {
            if constexpr (!(EVAC_ABLATE & 16)) {
                if (staged == 0) stage_t0 = t;
                if (w.i == 0) {
                    float* st = sm.stage[w.slot][staged];
                    *(f4*)(st + 0) = f4{o6[0], o6[1], o6[2], o6[3]};
                    *(f4*)(st + 4) = f4{o6[4], o6[5], o.reward, f_term};
                    st[8] = f_trunc;
                }
                ++staged;
                if (staged == kStageSteps || t == n_steps - 1) {
                    if (w.wave_in_env == 0) {   // the wave that staged them: in-order LDS, no barrier needed
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if (fl_s < staged) {
                            const float v = sm.stage[w.slot][fl_s][fl_k];
                            slab_out[((size_t)(stage_t0 + fl_s) * E + w.env) * kGravRow + fl_k] = v;
                        }
                    }
                    staged = 0;
                }
            }
        } else {
            if constexpr (!(EVAC_ABLATE & 2)) write_obs<WPE, GRAV>(p, sm, w.slot, w.wave_in_env, w.lane, w.i, active, q, e, rowp);
            if (w.i == 0 && !(EVAC_ABLATE & 16)) {
                rowp[p.obs_dim + 0] = o.reward;
                rowp[p.obs_dim + 1] = f_term;
                rowp[p.obs_dim + 2] = f_trunc;
            }
        }
        EVAC_T(7);   // autoreset check, observation epilogue, output stores
    }
#ifdef EVAC_STAMP
    if (w.lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_stamps[k], stamp_acc[k]);
#endif
    store_env(p, w.env, w.i, active, q, e);
}

// The default face carries no capture / action-recording code at all; the diagnostic face is used by
// rollout(capture_envs=K) and rollout(record_actions=True).
template <int WPE, bool GRAV>
__global__ __launch_bounds__(Geometry<WPE>::kBlock, 4) void k_rollout(
    Params p, int n_steps, const float2* __restrict__ actions, float* __restrict__ slab_out,
    evac_episode_stats_t* __restrict__ final_stats) {
    __shared__ Smem<WPE> sm;
    rollout_body<WPE, GRAV, false>(sm, p, n_steps, actions, nullptr, slab_out, final_stats, 0, nullptr);
}
template <int WPE, bool GRAV>
__global__ __launch_bounds__(Geometry<WPE>::kBlock, 4) void k_rollout_capture(
    Params p, int n_steps, const float2* __restrict__ actions, float2* __restrict__ actions_out,
    float* __restrict__ slab_out, evac_episode_stats_t* __restrict__ final_stats, int capture_envs,
    float* __restrict__ capture) {
    __shared__ Smem<WPE> sm;
    rollout_body<WPE, GRAV, true>(sm, p, n_steps, actions, actions_out, slab_out, final_stats, capture_envs, capture);
}

template <int WPE, bool GRAV>
__global__ __launch_bounds__(Geometry<WPE>::kBlock, 4) void k_reset(Params p, const uint8_t* __restrict__ mask,
                                                                const float4* __restrict__ draws,
                                                                float* __restrict__ obs_out) {
    __shared__ Smem<WPE> sm;
    const Who<WPE> w;
    if (w.env >= p.n_envs) return;
    if (mask && !mask[w.env]) return;   // per env: uniform over the env's waves
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    const uint32_t gid = p.env_id_offset + (uint32_t)w.env;
    float4 d = make_float4(0.f, 0.f, 1.f, 0.f);
    if (active) d = draws ? draws[(size_t)w.env * p.n_ped + w.i] : philox_reset_draw(p, gid, w.i, e.n_resets);
    reset_env(p, active, d, q, e);
    if (obs_out) write_obs<WPE, GRAV>(p, sm, w.slot, w.wave_in_env, w.lane, w.i, active, q, e, obs_out + (size_t)w.env * p.obs_dim);
    store_env(p, w.env, w.i, active, q, e);
}

template <int WPE, bool GRAV>
__global__ __launch_bounds__(Geometry<WPE>::kBlock, 4) void k_observe(Params p, float* __restrict__ obs_out) {
    __shared__ Smem<WPE> sm;
    const Who<WPE> w;
    if (w.env >= p.n_envs) return;
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    write_obs<WPE, GRAV>(p, sm, w.slot, w.wave_in_env, w.lane, w.i, active, q, e, obs_out + (size_t)w.env * p.obs_dim);
}

// state exchange in the reference's shapes
__global__ void k_get_state(Params p, float2* pos, float2* dir, uint8_t* status, float2* apos, float2* adir, int32_t* now) {
    const size_t n = (size_t)p.n_envs * p.n_ped;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p.ped[k];
        if (pos) pos[k] = make_float2(v.x, v.y);
        if (dir) dir[k] = make_float2(v.z, v.w);
        if (status) status[k] = p.status[k];
        if (k < (size_t)p.n_envs) {
            const float4 a = p.agent[k];
            if (apos) apos[k] = make_float2(a.x, a.y);
            if (adir) adir[k] = make_float2(a.z, a.w);
            if (now) now[k] = p.clock[k].x;
        }
    }
}
__global__ void k_set_state(Params p, const float2* pos, const float2* dir, const uint8_t* status, const float2* apos,
                            const float2* adir, const int32_t* now) {
    const size_t n = (size_t)p.n_envs * p.n_ped;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        float4 v = p.ped[k];
        if (pos) { v.x = pos[k].x; v.y = pos[k].y; }
        if (dir) { v.z = dir[k].x; v.w = dir[k].y; }
        p.ped[k] = v;
        if (status) p.status[k] = status[k];
        if (k < (size_t)p.n_envs) {
            float4 a = p.agent[k];
            if (apos) { a.x = apos[k].x; a.y = apos[k].y; }
            if (adir) { a.z = adir[k].x; a.w = adir[k].y; }
            p.agent[k] = a;
            if (now) {
                int4 c = p.clock[k];
                c.x = now[k];
                p.clock[k] = c;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The trainer's per-env wrapper chain as a device epilogue (rpo_agent.py:24-33): NormalizeObservation,
// clip, NormalizeReward(gamma), clip.  gymnasium's RunningMeanStd update for a batch of one sample,
// float64 like gymnasium's.  norm_state per env: obs_mean[D] | obs_var[D] | obs_count[D] | ret_mean |
// ret_var | ret_count | returns  (the count is replicated per feature so that threads never share a word).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rms_update1(double& mean, double& var, double& count, double x) {
    const double delta = x - mean;
    const double tot = count + 1.0;
    const double new_mean = mean + delta / tot;
    const double m2 = var * count + delta * delta * count / tot;
    mean = new_mean;
    var = m2 / tot;
    count = tot;
}
__device__ __forceinline__ float norm_clip(double x, double mean, double var, double eps, float clip) {
    const double v = (x - mean) / sqrt(var + eps);
    return (float)fmin(fmax(v, -(double)clip), (double)clip);
}

__global__ void k_norm_init(int n_envs, int D, double* __restrict__ st) {
    const int W = 3 * D + 4;
    const size_t n = (size_t)n_envs * W;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(k % W);
        double v = 0.0;                                   // means, returns
        if (c >= D && c < 2 * D) v = 1.0;                 // obs_var
        else if (c >= 2 * D && c < 3 * D) v = 1e-4;       // obs_count (RunningMeanStd epsilon)
        else if (c == 3 * D + 1) v = 1.0;                 // ret_var
        else if (c == 3 * D + 2) v = 1e-4;                // ret_count
        st[k] = v;
    }
}

// One thread per (env, feature) plus one per env for the reward.  On a finished env (same-step autoreset)
// the terminal observation is normalised first (and counted), then the reset observation -- the order in
// which SyncVectorEnv runs the wrapped step() and reset().
__global__ void k_norm_step(int n_envs, int D, float* __restrict__ obs, float* __restrict__ final_obs,
                            float* __restrict__ reward, const uint8_t* __restrict__ terminated,
                            const uint8_t* __restrict__ truncated, const uint8_t* __restrict__ reset_mask,
                            double* __restrict__ st, float gamma, float obs_clip, float reward_clip, float eps,
                            int reset_only) {
    const int W = 3 * D + 4;
    const size_t n = (size_t)n_envs * (D + 1);
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(k / (D + 1)), d = (int)(k % (D + 1));
        double* s = st + (size_t)e * W;
        if (reset_only && reset_mask && !reset_mask[e]) continue;
        if (d < D) {
            double mean = s[d], var = s[D + d], cnt = s[2 * D + d];
            const bool done = !reset_only && ((terminated && terminated[e]) || (truncated && truncated[e]));
            if (done && final_obs) {
                const double x = final_obs[(size_t)e * D + d];
                rms_update1(mean, var, cnt, x);
                final_obs[(size_t)e * D + d] = norm_clip(x, mean, var, eps, obs_clip);
            }
            const double x = obs[(size_t)e * D + d];
            rms_update1(mean, var, cnt, x);
            obs[(size_t)e * D + d] = norm_clip(x, mean, var, eps, obs_clip);
            s[d] = mean; s[D + d] = var; s[2 * D + d] = cnt;
        } else if (!reset_only) {
            double mean = s[3 * D], var = s[3 * D + 1], cnt = s[3 * D + 2], ret = s[3 * D + 3];
            const double r = reward[e];
            ret = ret * (double)gamma * (1.0 - ((terminated && terminated[e]) ? 1.0 : 0.0)) + r;
            rms_update1(mean, var, cnt, ret);
            const double v = r / sqrt(var + (double)eps);
            reward[e] = (float)fmin(fmax(v, -(double)reward_clip), (double)reward_clip);
            s[3 * D] = mean; s[3 * D + 1] = var; s[3 * D + 2] = cnt; s[3 * D + 3] = ret;
        }
    }
}

}  // namespace evac

namespace evac {
// layout guards: the tile and the staging rows are accessed with 16-byte LDS instructions
static_assert(offsetof(Smem<1>, stage) % 16 == 0 && offsetof(Smem<2>, stage) % 16 == 0 && offsetof(Smem<4>, stage) % 16 == 0 &&
              offsetof(Smem<8>, stage) % 16 == 0 && offsetof(Smem<16>, stage) % 16 == 0, "stage rows must be 16-byte aligned");
static_assert(offsetof(Smem<1>, tile) == 0 && alignof(Smem<1>) >= 16 && alignof(Smem<16>) >= 16, "tile must be 16-byte aligned");
}  // namespace evac
