// Device code of libevac, part 1: constants, parameter block, Philox, wave helpers and the per-lane pieces of the
// evacuation-env step that every kernel family shares (gfx950 / CDNA4, wave64).
//
// Lane i of an env owns pedestrian i in registers.  How the lanes of an env are grouped differs per family
// (evac_families.h): 16 or 32 lanes of a wave (small rooms), one wave, or a workgroup of 2..16 waves; the
// arithmetic in this file is identical for all of them, and the ONE step body (step_env, evac_device.h) is
// written against the family interface.
//
// Built with -ffp-contract=off: every fused multiply-add is written explicitly (fmaf), so what is fused is
// a decision of these files, not of the compiler.  Divisions and square roots use the 1-ulp hardware
// v_rcp / v_rsq / v_sqrt (see frcp / frsq / fsqrt); the parity bar is 1e-5 absolute.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/evac.h"

// Profiling-only phase ablation (tools/ablate.sh builds side libraries with -DEVAC_ABLATE=mask; the
// shipped library is always built with 0).  1: no pair loop, 2: no observation epilogue,
// 4: no Philox (constant action / noise), 8: no status/reward reductions, 16: no per-step stores.  (Round 4's mask 32 -- waves
// without a row to evaluate skip their step -- is gone: a skipped env's clock stops, it is never reset, and over a benchmark's
// sweeps the batch drifts into that frozen state; what it timed was another workload, not a bound: DESIGN.md 9.)
#ifndef EVAC_ABLATE
#define EVAC_ABLATE 0
#endif

// Diagnostic build only (-DEVAC_STAMP, tools/stamps.sh): s_memtime stamps around the phases of a step,
// summed per phase over all waves into g_stamps.  No stamp executes in the shipped library.
// Diagnostic build only (-DEVAC_STEP_TIMES, tools/step_times.py): s_memrealtime at the top of every step of the 16 waves of
// workgroup 0 -- how the waves of one CU progress through a launch.  (Separate from EVAC_STAMP: the phase stamps end every wave
// with atomics on 16 shared words, which stall the waves still running and distort exactly this picture.)
#ifdef EVAC_STEP_TIMES
#ifndef EVAC_STEP_TIMES_BLOCK
#define EVAC_STEP_TIMES_BLOCK 0      // (the workgroup whose waves are stamped)
#endif
__device__ unsigned long long g_step_times[16][128];
// ... and, for the waves of workgroups 0 and 100, the 100 MHz clock at kernel entry / at the top of the first step / after the last step /
// in front of the state write-back (+ four marks inside the prologue), for the last 64 launches (tools/launch_edges.py: what a launch boundary is made of)
__device__ unsigned long long g_launch_marks[64][2][16][8];
__device__ unsigned long long g_launch_span[64][256][2];      // ... and entry / exit of wave 0 of EVERY workgroup: the true kernel boundary
#endif
#ifdef EVAC_STAMP
__device__ unsigned long long g_stamps[16];
#ifdef EVAC_STAMP_WAVES
__device__ unsigned long long g_wave_stamps[16][16];
__device__ int g_stamp_block;                 // the workgroup whose waves report (tools/wave_stamps.py sets it to the slowest one of a first pass)
__device__ unsigned long long g_slowest;      // max over workgroups of (lifetime of wave 0 << 20 | workgroup)
#endif
struct StampState {
    unsigned long long acc[16] = {};
    unsigned long long last = 0;
};
#define EVAC_T(c, k)                                                                      \
    do {                                                                                  \
        unsigned long long now_;                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                                \
        (c).stamp.acc[k] += now_ - (c).stamp.last;                                        \
        (c).stamp.last = now_;                                                            \
    } while (0)
#else
#define EVAC_T(c, k) do { } while (0)
#endif

namespace evac {

constexpr int kViscek = 1, kFollower = 2, kExiting = 3, kEscaped = 4;   // statuses.py:16-27
constexpr float kExitX = 0.0f, kExitY = -1.0f;                           // area.py:39
// constants.py:35-38 (not configurable in the reference either).  Squared radii are rounded from the double
// product.
constexpr float kRLeader2 = (float)(0.2 * 0.2), kRPed2 = (float)(0.1 * 0.1), kRExit = 0.4f, kREscape = 0.01f;
constexpr int kTeamExactBatch = 8;                       // integer headings of a room with 513..1024 pedestrians (|h| <= 2^31 / N < 2^22) that sum exactly in f32
constexpr float kTileScale = 0x1.0p40f;                 // tile coordinates are stored times 2^40 (exact)
constexpr float kRPed2Big = kRPed2 * 0x1.0p80f;        // r_ped^2 * 2^80, exact: the pair test in scaled units
// boolean options packed into Params::flags (one SGPR instead of seven)
constexpr uint32_t kFlagNewExitingReward = 1u, kFlagNewFollowersReward = 2u, kFlagTermOnWall = 4u, kFlagNanGuard = 8u,
                   kFlagClipAction = 16u;
constexpr int kWave = 64;
constexpr int kStageSteps = 7, kGravRow = 9;   // 7 steps x (6 obs + reward + terminated + truncated) = 63 words <= 64 lanes
// Cell list of the workgroup-per-env kernels (evac_families.h, Cells): kCellsX x kCellsY cells over the room.
// A cell is at least 0.125 wide (host: evac_create), i.e. wider than the pedestrian radius 0.1 by far more than any
// rounding of the cell index, so two pedestrians within the radius are never more than one cell apart.
constexpr int kCellsX = 16, kCellsY = 16, kCells = kCellsX * kCellsY;
// native 16-byte vector: loads/stores of it are single ds_read_b128 / ds_write_b128 (HIP's float4 is
// copied member-wise and re-merged only to 8-byte alignment, i.e. ds_read2_b64 at half the LDS rate)
using f4 = float __attribute__((ext_vector_type(4)));
using i4 = int __attribute__((ext_vector_type(4)));
using i2 = int __attribute__((ext_vector_type(2)));

// Philox stream ids (counter word 3)
constexpr uint32_t kStreamNoise = 0x4e4f4953u;   // 'NOIS'
constexpr uint32_t kStreamReset = 0x52455345u;   // 'RESE'
constexpr uint32_t kStreamAction = 0x41435449u;  // 'ACTI'

struct Params {
    int32_t n_envs, n_ped;
    // envs per time step of the caller's time-major buffers (slab [T][slab_envs][D+3], episode stats, given actions): n_envs, or the
    // whole batch's when this handle is one PART of it (evac_options_t.parts: the part's pointers are offset to its first env)
    int32_t slab_envs;
    float width, height, step_size, noise_coef, eps;
    float ens, one_minus_ens;
    float init_reward, intrinsic_coef;
    int32_t max_timesteps;
    uint32_t flags;                                 // kFlag*
    float inv_n, inv_200n;                          // 1/N, 1/(200 N)
    int32_t obs_pos, obs_stat, obs_box, obs_dim;
    float alpha, neg_alpha, grav_pow;               // grav_pow = alpha + 2
    int32_t grav_pow_int;                           // alpha+2 if it is an integer in [1,63], else 0
    int32_t small_noise;                            // sin/cos regime: 2 short Taylor, 1 long Taylor, 0 ocml sincosf (noise_sincos)
    uint32_t seed_lo, seed_hi, env_id_offset;
    int32_t fair;                                   // rollouts rotate the wave priorities (launches of one or two rounds: evac_create)
    // cell list (Cells family): cell = (int)((x + cell_ox) * cell_inv_hx) clamped to [0, 15], same in y;
    // head_scale = min(2^23 - 1, (2^31 - 1) / N): unit headings are summed as integers (exact, order-independent)
    float cell_ox, cell_oy, cell_inv_hx, cell_inv_hy, head_scale;
    // bound state
    float4* ped;
    uint8_t* status;
    float4* agent;
    int4* clock;
    float4* acc;
    // exchange areas of the team kernels (evac_team.h, exchange()), inside the caller's workspace: two slot sets of
    // rec [2][E][32] x 16 B and tile [2][E][1024] x 16 B, contiguous, filled with 0xff bytes (= the tag of no round) by the host in front of
    // every team launch; team_err: the handle's host-mapped error word
    void *team_tile, *team_rec;
    unsigned* team_err;
};

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; Random123).  Restated in oracle/philox.py and checked there
// against the Random123 known-answer vectors.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint32_t k0, uint32_t k1) {
    // Keep the ten round keys from being hoisted out of the caller's loop as 20 live SGPRs (the step loop is
    // already over the scalar-register budget); recomputing them is 20 s_add per call.
    asm volatile("" : "+s"(k0), "+s"(k1));
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
// (k * 2^-23 is exact, so the fused form rounds once like 2 * u01 - 1 does: same bits, one instruction less)
__device__ __forceinline__ float usym(uint32_t x) { return __builtin_fmaf((float)(x >> 8), 0x1.0p-23f, -1.0f); }
// u01 - 0.5, likewise
__device__ __forceinline__ float u01_centred(uint32_t x) { return __builtin_fmaf((float)(x >> 8), 0x1.0p-24f, -0.5f); }

// ------------------------------------------------------------------------------------------------
// wave-level helpers
// ------------------------------------------------------------------------------------------------
// DPP add step: v + (v moved by `ctrl`), lanes without a source (or in rows masked off) add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_addi(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
// Sums over the 64 lanes, results wave-uniform (SGPRs).  row_shr 1/2/4/8 leave each row's total in its lane 15;
// row_bcast:15 / row_bcast:31 fold the rows into lane 63 (the rocPRIM gfx9 scheme), ~2.5x cheaper than six
// ds_bpermute butterflies (tools/microbench/valu_rates.hip).
// Three DPP chains interleaved step by step (the sub-wave families' group sums): a DPP source written by the previous VALU
// instruction costs wait states; with three independent chains in lock-step the hazard is covered by real work.
#define EVAC_DPP3(CTRL, MASK) a = dpp_add<CTRL, MASK>(a); b = dpp_add<CTRL, MASK>(b); c = dpp_add<CTRL, MASK>(c);
template <int LANE>
__device__ __forceinline__ float readlane_const(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), LANE));
}
__device__ __forceinline__ void wave_sum3(float& a, float& b, float& c) {
    // gfx950's half / row swaps PACK the three sums into one register before a single DPP chain runs:
    //   v_permlane32_swap a, b : a' = [a.lo | b.lo], b' = [a.hi | b.hi]  ->  a' + b' = 32-lane partial sums [a | b]
    //   the same with (c, 0)                                             ->                                  [c | 0]
    //   v_permlane16_swap of the two (odd rows of the first with even rows of the second), added
    //                                                                   ->  16-lane partial sums, rows [a, c, b, 0]
    //   row_shr 1/2/4/8 on that ONE register: the totals land in lanes 15 (a), 31 (c), 47 (b).
    // 3 swaps + 3 adds + 4 DPP adds + 3 readlanes instead of 18 DPP adds + 3 readlanes (a DPP add occupies the SIMD ~2.5x as
    // long as a plain one: tools/microbench/valu_rates.hip).  A fixed tree, shared by every kernel family and entry point.
    // (inline asm: with the __builtin_amdgcn_permlane*_swap builtins hipcc 7.2 lost the second result of a swap inside the
    // step body -- both adds read the first -- although a small test kernel compiled correctly.  The leading s_nop covers the
    // "VALU write -> v_permlane read" hazard of gfx950 for whatever instruction precedes.)
    float z = 0.0f;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(z));
    float pab = a + b, pcz = c + z;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(pab), "+v"(pcz));
    float s = pab + pcz;
    s = dpp_add<0x111, 0xf>(s);
    s = dpp_add<0x112, 0xf>(s);
    s = dpp_add<0x114, 0xf>(s);
    s = dpp_add<0x118, 0xf>(s);
    a = readlane_const<15>(s);
    c = readlane_const<31>(s);
    b = readlane_const<47>(s);
}
// Two integer sums over the 64 lanes, the totals in LANE 63 (the row_shr / row_bcast scheme of wave_sum3, two chains
// interleaved; the s_nop between the bcast steps is the DPP read-after-VALU-write wait state the third chain covers there).
__device__ __forceinline__ void wave_sum2_int_lane63(int& a, int& b) {
    a = dpp_addi<0x111, 0xf>(a); b = dpp_addi<0x111, 0xf>(b);
    a = dpp_addi<0x112, 0xf>(a); b = dpp_addi<0x112, 0xf>(b);
    a = dpp_addi<0x114, 0xf>(a); b = dpp_addi<0x114, 0xf>(b);
    a = dpp_addi<0x118, 0xf>(a); b = dpp_addi<0x118, 0xf>(b);
    asm volatile(
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}
// Inclusive prefix sum over the 64 lanes (Hillis-Steele inside each 16-lane row with row_shr 1/2/4/8, then the
// row totals carried across with row_bcast:15 / row_bcast:31).
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    v = dpp_addi<0x111, 0xf>(v);
    v = dpp_addi<0x112, 0xf>(v);
    v = dpp_addi<0x114, 0xf>(v);
    v = dpp_addi<0x118, 0xf>(v);
    v = dpp_addi<0x142, 0xa>(v);
    v = dpp_addi<0x143, 0xc>(v);
    return v;
}
// Lane mask of a predicate.  HIP's __ballot(int) first materialises the bool as 0/1 in a VGPR and compares it with 0
// again (v_cndmask + v_cmp per call); the builtin takes the condition's mask as it is.
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Lanes whose value equals K, as ONE v_cmp into a scalar register pair.  ballot(v == K) is that too when the comparison is
// its only user; when the compiler has the same comparison at hand as a lane bool (a select elsewhere) it turns the bool
// into the mask with v_cndmask + v_cmp instead.
template <int K>
__device__ __forceinline__ unsigned long long mask_eq(int v) {
    unsigned long long m;
    asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m) : "n"(K), "v"(v));
    return m;
}
// Set bits of a ballot as a 32-bit value the compiler knows nothing else about: otherwise (float)count is expanded as a
// 64-bit integer conversion (s_lshl_b64 / s_min / s_or / v_cvt / v_ldexp) because ctpop's operand is 64 bits wide.
__device__ __forceinline__ int mask_count(unsigned long long m) {
    int n = __popcll(m);
    asm("" : "+s"(n));        // (not volatile: an unused count still disappears)
    return n;
}
__device__ __forceinline__ int wave_count(bool p) { return mask_count(ballot(p)); }
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

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

// One (i, j) pair of the neighbour sum: 2 subtractions, 2 FMAs for the 0/1 weight, 2 FMAs (packed by the
// compiler) for the heading sum.  (XI, YI) and t.x, t.y are the 2^40-scaled coordinates.
__device__ __forceinline__ void pair_accumulate(float XI, float YI, f4 t, float r2b, float& sx, float& sy) {
    const float w = neighbour_weight(XI - t.x, YI - t.y, r2b);
    sx = fmaf(w, t.z, sx);
    sy = fmaf(w, t.w, sy);
}
// TWO columns of the neighbour sum at once in packed f32 arithmetic (the one-wave-per-env loops): the tile holds the columns
// in pairs, xy = (X_j, X_j+1, Y_j, Y_j+1) and uv = (ux_j, ux_j+1, uy_j, uy_j+1); six v_pk_* instructions do what ten plain ones
// do for two columns -- the same operations per column (subtract, two FMAs for the 0/1 weight, one FMA per heading component),
// with the heading sums kept as (even columns, odd columns) halves that the caller adds at the end.
using f2 = float __attribute__((ext_vector_type(2)));
// `P` = (X_i, Y_i) of the lane's own row in ONE register pair: op_sel broadcasts its low half against (X_j, X_j+1) and its
// high half against (Y_j, Y_j+1) -- no duplicated (X_i, X_i) / (Y_i, Y_i) pairs to build every step.
__device__ __forceinline__ void pair2_accumulate(f2 P, f4 xy, f4 uv, f2 r2b2, f2& sx2, f2& sy2) {
    const f2 xx = __builtin_shufflevector(xy, xy, 0, 1), yy = __builtin_shufflevector(xy, xy, 2, 3);
    f2 dx, dy;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(P), "v"(xx));                 // X_i - (X_j, X_j+1)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(P), "v"(yy));    // Y_i - (Y_j, Y_j+1)
    const f2 a = __builtin_elementwise_fma(-dy, dy, r2b2);
    f2 w;
    asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(w) : "v"(dx), "v"(a));
    sx2 = __builtin_elementwise_fma(w, __builtin_shufflevector(uv, uv, 0, 1), sx2);
    sy2 = __builtin_elementwise_fma(w, __builtin_shufflevector(uv, uv, 2, 3), sy2);
}

// x + y of a packed pair as ONE v_add_f32 on the two halves of the register pair: the empty asm hides one half from the
// vectoriser, which otherwise moves three registers around to use v_pk_add_f32 on two such sums at once.
__device__ __forceinline__ float hsum2(f2 v) {
    float lo = v.x;
    asm("" : "+v"(lo));
    return lo + v.y;
}

// TWO ROWS of one lane against one column in packed f32 arithmetic (the multi-wave all-pairs sweeps, where a lane carries up
// to four rows): (X, X') and (Y, Y') of the two rows against the column's t.x / t.y, broadcast into both halves by op_sel; six
// v_pk_* instructions for two rows instead of ten plain ones, the same operations per row in the same order -- bit-identical
// to two pair_accumulate calls.
__device__ __forceinline__ void pair_accumulate_rows2(f2 X2, f2 Y2, f4 t, f2 r2b2, f2& ax2, f2& ay2) {
    const f2 txy = __builtin_shufflevector(t, t, 0, 1), tzw = __builtin_shufflevector(t, t, 2, 3);
    f2 dx, dy, w;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(X2), "v"(txy));                 // X - t.x
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(Y2), "v"(txy));    // Y - t.y
    const f2 a = __builtin_elementwise_fma(-dy, dy, r2b2);
    asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(w) : "v"(dx), "v"(a));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ax2) : "v"(w), "v"(tzw));                                     // += w * t.z
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(ay2) : "v"(w), "v"(tzw));                      // += w * t.w
}

// ... the first column of a batch: the sums START at w * heading (two packed multiplies where zeroed accumulators and two FMAs
// would be four instructions)
__device__ __forceinline__ void pair_start_rows2(f2 X2, f2 Y2, f4 t, f2 r2b2, f2& ax2, f2& ay2) {
    const f2 txy = __builtin_shufflevector(t, t, 0, 1), tzw = __builtin_shufflevector(t, t, 2, 3);
    f2 dx, dy, w;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(X2), "v"(txy));                 // X - t.x
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(Y2), "v"(txy));    // Y - t.y
    const f2 a = __builtin_elementwise_fma(-dy, dy, r2b2);
    asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(w) : "v"(dx), "v"(a));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(ax2) : "v"(w), "v"(tzw));                                           // = w * t.z
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(ay2) : "v"(w), "v"(tzw));                              // = w * t.w
}

// The same pair with the unit heading stored as integers (heading * Params::head_scale, rounded): the weight's
// bit pattern (0x3f800000 or 0) shifted down is the integer 1 or 0, and the sums are integer multiply-adds
// (v_mad_i32_i24: |heading| < 2^23; N of them fit an int32).  Integer addition is exact, so the sum does not depend on the order in
// which the peers are visited -- which is what lets the cell-list kernels place peers with LDS atomics.
__device__ __forceinline__ void pair_accumulate_int(float XI, float YI, f4 t, float r2b, int& sx, int& sy) {
    const float w = neighbour_weight(XI - t.x, YI - t.y, r2b);
    const int wi = (int)(__builtin_bit_cast(unsigned, w) >> 29);
    const float hz = t.z, hw = t.w;   // (bit_cast straight from a vector element picks element 0 with this compiler)
    sx = __mul24(wi, __float_as_int(hz)) + sx;
    sy = __mul24(wi, __float_as_int(hw)) + sy;
}

// Two ROWS of a lane against one column with integer heading sums (the teams' many-rows sweep): the two 0/1 weights come out
// of four packed instructions (as pair_accumulate_rows2), then one shift and two integer multiply-adds per row -- 10
// instructions where two pair_accumulate_int calls take 14; the same weights, the same integer sums.
__device__ __forceinline__ void pair_accumulate_int_rows2(f2 X2, f2 Y2, f4 t, f2 r2b2, int& ax0, int& ay0, int& ax1, int& ay1) {
    const f2 txy = __builtin_shufflevector(t, t, 0, 1);
    f2 dx, dy, w;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(X2), "v"(txy));                 // X - t.x
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(Y2), "v"(txy));    // Y - t.y
    const f2 a = __builtin_elementwise_fma(-dy, dy, r2b2);
    asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(w) : "v"(dx), "v"(a));
    const float wx = w.x, wy = w.y, hz = t.z, hw = t.w;   // (bit_cast straight from a vector element picks element 0 with this compiler)
    const int w0 = (int)(__builtin_bit_cast(unsigned, wx) >> 29), w1 = (int)(__builtin_bit_cast(unsigned, wy) >> 29);
    ax0 = __mul24(w0, __float_as_int(hz)) + ax0;
    ay0 = __mul24(w0, __float_as_int(hw)) + ay0;
    ax1 = __mul24(w1, __float_as_int(hz)) + ax1;
    ay1 = __mul24(w1, __float_as_int(hw)) + ay1;
}

// x^k for a wave-uniform integer k in [1,63] (k = alpha + 2).  The powers the reference's experiments use
// (alpha = 2, 3, 5: run_scripts/) take a uniform branch to a straight product; anything else binary powering with
// uniform branches -- no per-bit select masks held in scalar registers through the step loop.  A few ulp.
__device__ __forceinline__ float powi(float x, int k) {
    const float x2 = x * x, x4 = x2 * x2;
    if (k == 5) return x4 * x;
    if (k == 4) return x4;
    if (k == 7) return x4 * x2 * x;
    float r = (k & 1) ? x : 1.0f;
    if (k & 2) r *= x2;
    if (k & 4) r *= x4;
    if (k & 56) {   // rare: alpha >= 6
        const float x8 = x4 * x4, x16 = x8 * x8;
        if (k & 8) r *= x8;
        if (k & 16) r *= x16;
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
    uint32_t total;                   // steps since creation (Philox counter; Time.overall_timesteps)
    float acc_ret, acc_intr, acc_stat;   // env.py:65-67
};
struct StepOut {
    float reward;
    bool terminated, truncated, done;     // done = terminated | truncated
    int n_escaped, n_exiting, n_follower, n_viscek;
    float gx, gy, ex, ey;   // gravity observation of the post-step state (GRAV kernels): ped sums, exit term * n_followers
};
struct Sums {
    float f0, f1, f2;
    int i[8];
};

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

// Does this pedestrian's row of the neighbour sum have to be evaluated (see step_env)?  VISCEK always; FOLLOWER unless the
// enslaving blend multiplies its Vicsek heading by exactly 0 (area.py:139-142 with enslaving_degree = 1).
__device__ __forceinline__ bool needs_row(const Params& p, int st) {
    return st == kViscek || (st == kFollower && p.one_minus_ens != 0.0f);
}

// The head of Area.pedestrians_step (area.py:79-101), per pedestrian: escaped pedestrians are pinned to the exit, exiting
// ones head for it, and every moving pedestrian gets its unit heading.  Mutates q (position / direction of escaped and
// exiting pedestrians).  A function of the pedestrian's own state only -- the team kernels evaluate it a second time, on a
// copy of the post-step state, to publish the next step's tile entry together with this step's reduction (evac_team.h).
struct PrePair {
    bool efv, fv, fol, row;   // moves (V | F | E); has a row in the reference (V | F); FOLLOWER; row must be evaluated
    float ux, uy;             // unit heading (NaN for a zero direction, as the reference's 0/0)
};
__device__ __forceinline__ PrePair pre_pair(const Params& p, Ped& q) {
    PrePair r;
    const bool esc = q.st == kEscaped, exi = q.st == kExiting;
    q.x = esc ? kExitX : q.x;                                               // area.py:79-81
    q.y = esc ? kExitY : q.y;
    q.dx = esc ? 0.0f : q.dx;
    q.dy = esc ? 0.0f : q.dy;
    if (ballot(exi) != 0ull) {                                            // area.py:84-90 (area.py:85 `if any(exiting)`)
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
    r.efv = (unsigned)(q.st - kViscek) < 3u;                                // area.py:99  (V | F | E) = codes 1..3
    r.fv = (unsigned)(q.st - kViscek) < 2u;                                 // area.py:104 (V | F) = codes 1..2
    r.fol = q.st == kFollower;
    // Which pedestrians need their row of the distance matrix evaluated.  The reference evaluates FOLLOWER and VISCEK rows
    // (area.py:104) and then blends a follower's new heading as e * leader + (1 - e) * heading (area.py:139-142): with
    // enslaving_degree = 1 -- the reference's default (config.py:32) -- the follower's own Vicsek mean is multiplied by
    // exactly 0, so only the VISCEK rows are evaluated (late in an episode most moving pedestrians are followers:
    // tools/moving_distribution.py).  A follower lane then sees a zero sum -> a finite heading -> times 0; the one way the
    // product is not 0, the reference's NaN poisoning (any NaN heading makes every row NaN, area.py:118-119), is kept
    // by the families (a flag wherever rows are skipped).
    r.row = needs_row(p, q.st);
    // unit headings of the moving pedestrians: area.py:100-101.  0 * rsq(0) = 0 * inf = NaN, as 0/0.
    // A NaN heading reaches every FOLLOWER/VISCEK pedestrian's sum (w * NaN = NaN even for w = 0 in the all-pairs
    // families, a flag in the cell-list and team families) -- exactly the reference's (intersection * u).sum() with
    // NaN * 0 = NaN (area.py:118-119).  nan_guard (non-reference) zeroes it instead.
    const float inrm = frsq(q.dx * q.dx + q.dy * q.dy);
    r.ux = q.dx * inrm;
    r.uy = q.dy * inrm;
    if (p.flags & kFlagNanGuard) {          // uniform
        r.ux = (r.ux != r.ux) ? 0.0f : r.ux;
        r.uy = (r.uy != r.uy) ? 0.0f : r.uy;
    }
    return r;
}

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

// ------------------------------------------------------------------------------------------------
// Running statistics of the trainer's wrapper chain (rpo_agent.py:24-33: NormalizeObservation / NormalizeReward):
// gymnasium's RunningMeanStd update for a batch of one sample, float64 like gymnasium's.
// norm_state per env: obs_mean[D] | obs_var[D] | obs_count[D] | ret_mean | ret_var | ret_count | returns
// (the count is replicated per feature so that threads never share a word).
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
// what evac_step_normalized passes to the step kernel (state == nullptr: plain evac_step)
struct NormArgs {
    double* state;
    float gamma, obs_clip, reward_clip, eps;
};
// Store policies of the observation epilogue: plain, or counted + normalised + clipped in place (one thread owns one
// feature of one env, for the terminal observation and the reset observation alike, so the two updates are ordered).
struct StorePlain {
    float* __restrict__ p;
    __device__ __forceinline__ void operator()(int idx, float v) const { p[idx] = v; }
};
struct StoreNorm {
    float* __restrict__ p;
    double* __restrict__ s;     // norm_state of this env
    int D;
    float eps, clip;
    __device__ __forceinline__ void operator()(int idx, float v) const {
        double mean = s[idx], var = s[D + idx], cnt = s[2 * D + idx];
        rms_update1(mean, var, cnt, (double)v);
        s[idx] = mean; s[D + idx] = var; s[2 * D + idx] = cnt;
        p[idx] = norm_clip((double)v, mean, var, (double)eps, clip);
    }
};

// Positions / statuses observations (abs | rel) x (no | ohe | cat) x (Dict | Box): env.py:98-104, wrappers.py:8-96.
// Purely per-lane writes (lane i owns pedestrian row i; lane 0 also writes the agent and exit rows).
template <class Store>
__device__ __forceinline__ void write_obs_generic(const Params& p, int i, bool active, const Ped& q, const Env& e,
                                                  Store obs) {
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
            obs(0, e.ax);
            obs(1, e.ay);
            obs(C + 0, ex);
            obs(C + 1, ey);
            if (p.obs_stat == EVAC_STAT_OHE) {
                obs(2, 0.0f); obs(3, 0.0f); obs(4, 0.0f); obs(5, 0.0f);
                obs(C + 2, 1.0f);
                obs(C + 3, 0.0f); obs(C + 4, 0.0f); obs(C + 5, 0.0f);
            } else if (p.obs_stat == EVAC_STAT_CAT) {
                obs(2, 0.0f);
                obs(C + 2, 1.0f);
            }
        }
        if (active) {
            const int row = (i + 2) * C;
            obs(row + 0, px);
            obs(row + 1, py);
            if (p.obs_stat == EVAC_STAT_OHE) {
                obs(row + 2, code == 0 ? 1.0f : 0.0f);
                obs(row + 3, code == 1 ? 1.0f : 0.0f);
                obs(row + 4, code == 2 ? 1.0f : 0.0f);
                obs(row + 5, code == 3 ? 1.0f : 0.0f);
            } else if (p.obs_stat == EVAC_STAT_CAT) {
                obs(row + 2, (float)code * 0.25f);
            }
        }
        return;
    }
    // Dict, flattened in gymnasium key order: agent, exit, pedestrians_positions, pedestrians_statuses
    const int N = p.n_ped;
    if (i == 0) {
        obs(0, e.ax);
        obs(1, e.ay);
        obs(2, ex);
        obs(3, ey);
    }
    if (active) {
        obs(4 + 2 * i, px);
        obs(5 + 2 * i, py);
        const int st = 4 + 2 * N;
        if (p.obs_stat == EVAC_STAT_OHE) {                            // wrappers.py:50-54
            obs(st + 4 * i + 0, code == 0 ? 1.0f : 0.0f);
            obs(st + 4 * i + 1, code == 1 ? 1.0f : 0.0f);
            obs(st + 4 * i + 2, code == 2 ? 1.0f : 0.0f);
            obs(st + 4 * i + 3, code == 3 ? 1.0f : 0.0f);
        } else if (p.obs_stat == EVAC_STAT_CAT) {                     // wrappers.py:55-56
            obs(st + i, (float)code * 0.25f);
        }
    }
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
    return u01_centred(w) * p.noise_coef;                              // area.py:124: U(-c/2, c/2)
}
__device__ __forceinline__ float2 philox_action(const Params& p, uint32_t env_gid, uint32_t total) {
    const uint4 r = philox4x32_10(make_uint4(env_gid, 0u, total, kStreamAction), p.seed_lo, p.seed_hi);
    return make_float2(usym(r.x), usym(r.y));
}

// area.py:189-192: a /= |a| + eps ; agent.direction = step_size * a
__device__ __forceinline__ float2 agent_direction(const Params& p, float act_x, float act_y) {
    if (p.flags & kFlagClipAction) {                                  // gym.wrappers.ClipAction (rpo_agent.py:27), wave-uniform
        act_x = __builtin_amdgcn_fmed3f(act_x, -1.0f, 1.0f);
        act_y = __builtin_amdgcn_fmed3f(act_y, -1.0f, 1.0f);
    }
    const float inrm = frcp(fsqrt(act_x * act_x + act_y * act_y) + p.eps);   // area.py:190
    return make_float2(p.step_size * (act_x * inrm), p.step_size * (act_y * inrm));
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
// `owner`: the lane that writes the per-env words
__device__ __forceinline__ void store_env(const Params& p, int env, int i, bool active, bool owner, const Ped& q, const Env& e) {
    if (active) {
        p.ped[(size_t)env * p.n_ped + i] = make_float4(q.x, q.y, q.dx, q.dy);
        p.status[(size_t)env * p.n_ped + i] = (uint8_t)q.st;
    }
    if (owner) {
        p.agent[env] = make_float4(e.ax, e.ay, e.adx, e.ady);
        p.clock[env] = make_int4(e.now, e.n_resets, (int)e.total, 0);
        p.acc[env] = make_float4(e.acc_ret, e.acc_intr, e.acc_stat, 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// CHAINED rollout launches (evac_options_t.chain; rollout_body<..., CHAIN>): consecutive launches of one handle go to two
// hardware queues alternately and OVERLAP -- a wave of launch g + 1 starts as soon as ITS env's state of launch g is in memory,
// not when launch g's slowest env is done (the in-order queue's kernel boundary).  Launches that overlap share no kernel
// boundary, so nothing flushes or invalidates caches between them, and the XCDs' L2s are not coherent with each other: the
// state travels through an EXCHANGE RECORD per env in the caller's workspace, by the hand-off form measured valid on gfx950
// without fences (MI355X_MICROARCH.md, inter-workgroup visibility: every store of the bytes `sc1` -- write-through, the line is
// dropped from the writer's L2 --, each 128-byte line written WHOLE by one store instruction of one wave, the storing wave's
// `s_waitcnt vmcnt(0)`, then its signal; every load of the bytes a `global_load ... sc1` to registers; nothing on the scalar
// path).  The caller's own state arrays cannot serve: neighbouring envs share their lines (960 bytes of pedestrians, 60 status
// bytes per env), and a line that two waves on two XCDs each write a part of came back wrong in 6 % of the envs
// (tools/chain_debug.py on the first version).  A record's reader is also its next writer, and an `sc1` store drops the line,
// so no L2 keeps a copy across launches.
//   record(env), 12 lines of 128 bytes:  [0, 1024) (x, y, dx, dy) of lane i as one dwordx4 per lane (8 lines, one instruction)
//                                         [1024, 1280) status of lane i as a dword (2 lines, one instruction)
//                                         [1280, 1408) leader | clock | episode sums, 16 bytes each, by lanes 0..7 (1 line)
//                                         [1408, 1536) dword 0: the GENERATION of the record (a line of its own: no false sharing)
// Launch g waits for generation g, and publishes g + 1 behind its record stores.  k_chain_import fills the records from the
// caller's arrays when a chain (re)starts; k_chain_export writes them back when the caller's stream joins (evac_join).
// ------------------------------------------------------------------------------------------------
// (the trailing s_nop: a store of more than 64 bits reads its data registers for a cycle or two after it issues, and the compiler, which
// does not know that this asm is a store, may let the next vector instruction overwrite them -- cdna_hip_programming.md 5.7; seen here as
// records that came back wrong whenever a build's register allocation happened to reuse the data registers at once)
__device__ __forceinline__ void store_dev(void* ptr, f4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory"); }
__device__ __forceinline__ void store_dev_i32(void* ptr, int v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory"); }
__device__ __forceinline__ void load_dev(f4& v, const void* ptr) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(ptr) : "memory"); }
__device__ __forceinline__ void load_dev_i32(int& v, const void* ptr) { asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(ptr) : "memory"); }
__device__ __forceinline__ void wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int load_dev_i32_now(const void* ptr) {      // (issued and waited for: a poll)
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
// (T = lanes per env: 64 for one-wave envs -- the layout above --, 256 for four-wave envs: 16 T bytes of pedestrians, 4 T of statuses,
// one line of env words, one line for the generation)
template <int T>
struct Xchg {
    static constexpr int kStatus = 16 * T, kEnv = 20 * T, kGen = 20 * T + 128, kBytes = 20 * T + 256;
};
constexpr int xchg_bytes(int lanes_per_env) { return 20 * lanes_per_env + 256; }
struct ChainArgs {
    char* xchg;          // [E] exchange records (kXchgBytes each, 128-byte aligned)
    int gen;             // this launch: waits for generation `gen`, publishes `gen + 1`
    unsigned* abort;     // device word (workspace, a line of its own): a wait of some launch timed out -- every later wait gives up at once
    unsigned* err;       // the handle's host-mapped error word (as the teams')
    int deal_mode;       // diagnostic (EVAC_CHAIN_DEAL): how workgroup 0 deals the launch after next (schedule_slot); 0 in the product
    unsigned long long* started;   // workgroups of the chain's launches that have started (since the chain's last restart): what the
                                   // NEXT launch's queue waits for before it dispatches (evac_api.hip, "the invariant of the chain")
    int resume;                    // persistent kernels: 1 = every env takes up at ITS OWN next command (the word the kernel before left for it)
    int stop_at;                   // persistent kernels: the index of a STOP command that is known to be in the ring (the finisher of a join), else INT_MAX
};
constexpr int kChainMaxPolls = 1 << 19;     // bounded wait: ~0.3 s of polls with the back-off below; a launch that gives up voids the run
// Wait until the env's record holds generation `gen`.  Wave-uniform (all lanes poll the same word).  false: timed out.
template <int T>
__device__ __forceinline__ bool chain_wait(const ChainArgs& ch, int env) {
    const char* addr = ch.xchg + (size_t)env * Xchg<T>::kBytes + Xchg<T>::kGen;
    int pause = 0;
    for (int polls = 0; polls < kChainMaxPolls; ++polls) {
        const int v = __builtin_amdgcn_readfirstlane(load_dev_i32_now(addr));
        if (v == ch.gen) return true;
        if ((polls & 255) == 255 && __builtin_amdgcn_readfirstlane(load_dev_i32_now(ch.abort)) != 0) return false;
        // back off: the first polls come fast (a hand-off in flight); a long wait -- a workgroup dispatched well before its
        // heavy env is done -- must not keep the fabric busy (many pollers slow everybody's round trips: evac_team.h)
        if (pause < 4) __builtin_amdgcn_s_sleep(2); else if (pause < 16) __builtin_amdgcn_s_sleep(6); else __builtin_amdgcn_s_sleep(20);
        pause += 1;
    }
    return false;
}
// (the abort line keeps what the last wave to give up waited for: word 1 the generation, word 2 the env -- diagnostics of a run that
// is void anyway.  Kept this small on purpose: a version that also carried the last value polled out of chain_wait -- one more live
// scalar through the prologue -- left the kernel, instruction for instruction the same in its step loop, 15-45 % slower in its steps
// in every run (tools/chain_rhythm.py; profiles/r06_*_chain_bisect.txt): this kernel's register allocation is at its limits.)
__device__ __forceinline__ void chain_give_up(const ChainArgs& ch, int lane, int env) {
    if (lane == 0) {
        store_dev_i32(ch.abort + 1, ch.gen);
        store_dev_i32(ch.abort + 2, env);
        store_dev_i32(ch.abort, 1);
        __hip_atomic_store(ch.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// the env's record -> registers (every lane loads its pedestrian `idx`; the per-env words come from lanes 0..2 of the env-word line)
template <int T>
__device__ __forceinline__ void load_record(const char* rec, int idx, int lane, bool active, Ped& q, Env& e) {
    f4 v, w;
    int st;
    load_dev(v, rec + idx * 16);
    load_dev_i32(st, rec + Xchg<T>::kStatus + idx * 4);
    load_dev(w, rec + Xchg<T>::kEnv + (lane & 7) * 16);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v), "+v"(w), "+v"(st)::"memory");
    const float wx = w.x, wy = w.y, wz = w.z, ww = w.w;
    e.ax = readlane_const<0>(wx); e.ay = readlane_const<0>(wy); e.adx = readlane_const<0>(wz); e.ady = readlane_const<0>(ww);
    e.now = __builtin_bit_cast(int, readlane_const<1>(wx)); e.n_resets = __builtin_bit_cast(int, readlane_const<1>(wy));
    e.total = __builtin_bit_cast(uint32_t, readlane_const<1>(wz));
    e.acc_ret = readlane_const<2>(wx); e.acc_intr = readlane_const<2>(wy); e.acc_stat = readlane_const<2>(wz);
    q.x = active ? v.x : 0.0f; q.y = active ? v.y : 0.0f; q.dx = active ? v.z : 0.0f; q.dy = active ? v.w : 0.0f;
    q.st = active ? st : 0;
}
// registers -> the env's record: three store instructions, every line written whole by one of them (all 64 lanes of a wave store:
// 8 lines of pedestrians, 2 of statuses; `words`: this wave also writes the line of env words -- the env's first wave)
template <int T>
__device__ __forceinline__ void store_record(char* rec, int idx, int lane, bool words, bool active, const Ped& q, const Env& e) {
    store_dev(rec + idx * 16, active ? f4{q.x, q.y, q.dx, q.dy} : f4{0.0f, 0.0f, 0.0f, 0.0f});
    store_dev_i32(rec + Xchg<T>::kStatus + idx * 4, active ? q.st : 0);
    if (words && lane < 8) {
        const f4 a = f4{e.ax, e.ay, e.adx, e.ady};
        const f4 c = f4{__builtin_bit_cast(float, e.now), __builtin_bit_cast(float, e.n_resets), __builtin_bit_cast(float, (int)e.total), 0.0f};
        const f4 k = f4{e.acc_ret, e.acc_intr, e.acc_stat, 0.0f};
        store_dev(rec + Xchg<T>::kEnv + lane * 16, lane == 0 ? a : (lane == 1 ? c : (lane == 2 ? k : f4{0.0f, 0.0f, 0.0f, 0.0f})));
    }
}

// ---- ONE PERSISTENT KERNEL PER JOIN (evac_options_t.chain = 2): the rollout kernel stays resident between evac_rollout calls and takes
// every call as a COMMAND from a ring in uncached device memory that the host writes through the PCIe BAR (no stream operation, no
// kernel: nothing that would need a CU the resident kernel holds).  A command is one 64-byte segment: the payload, then -- behind a
// store fence -- its sequence number in the same segment; a wave reads the segment with ONE load instruction (four lanes x 16 bytes;
// a 64-byte segment is never seen torn: tools/microbench/seg_atomicity.hip) and takes the command when the sequence number is the one it
// waits for.  n_steps = 0 is STOP: the waves store their state and the kernel ends (evac_join).
struct PersistCmd {
    unsigned long long slab, stats, actions;     // the call's output / input pointers (stats, actions: 0 = none)
    int n_steps;                                 // 0: stop
    int pad_[8];
    unsigned seq;                                // written LAST (the segment's 16th word)
};
static_assert(sizeof(PersistCmd) == 64, "one 64-byte segment");
constexpr int kPersistRing = 1024;               // commands per ring (the host joins before it would lap the kernel)
constexpr int kPersistIdlePolls = 256;           // a wave that finds no command for this many polls (~150 us) LEAVES: it stores its env's state and the
                                                 // index of the command it was waiting for, and the kernel ends when all its waves have left -- a
                                                 // resident kernel never outlives the caller's attention (a host that waits for the device, another
                                                 // kernel that needs the CUs), and the next evac_rollout / evac_join starts a kernel that takes up
                                                 // every env where it stopped (ChainArgs.resume)
// (behind the ring, in the same allocation: a 128-byte line of diagnostics, then next_cmd[E] -- per env, the index of the first command it
// has NOT run; written by a leaving wave, read by the waves of a kernel started with resume = 1)
__device__ __forceinline__ int* persist_next_cmd(const char* ring) { return (int*)(ring + (size_t)kPersistRing * 64 + 128); }
// TEAMS (one env on K workgroups): the members read the ring each for itself, and must not disagree on whether command idx came in time
// -- one that runs it would wait in the team's exchange for one that has left.  decision[env] (behind next_cmd[E], zeroed before every
// kernel) holds the team's verdict for the r-th command of this kernel (r = 1, 2, ...): 2 r + 1 = run it, 2 r = leave before it; the first
// member to have an outcome of its own makes it the team's (compare-and-swap from the verdict of command r - 1, which was "run"), the
// others take it: a member that found nothing keeps reading until the command it was told to run shows, one that found the command after
// the team has left discards it (it stays in the ring for the kernel that resumes).
__device__ __forceinline__ int* persist_decision(const char* ring, int n_envs) { return persist_next_cmd(ring) + n_envs; }
__device__ __forceinline__ int persist_team_decide(int* word, int r, bool mine_run) {
    int expected = r == 1 ? 0 : 2 * (r - 1) + 1;
    const int want = 2 * r + (mine_run ? 1 : 0);
    if (__hip_atomic_compare_exchange_strong(word, &expected, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return want;
    return expected;          // (what another member decided: 2 r or 2 r + 1)
}
// Wait for command `idx` (sequence number idx + 1) of the ring.  Wave-uniform results.  false: none came within the idle bound.
__device__ __forceinline__ bool persist_wait(const char* ring, int idx, int lane, int& n_steps, unsigned long long& slab,
                                             unsigned long long& stats) {
    const char* addr = ring + (size_t)(idx & (kPersistRing - 1)) * 64 + (lane & 3) * 16;
    for (int polls = 0; polls < kPersistIdlePolls; ++polls) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        const unsigned seq = (unsigned)__builtin_bit_cast(int, readlane_const<3>(v.w));
        if (seq == (unsigned)idx + 1u) {
            const unsigned lo0 = (unsigned)__builtin_bit_cast(int, readlane_const<0>(v.x)), hi0 = (unsigned)__builtin_bit_cast(int, readlane_const<0>(v.y));
            const unsigned lo1 = (unsigned)__builtin_bit_cast(int, readlane_const<0>(v.z)), hi1 = (unsigned)__builtin_bit_cast(int, readlane_const<0>(v.w));
            slab = ((unsigned long long)hi0 << 32) | lo0;
            stats = ((unsigned long long)hi1 << 32) | lo1;
            n_steps = __builtin_bit_cast(int, readlane_const<1>(v.z));
            return true;
        }
        if (polls < 8) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(4);
    }
    return false;
}

// the episode record: env.py:115-125 (nine keys) + Time.n_episodes
__device__ __forceinline__ void write_stats(evac_episode_stats_t* dst, const Env& e, const StepOut& o) {
    dst->episode_reward = e.acc_ret;
    dst->episode_length = (float)e.now;
    dst->episode_intrinsic_reward = e.acc_intr;
    dst->episode_status_reward = e.acc_stat;
    dst->escaped_pedestrians = (float)o.n_escaped;
    dst->exiting_pedestrians = (float)o.n_exiting;
    dst->following_pedestrians = (float)o.n_follower;
    dst->viscek_pedestrians = (float)o.n_viscek;
    dst->overall_timesteps = (int32_t)e.total;
    dst->n_episodes = e.n_resets;
}

}  // namespace evac
