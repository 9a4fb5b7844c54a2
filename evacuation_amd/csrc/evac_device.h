// Device code of libevac, part 3: THE step body (one for every kernel family), the observation / reset epilogues
// and the kernels of the wave- and workgroup-per-env families (the sub-wave kernels are in evac_subwave.h).
//
// The only O(N^2) part -- the Vicsek neighbour average, area.py:104-119 of the reference -- is the family's
// neighbour_sum (evac_families.h).  Everything else is O(N) per-lane work plus reductions over the env's lanes
// (v_cmp ballots + s_bcnt1 for the counts, DPP trees for the float sums).
// No MFMA: there is no dense contraction here (output width 2).
#pragma once

#include <type_traits>

#include "evac_common.h"
#include "evac_families.h"


namespace evac {

// ------------------------------------------------------------------------------------------------
// Observation epilogue: env.py:98-104 through the wrapper chain of wrappers/config.py:46-93.
// ------------------------------------------------------------------------------------------------
// Gravity observation of the CURRENT state by a full reduction: used by reset / observe and after an
// in-kernel autoreset (the per-step path gets the same numbers fused into step_env's reduction).
// o6 = [agent(2), grad_potential_exit(2), grad_potential_pedestrians(2)], valid in the env's owner lane.
template <class F>
__device__ __forceinline__ void grav_observation(const Params& p, typename F::Ctx& c, bool active, const Ped& q,
                                                 const Env& e, float (&o6)[6]) {
    Sums s{};
    float gx = 0.0f, gy = 0.0f;
    const bool visc = active && q.st == kViscek;
    grav_term(p, e.ax - q.x, e.ay - q.y, gx, gy);                   // gravity_encoding.py:8-25
    s.f0 = visc ? gx : 0.0f;
    s.f1 = visc ? gy : 0.0f;
    s.f2 = 0.0f;
    const unsigned long long pred[8] = {ballot(active && q.st == kFollower), 0, 0, 0, 0, 0, 0, 0};
    F::template reduce<true>(p, c, s, pred);
    float ex, ey;
    grav_term(p, e.ax - kExitX, e.ay - kExitY, ex, ey);              // gravity_encoding.py:28-38
    const float nf = (float)s.i[0];
    o6[0] = e.ax; o6[1] = e.ay; o6[2] = ex * nf; o6[3] = ey * nf; o6[4] = s.f0; o6[5] = s.f1;
}

template <class F, bool GRAV>
__device__ __forceinline__ void write_obs(const Params& p, typename F::Ctx& c, bool active, bool store, const Ped& q,
                                          const Env& e, float* __restrict__ obs) {
    if constexpr (GRAV) {
        float o6[6];
        grav_observation<F>(p, c, active, q, e, o6);
        if (store && c.owner) {
#pragma unroll
            for (int k = 0; k < 6; ++k) obs[k] = o6[k];
        }
    } else {
        if (store) write_obs_generic(p, c.i, active, q, e, StorePlain{obs});
    }
}

// exiting / viscek counts of the final state for the episode record (env.py:120-123); called by all lanes
// of the env when an episode ends (rare), so the per-step reduction does not carry them.
template <class F>
__device__ __forceinline__ void finish_counts(const Params& p, typename F::Ctx& c, const Ped& q, StepOut& o) {
    Sums s{};
    const unsigned long long pred[8] = {ballot(q.st == kExiting), ballot(q.st == kViscek), 0, 0, 0, 0, 0, 0};
    F::template reduce<true>(p, c, s, pred);
    o.n_exiting = s.i[0];
    o.n_viscek = s.i[1];
}

// ------------------------------------------------------------------------------------------------
// One env step: env.py:141-171.  All lanes of the env call this together.  The body is the same for every
// family; what differs (neighbour sum, reductions, cross-lane fetches) is behind F.
// ------------------------------------------------------------------------------------------------
template <class F, bool GRAV>
__device__ __forceinline__ void step_env(const Params& p, typename F::Ctx& c, bool active, Ped& q, Env& e, float2 adir,
                                         float noise, StepOut& out) {
    // The body is branch-free: every lane runs every instruction and the results are merged with
    // selects.  Divergent `if` blocks cost s_and_saveexec / s_cbranch pairs and fence the scheduler;
    // with 4 waves per SIMD at C2 the per-wave instruction stream is what bounds the step.
    const int i = c.i;

    // ---- Time.step: area.py:53-59 ----
    e.now += 1;
    e.total += 1u;
    out.truncated = e.now >= p.max_timesteps;

    // ---- Area.agent_step: area.py:182-210 (uniform over the env, every lane computes the same values) ----
    e.adx = adir.x;                                                         // area.py:192
    e.ady = adir.y;
    const float tx = e.ax + e.adx, ty = e.ay + e.ady;                       // area.py:201
    const bool hit = fabsf(tx) > p.width || fabsf(ty) > p.height;           // area.py:203-206: < -W or > W
    e.ax = hit ? e.ax : tx;                                                 // area.py:195
    e.ay = hit ? e.ay : ty;
    const float r_agent = hit ? -5.0f : 0.0f;                               // area.py:198
    const bool term_agent = hit && (p.flags & kFlagTermOnWall) != 0;

    // ---- Area.pedestrians_step: area.py:76-180 ----
    // (team kernels: a wave without pedestrians -- a "helper" wave -- skips the per-pedestrian arithmetic and only takes part
    // in the neighbour sum and the reductions; `work` is constant true for every other family)
    bool work = true;
    if constexpr (F::kHelpers) work = !c.helper;
    const int old_st = q.st;
    PrePair pp{};
    if (work) pp = pre_pair(p, q);
    const bool efv = pp.efv, fv = pp.fv, fol = pp.fol, row = pp.row;
    const float ux = pp.ux, uy = pp.uy;
    EVAC_T(c, 1);   // leader + per-lane pre-pair work

    // ---- neighbour sum: area.py:104-119 ----
    float sx, sy;
    F::neighbour_sum(p, c, q, efv, row, ux, uy, sx, sy);
    EVAC_T(c, 3);   // neighbour sum

    Sums s{};
    unsigned long long pred[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float gx = 0.0f, gy = 0.0f;
    bool exit_lane = false;
    if (work) {
    // ---- new heading = mean heading rotated by the noise: area.py:120-136.
    // cos/sin(arctan2(my,mx)+eta) = rotation of (mx,my)/|m| by eta; arctan2(0,0) = 0.
    {
        // Only the lanes whose row was evaluated use the result (a follower's is multiplied by 0 below).  A wave without any
        // such lane -- late in an episode most waves: tools/moving_distribution.py -- skips the block (and its caller the
        // noise draw, see rollout_body): the neighbour sum it would normalise is 0 there, or the NaN of a poisoned env, and
        // 0 * (a finite heading) is the 0 that 0 * 0 is (up to the sign of a zero direction component).
        float ndx = sx, ndy = sy;
        if (ballot(row) != 0ull) {
            asm volatile("");                 // (a real uniform branch: keep the compiler from if-converting 25 instructions)
            const bool zero_mean = sx == 0.0f && sy == 0.0f;
            float sn, cs;
            noise_sincos(noise, p.small_noise, sn, cs);
            // step_size * unit mean heading, then ONE rotation: 7 instructions (scale the normaliser, two products, two
            // multiply-adds) where normalise / rotate / scale took 10
            const float ils = frsq(sx * sx + sy * sy) * p.step_size;
            const float cx = zero_mean ? p.step_size : sx * ils;
            const float cy = zero_mean ? 0.0f : sy * ils;
            ndx = __builtin_fmaf(cx, cs, -(cy * sn));
            ndy = __builtin_fmaf(cy, cs, cx * sn);
        }
        q.dx = fv ? ndx : q.dx;                                             // area.py:136
        q.dy = fv ? ndy : q.dy;
        float bdx, bdy;                                                     // area.py:139-142
        if (__builtin_constant_p(p.one_minus_ens) && p.one_minus_ens == 0.0f) {
            // (the default-configuration kernels: 0 * heading is exact, so the fused form has the same bits -- NaN * 0 included)
            bdx = __builtin_fmaf(0.0f, q.dx, p.ens * e.adx);
            bdy = __builtin_fmaf(0.0f, q.dy, p.ens * e.ady);
        } else {
            bdx = p.ens * e.adx + p.one_minus_ens * q.dx;
            bdy = p.ens * e.ady + p.one_minus_ens * q.dy;
        }
        q.dx = fol ? bdx : q.dx;
        q.dy = fol ? bdy : q.dy;
        q.x += efv ? q.dx : 0.0f;                                           // area.py:145
        q.y += efv ? q.dy : 0.0f;
    }
    {   // area.py:148-152.  med3 of a NaN is not NaN, but then miss = NaN - finite = NaN, so the position,
        // the `miss != 0` test and the flipped direction come out exactly as with np.clip.
        // (a uniform branch around the block for the steps on which no pedestrian of the wave is outside: measured, no gain --
        // profiles/r04_f_c2_ab_tail12_reflect_skip_no_gain.txt)
        const float cx = __builtin_amdgcn_fmed3f(q.x, -p.width, p.width);
        const float cy = __builtin_amdgcn_fmed3f(q.y, -p.height, p.height);
        const float mx = q.x - cx, my = q.y - cy;
        q.x = fmaf(-2.0f, mx, q.x);                                         // 2*miss is exact: same rounding as pos - 2*miss
        q.y = fmaf(-2.0f, my, q.y);
        q.dx = (mx != 0.0f) ? -q.dx : q.dx;
        q.dy = (my != 0.0f) ? -q.dy : q.dy;
    }
    EVAC_T(c, 4);   // heading, blend, move, reflect

    // ---- statuses, rewards, termination: area.py:155-178, statuses.py:29-48, reward.py:19-47 ----
    // The first idle lane (i == N, if the env does not fill its lanes) stands on the exit: it evaluates the
    // gravity exit term (gravity_encoding.py:28-38) with the very same instructions as the pedestrians'
    // terms instead of a separate single-lane block.  Its classifier result is discarded (inactive).
    exit_lane = GRAV && F::kExitLane && i == p.n_ped;
    const float px = exit_lane ? kExitX : q.x, py = exit_lane ? kExitY : q.y;
    float de, lx, ly, dl2;
    const int cls = classify(p, px, py, e.ax, e.ay, de, lx, ly, dl2);
    const int new_st = active ? cls : 0;
    q.st = new_st;
    s.f0 = active ? de : 0.0f;
    // gravity observation of the post-step state, fused into the same reduction (gravity_encoding.py:8-25);
    // R = agent - pos = -(pos - agent) reuses the classifier's offset and squared distance.
    if constexpr (GRAV) {
        grav_term2(p, -lx, -ly, dl2, gx, gy);
        const bool visc = new_st == kViscek;
        s.f1 = visc ? gx : 0.0f;
        s.f2 = visc ? gy : 0.0f;
    }
    // per-step counts: the two reward transitions, escaped (termination) and followers (gravity exit term);
    // exiting / viscek counts are only part of the episode record and are taken at episode end.
    // (conjunctions are taken on the MASKS: the ballot of a plain comparison is the comparison's own result register,
    // the ballot of `a && b` costs a select and a second comparison)
    const unsigned long long now_follower = ballot(new_st == kFollower);
    pred[0] = ballot((unsigned)(old_st - kViscek) < 2u) & ballot(new_st == kExiting);      // reward.py:35-39
    pred[1] = mask_eq<kViscek>(old_st) & now_follower;                                      // reward.py:43-46
    pred[2] = mask_eq<kEscaped>(new_st);
    pred[3] = ballot((unsigned)(new_st - kViscek) < 3u);      // moves at the next step (only the multi-wave all-pairs family uses it)
    pred[4] = now_follower;
    pred[5] = ballot(needs_row(p, new_st));                   // its row is needed at the next step (multi-wave all-pairs family)
    }   // work
    if constexpr (GRAV) F::exit_publish(c, exit_lane, gx, gy);
    if constexpr (F::kPipelined) F::stage_next(p, c, q, work);   // team kernels: the next step's tile entry travels with this reduction
    if constexpr (!(EVAC_ABLATE & 8)) {
        if constexpr (F::kPipelined) {
            // team kernels: the reduction is a round trip through global memory; the step's observation row -- per-lane data that does
            // not depend on it -- is stored between the publish and the poll, under that latency (rollout_body stores it again, over
            // this one, on the rare step that ends an episode: the reset observation)
            F::reduce_publish(p, c, s, pred);
            if constexpr (!GRAV) {
                if (work && c.obs_dst != nullptr) write_obs_generic(p, c.i, active, q, e, StorePlain{c.obs_dst});
            }
            F::reduce_collect(p, c, s);
        } else {
            F::template reduce<false>(p, c, s, pred);
        }
    }
    out.reward = out.gx = out.gy = out.ex = out.ey = 0.0f;
    if (GRAV && work) {                       // (helper waves store no observation)
        float ex = 0.0f, ey = 0.0f;
        if (F::kExitLane && p.n_ped < F::kThreadsPerEnv) {    // uniform: the env leaves a lane idle
            F::exit_fetch(c, gx, gy, p.n_ped, ex, ey);
        } else {                              // the env fills its lanes
            grav_term(p, e.ax - kExitX, e.ay - kExitY, ex, ey);
        }
        const float nf = (float)s.i[4];
        out.ex = ex * nf;
        out.ey = ey * nf;
        out.gx = s.f1;
        out.gy = s.f2;
    }
    EVAC_T(c, 5);   // classify + reductions
    out.n_escaped = s.i[2];
    out.n_follower = s.i[4];
    out.n_exiting = out.n_viscek = 0;   // filled by finish_counts() when the episode ends

    out.terminated = term_agent || (s.i[2] == p.n_ped);   // area.py:175-178, env.py:171
    // (the episode ends: ONE scalar condition for the callers' rare branch -- min of the pedestrians still inside and the steps
    // left -- instead of two compares, two mask selects and an OR every step)
    out.done = term_agent || min(p.n_ped - s.i[2], p.max_timesteps - e.now) <= 0;
    if (!work) return;                        // helper waves: the flags steer them, rewards and episode sums are the ped waves'
    float r_ped = p.init_reward;
    if constexpr (F::kEnvUniform) {
        // Real uniform branches (the empty asm keeps the compiler from if-converting them into always-executed
        // arithmetic + selects): on most steps nobody changes status and the bonus terms are skipped.
        bool moved;
        if constexpr (F::kThreadsPerEnv == kWave) {
            // one wave per env: the test is taken on the ballots themselves (the reward switches as loop-invariant masks), the
            // two popcounts move into the branch
            unsigned long long m0 = (p.flags & kFlagNewExitingReward) ? ~0ull : 0ull, m1 = (p.flags & kFlagNewFollowersReward) ? ~0ull : 0ull;
            asm("" : "+s"(m0), "+s"(m1));      // (opaque: two s_and_b64 on the ballots, not a select per half of each)
            moved = ((pred[0] & m0) | (pred[1] & m1)) != 0ull;
        } else {
            moved = (((p.flags & kFlagNewExitingReward) ? s.i[0] : 0) | ((p.flags & kFlagNewFollowersReward) ? s.i[1] : 0)) != 0;
        }
        if (moved) {
            asm volatile("");
            const float tf = 1.0f - (float)e.now * p.inv_200n;              // reward.py:26
            if (p.flags & kFlagNewExitingReward) r_ped += (15.0f + 10.0f * tf) * (float)s.i[0];
            if (p.flags & kFlagNewFollowersReward) r_ped += (10.0f + 5.0f * tf) * (float)s.i[1];
        }
    } else {                          // counts differ between the envs of a wave: selects
        const float tf = 1.0f - (float)e.now * p.inv_200n;                  // reward.py:26
        r_ped += (p.flags & kFlagNewExitingReward) ? (15.0f + 10.0f * tf) * (float)s.i[0] : 0.0f;
        r_ped += (p.flags & kFlagNewFollowersReward) ? (10.0f + 5.0f * tf) * (float)s.i[1] : 0.0f;
    }
    const float intrinsic = 0.0f - s.f0 * p.inv_n;                          // reward.py:19-21
    out.reward = r_agent + r_ped + p.intrinsic_coef * intrinsic;           // env.py:158
    e.acc_ret += out.reward;                                                // env.py:168-170
    e.acc_intr += intrinsic;
    e.acc_stat += r_agent + r_ped;
    EVAC_T(c, 6);   // rewards, flags
}

// ------------------------------------------------------------------------------------------------
// Even progress of the waves that share a SIMD.  With equal priorities the SIMD's issue arbiter serves its OLDEST wave
// first: of the four one-wave envs of a SIMD the first runs at the speed of a lone wave and leaves early, the last ends
// the launch alone on a mostly idle SIMD (wave lifetimes of one launch spread 1 : 4).  s_setprio outranks age, so:
//   * kernels whose SIMD-mates sit in other workgroups rotate the priorities by the step counter, offset by the wave's
//     slot in its SIMD (HW_ID): over four steps every wave has held every priority once (C2: 3.02 -> 2.74 us per step);
//     the workgroup-per-env kernels use the env's load as its priority instead (see rollout_body);
//   * the CU-wide workgroup (Wave<1, 1024>) knows its SIMD-mates (waves w, w+4, w+8, w+12): every wave publishes its step
//     counter in LDS and takes as priority the number of mates that are ahead of it.
// A hint only: results do not depend on it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void set_wave_priority(int k) {   // k in 0..3, wave-uniform
    // (s_setprio takes an immediate: a two-level compare tree, four or five scalar instructions on every path -- the switch the
    // compiler builds from the four builtin calls runs up to thirteen)
    k &= 3;
    asm volatile(
        "s_cmp_lt_u32 %0, 2\n\t"
        "s_cbranch_scc1 1f\n\t"
        "s_cmp_eq_u32 %0, 3\n\t"
        "s_cbranch_scc1 3f\n\t"
        "s_setprio 2\n\t"
        "s_branch 9f\n"
        "3:\n\t"
        "s_setprio 3\n\t"
        "s_branch 9f\n"
        "1:\n\t"
        "s_cmp_eq_u32 %0, 0\n\t"
        "s_cbranch_scc1 0f\n\t"
        "s_setprio 1\n\t"
        "s_branch 9f\n"
        "0:\n\t"
        "s_setprio 0\n"
        "9:"
        :
        : "s"(k)
        : "scc");
}
__device__ __forceinline__ int simd_wave_slot() {   // slot of this wave among the waves of its SIMD (HW_ID bits 3:0)
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return (int)(hw & 3u);
}

// ------------------------------------------------------------------------------------------------
// The schedule of the CU-wide rollout workgroups: envs sorted by the pedestrians still moving (the length of their pair
// loop, as a launch left it in `moving`) and dealt to the SIMDs in snake order, so that the four envs of every SIMD -- and
// the sixteen of every CU -- carry about the same load.  A launch lasts as long as its slowest SIMD; with random placement
// that is 1.25-1.35x the mean load for most of an episode (tools/moving_distribution.py).
// perm[slot] = env.  per_wg = 16: one-wave envs (slot = workgroup * 16 + wave, SIMD = wave % 4); per_wg = 4: four-wave envs
// (slot = workgroup * 4 + env of the workgroup; every SIMD runs one wave of each of the four) -- one env of each load quartile
// per SIMD / per workgroup, the heaviest in the SIMD's first (oldest) wave.  Loads are binned in 65 steps of `unit` pedestrians;
// ties are placed in arrival order (LDS atomics): the permutation may differ from run to run, the results cannot.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int schedule_bin(int load, int unit) { return min(max(load / unit, 0), kWave); }
// rank by load (ascending) -> slot.  Workgroup 0 takes the per_wg LIGHTEST envs of the batch (it deals the next launch's envs
// before it starts stepping -- rollout_body -- and must still be done before the others); the rest is dealt in snake order.
// (mode: a run-time switch of the CHAINED launches' deal, for A/B runs of ONE binary -- this kernel's speed depends on its register
// allocation more than on most source changes, so compile-time variants cannot be compared: 1 = envs of similar load share a
// workgroup, lightest workgroups first; 2 = heaviest first; 0 = the snake deal below)
__device__ __forceinline__ int schedule_slot(int r, int n_envs, int per_wg, int mode = 0) {
    if (mode == 1) return r;
    if (mode == 2) return n_envs - 1 - r;
    const int e16 = n_envs & ~(per_wg - 1);  // the last, partial workgroup: as ranked
    if (r < per_wg || r >= e16) return r;
    r -= per_wg;
    const int G = (e16 - per_wg) >> 2;       // groups of four (one SIMD / one workgroup each) among the other full workgroups
    const int k = r / G, j = r - k * G;
    const int g = (k & 1) ? G - 1 - j : j;   // snake: quarters 0 and 2 ascending, 1 and 3 descending
    const int kk = 3 - k;                    // (the heaviest quartile goes to the SIMDs' first -- oldest -- waves, see rollout_body)
    return per_wg + (per_wg == 16 ? (g >> 2) * 16 + kk * 4 + (g & 3)      // workgroup 1 + g / 4, SIMD g % 4, the SIMD's kk-th wave
                                  : g * 4 + kk);                          // workgroup 1 + g, its kk-th env (its waves 4 kk .. 4 kk + 3: one per SIMD)
}
// The counting sort by ONE workgroup of 1024 threads (hist: kWave + 2 ints of LDS; all 16 waves call it together): as a launch
// of its own (k_schedule) and at the start of workgroup 0 of a rollout launch (rollout_body).
__device__ __forceinline__ void schedule_envs_by_workgroup(int* hist, int tid, int n_envs, const int* __restrict__ loads,
                                                           int* __restrict__ perm, int per_wg, int unit, int mode = 0) {
    constexpr int kPer = 4;                  // envs per thread and pass: their loads are fetched back to back (one memory latency, not four)
    if (tid < kWave + 2) hist[tid] = 0;
    __syncthreads();
    for (int base = 0; base < n_envs; base += kPer * 1024) {
        int b[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int e = base + k * 1024 + tid;
            b[k] = e < n_envs ? loads[e] : -1;
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (b[k] >= 0) atomicAdd(&hist[schedule_bin(b[k], unit)], 1);
    }
    __syncthreads();
    if (tid < kWave) {                       // exclusive prefix over the 65 bins (bin 64 = everything at or above 64)
        const int v = hist[tid];
        const int incl = wave_inclusive_scan(v);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        hist[tid] = incl - v;
        if (tid == kWave - 1) hist[kWave] = incl;
    }
    __syncthreads();
    for (int base = 0; base < n_envs; base += kPer * 1024) {
        int r[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int e = base + k * 1024 + tid;
            r[k] = e < n_envs ? loads[e] : -1;
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k)
            if (r[k] >= 0) r[k] = atomicAdd(&hist[schedule_bin(r[k], unit)], 1);            // rank by load, ascending
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int e = base + k * 1024 + tid;
            if (e < n_envs) perm[schedule_slot(r[k], n_envs, per_wg, mode)] = e;
        }
    }
}

// Pace keeping of the CU-wide workgroups: a wave publishes its step counter and takes as priority the number of its
// SIMD-mates that are ahead of it (one-wave envs: wave w runs on SIMD w % 4; four-wave envs: wave k of every env on SIMD k).
// Cost per step: one LDS write (all lanes, no exec mask: rollout_body), one LDS read whose result is used a whole step later (`seen`: a hint may be a step old, and
// the wave never waits for the round trip), one compare + popcount, and the s_setprio switch only when the rank changes.
__device__ __forceinline__ void pace_step(int* slot, const int* mine, int lane, int t, int& seen, int& prio) {   // `mine`: the SIMD's four counters
    *slot = t;                                                         // (lane 0: the wave's counter; the others: the sink)
    const unsigned ahead_mask = (unsigned)ballot(seen >= t) & 0xfu;   // lanes 0..3 hold the four counters as read one step ago
    seen = mine[lane & 3];
    const int ahead = __popc(ahead_mask) & 3;                          // (t = 0: the zero-initialised `seen` counts all four: & 3)
    if (ahead != prio) {
        prio = ahead;
        set_wave_priority(ahead);
    }
}


// ------------------------------------------------------------------------------------------------
// Kernels of the Wave / Cells families.  __launch_bounds__(block, 4): at least 4 waves per SIMD, i.e. at most
// 128 VGPRs -- the 4-wave kernel once grew to 135 and silently lost a quarter of its occupancy.
// ------------------------------------------------------------------------------------------------
// The observation / reward outputs of one env step, optionally through the trainer's normalisation chain (NORM; the
// fused form of evac_norm_step): every thread normalises the observation elements it writes, terminal observation
// before reset observation (the order in which SyncVectorEnv runs the wrapped step() and reset()); for the gravity
// observation lanes 0..5 of the env take one feature each, lane 6 the reward.
template <class F, bool GRAV, bool NORM>
__device__ __forceinline__ void step_outputs(const Params& p, typename F::Ctx& w, bool active, Ped& q, Env& e, StepOut& o,
                                             uint32_t gid, int autoreset, float* __restrict__ obs_out,
                                             float* __restrict__ reward_out, uint8_t* __restrict__ term_out,
                                             uint8_t* __restrict__ trunc_out, float* __restrict__ final_obs,
                                             evac_episode_stats_t* __restrict__ final_stats, const NormArgs& na) {
    const bool done = o.terminated || o.truncated;
    const int D = p.obs_dim;
    float* obs = obs_out + (size_t)w.env * D;
    float* fo = final_obs ? final_obs + (size_t)w.env * D : nullptr;
    double* ns = NORM ? na.state + (size_t)w.env * (3 * D + 4) : nullptr;
    float o6[6] = {e.ax, e.ay, o.ex, o.ey, o.gx, o.gy};   // GRAV: the observation came out of step_env's reduction
    float t6[6] = {o6[0], o6[1], o6[2], o6[3], o6[4], o6[5]};
    const bool reset_now = done && autoreset;
    if (reset_now) {
        if constexpr (!GRAV) {
            if (fo) {
                if constexpr (NORM) write_obs_generic(p, w.i, active, q, e, StoreNorm{fo, ns, D, na.eps, na.obs_clip});
                else write_obs_generic(p, w.i, active, q, e, StorePlain{fo});
            }
        }
        if (final_stats) {
            finish_counts<F>(p, w, q, o);
            if (w.owner) write_stats(final_stats + w.env, e, o);
        }
        reset_env(p, active, philox_reset_draw(p, gid, w.i, e.n_resets), q, e);
        if constexpr (GRAV) grav_observation<F>(p, w, active, q, e, o6);
    }
    if constexpr (GRAV) {
        if constexpr (!NORM) {
            if (w.owner) {
                if (reset_now && fo) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) fo[k] = t6[k];
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) obs[k] = o6[k];
            }
        } else if constexpr (F::kEnvUniform) {
            // the six numbers are uniform over the env's first wave: lane k owns feature k
            if (w.i < 6) {
                const int k = w.i;
                float tv = t6[0], ov = o6[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) { tv = k == j ? t6[j] : tv; ov = k == j ? o6[j] : ov; }
                if (reset_now && fo) StoreNorm{fo, ns, D, na.eps, na.obs_clip}(k, tv);
                StoreNorm{obs, ns, D, na.eps, na.obs_clip}(k, ov);
            }
        } else if (w.owner) {   // several envs per wave: the sums are valid in the owner lane only
            if (reset_now && fo) {
#pragma unroll
                for (int k = 0; k < 6; ++k) StoreNorm{fo, ns, D, na.eps, na.obs_clip}(k, t6[k]);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) StoreNorm{obs, ns, D, na.eps, na.obs_clip}(k, o6[k]);
        }
    } else {
        if constexpr (NORM) write_obs_generic(p, w.i, active, q, e, StoreNorm{obs, ns, D, na.eps, na.obs_clip});
        else write_obs_generic(p, w.i, active, q, e, StorePlain{obs});
    }
    const bool reward_lane = (NORM && GRAV && F::kEnvUniform) ? w.i == 6 : w.owner;
    if (reward_lane) {
        float r = o.reward;
        if constexpr (NORM) {   // gymnasium NormalizeReward: returns = returns * gamma * (1 - terminated) + r; r / sqrt(var + eps)
            double mean = ns[3 * D], var = ns[3 * D + 1], cnt = ns[3 * D + 2], ret = ns[3 * D + 3];
            ret = ret * (double)na.gamma * (o.terminated ? 0.0 : 1.0) + (double)r;
            rms_update1(mean, var, cnt, ret);
            const double v = (double)r / sqrt(var + (double)na.eps);
            r = (float)fmin(fmax(v, -(double)na.reward_clip), (double)na.reward_clip);
            ns[3 * D] = mean; ns[3 * D + 1] = var; ns[3 * D + 2] = cnt; ns[3 * D + 3] = ret;
        }
        reward_out[w.env] = r;
        term_out[w.env] = o.terminated ? 1 : 0;
        trunc_out[w.env] = o.truncated ? 1 : 0;
    }
}

template <class F, bool GRAV, bool NORM>
__device__ __forceinline__ void step_kernel_body(
    typename F::Smem& sm, const Params& p, const float2* __restrict__ actions, const float* __restrict__ noise_in,
    float* __restrict__ obs_out, float* __restrict__ reward_out, uint8_t* __restrict__ term_out,
    uint8_t* __restrict__ trunc_out, int autoreset, float* __restrict__ final_obs,
    evac_episode_stats_t* __restrict__ final_stats, const NormArgs& na) {
    typename F::Ctx w(sm);
    if (w.env >= p.n_envs) return;   // whole waves (WPE == 1) or whole workgroups: no barrier is skipped by a subset
    F::init(w);
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    const uint32_t gid = p.env_id_offset + (uint32_t)w.env;
    const float2 a = actions[w.env];
    float nz = 0.0f;
    if (noise_in) {
        if (active) nz = noise_in[(size_t)w.env * p.n_ped + w.i];
    } else if (ballot(needs_row(p, q.st)) != 0ull) {   // (only a lane whose row is evaluated uses its draw: step_env)
        nz = philox_noise(p, gid, w.i, e.total);
    }
    StepOut o;
    step_env<F, GRAV>(p, w, active, q, e, agent_direction(p, a.x, a.y), nz, o);
    step_outputs<F, GRAV, NORM>(p, w, active, q, e, o, gid, autoreset, obs_out, reward_out, term_out, trunc_out, final_obs,
                                final_stats, na);
    store_env(p, w.env, w.i, active, w.owner, q, e);
}
// The reference's default configuration as compile-time constants (see k_rollout_default_config below).
template <bool GRAV>
__device__ __forceinline__ Params default_config_constants(Params p) {
    // no wall termination, no NaN guard; the two status rewards and ClipAction stay run-time options (the reference's scripts
    // differ in them: run_scripts/*.sh; the trainer clips)
    p.flags = p.flags & (kFlagClipAction | kFlagNewExitingReward | kFlagNewFollowersReward);
    p.small_noise = 2;
    p.ens = 1.0f;
    p.one_minus_ens = 0.0f;
    if constexpr (GRAV) {
        p.grav_pow_int = 5;                       // gravity observation with alpha = 3 (wrappers/config.py default)
    } else {
        p.obs_pos = EVAC_POS_REL;                 // the Box observation of BASELINE config 5: relative positions + one-hot statuses
        p.obs_stat = EVAC_STAT_OHE;
        p.obs_box = 1;
    }
    return p;
}
// evac_step / evac_step_normalized
#define EVAC_STEP_KERNEL(NAME, NORM_)                                                                                          \
    template <class F, bool GRAV>                                                                                              \
    __global__ __launch_bounds__(F::kBlock, 4) void NAME(                                                                       \
        Params p, const float2* __restrict__ actions, const float* __restrict__ noise_in, float* __restrict__ obs_out,          \
        float* __restrict__ reward_out, uint8_t* __restrict__ term_out, uint8_t* __restrict__ trunc_out, int autoreset,         \
        float* __restrict__ final_obs, evac_episode_stats_t* __restrict__ final_stats, NormArgs na) {                           \
        __shared__ typename F::Smem sm;                                                                                         \
        step_kernel_body<F, GRAV, NORM_>(sm, p, actions, noise_in, obs_out, reward_out, term_out, trunc_out, autoreset, final_obs, \
                                         final_stats, na);                                                                      \
    }
EVAC_STEP_KERNEL(k_step_raw, false)
EVAC_STEP_KERNEL(k_step_norm, true)
#undef EVAC_STEP_KERNEL
#define EVAC_STEP_KERNEL_DEFAULT(NAME, NORM_)                                                                                  \
    template <class F, bool GRAV>                                                                                              \
    __global__ __launch_bounds__(F::kBlock, 4) void NAME(                                                                       \
        Params p, const float2* __restrict__ actions, const float* __restrict__ noise_in, float* __restrict__ obs_out,          \
        float* __restrict__ reward_out, uint8_t* __restrict__ term_out, uint8_t* __restrict__ trunc_out, int autoreset,         \
        float* __restrict__ final_obs, evac_episode_stats_t* __restrict__ final_stats, NormArgs na) {                           \
        __shared__ typename F::Smem sm;                                                                                         \
        const Params q = default_config_constants<GRAV>(p);                                                                     \
        step_kernel_body<F, GRAV, NORM_>(sm, q, actions, noise_in, obs_out, reward_out, term_out, trunc_out, autoreset, final_obs, \
                                         final_stats, na);                                                                      \
    }
EVAC_STEP_KERNEL_DEFAULT(k_step_default_config, false)
EVAC_STEP_KERNEL_DEFAULT(k_step_norm_default_config, true)
#undef EVAC_STEP_KERNEL_DEFAULT

// T steps per launch, state in registers (rpo_agent.py:180-203 rollout loop, RandomAgent or given actions).
// Output: ONE packed f32 slab [T][E][D+3] = [obs(D) | reward | terminated | truncated] -- a single message
// for the all-gather and a single coalesced store stream for the kernel.  GRAV kernels stage the 9 words
// of up to 7 steps in LDS and flush them with one 64-lane store (five single-lane stores per step cost a
// third of the step before: profiles/r01_e_*).
// DIAG: the diagnostic face (trajectory capture, action recording, injected noise); the default face carries none
// of that code.
// CHAIN: the launch is one of a chain of overlapping launches on two queues (evac_common.h, ChainArgs): every wave first waits
// for ITS env's generation word, exchanges the state by device-scope accesses and publishes the next generation at its end.
// PERSIST: the kernel stays resident and takes every evac_rollout call as a command from a ring (evac_common.h, PersistCmd): the state
// stays in registers from call to call -- no launch boundary, no prologue, no hand-off -- until a STOP command (evac_join).
template <class F, bool GRAV, bool DIAG, bool CHAIN = false, bool PERSIST = false>
__device__ __forceinline__ void rollout_body(
    typename F::Smem& sm, const Params& p, int n_steps, const float2* __restrict__ actions, float2* __restrict__ actions_out,
    float* __restrict__ slab_out, evac_episode_stats_t* __restrict__ final_stats, int capture_envs,
    float* __restrict__ capture, const float* __restrict__ noise_in, const int* __restrict__ perm = nullptr,
    int* __restrict__ moving_out = nullptr, const int* __restrict__ deal_loads = nullptr, int* __restrict__ deal_perm = nullptr,
    ChainArgs chain = ChainArgs{nullptr, 0, nullptr, nullptr}) {
#ifdef EVAC_STEP_TIMES
    unsigned long long mark_entry_, mark_loop_ = 0, mark_done_, mark_perm_, mark_init_, mark_act_, mark_state_;
#define EVAC_MARK(M) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(M)::"memory")
    EVAC_MARK(mark_entry_);
#endif
    if constexpr (CHAIN) {      // this workgroup has its CU: counted for the gate in front of the next launch (system scope: the queue's processor reads it)
        if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(chain.started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    typename F::Ctx w(sm);
    if (w.env >= p.n_envs) return;
    // the schedule of the CU-wide workgroups: which env this wave carries; any permutation gives the same results
    const int my_slot = w.env;
    if (perm) w.env = __builtin_amdgcn_readfirstlane(perm[w.env]);
#ifdef EVAC_STEP_TIMES
    asm volatile("" ::"s"(w.env));
    EVAC_MARK(mark_perm_);
#endif
    F::init(w);
#ifdef EVAC_STEP_TIMES
    EVAC_MARK(mark_init_);
#endif
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    if constexpr (!CHAIN) load_env(p, w.env, w.i, active, q, e);
    // THE DEAL OF THE NEXT LAUNCH IS MADE INSIDE THIS ONE, by workgroup 0 before it starts stepping: it carries the lightest
    // envs of the batch (schedule_slot), which are done 20-30 % before the launch ends (tools/step_times.py) -- the ~3 us of the
    // sort disappear in that slack.  It sorts by the loads the PREVIOUS launch left (complete, unlike this launch's) into the
    // permutation buffer this launch does not read.  A launch of its own for the sort costs 6.5 us next to a 45 us rollout
    // launch, so rounds 2-3 dealt only every 50-200 env steps; a fresh deal is worth 3 us per launch late in an episode
    // (profiles/r04_b_c2_schedule_frequency.txt).  (Tried and dropped: one light wave sorting at the END of the launch -- 6 us
    // on every launch's tail; every env drawing a ticket with a global atomic -- 4096 atomics on a handful of addresses, 48 us.)
    // Right after a reset every env is as heavy as every other: workgroup 0 has no slack to hide the sort in (+3.5 us on those
    // launches), and no deal is better than another -- so it deals only once its lightest env (slot 0 of the deal at hand) is
    // down to three quarters of the pedestrians, and otherwise leaves the other buffer's older permutation in place.
    if constexpr (F::kPace) {
        if (deal_perm && blockIdx.x == 0) {    // (workgroup-uniform; the host passes deal_perm only for batches of >= one full workgroup)
            const int lightest = deal_loads[perm[0]];
            if (4 * lightest <= 3 * p.n_ped)
                schedule_envs_by_workgroup(sm.deal_hist, (int)threadIdx.x, p.n_envs, deal_loads, deal_perm, F::kEnvsPerBlock, F::WPE == 1 ? 1 : 4,
                                           CHAIN ? chain.deal_mode : 0);
        }
    }
    if constexpr (CHAIN) {
        // (a chained launch deals -- above -- BEFORE it waits: the sort of workgroup 0 runs under the hand-off of its envs.  The
        // permutation it writes is read two launches later, by the next launch of THIS queue: in order, no flag needed.)
        static_assert(!F::kHelpers && (F::kPace || F::WPE == 1), "chained launches: one-wave envs (any workgroup size), and four-wave envs in CU-wide workgroups (a barrier per env)");
        constexpr int T = F::kThreadsPerEnv;
        if constexpr (F::WPE == 1) {
            if (!chain_wait<T>(chain, w.env)) {      // (wave-uniform) the env's state never came: void run, the host is told
                chain_give_up(chain, w.lane, w.env);
                return;
            }
            load_record<T>(chain.xchg + (size_t)w.env * Xchg<T>::kBytes, w.lane, w.lane, active, q, e);
        } else {
            // several waves per env: its first wave waits, the others meet it at the env's barrier and take ITS verdict -- one
            // poller per env, and all waves of the env leave together should the wait give up
            if (w.wave_in_env == 0) {
                const bool ok = chain_wait<T>(chain, w.env);
                if (w.lane == 0) sm.chain_ok[w.slot] = ok ? 1 : 0;
            }
            F::sync(w);
            if (sm.chain_ok[w.slot] == 0) {
                if (w.wave_in_env == 0) chain_give_up(chain, w.lane, w.env);
                return;
            }
            load_record<T>(chain.xchg + (size_t)w.env * Xchg<T>::kBytes, w.i, w.lane, active, q, e);
        }
    }
    const uint32_t gid = p.env_id_offset + (uint32_t)w.env;
    const size_t E = (size_t)p.slab_envs;
    const int row = p.obs_dim + 3;
    // The four noise words of Philox block `noise_group` (-1: none), ROTATED every step so that nzw[0] is the word of the step
    // at hand: three register moves per step where picking word `total & 3` took three scalar compare / select pairs and
    // three v_cndmask.
    uint32_t nzw[4] = {0u, 0u, 0u, 0u};
    int noise_group = -1;
    // RandomAgent actions and the leader directions they give (area.py:189-192) are produced 64 steps at a
    // time, one step per LANE (a per-wave scalar Philox would cost ~100 SALU instructions every step),
    // and fetched per step with v_readlane.  The first block is drawn HERE, under the state loads (only the env's clock
    // word is needed for it: a scalar load); the later ones at the bottom of the step in front of theirs.
    float2 lane_act = make_float2(0.f, 0.f), lane_adir = make_float2(0.f, 0.f);
    auto draw_actions = [&](int t_first) {
        if (actions) {
            if (t_first + w.lane < n_steps) lane_act = actions[(size_t)(t_first + w.lane) * E + w.env];
        } else if constexpr (EVAC_ABLATE & 4) {
            lane_act = make_float2(0.3f, -0.7f);
        } else {
            lane_act = philox_action(p, gid, e.total + (uint32_t)w.lane);   // e.total grows by exactly 1 per step
        }
        lane_adir = agent_direction(p, lane_act.x, lane_act.y);
    };
    if constexpr (!PERSIST) draw_actions(0);
#ifdef EVAC_STEP_TIMES
    asm volatile("" ::"v"(lane_adir.x), "v"(lane_adir.y));
    EVAC_MARK(mark_act_);
#endif
    // flush mapping of the staged outputs: lane l carries word l % 9 of staged step l / 9
    const int fl_s = w.lane / kGravRow, fl_k = w.lane - fl_s * kGravRow;
    // Retire the state loads HERE, or their first use inside the loop puts `s_waitcnt vmcnt(0)` -- which
    // also waits for the previous step's stores -- into every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only
#ifdef EVAC_STEP_TIMES
    EVAC_MARK(mark_state_);
#endif
    // (the wave's place among its SIMD-mates: recomputed where it is used rather than held in two more scalar registers)
#define EVAC_PACE_SIMD (F::WPE == 1 ? (w.slot & 3) : (w.wave_in_env & 3))
#define EVAC_PACE_K (F::WPE == 1 ? (w.slot >> 2) : w.slot)
    // (pace keeping: lane 0 publishes the wave's step counter; the other lanes store theirs to a sink row of the wave's own, so
    // that the store needs no exec mask -- one address register, computed once, instead of a save / restore pair per step)
    int* pace_slot = nullptr;
    if constexpr (F::kPace) {
        pace_slot = w.lane == 0 ? &sm.progress[EVAC_PACE_SIMD * 4 + EVAC_PACE_K] : &sm.pace_sink[threadIdx.x >> 6][w.lane];
        *pace_slot = 0;
    }
    constexpr bool kRotate = !F::kPace && std::is_same<F, Wave<F::WPE>>::value && F::WPE < 16;
    int prio_slot = 0;
    if constexpr (kRotate) prio_slot = simd_wave_slot();
#ifdef EVAC_STAMP
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w.stamp.last)::"memory");
    unsigned long long rt0_;   // constant 100 MHz counter next to the shader-clock one: their ratio is the clock the kernel ran at
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0_)::"memory");
    const unsigned long long ck0_ = w.stamp.last;
#endif
    // Start-of-launch priorities.  With a schedule (k_schedule) the SIMD's k-th wave carries an env of the k-th load quartile
    // (the heaviest in the SIMD's OLDEST wave, so that age -- the arbiter's tie-break -- works for it too),
    // and the heaviest wave of a SIMD ends the launch: it starts with the highest priority instead of earning it over the
    // first steps (pace_step's information is a step old; with equal priorities the twelve lighter waves of the CU ran their
    // first steps first and the heavy ones began 3-4 us late: tools/step_times.py).  `pace_seen` is seeded so that step 0
    // confirms that rank.
    int pace_seen = 0, pace_prio = 0;
    if constexpr (F::kPace && F::WPE == 1) {
        if (perm) {
            pace_prio = 3 - (w.slot >> 2);
            set_wave_priority(pace_prio);
            pace_seen = (w.lane & 3) < pace_prio ? (1 << 30) : -1;
        }
    }
    // PERSIST: one pass of the loop below per command; `t_base` = the steps of the commands before (the pace counters run on)
    int cmd_index = PERSIST ? chain.gen : 0, t_base = 0;
    if constexpr (PERSIST) {      // (a kernel that takes up where another one left: every env at the command IT had reached)
        if (chain.resume) cmd_index = __builtin_amdgcn_readfirstlane(load_dev_i32_now(persist_next_cmd(chain.xchg) + w.env));
    }
    [[maybe_unused]] const int cmd_first = cmd_index;   // (teams: their verdicts are numbered from the kernel's first command)
    for (;;) {
    if constexpr (PERSIST) {
        unsigned long long c_slab = 0, c_stats = 0;
        int c_steps = 0;
        bool got = false;
        if (cmd_index > chain.stop_at) break;           // (the finisher of a join: this env has run everything up to its STOP)
        if constexpr (F::kHelpers) {
            if (F::aborted(w)) break;                   // (a team that lost a member in the command before: void, nothing more is run)
            // a team: every member workgroup reads the ring through its first wave, and the TEAM decides once (evac_common.h)
            if (w.wave == 0) {
                got = persist_wait(chain.xchg, cmd_index, w.lane, c_steps, c_slab, c_stats);
                int verdict = 0;
                if (w.lane == 0) verdict = persist_team_decide(persist_decision(chain.xchg, p.n_envs) + w.env, cmd_index - cmd_first + 1, got);
                verdict = __builtin_amdgcn_readfirstlane(verdict);
                const bool run = (verdict & 1) != 0;
                if (run && !got)                        // the team runs it: it is in the ring, or about to show
                    for (int tries = 0; tries < 8192 && !got; ++tries) got = persist_wait(chain.xchg, cmd_index, w.lane, c_steps, c_slab, c_stats);
                if (!run) got = false;                  // the team has left before this command
                if (w.lane == 0) {
                    int* pc = sm.persist_cmd[0];
                    pc[0] = got ? 1 : 0; pc[1] = c_steps;
                    pc[2] = (int)(unsigned)c_slab; pc[3] = (int)(unsigned)(c_slab >> 32);
                    pc[4] = (int)(unsigned)c_stats; pc[5] = (int)(unsigned)(c_stats >> 32);
                }
            }
            __syncthreads();
            const int* pc = sm.persist_cmd[0];
            got = __builtin_amdgcn_readfirstlane(pc[0]) != 0;
            c_steps = __builtin_amdgcn_readfirstlane(pc[1]);
            c_slab = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(pc[3]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(pc[2]);
            c_stats = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(pc[5]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(pc[4]);
            // (the next write to persist_cmd lies behind this command's steps and their workgroup barriers)
        } else if constexpr (F::WPE == 1) {
            got = persist_wait(chain.xchg, cmd_index, w.lane, c_steps, c_slab, c_stats);
        } else {
            // several waves per env: its first wave reads the ring, the others take ITS verdict and command at the env's barrier -- the
            // waves of an env must not disagree on whether a command came in time
            if (w.wave_in_env == 0) {
                got = persist_wait(chain.xchg, cmd_index, w.lane, c_steps, c_slab, c_stats);
                if (w.lane == 0) {
                    int* pc = sm.persist_cmd[w.slot];
                    pc[0] = got ? 1 : 0; pc[1] = c_steps;
                    pc[2] = (int)(unsigned)c_slab; pc[3] = (int)(unsigned)(c_slab >> 32);
                    pc[4] = (int)(unsigned)c_stats; pc[5] = (int)(unsigned)(c_stats >> 32);
                }
            }
            F::sync(w);
            const int* pc = sm.persist_cmd[w.slot];
            got = __builtin_amdgcn_readfirstlane(pc[0]) != 0;
            c_steps = __builtin_amdgcn_readfirstlane(pc[1]);
            c_slab = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(pc[3]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(pc[2]);
            c_stats = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(pc[5]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(pc[4]);
            // (the next write to persist_cmd lies behind this command's steps and their barriers: no second barrier here)
        }
        if (!got) break;                                // IDLE: nothing came for ~150 us -- the env's state and its place in the ring are stored, the wave leaves
        if (c_steps == 0) {                             // STOP (evac_join): consumed
            cmd_index += 1;
            break;
        }
        n_steps = c_steps;
        slab_out = (float*)c_slab;
        final_stats = (evac_episode_stats_t*)c_stats;
        draw_actions(0);                                 // (RandomAgent actions: calls with GIVEN actions take the plain path, evac_api.hip)
    }
    // Staging block of the slab rows (GRAV kernels): the step's row goes to byte offset `stage_off`; the block is flushed after
    // step `flush_t` (its seventh row, or the launch's last); `stage_t0` = the step its first row belongs to.
    int stage_off = 0, stage_t0 = 0, flush_t = min(kStageSteps, n_steps) - 1;
    constexpr int kStageRowBytes = (int)sizeof(sm.stage[0][0]);
    for (int t = 0; t < n_steps; ++t) {
#ifdef EVAC_STEP_TIMES
        if (w.lane == 0 && t < 128 && blockIdx.x == EVAC_STEP_TIMES_BLOCK && threadIdx.x < 1024) {
            unsigned long long now_;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");
            g_step_times[threadIdx.x >> 6][t] = now_;
        }
        if (t == 0) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mark_loop_)::"memory");
#endif
        if constexpr (kRotate) {
            if (p.fair) {
                // workgroup-per-env kernels: longest job first -- the env's load (its moving pedestrians, known from the last
                // reduction) is its priority, so the densest env of a CU, which ends the launch, runs nearly unimpeded
                // (C3: 1.78e8 against 1.73e8 env-steps/s with the rotation, 1.62e8 without priorities)
                if constexpr (F::WPE > 1) {
                    if (w.have_next) set_wave_priority(min(3, (4 * w.next_cols) / max(1, p.n_ped)));
                    else set_wave_priority(t + prio_slot);
                } else {
                    set_wave_priority(t + prio_slot);
                }
            }
        }
        if constexpr (F::kPace) pace_step(pace_slot, &sm.progress[EVAC_PACE_SIMD * 4], w.lane, t_base + t, pace_seen, pace_prio);
        EVAC_T(w, 12);  // (sub-phase of the diagnostic build: loop top + pace keeping)
#undef EVAC_PACE_SIMD
#undef EVAC_PACE_K
        const int slot64 = t & 63;
        float2 a = make_float2(0.f, 0.f), adir;
        if constexpr (DIAG) {
            a.x = readlane_f(lane_act.x, slot64);
            a.y = readlane_f(lane_act.y, slot64);
        }
        adir.x = readlane_f(lane_adir.x, slot64);
        adir.y = readlane_f(lane_adir.y, slot64);
        if (DIAG && actions_out && w.owner) actions_out[(size_t)t * E + w.env] = a;   // diagnostic face only
        // One Philox call serves four consecutive steps of this pedestrian: word (total & 3) of the block with counter
        // total >> 2.  The draw is LAZY: only a lane whose row is evaluated uses its noise (step_env), so a wave without such a
        // lane -- most waves late in an episode -- neither draws nor, at the next group of four steps, calls Philox at all;
        // `noise_group` is the block the four words in registers belong to; they are rotated by one at the bottom of every
        // step (drawn or not), so the step's word is nzw[0] -- and a block first drawn in the middle of its group (the launch
        // did not start on a multiple of four, or the wave skipped its first steps) is rotated into place where it is drawn.
        bool draws = true;                      // (team kernels: waves without pedestrians draw no noise)
        if constexpr (F::kHelpers) draws = !w.helper;
        float nz = 0.0f;
        const bool wants_noise = ballot(needs_row(p, q.st)) != 0ull;
        if (draws && wants_noise && !(EVAC_ABLATE & 4)) {
            const int group = (int)(e.total >> 2);
            if (group != noise_group) {
                asm volatile("");
                const uint4 r = philox4x32_10(make_uint4(gid, (uint32_t)w.i, e.total >> 2, kStreamNoise), p.seed_lo, p.seed_hi);
                nzw[0] = r.x; nzw[1] = r.y; nzw[2] = r.z; nzw[3] = r.w;
                noise_group = group;
                for (unsigned k = e.total & 3u; k != 0u; --k) {      // (rare: see above)
                    asm volatile("");
                    const uint32_t f = nzw[0];
                    nzw[0] = nzw[1]; nzw[1] = nzw[2]; nzw[2] = nzw[3]; nzw[3] = f;
                }
            }
            nz = u01_centred(nzw[0]) * p.noise_coef;
        }
        nzw[0] = nzw[1]; nzw[1] = nzw[2]; nzw[2] = nzw[3];           // the next step's word moves up
        if constexpr (DIAG) {
            if (noise_in) nz = active ? noise_in[((size_t)t * E + w.env) * p.n_ped + w.i] : 0.0f;   // injection mode
        }
        StepOut o{};
        EVAC_T(w, 0);   // action fetch + noise Philox
        float* rowp = slab_out + ((size_t)t * E + w.env) * row;
        if constexpr (F::kPipelined && !GRAV && !(EVAC_ABLATE & 2)) w.obs_dst = rowp;     // (team kernels: step_env stores the observation row itself)
        step_env<F, GRAV>(p, w, active, q, e, adir, nz, o);
        // trajectory capture for rendering (Pedestrians.save / Agent.save, pedestrians.py:33-35, area.py:32-33):
        // the post-step, pre-reset state of the first `capture_envs` envs; row N holds the leader.
        if (DIAG && capture && w.env < capture_envs) {   // wave-/workgroup-uniform; compiled out of the default kernel
            float* cp = capture + (((size_t)t * capture_envs + w.env) * (p.n_ped + 1)) * 3;
            if (active) {
                cp[3 * w.i + 0] = q.x;
                cp[3 * w.i + 1] = q.y;
                cp[3 * w.i + 2] = (float)q.st;
            }
            if (w.owner) {
                cp[3 * p.n_ped + 0] = e.ax;
                cp[3 * p.n_ped + 1] = e.ay;
                cp[3 * p.n_ped + 2] = 0.0f;
            }
        }
        float o6[6] = {e.ax, e.ay, o.ex, o.ey, o.gx, o.gy};
        float f_term = 0.0f, f_trunc = 0.0f;
        if (o.done) {                        // wave-/workgroup-uniform, rare
            asm volatile("");                // (a real branch: the two flags are constants on the other path)
            f_term = o.terminated ? 1.0f : 0.0f;
            f_trunc = o.truncated ? 1.0f : 0.0f;
            if (final_stats) {
                finish_counts<F>(p, w, q, o);
                if (w.owner) write_stats(final_stats + (size_t)t * E + w.env, e, o);
            }
            reset_env(p, active, philox_reset_draw(p, gid, w.i, e.n_resets), q, e);
            F::invalidate(w);
            if constexpr (GRAV) grav_observation<F>(p, w, active, q, e, o6);
        }
        if constexpr (GRAV) {
            if constexpr (!(EVAC_ABLATE & 16)) {
                if (w.owner) {
                    float* st = (float*)((char*)sm.stage[w.slot][0] + stage_off);
                    *(f4*)(st + 0) = f4{o6[0], o6[1], o6[2], o6[3]};
                    *(f4*)(st + 4) = f4{o6[4], o6[5], o.reward, f_term};
                    st[8] = f_trunc;
                }
                stage_off += kStageRowBytes;
                EVAC_T(w, 14);  // (sub-phase: end-of-episode check, staging of the step's row)
                if (t == flush_t) {
                    if (w.wave_in_env == 0) {   // the wave that staged them: in-order LDS, no barrier needed
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if (fl_s <= t - stage_t0) {
                            const float v = sm.stage[w.slot][fl_s][fl_k];
                            float* dst = &slab_out[((size_t)(stage_t0 + fl_s) * E + w.env) * kGravRow + fl_k];
                            *dst = v;
                        }
                    }
                    stage_off = 0;
                    stage_t0 = t + 1;
                    flush_t = min(t + kStageSteps, n_steps - 1);
                }
            }
        } else {
            if constexpr (!(EVAC_ABLATE & 2)) {
                bool stores = true;             // (team kernels: waves without pedestrians store no observation)
                if constexpr (F::kHelpers) stores = !w.helper;
                if constexpr (F::kPipelined) stores = stores && o.done;   // (... and the row went out inside step_env: only the reset observation of a finished episode is left)
                if (stores) write_obs_generic(p, w.i, active, q, e, StorePlain{rowp});
            }
            if (w.owner && !(EVAC_ABLATE & 16)) {
                rowp[p.obs_dim + 0] = o.reward;
                rowp[p.obs_dim + 1] = f_term;
                rowp[p.obs_dim + 2] = f_trunc;
            }
        }
        if (slot64 == 63) {                  // the actions of the next 64 steps
            asm volatile("");
            if (t + 1 < n_steps) draw_actions(t + 1);
        }
        EVAC_T(w, 7);   // autoreset check, observation epilogue, output stores
    }
    if constexpr (!PERSIST) break;
    cmd_index += 1;
    t_base += n_steps;
    }
#ifdef EVAC_STAMP
    unsigned long long rt1_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1_)::"memory");
    bool stamp_me = w.lane == 0;
    if constexpr (F::kHelpers) stamp_me = stamp_me && !w.helper;   // team kernels: the critical path runs through the ped waves
#ifdef EVAC_STAMP_WAVES   // per wave of workgroup 0, plain stores: the heaviest wave's own phase breakdown (tools/wave_stamps.py)
    if (w.lane == 0 && threadIdx.x == 0) atomicMax(&g_slowest, ((rt1_ - rt0_) << 20) | (unsigned long long)blockIdx.x);
    if (w.lane == 0 && (int)blockIdx.x == g_stamp_block && threadIdx.x < 1024) {
        for (int k = 0; k < 16; ++k) g_wave_stamps[threadIdx.x >> 6][k] = w.stamp.acc[k];
        g_wave_stamps[threadIdx.x >> 6][8] = w.stamp.last - ck0_;
        g_wave_stamps[threadIdx.x >> 6][9] = rt1_ - rt0_;
    }
    stamp_me = false;
#endif
    if (stamp_me) {
        for (int k = 0; k < 8; ++k) atomicAdd(&g_stamps[k], w.stamp.acc[k]);
        for (int k = 12; k < 16; ++k) atomicAdd(&g_stamps[k], w.stamp.acc[k]);   // family-specific sub-phases
        atomicAdd(&g_stamps[8], w.stamp.last - ck0_);
        atomicAdd(&g_stamps[9], rt1_ - rt0_);
        atomicMax(&g_stamps[10], rt1_ - rt0_);                      // slowest / fastest wave of the launch
        atomicMax(&g_stamps[11], ~0ull - (rt1_ - rt0_));
    }
#endif
#ifdef EVAC_STEP_TIMES
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mark_done_)::"memory");
#endif
    if constexpr (F::kEnvBarrier) {               // (multi-wave envs: the load is the column count the last reduction delivered)
        if (moving_out && w.owner) moving_out[w.env] = w.have_next ? w.next_cols : p.n_ped;
    }
    if constexpr (F::kThreadsPerEnv == kWave) {   // what the envs of the next launches are dealt by
        if (moving_out) {   // the length of the env's pair loop: its moving pedestrians, or 0 if no row needs evaluating
            const int nm = ballot(needs_row(p, q.st)) != 0ull ? wave_count((unsigned)(q.st - kViscek) < 3u) : 0;
            if (w.owner) moving_out[w.env] = nm;
        }
    }
    if constexpr (F::kHelpers) {
        if (F::aborted(w)) return;      // a team that lost a member: void results, the env keeps its pre-launch state
    }
    if constexpr (CHAIN) {
        constexpr int T = F::kThreadsPerEnv;
        char* rec = chain.xchg + (size_t)w.env * Xchg<T>::kBytes;
        if constexpr (F::WPE == 1) {
            store_record<T>(rec, w.lane, w.lane, true, active, q, e);
            wait_vmem();                                       // every lane's record stores are acknowledged ...
            if (w.lane == 0) store_dev_i32(rec + Xchg<T>::kGen, chain.gen + 1);      // ... before the next launch's wave may load them
        } else {
            store_record<T>(rec, w.i, w.lane, w.wave_in_env == 0, active, q, e);
            wait_vmem();
            F::sync(w);                                        // ... of EVERY wave of the env
            if (w.wave_in_env == 0 && w.lane == 0) store_dev_i32(rec + Xchg<T>::kGen, chain.gen + 1);
        }
    } else {
        store_env(p, w.env, w.i, active, w.owner, q, e);
        if constexpr (PERSIST) {      // where a kernel started with resume = 1 takes this env up
            if (w.owner) store_dev_i32(persist_next_cmd(chain.xchg) + w.env, cmd_index);
        }
    }
#ifdef EVAC_STEP_TIMES
    if (w.lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100) && threadIdx.x < 1024 && n_steps > 0) {
        unsigned long long mark_exit_;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mark_exit_)::"memory");   // (the state stores have left)
        unsigned long long* m = g_launch_marks[((e.total - 1u) / (uint32_t)n_steps) & 63][blockIdx.x == 0 ? 0 : 1][threadIdx.x >> 6];
        m[0] = mark_entry_;
        m[1] = mark_loop_;
        m[2] = mark_done_;
        m[3] = mark_exit_;
        m[4] = mark_perm_;
        m[5] = mark_init_;
        m[6] = mark_act_;
        m[7] = mark_state_;
    }
    if (threadIdx.x == 0 && blockIdx.x < 256 && n_steps > 0) {
        unsigned long long mark_exit_;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mark_exit_)::"memory");
        unsigned long long* m = g_launch_span[((e.total - 1u) / (uint32_t)n_steps) & 63][blockIdx.x];
        m[0] = mark_entry_;
        m[1] = mark_exit_;
    }
#endif
}

template <class F, bool GRAV>
__global__ __launch_bounds__(F::kBlock, 4) void k_rollout(
    Params p, int n_steps, const float2* __restrict__ actions, float* __restrict__ slab_out,
    evac_episode_stats_t* __restrict__ final_stats, const int* __restrict__ perm, int* __restrict__ moving_out,
    const int* __restrict__ deal_loads, int* __restrict__ deal_perm) {
    __shared__ typename F::Smem sm;
    rollout_body<F, GRAV, false>(sm, p, n_steps, actions, nullptr, slab_out, final_stats, 0, nullptr, nullptr, perm, moving_out,
                                 deal_loads, deal_perm);
}

// The same kernel specialised for the reference's default configuration (what its training scripts and the benchmark run):
// |noise| <= 0.2, enslaving_degree 1, no wall termination, no NaN guard; and either the gravity observation
// with alpha = 3 or the Box observation of relative positions + one-hot statuses.  The options are wave-uniform branches in the generic kernel -- a compare, a
// branch and often a taken jump each, ~50 scalar instructions of a step whose cost for a lone wave is its instruction count
// times ~8 cycles; here they are constants the compiler folds.  Same arithmetic on the path taken: bit-identical results
// (tests/test_gpu_schedule.py runs both).  The host picks it when the handle's configuration matches (evac_create).
template <class F, bool GRAV>
__global__ __launch_bounds__(F::kBlock, 4) void k_rollout_default_config(
    Params p, int n_steps, const float2* __restrict__ actions, float* __restrict__ slab_out,
    evac_episode_stats_t* __restrict__ final_stats, const int* __restrict__ perm, int* __restrict__ moving_out,
    const int* __restrict__ deal_loads, int* __restrict__ deal_perm) {
    __shared__ typename F::Smem sm;
    const Params q = default_config_constants<GRAV>(p);
    rollout_body<F, GRAV, false>(sm, q, n_steps, actions, nullptr, slab_out, final_stats, 0, nullptr, nullptr, perm, moving_out,
                                 deal_loads, deal_perm);
}

// Chained launches (evac_options_t.chain): the same two kernels with the generation hand-off of rollout_body<..., CHAIN>.
template <class F, bool GRAV>
__global__ __launch_bounds__(F::kBlock, 4) void k_rollout_chain(
    Params p, int n_steps, const float2* __restrict__ actions, float* __restrict__ slab_out,
    evac_episode_stats_t* __restrict__ final_stats, const int* __restrict__ perm, int* __restrict__ moving_out,
    const int* __restrict__ deal_loads, int* __restrict__ deal_perm, ChainArgs chain) {
    __shared__ typename F::Smem sm;
    rollout_body<F, GRAV, false, true>(sm, p, n_steps, actions, nullptr, slab_out, final_stats, 0, nullptr, nullptr, perm, moving_out,
                                       deal_loads, deal_perm, chain);
}
template <class F, bool GRAV>
__global__ __launch_bounds__(F::kBlock, 4) void k_rollout_chain_default_config(
    Params p, int n_steps, const float2* __restrict__ actions, float* __restrict__ slab_out,
    evac_episode_stats_t* __restrict__ final_stats, const int* __restrict__ perm, int* __restrict__ moving_out,
    const int* __restrict__ deal_loads, int* __restrict__ deal_perm, ChainArgs chain) {
    __shared__ typename F::Smem sm;
    const Params q = default_config_constants<GRAV>(p);
    rollout_body<F, GRAV, false, true>(sm, q, n_steps, actions, nullptr, slab_out, final_stats, 0, nullptr, nullptr, perm, moving_out,
                                       deal_loads, deal_perm, chain);
}
// One persistent kernel per join (evac_options_t.chain = 2): rollout_body<..., PERSIST> -- n_steps, the slab, the episode records and the
// actions come with every command of the ring at chain.xchg, the first of them command chain.gen.
template <class F, bool GRAV>
__global__ __launch_bounds__(F::kBlock, 4) void k_rollout_persist(
    Params p, const int* __restrict__ perm, int* __restrict__ moving_out, const int* __restrict__ deal_loads, int* __restrict__ deal_perm,
    ChainArgs chain) {
    __shared__ typename F::Smem sm;
    rollout_body<F, GRAV, false, false, true>(sm, p, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, perm, moving_out,
                                              deal_loads, deal_perm, chain);
}
template <class F, bool GRAV>
__global__ __launch_bounds__(F::kBlock, 4) void k_rollout_persist_default_config(
    Params p, const int* __restrict__ perm, int* __restrict__ moving_out, const int* __restrict__ deal_loads, int* __restrict__ deal_perm,
    ChainArgs chain) {
    __shared__ typename F::Smem sm;
    const Params q = default_config_constants<GRAV>(p);
    rollout_body<F, GRAV, false, false, true>(sm, q, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, perm, moving_out,
                                              deal_loads, deal_perm, chain);
}
// (re)start of a chain: every env's generation word (device-scope stores, like the launches' own), and one permutation copied to
// the three other buffers of the four-deep rotation
// one wave per env: the caller's state arrays -> the env's exchange record at generation `gen` (a chain starts) ...
__global__ __launch_bounds__(256) void k_chain_import(Params p, char* __restrict__ xchg, int gen, unsigned* __restrict__ abort_word,
                                                      unsigned long long started_so_far, int wpe) {
    // (one wave per 64 pedestrians of an env: wave `part` of env `env`; T = 64 wpe lanes per env)
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), env = wv / wpe, part = wv - env * wpe, lane = threadIdx.x & 63, T = 64 * wpe;
    if (blockIdx.x == 0 && threadIdx.x < 8) abort_word[threadIdx.x] = 0u;      // (the abort word and its diagnostics)
    // the started-workgroups counter behind them is SET to what the host has enqueued so far -- every one of those workgroups has
    // run: a restart follows a join --, so that a caller who restored a snapshot of the workspace (bench.py's replays) and told the
    // library (evac_reschedule) does not leave the gates waiting for counts that were rolled back.  A gate of the other queue that
    // reads the word before this store sees an older, smaller value: it waits, it cannot pass early.
    if (blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store((unsigned long long*)(abort_word + 8), started_so_far, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (env >= p.n_envs) return;
    const int idx = part * 64 + lane;
    const bool active = idx < p.n_ped;
    Ped q;
    Env e;
    load_env(p, env, idx, active, q, e);
    char* rec = xchg + (size_t)env * xchg_bytes(T);
    const int kXchgStatus = 16 * T, kXchgEnv = 20 * T, kXchgGen = 20 * T + 128;
    // PLAIN stores, made visible by the kernel boundary like any kernel's output.  (`sc1` stores here were wrong: the workspace comes
    // zero-filled by plain stores, and in the first launch after an import ~1.5 of a record's 12 lines read back as zeros -- a copy a
    // plain store left in some XCD's L2 is not refreshed by another XCD's write-through store; tools/chain_debug.py.  Inside the chain
    // a record's reader is its next writer and its `sc1` store drops the line, so no such copy exists: 10^6 chained steps against the
    // plain kernels, with restarts and joins in between, bit for bit -- tools/soak_variants.py.)
    *(f4*)(rec + idx * 16) = active ? f4{q.x, q.y, q.dx, q.dy} : f4{0.0f, 0.0f, 0.0f, 0.0f};
    *(int*)(rec + kXchgStatus + idx * 4) = active ? q.st : 0;
    if (idx == 0) {
        *(f4*)(rec + kXchgEnv) = f4{e.ax, e.ay, e.adx, e.ady};
        *(f4*)(rec + kXchgEnv + 16) = f4{__builtin_bit_cast(float, e.now), __builtin_bit_cast(float, e.n_resets), __builtin_bit_cast(float, (int)e.total), 0.0f};
        *(f4*)(rec + kXchgEnv + 32) = f4{e.acc_ret, e.acc_intr, e.acc_stat, 0.0f};
        *(int*)(rec + kXchgGen) = gen;
    }
}
// ... and back (the caller's stream joins: its arrays are the state again)
__global__ __launch_bounds__(256) void k_chain_export(Params p, const char* __restrict__ xchg, int wpe) {
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), env = wv / wpe, part = wv - env * wpe, lane = threadIdx.x & 63;
    if (env >= p.n_envs) return;
    const int idx = part * 64 + lane;
    const bool active = idx < p.n_ped;
    Ped q;
    Env e;
    if (wpe == 1) load_record<64>(xchg + (size_t)env * Xchg<64>::kBytes, idx, lane, active, q, e);
    else load_record<256>(xchg + (size_t)env * Xchg<256>::kBytes, idx, lane, active, q, e);
    store_env(p, env, idx, active, idx == 0, q, e);
}
__global__ void k_copy_perm3(int n_envs, const int* __restrict__ src, int* __restrict__ a, int* __restrict__ b, int* __restrict__ c) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_envs; e += gridDim.x * blockDim.x) {
        const int v = src[e];
        a[e] = v; b[e] = v; c[e] = v;
    }
}

// The same deal as a launch of its own (one workgroup): the first deal of a handle, and evac_reschedule.
// (perm_other: the second permutation buffer of the schedule, brought to the same deal -- a rollout launch that skips its own
// deal leaves whatever that buffer holds to the launch after it)
__global__ __launch_bounds__(1024) void k_schedule(int n_envs, const int* __restrict__ moving, int* __restrict__ perm, int* __restrict__ perm_other,
                                                   int per_wg, int unit, int mode = 0) {
    __shared__ int hist[kWave + 2];
    schedule_envs_by_workgroup(hist, (int)threadIdx.x, n_envs, moving, perm, per_wg, unit, mode);
    if (perm_other) {
        __syncthreads();                     // (this workgroup's own global stores are visible to it after the barrier)
        for (int e = threadIdx.x; e < n_envs; e += 1024) perm_other[e] = perm[e];
    }
}
template <class F, bool GRAV>
__global__ __launch_bounds__(F::kBlock, 4) void k_rollout_diag(
    Params p, int n_steps, const float2* __restrict__ actions, float2* __restrict__ actions_out,
    float* __restrict__ slab_out, evac_episode_stats_t* __restrict__ final_stats, int capture_envs,
    float* __restrict__ capture, const float* __restrict__ noise_in) {
    __shared__ typename F::Smem sm;
    rollout_body<F, GRAV, true>(sm, p, n_steps, actions, actions_out, slab_out, final_stats, capture_envs, capture, noise_in);
}

template <class F, bool GRAV>
__global__ __launch_bounds__(F::kBlock, 4) void k_reset(Params p, const uint8_t* __restrict__ mask,
                                                      const float4* __restrict__ draws, float* __restrict__ obs_out) {
    __shared__ typename F::Smem sm;
    typename F::Ctx w(sm);
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
    if (obs_out) write_obs<F, GRAV>(p, w, active, true, q, e, obs_out + (size_t)w.env * p.obs_dim);
    store_env(p, w.env, w.i, active, w.owner, q, e);
}

template <class F, bool GRAV>
__global__ __launch_bounds__(F::kBlock, 4) void k_observe(Params p, float* __restrict__ obs_out) {
    __shared__ typename F::Smem sm;
    typename F::Ctx w(sm);
    if (w.env >= p.n_envs) return;
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    write_obs<F, GRAV>(p, w, active, true, q, e, obs_out + (size_t)w.env * p.obs_dim);
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
// The trainer's per-env wrapper chain as a SEPARATE epilogue (rpo_agent.py:24-33): used for resets (evac_norm_reset) and
// kept as the unfused reference of evac_step_normalized (tests/test_gpu_wrappers.py); the statistics themselves are in
// evac_common.h (rms_update1, norm_clip).
// ------------------------------------------------------------------------------------------------
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

// layout guards: the tile and the staging rows are accessed with 16-byte LDS instructions
static_assert(offsetof(Wave<1>::Smem, stage) % 16 == 0 && offsetof(Wave<2>::Smem, stage) % 16 == 0 &&
              offsetof(Wave<4>::Smem, stage) % 16 == 0 && offsetof(Wave<8>::Smem, stage) % 16 == 0 &&
              offsetof(Wave<16>::Smem, stage) % 16 == 0, "stage rows must be 16-byte aligned");
static_assert(offsetof(Cells<2>::Smem, stage) % 16 == 0 && offsetof(Cells<4>::Smem, stage) % 16 == 0 &&
              offsetof(Cells<8>::Smem, stage) % 16 == 0 && offsetof(Cells<16>::Smem, stage) % 16 == 0 &&
              offsetof(Cells<4>::Smem, cnt) % 16 == 0 && offsetof(Cells<4>::Smem, start) % 16 == 0,
              "stage rows and the cell tables must be 16-byte aligned");
static_assert(sizeof(Wave<4>::Smem::tile) == 2 * 256 * 16, "two tiles for the multi-wave all-pairs kernels");
static_assert(offsetof(Wave<1>::Smem, tile) == 0 && alignof(Wave<1>::Smem) >= 16 && alignof(Cells<16>::Smem) >= 16,
              "tile must be 16-byte aligned");

}  // namespace evac
