// Sub-wave kernels for small rooms: G lanes per env (G = 32 for N <= 32, G = 16 for N <= 16), i.e. 2 or 4
// independent envs per 64-lane wave.  The reference's default is number_of_pedestrians = 10
// (src/env/env/config.py:11); with one wave per env 54 of 64 lanes would idle.
//
// Same arithmetic as evac_device.h (the helpers are shared: classify, grav_term2, pair_accumulate,
// noise_sincos, Philox streams, reset_env, write_obs_generic), so results are bit-identical to the
// one-wave-per-env kernels except for the summation trees.  What changes is everything that was
// "wave-uniform" there and is "group-uniform" here:
//   * votes / counts: the 64-bit ballot is masked to the group's lanes (popcount / mbcnt on the masked word);
//   * float sums: DPP row_shr 1/2/4/8 reduce each 16-lane row (= a whole G=16 group), one row_bcast:15 joins
//     the two rows of a G=32 group; the total lands in the group's LAST lane, which therefore owns the
//     per-env outputs (reward, observation, state write-back);
//   * values that were fetched with v_readlane (actions, gravity exit term) come through ds_bpermute from a
//     per-group source lane;
//   * episode ends are per group: the autoreset path runs when ANY group of the wave finished and is merged
//     with selects.
// There is no workgroup barrier anywhere (a wave is in lock-step).
#pragma once

#include "evac_device.h"

namespace evac {

template <int G>
struct SubGeo {
    static_assert(G == 16 || G == 32, "sub-wave groups are 16 or 32 lanes");
    static constexpr int kEnvsPerWave = kWave / G;
    static constexpr int kBlock = 256;
    static constexpr int kEnvsPerBlock = (kBlock / kWave) * kEnvsPerWave;
};

template <int G>
struct SmemSub {
    f4 tile[SubGeo<G>::kEnvsPerBlock][G];   // per env: moving pedestrians first, then zero-weight padding
};

template <int G>
struct WhoSub {
    int env, slot, lane, sub, i;
    unsigned long long gmask;   // this group's lanes in a 64-bit ballot
    __device__ __forceinline__ WhoSub() {
        const int t = threadIdx.x;
        lane = t & (kWave - 1);
        sub = lane / G;
        i = lane - sub * G;
        slot = (t / kWave) * SubGeo<G>::kEnvsPerWave + sub;
        env = blockIdx.x * SubGeo<G>::kEnvsPerBlock + slot;
        gmask = (G == 32 ? 0xffffffffull : 0xffffull) << (sub * G);
    }
};

__device__ __forceinline__ int group_count(unsigned long long m, unsigned long long gmask) { return __popcll(m & gmask); }
// number of set lanes of the group below this lane
__device__ __forceinline__ int group_rank(unsigned long long m, unsigned long long gmask) {
    const unsigned long long g = m & gmask;
    return __builtin_amdgcn_mbcnt_hi((unsigned)(g >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)g, 0u));
}
// Sum over the group; valid in the group's last lane (i == G-1).  Three chains interleaved (see wave_sum3).
template <int G>
__device__ __forceinline__ void group_sum3(float& a, float& b, float& c) {
#define EVAC_DPP3(CTRL, MASK) a = dpp_add<CTRL, MASK>(a); b = dpp_add<CTRL, MASK>(b); c = dpp_add<CTRL, MASK>(c);
    EVAC_DPP3(0x111, 0xf)
    EVAC_DPP3(0x112, 0xf)
    EVAC_DPP3(0x114, 0xf)
    EVAC_DPP3(0x118, 0xf)
    if constexpr (G == 32) { EVAC_DPP3(0x142, 0xa) }
#undef EVAC_DPP3
}
// value held by lane `src_i` of this lane's group
__device__ __forceinline__ float group_fetch(float v, int sub_base, int src_i) { return __shfl(v, sub_base + src_i, kWave); }

// grav observation of the current state: valid in the group's last lane
template <int G>
__device__ __forceinline__ void grav_observation_sub(const Params& p, const WhoSub<G>& w, bool active, const Ped& q,
                                                     const Env& e, float (&o6)[6]) {
    float gx, gy, zero = 0.0f;
    grav_term(p, e.ax - q.x, e.ay - q.y, gx, gy);
    const bool visc = active && q.st == kViscek;
    float sgx = visc ? gx : 0.0f, sgy = visc ? gy : 0.0f;
    group_sum3<G>(sgx, sgy, zero);
    const float nf = (float)group_count(__ballot(active && q.st == kFollower), w.gmask);
    float ex, ey;
    grav_term(p, e.ax - kExitX, e.ay - kExitY, ex, ey);
    o6[0] = e.ax; o6[1] = e.ay; o6[2] = ex * nf; o6[3] = ey * nf; o6[4] = sgx; o6[5] = sgy;
}

// One env step for every group of the wave (env.py:141-171); see step_env in evac_device.h for the
// line-by-line references -- the body is the same sequence of operations.
template <int G, bool GRAV>
__device__ __forceinline__ void step_env_sub(const Params& p, SmemSub<G>& sm, const WhoSub<G>& w, bool active, Ped& q,
                                             Env& e, float2 adir, float noise, StepOut& out) {
    const int i = w.i;
    e.now += 1;
    e.total += 1u;
    out.truncated = e.now >= p.max_timesteps;

    e.adx = adir.x;
    e.ady = adir.y;
    const float tx = e.ax + e.adx, ty = e.ay + e.ady;
    const bool hit = fabsf(tx) > p.width || fabsf(ty) > p.height;
    e.ax = hit ? e.ax : tx;
    e.ay = hit ? e.ay : ty;
    const float r_agent = hit ? -5.0f : 0.0f;
    const bool term_agent = hit && (p.flags & kFlagTermOnWall) != 0;

    const int old_st = q.st;
    const bool esc = q.st == kEscaped, exi = q.st == kExiting;
    q.x = esc ? kExitX : q.x;
    q.y = esc ? kExitY : q.y;
    q.dx = esc ? 0.0f : q.dx;
    q.dy = esc ? 0.0f : q.dy;
    if (__ballot(exi) != 0ull) {
        const float vx = kExitX - q.x, vy = kExitY - q.y;
        const float l2 = vx * vx + vy * vy;
        const float il = frsq(l2);
        const float ln = l2 * il;
        const float sz = ln > p.step_size ? p.step_size : ln;
        const float k = il * sz;
        q.dx = exi ? vx * k : q.dx;
        q.dy = exi ? vy * k : q.dy;
    }
    const bool efv = (unsigned)(q.st - kViscek) < 3u;       // V | F | E; lanes beyond n_ped carry status 0
    const bool fv = (unsigned)(q.st - kViscek) < 2u;        // V | F
    const bool fol = q.st == kFollower;

    const float inrm = frsq(q.dx * q.dx + q.dy * q.dy);
    float ux = q.dx * inrm, uy = q.dy * inrm;
    if (p.flags & kFlagNanGuard) {
        ux = (ux != ux) ? 0.0f : ux;
        uy = (uy != uy) ? 0.0f : uy;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // tile readers of the previous step are done
    __builtin_amdgcn_wave_barrier();
    const unsigned long long m_efv = __ballot(efv);
    const int n_cols = group_count(m_efv, w.gmask);
    {
        const int before = group_rank(m_efv, w.gmask);
        const int idx = efv ? before : n_cols + (i - before);     // a bijection onto the group's G slots
        sm.tile[w.slot][idx] = f4{efv ? q.x * kTileScale : __builtin_inff(), q.y * kTileScale, efv ? ux : 0.0f, efv ? uy : 0.0f};
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    float sx = 0.0f, sy = 0.0f;
    {
        // the loop runs to the largest column count of the wave's groups (wave-uniform); the padding entries
        // of smaller groups weigh 0
        int nmax = 0;
#pragma unroll
        for (int g = 0; g < SubGeo<G>::kEnvsPerWave; ++g) {
            const int c = __popcll((m_efv >> (g * G)) & (G == 32 ? 0xffffffffull : 0xffffull));
            nmax = c > nmax ? c : nmax;
        }
        const int n4 = (__ballot(fv) != 0ull) ? ((nmax + 3) & ~3) : 0;
        const f4* __restrict__ tile = sm.tile[w.slot];           // per lane: its group's tile
        const float XI = q.x * kTileScale, YI = q.y * kTileScale;
        int j = 0;
        for (; j + 8 <= n4; j += 8) {
            f4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = tile[j + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) pair_accumulate(XI, YI, t[k], kRPed2Big, sx, sy);
        }
        for (; j < n4; j += 4) {
            f4 t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = tile[j + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) pair_accumulate(XI, YI, t[k], kRPed2Big, sx, sy);
        }
    }
    {
        const bool zero_mean = sx == 0.0f && sy == 0.0f;
        const float il = frsq(sx * sx + sy * sy);
        const float cx = zero_mean ? 1.0f : sx * il;
        const float cy = zero_mean ? 0.0f : sy * il;
        float sn, cs;
        noise_sincos(noise, p.small_noise, sn, cs);
        const float ndx = (cx * cs - cy * sn) * p.step_size;
        const float ndy = (cy * cs + cx * sn) * p.step_size;
        q.dx = fv ? ndx : q.dx;
        q.dy = fv ? ndy : q.dy;
        const float bdx = p.ens * e.adx + p.one_minus_ens * q.dx;
        const float bdy = p.ens * e.ady + p.one_minus_ens * q.dy;
        q.dx = fol ? bdx : q.dx;
        q.dy = fol ? bdy : q.dy;
        q.x += efv ? q.dx : 0.0f;
        q.y += efv ? q.dy : 0.0f;
    }
    {
        const float cx = __builtin_amdgcn_fmed3f(q.x, -p.width, p.width);
        const float cy = __builtin_amdgcn_fmed3f(q.y, -p.height, p.height);
        const float mx = q.x - cx, my = q.y - cy;
        q.x = fmaf(-2.0f, mx, q.x);
        q.y = fmaf(-2.0f, my, q.y);
        q.dx = (mx != 0.0f) ? -q.dx : q.dx;
        q.dy = (my != 0.0f) ? -q.dy : q.dy;
    }

    const bool has_idle = p.n_ped < G;                      // wave-uniform
    const bool exit_lane = GRAV && has_idle && i == p.n_ped;
    const float px = exit_lane ? kExitX : q.x, py = exit_lane ? kExitY : q.y;
    float de, lx, ly, dl2;
    const int cls = classify(p, px, py, e.ax, e.ay, de, lx, ly, dl2);
    const int new_st = active ? cls : 0;
    q.st = new_st;
    float s0 = active ? de : 0.0f, s1 = 0.0f, s2 = 0.0f, gx = 0.0f, gy = 0.0f;
    if constexpr (GRAV) {
        grav_term2(p, -lx, -ly, dl2, gx, gy);
        const bool visc = new_st == kViscek;
        s1 = visc ? gx : 0.0f;
        s2 = visc ? gy : 0.0f;
    }
    group_sum3<G>(s0, s1, s2);                               // totals in the group's last lane
    const int n_new_exit = group_count(__ballot((old_st == kViscek || old_st == kFollower) && new_st == kExiting), w.gmask);
    const int n_new_fol = group_count(__ballot(old_st == kViscek && new_st == kFollower), w.gmask);
    const int n_esc = group_count(__ballot(new_st == kEscaped), w.gmask);
    const int n_fol = group_count(__ballot(new_st == kFollower), w.gmask);
    if constexpr (GRAV) {
        float ex, ey;
        if (has_idle) {
            ex = group_fetch(gx, w.sub * G, p.n_ped);
            ey = group_fetch(gy, w.sub * G, p.n_ped);
        } else {
            grav_term(p, e.ax - kExitX, e.ay - kExitY, ex, ey);
        }
        const float nf = (float)n_fol;
        out.ex = ex * nf;
        out.ey = ey * nf;
        out.gx = s1;
        out.gy = s2;
    }
    out.n_escaped = n_esc;
    out.n_follower = n_fol;
    out.n_exiting = out.n_viscek = 0;

    const float tf = 1.0f - (float)e.now * p.inv_200n;
    float r_ped = p.init_reward;
    r_ped += (p.flags & kFlagNewExitingReward) ? (15.0f + 10.0f * tf) * (float)n_new_exit : 0.0f;
    r_ped += (p.flags & kFlagNewFollowersReward) ? (10.0f + 5.0f * tf) * (float)n_new_fol : 0.0f;
    const float intrinsic = 0.0f - s0 * p.inv_n;              // (valid in the owner lane)
    out.reward = r_agent + r_ped + p.intrinsic_coef * intrinsic;
    out.terminated = term_agent || (n_esc == p.n_ped);        // group-uniform: counts come from ballots
    e.acc_ret += out.reward;
    e.acc_intr += intrinsic;
    e.acc_stat += r_agent + r_ped;
}

// state <-> HBM: per-env words are written by the group's OWNER lane (the last one), whose copies of the
// float sums are the valid ones
template <int G>
__device__ __forceinline__ void store_env_sub(const Params& p, const WhoSub<G>& w, bool active, bool valid, const Ped& q,
                                              const Env& e) {
    if (valid && active) {
        p.ped[(size_t)w.env * p.n_ped + w.i] = make_float4(q.x, q.y, q.dx, q.dy);
        p.status[(size_t)w.env * p.n_ped + w.i] = (uint8_t)q.st;
    }
    if (valid && w.i == G - 1) {
        p.agent[w.env] = make_float4(e.ax, e.ay, e.adx, e.ady);
        p.clock[w.env] = make_int4(e.now, e.n_resets, (int)e.total, 0);
        p.acc[w.env] = make_float4(e.acc_ret, e.acc_intr, e.acc_stat, 0.0f);
    }
}

// Episode end for the groups with `done` set: episode record, Philox reset, fresh observation; merged by selects.
template <int G, bool GRAV>
__device__ __forceinline__ void autoreset_sub(const Params& p, const WhoSub<G>& w, bool active, bool done, uint32_t gid,
                                              Ped& q, Env& e, StepOut& o, float (&o6)[6], evac_episode_stats_t* stats_row) {
    o.n_exiting = group_count(__ballot(q.st == kExiting), w.gmask);
    o.n_viscek = group_count(__ballot(q.st == kViscek), w.gmask);
    if (done && stats_row && w.i == G - 1) write_stats(stats_row, e, o);
    Ped nq = q;
    Env ne = e;
    reset_env(p, active, philox_reset_draw(p, gid, w.i, e.n_resets), nq, ne);
    float n6[6] = {ne.ax, ne.ay, 0.f, 0.f, 0.f, 0.f};
    if constexpr (GRAV) grav_observation_sub<G>(p, w, active, nq, ne, n6);
    if (done) {
        q = nq;
        e = ne;
#pragma unroll
        for (int k = 0; k < 6; ++k) o6[k] = n6[k];
    }
}

template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_step_sub(
    Params p, const float2* __restrict__ actions, const float* __restrict__ noise_in, float* __restrict__ obs_out,
    float* __restrict__ reward_out, uint8_t* __restrict__ term_out, uint8_t* __restrict__ trunc_out, int autoreset,
    float* __restrict__ final_obs, evac_episode_stats_t* __restrict__ final_stats) {
    __shared__ SmemSub<G> sm;
    WhoSub<G> w;
    const bool valid = w.env < p.n_envs;
    if (!valid) w.env = p.n_envs - 1;          // idle groups shadow the last env (same wave must stay converged); no stores
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    const uint32_t gid = p.env_id_offset + (uint32_t)w.env;
    const float2 a = actions[w.env];
    float nz = 0.0f;
    if (active) nz = noise_in ? noise_in[(size_t)w.env * p.n_ped + w.i] : philox_noise(p, gid, w.i, e.total);
    StepOut o;
    step_env_sub<G, GRAV>(p, sm, w, active, q, e, agent_direction(p, a.x, a.y), nz, o);
    const bool done = (o.terminated || o.truncated) && autoreset;
    const bool owner = valid && w.i == G - 1;
    float o6[6] = {e.ax, e.ay, o.ex, o.ey, o.gx, o.gy};
    if (__ballot(done) != 0ull) {              // wave-uniform: some group finished
        if (done && valid && final_obs) {
            float* fo = final_obs + (size_t)w.env * p.obs_dim;
            if constexpr (GRAV) {
                if (owner) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) fo[k] = o6[k];
                }
            } else {
                write_obs_generic(p, w.i, active, q, e, fo);
            }
        }
        autoreset_sub<G, GRAV>(p, w, active, done, gid, q, e, o, o6, (final_stats && valid) ? final_stats + w.env : nullptr);
    }
    if (valid) {
        float* obs = obs_out + (size_t)w.env * p.obs_dim;
        if constexpr (GRAV) {
            if (owner) {
#pragma unroll
                for (int k = 0; k < 6; ++k) obs[k] = o6[k];
            }
        } else {
            write_obs_generic(p, w.i, active, q, e, obs);
        }
    }
    store_env_sub<G>(p, w, active, valid, q, e);
    if (owner) {
        reward_out[w.env] = o.reward;
        term_out[w.env] = o.terminated ? 1 : 0;
        trunc_out[w.env] = o.truncated ? 1 : 0;
    }
}

template <int G, bool GRAV, bool CAPTURE>
__device__ __forceinline__ void rollout_body_sub(SmemSub<G>& sm, const Params& p, int n_steps,
                                                 const float2* __restrict__ actions, float2* __restrict__ actions_out,
                                                 float* __restrict__ slab_out, evac_episode_stats_t* __restrict__ final_stats,
                                                 int capture_envs, float* __restrict__ capture) {
    WhoSub<G> w;
    const bool valid = w.env < p.n_envs;
    if (!valid) w.env = p.n_envs - 1;
    const bool active = w.i < p.n_ped;
    const bool owner = valid && w.i == G - 1;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    const uint32_t gid = p.env_id_offset + (uint32_t)w.env;
    const size_t E = (size_t)p.n_envs;
    const int row = p.obs_dim + 3;
    uint4 nzr = make_uint4(0, 0, 0, 0);
    bool have = false;
    float2 lane_act = make_float2(0.f, 0.f), lane_adir = make_float2(0.f, 0.f);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep the state loads' wait out of the loop
    for (int t = 0; t < n_steps; ++t) {
        const int slotG = t & (G - 1);
        if (slotG == 0) {                 // actions of the next G steps, one step per lane of the group
            if (actions) {
                if (t + w.i < n_steps) lane_act = actions[(size_t)(t + w.i) * E + w.env];
            } else {
                lane_act = philox_action(p, gid, e.total + (uint32_t)w.i);
            }
            lane_adir = agent_direction(p, lane_act.x, lane_act.y);
        }
        float2 adir;
        adir.x = group_fetch(lane_adir.x, w.sub * G, slotG);
        adir.y = group_fetch(lane_adir.y, w.sub * G, slotG);
        if constexpr (CAPTURE) {
            if (actions_out) {
                const float ax = group_fetch(lane_act.x, w.sub * G, slotG), ay = group_fetch(lane_act.y, w.sub * G, slotG);
                if (owner) actions_out[(size_t)t * E + w.env] = make_float2(ax, ay);
            }
        }
        const uint32_t sel = e.total & 3u;
        if (!have || sel == 0u) {
            nzr = philox4x32_10(make_uint4(gid, (uint32_t)w.i, e.total >> 2, kStreamNoise), p.seed_lo, p.seed_hi);
            have = true;
        }
        const uint32_t wsel = sel == 0 ? nzr.x : (sel == 1 ? nzr.y : (sel == 2 ? nzr.z : nzr.w));
        const float nz = (u01(wsel) - 0.5f) * p.noise_coef;
        StepOut o;
        step_env_sub<G, GRAV>(p, sm, w, active, q, e, adir, nz, o);
        if (CAPTURE && capture && valid && w.env < capture_envs) {
            float* cp = capture + (((size_t)t * capture_envs + w.env) * (p.n_ped + 1)) * 3;
            if (active) {
                cp[3 * w.i + 0] = q.x;
                cp[3 * w.i + 1] = q.y;
                cp[3 * w.i + 2] = (float)q.st;
            }
            if (w.i == G - 1) {
                cp[3 * p.n_ped + 0] = e.ax;
                cp[3 * p.n_ped + 1] = e.ay;
                cp[3 * p.n_ped + 2] = 0.0f;
            }
        }
        float o6[6] = {e.ax, e.ay, o.ex, o.ey, o.gx, o.gy};
        const bool done = o.terminated || o.truncated;
        if (__ballot(done) != 0ull)
            autoreset_sub<G, GRAV>(p, w, active, done, gid, q, e, o, o6,
                                   (final_stats && valid) ? final_stats + (size_t)t * E + w.env : nullptr);
        if (valid) {
            float* rowp = slab_out + ((size_t)t * E + w.env) * row;
            if constexpr (GRAV) {
                if (owner) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) rowp[k] = o6[k];
                }
            } else {
                write_obs_generic(p, w.i, active, q, e, rowp);
            }
            if (owner) {
                rowp[p.obs_dim + 0] = o.reward;
                rowp[p.obs_dim + 1] = o.terminated ? 1.0f : 0.0f;
                rowp[p.obs_dim + 2] = o.truncated ? 1.0f : 0.0f;
            }
        }
    }
    store_env_sub<G>(p, w, active, valid, q, e);
}

template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_rollout_sub(Params p, int n_steps, const float2* __restrict__ actions,
                                                     float* __restrict__ slab_out,
                                                     evac_episode_stats_t* __restrict__ final_stats) {
    __shared__ SmemSub<G> sm;
    rollout_body_sub<G, GRAV, false>(sm, p, n_steps, actions, nullptr, slab_out, final_stats, 0, nullptr);
}
template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_rollout_capture_sub(Params p, int n_steps, const float2* __restrict__ actions,
                                                             float2* __restrict__ actions_out, float* __restrict__ slab_out,
                                                             evac_episode_stats_t* __restrict__ final_stats,
                                                             int capture_envs, float* __restrict__ capture) {
    __shared__ SmemSub<G> sm;
    rollout_body_sub<G, GRAV, true>(sm, p, n_steps, actions, actions_out, slab_out, final_stats, capture_envs, capture);
}

template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_reset_sub(Params p, const uint8_t* __restrict__ mask, const float4* __restrict__ draws,
                                                   float* __restrict__ obs_out) {
    WhoSub<G> w;
    bool valid = w.env < p.n_envs;
    if (!valid) w.env = p.n_envs - 1;
    if (mask && !mask[w.env]) valid = false;                 // masked-out groups compute along but store nothing
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    const uint32_t gid = p.env_id_offset + (uint32_t)w.env;
    float4 d = make_float4(0.f, 0.f, 1.f, 0.f);
    if (active) d = draws ? draws[(size_t)w.env * p.n_ped + w.i] : philox_reset_draw(p, gid, w.i, e.n_resets);
    reset_env(p, active, d, q, e);
    if (obs_out) {
        float* obs = obs_out + (size_t)w.env * p.obs_dim;
        if constexpr (GRAV) {
            float o6[6];
            grav_observation_sub<G>(p, w, active, q, e, o6);
            if (valid && w.i == G - 1) {
#pragma unroll
                for (int k = 0; k < 6; ++k) obs[k] = o6[k];
            }
        } else if (valid) {
            write_obs_generic(p, w.i, active, q, e, obs);
        }
    }
    store_env_sub<G>(p, w, active, valid, q, e);
}

template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_observe_sub(Params p, float* __restrict__ obs_out) {
    WhoSub<G> w;
    const bool valid = w.env < p.n_envs;
    if (!valid) w.env = p.n_envs - 1;
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    float* obs = obs_out + (size_t)w.env * p.obs_dim;
    if constexpr (GRAV) {
        float o6[6];
        grav_observation_sub<G>(p, w, active, q, e, o6);
        if (valid && w.i == G - 1) {
#pragma unroll
            for (int k = 0; k < 6; ++k) obs[k] = o6[k];
        }
    } else if (valid) {
        write_obs_generic(p, w.i, active, q, e, obs);
    }
}

}  // namespace evac
