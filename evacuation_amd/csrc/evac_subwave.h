// Kernels of the sub-wave family Sub<G> (evac_families.h): G lanes per env (G = 32 for N <= 32, G = 16 for
// N <= 16), i.e. 2 or 4 independent envs per 64-lane wave.  The step itself is the common step_env
// (evac_device.h); what is specific here is the kernel scaffolding of a wave that carries several envs:
//   * groups beyond the last env shadow the last env (the wave must stay converged) and store nothing;
//   * the group's LAST lane owns the per-env outputs (the DPP sums are valid there);
//   * episode ends are per group: the autoreset path runs when ANY group of the wave finished and is merged
//     with selects.
#pragma once

#include "evac_device.h"

namespace evac {

// Episode end for the groups with `done` set: episode record, Philox reset, fresh observation; merged by selects.
template <int G, bool GRAV>
__device__ __forceinline__ void autoreset_sub(const Params& p, typename Sub<G>::Ctx& w, bool active, bool done, uint32_t gid,
                                              Ped& q, Env& e, StepOut& o, float (&o6)[6], evac_episode_stats_t* stats_row) {
    finish_counts<Sub<G>>(p, w, q, o);
    if (done && stats_row && w.owner) write_stats(stats_row, e, o);
    Ped nq = q;
    Env ne = e;
    reset_env(p, active, philox_reset_draw(p, gid, w.i, e.n_resets), nq, ne);
    float n6[6] = {ne.ax, ne.ay, 0.f, 0.f, 0.f, 0.f};
    if constexpr (GRAV) grav_observation<Sub<G>>(p, w, active, nq, ne, n6);
    if (done) {
        q = nq;
        e = ne;
#pragma unroll
        for (int k = 0; k < 6; ++k) o6[k] = n6[k];
    }
}

template <int G, bool GRAV, bool NORM>
__device__ __forceinline__ void step_kernel_body_sub(
    typename Sub<G>::Smem& sm, const Params& p, const float2* __restrict__ actions, const float* __restrict__ noise_in,
    float* __restrict__ obs_out, float* __restrict__ reward_out, uint8_t* __restrict__ term_out,
    uint8_t* __restrict__ trunc_out, int autoreset, float* __restrict__ final_obs,
    evac_episode_stats_t* __restrict__ final_stats, const NormArgs& na) {
    using F = Sub<G>;
    typename F::Ctx w(sm);
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
    step_env<F, GRAV>(p, w, active, q, e, agent_direction(p, a.x, a.y), nz, o);
    const bool done = (o.terminated || o.truncated) && autoreset;
    const bool owner = valid && w.owner;
    const int D = p.obs_dim;
    double* ns = NORM ? na.state + (size_t)w.env * (3 * D + 4) : nullptr;   // the trainer's normalisation chain, fused (see step_outputs)
    float o6[6] = {e.ax, e.ay, o.ex, o.ey, o.gx, o.gy};
    if (ballot(done) != 0ull) {              // wave-uniform: some group finished
        if (done && valid && final_obs) {
            float* fo = final_obs + (size_t)w.env * D;
            if constexpr (GRAV) {
                if (owner) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        if constexpr (NORM) StoreNorm{fo, ns, D, na.eps, na.obs_clip}(k, o6[k]);
                        else fo[k] = o6[k];
                    }
                }
            } else {
                if constexpr (NORM) write_obs_generic(p, w.i, active, q, e, StoreNorm{fo, ns, D, na.eps, na.obs_clip});
                else write_obs_generic(p, w.i, active, q, e, StorePlain{fo});
            }
        }
        autoreset_sub<G, GRAV>(p, w, active, done, gid, q, e, o, o6, (final_stats && valid) ? final_stats + w.env : nullptr);
    }
    if (valid) {
        float* obs = obs_out + (size_t)w.env * D;
        if constexpr (GRAV) {
            if (owner) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    if constexpr (NORM) StoreNorm{obs, ns, D, na.eps, na.obs_clip}(k, o6[k]);
                    else obs[k] = o6[k];
                }
            }
        } else {
            if constexpr (NORM) write_obs_generic(p, w.i, active, q, e, StoreNorm{obs, ns, D, na.eps, na.obs_clip});
            else write_obs_generic(p, w.i, active, q, e, StorePlain{obs});
        }
    }
    store_env(p, w.env, w.i, valid && active, owner, q, e);
    if (owner) {
        float r = o.reward;
        if constexpr (NORM) {
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
#define EVAC_STEP_KERNEL_SUB(NAME, NORM_)                                                                                       \
    template <int G, bool GRAV>                                                                                                \
    __global__ __launch_bounds__(256) void NAME(                                                                                \
        Params p, const float2* __restrict__ actions, const float* __restrict__ noise_in, float* __restrict__ obs_out,          \
        float* __restrict__ reward_out, uint8_t* __restrict__ term_out, uint8_t* __restrict__ trunc_out, int autoreset,         \
        float* __restrict__ final_obs, evac_episode_stats_t* __restrict__ final_stats, NormArgs na) {                           \
        __shared__ typename Sub<G>::Smem sm;                                                                                    \
        step_kernel_body_sub<G, GRAV, NORM_>(sm, p, actions, noise_in, obs_out, reward_out, term_out, trunc_out, autoreset,      \
                                             final_obs, final_stats, na);                                                       \
    }
EVAC_STEP_KERNEL_SUB(k_step_raw_sub, false)
EVAC_STEP_KERNEL_SUB(k_step_norm_sub, true)
#undef EVAC_STEP_KERNEL_SUB
#define EVAC_STEP_KERNEL_SUB_DEFAULT(NAME, NORM_)                                                                               \
    template <int G, bool GRAV>                                                                                                \
    __global__ __launch_bounds__(256) void NAME(                                                                                \
        Params p, const float2* __restrict__ actions, const float* __restrict__ noise_in, float* __restrict__ obs_out,          \
        float* __restrict__ reward_out, uint8_t* __restrict__ term_out, uint8_t* __restrict__ trunc_out, int autoreset,         \
        float* __restrict__ final_obs, evac_episode_stats_t* __restrict__ final_stats, NormArgs na) {                           \
        __shared__ typename Sub<G>::Smem sm;                                                                                    \
        const Params q = default_config_constants<GRAV>(p);                                                                     \
        step_kernel_body_sub<G, GRAV, NORM_>(sm, q, actions, noise_in, obs_out, reward_out, term_out, trunc_out, autoreset,      \
                                             final_obs, final_stats, na);                                                       \
    }
EVAC_STEP_KERNEL_SUB_DEFAULT(k_step_default_config_sub, false)
EVAC_STEP_KERNEL_SUB_DEFAULT(k_step_norm_default_config_sub, true)
#undef EVAC_STEP_KERNEL_SUB_DEFAULT

template <int G, bool GRAV, bool DIAG>
__device__ __forceinline__ void rollout_body_sub(typename Sub<G>::Smem& sm, const Params& p, int n_steps,
                                                 const float2* __restrict__ actions, float2* __restrict__ actions_out,
                                                 float* __restrict__ slab_out, evac_episode_stats_t* __restrict__ final_stats,
                                                 int capture_envs, float* __restrict__ capture,
                                                 const float* __restrict__ noise_in) {
    using F = Sub<G>;
    typename F::Ctx w(sm);
    const bool valid = w.env < p.n_envs;
    if (!valid) w.env = p.n_envs - 1;
    const bool active = w.i < p.n_ped;
    const bool owner = valid && w.owner;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    const uint32_t gid = p.env_id_offset + (uint32_t)w.env;
    const size_t E = (size_t)p.slab_envs;
    const int row = p.obs_dim + 3;
    uint4 nzr = make_uint4(0, 0, 0, 0);
    bool have = false;
    float2 lane_act = make_float2(0.f, 0.f), lane_adir = make_float2(0.f, 0.f);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): keep the state loads' wait out of the loop
    const int prio_slot = simd_wave_slot();
    for (int t = 0; t < n_steps; ++t) {
        if (p.fair) set_wave_priority(t + prio_slot);   // even progress of the waves of a SIMD (evac_device.h)
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
        adir.x = F::fetch(lane_adir.x, w.sub * G, slotG);
        adir.y = F::fetch(lane_adir.y, w.sub * G, slotG);
        if constexpr (DIAG) {
            if (actions_out) {
                const float ax = F::fetch(lane_act.x, w.sub * G, slotG), ay = F::fetch(lane_act.y, w.sub * G, slotG);
                if (owner) actions_out[(size_t)t * E + w.env] = make_float2(ax, ay);
            }
        }
        const uint32_t sel = e.total & 3u;
        if (!have || sel == 0u) {
            nzr = philox4x32_10(make_uint4(gid, (uint32_t)w.i, e.total >> 2, kStreamNoise), p.seed_lo, p.seed_hi);
            have = true;
        }
        const uint32_t wsel = sel == 0 ? nzr.x : (sel == 1 ? nzr.y : (sel == 2 ? nzr.z : nzr.w));
        float nz = (u01(wsel) - 0.5f) * p.noise_coef;
        if constexpr (DIAG) {
            if (noise_in) nz = active ? noise_in[((size_t)t * E + w.env) * p.n_ped + w.i] : 0.0f;
        }
        StepOut o;
        step_env<F, GRAV>(p, w, active, q, e, adir, nz, o);
        if (DIAG && capture && valid && w.env < capture_envs) {
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
        const bool done = o.terminated || o.truncated;
        if (ballot(done) != 0ull)
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
                write_obs_generic(p, w.i, active, q, e, StorePlain{rowp});
            }
            if (owner) {
                rowp[p.obs_dim + 0] = o.reward;
                rowp[p.obs_dim + 1] = o.terminated ? 1.0f : 0.0f;
                rowp[p.obs_dim + 2] = o.truncated ? 1.0f : 0.0f;
            }
        }
    }
    store_env(p, w.env, w.i, valid && active, owner, q, e);
}

template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_rollout_sub(Params p, int n_steps, const float2* __restrict__ actions,
                                                     float* __restrict__ slab_out,
                                                     evac_episode_stats_t* __restrict__ final_stats, const int*, int*, const int*, int*) {
    __shared__ typename Sub<G>::Smem sm;
    rollout_body_sub<G, GRAV, false>(sm, p, n_steps, actions, nullptr, slab_out, final_stats, 0, nullptr, nullptr);
}
template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_rollout_default_config_sub(Params p, int n_steps, const float2* __restrict__ actions,
                                                                    float* __restrict__ slab_out,
                                                                    evac_episode_stats_t* __restrict__ final_stats, const int*, int*, const int*, int*) {
    __shared__ typename Sub<G>::Smem sm;
    const Params q = default_config_constants<GRAV>(p);
    rollout_body_sub<G, GRAV, false>(sm, q, n_steps, actions, nullptr, slab_out, final_stats, 0, nullptr, nullptr);
}
template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_rollout_diag_sub(Params p, int n_steps, const float2* __restrict__ actions,
                                                          float2* __restrict__ actions_out, float* __restrict__ slab_out,
                                                          evac_episode_stats_t* __restrict__ final_stats,
                                                          int capture_envs, float* __restrict__ capture,
                                                          const float* __restrict__ noise_in) {
    __shared__ typename Sub<G>::Smem sm;
    rollout_body_sub<G, GRAV, true>(sm, p, n_steps, actions, actions_out, slab_out, final_stats, capture_envs, capture, noise_in);
}

template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_reset_sub(Params p, const uint8_t* __restrict__ mask, const float4* __restrict__ draws,
                                                   float* __restrict__ obs_out) {
    using F = Sub<G>;
    __shared__ typename F::Smem sm;
    typename F::Ctx w(sm);
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
    if (obs_out) write_obs<F, GRAV>(p, w, active, valid, q, e, obs_out + (size_t)w.env * p.obs_dim);
    store_env(p, w.env, w.i, valid && active, valid && w.owner, q, e);
}

template <int G, bool GRAV>
__global__ __launch_bounds__(256) void k_observe_sub(Params p, float* __restrict__ obs_out) {
    using F = Sub<G>;
    __shared__ typename F::Smem sm;
    typename F::Ctx w(sm);
    const bool valid = w.env < p.n_envs;
    if (!valid) w.env = p.n_envs - 1;
    const bool active = w.i < p.n_ped;
    Ped q;
    Env e;
    load_env(p, w.env, w.i, active, q, e);
    write_obs<F, GRAV>(p, w, active, valid, q, e, obs_out + (size_t)w.env * p.obs_dim);
}

}  // namespace evac
