// Device code of libevac, part 3b (included by evac_device.h between the step body and the rollout scaffolding):
// PACKED rollouts of one-wave envs.
//
// Late in an episode most pedestrians of an env have escaped -- under a RandomAgent 48 of 60 at t = 1000 -- and escaped
// pedestrians are inert (pinned to the exit, direction 0: area.py:79-81; they count in no sum but the termination test).
// A 64-lane wave then spends the whole O(N) part of the step on 12 live lanes.  So at the start of a rollout launch an env
// whose moving pedestrians fit 32 lanes hands them -- compacted in ascending pedestrian order, with their ids -- to a
// 32-lane half of a wave, and two such envs of a workgroup share ONE wave running the step body of the sub-wave family
// (Sub<32>); the other wave of the pair exits and leaves its SIMD to the rest.  Same pedestrians, same column order of the
// neighbour sum, same Philox counters (keyed by env and pedestrian id): trajectories, statuses, flags and rewards are
// bit-identical to the unpacked kernel; the per-env float sums (gravity observation, intrinsic reward) are taken over
// different lanes and agree to f32 rounding.
//
// ELIGIBLE is a property of the env and the launch alone, so that results never depend on what else is in the batch:
//   * at most 32 moving pedestrians;
//   * no episode end inside the launch: now + T < max_timesteps (truncation), termination at a wall switched off, and some
//     moving pedestrian farther from the exit than T steps can carry it (|step| <= step_size: area.py:136-145, 189-192) --
//     so the env cannot terminate (area.py:175-178) and the packed loop never has to reset (60 fresh pedestrians would not
//     fit the half wave);
//   * gravity observation (the generic observations list every pedestrian, also the escaped ones).
// An eligible env without a partner (odd count in its workgroup) runs the same packed arithmetic with the other half empty.
#pragma once

namespace evac {

template <class FW, bool GRAV>
__device__ __forceinline__ bool try_pack(typename FW::Smem& sm, const Params& p, typename FW::Ctx& w, const Ped& q0, const Env& e0,
                                         bool active0, int n_steps, const float2* __restrict__ actions, float* __restrict__ slab_out,
                                         int* __restrict__ moving_out) {
    using F = Sub<32, FW::kBlock>;
    static_assert(sizeof(typename F::Smem) <= sizeof(sm.tile), "the packed tiles alias the waves' own tile regions");
    const int N = p.n_ped;
    // ---- eligibility ----
    const bool mov0 = (unsigned)(q0.st - kViscek) < 3u;
    const unsigned long long m_mov = ballot(mov0);
    const int m = __popcll(m_mov);
    const float ex = q0.x - kExitX, ey = q0.y - kExitY;
    const float reach = (float)n_steps * p.step_size + kREscape + 1e-3f;
    const bool far = ballot(mov0 && ex * ex + ey * ey > reach * reach) != 0ull;
    const bool elig = m <= 32 && far && e0.now + n_steps < p.max_timesteps && !(p.flags & kFlagTermOnWall);
    if (w.lane == 0) sm.pk_elig[w.slot] = elig ? 1 : 0;
    __syncthreads();                                                       // (1) everybody's eligibility
    // (waves beyond the batch's last env have left the kernel before this point and wrote nothing)
    // Pairs are formed in SIMD order (CU-wide workgroups: wave w runs on SIMD w % 4, so the order key is (w % 4, w / 4)): the
    // two waves of a pair then sit on the same SIMD wherever possible and every SIMD keeps its share of the surviving waves
    // (pairing waves 0-1, 2-3, ... would empty SIMDs 1 and 3).
    const int n_present = min(FW::kEnvsPerBlock, p.n_envs - (int)blockIdx.x * FW::kEnvsPerBlock);
    constexpr bool kBySimd = FW::kEnvsPerBlock == 16;
    const int key_slot = kBySimd ? (w.lane >> 2) + 4 * (w.lane & 3) : w.lane;          // the wave whose order key is `lane`
    const int my_key = kBySimd ? (w.slot & 3) * 4 + (w.slot >> 2) : w.slot;
    const int eg = (w.lane < FW::kEnvsPerBlock && key_slot < n_present) ? sm.pk_elig[key_slot] : 0;
    const unsigned mask = (unsigned)ballot(eg != 0);
    if (!elig) {
        __syncthreads();                                                   // (2) (kept in step with the packing waves)
        return false;
    }
    const int rank = __popc(mask & ((1u << my_key) - 1u)), n_elig = __popc(mask);
    const int pair = rank >> 1, half = rank & 1;
    // ---- the escaped pedestrians are done for this launch: pinned (area.py:79-81) and stored ----
    if (active0 && q0.st == kEscaped) p.ped[(size_t)w.env * N + w.i] = make_float4(kExitX, kExitY, 0.0f, 0.0f);
    // ---- the moving ones, compacted in ascending order, into this wave's half of the pair's exchange area ----
    const int r = __builtin_amdgcn_mbcnt_hi((unsigned)(m_mov >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_mov, 0u));
    if (mov0) {
        sm.pk_ped[pair][half * 32 + r] = f4{q0.x, q0.y, q0.dx, q0.dy};
        sm.pk_tag[pair][half * 32 + r] = i2{q0.st, w.i};
    }
    if (w.lane < 32 && w.lane >= m) {
        sm.pk_ped[pair][half * 32 + w.lane] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        sm.pk_tag[pair][half * 32 + w.lane] = i2{0, 1 << 20};
    }
    if (w.lane == 0) {
        sm.pk_slot[pair][half] = w.slot;
        sm.pk_env[pair][half][0] = f4{e0.ax, e0.ay, e0.adx, e0.ady};
        sm.pk_env[pair][half][1] = __builtin_bit_cast(f4, i4{e0.now, e0.n_resets, (int)e0.total, w.env});
        sm.pk_env[pair][half][2] = f4{e0.acc_ret, e0.acc_intr, e0.acc_stat, __builtin_bit_cast(float, N - m)};
    }
    if (half == 0 && rank + 1 == n_elig) {       // no partner: the other half stays empty and shadows this env (no stores)
        if (w.lane < 32) {
            sm.pk_ped[pair][32 + w.lane] = f4{0.0f, 0.0f, 0.0f, 0.0f};
            sm.pk_tag[pair][32 + w.lane] = i2{0, 1 << 20};
        }
        if (w.lane == 0) {
            sm.pk_slot[pair][1] = w.slot;
            sm.pk_env[pair][1][0] = f4{e0.ax, e0.ay, e0.adx, e0.ady};
            sm.pk_env[pair][1][1] = __builtin_bit_cast(f4, i4{e0.now, e0.n_resets, (int)e0.total, -1});
            sm.pk_env[pair][1][2] = f4{0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, N)};
        }
    }
    __syncthreads();                                                       // (2) the exchange areas are complete
    if (half == 1) {                             // this env lives on in the partner's wave; the wave leaves its SIMD to the others
        if constexpr (EVAC_PRIO && FW::kPace) {
            if (w.lane == 0) sm.progress[(w.slot & 3) * 4 + (w.slot >> 2)] = 0x7fffffff;   // (never "behind" its mates)
        }
        return true;
    }

    // ---- two envs in one wave: the step body of the sub-wave family over the compacted pedestrians ----
    typename F::Smem& psm = *reinterpret_cast<typename F::Smem*>(&sm.tile[0][0][0]);   // slots 2w, 2w+1 = wave w's own tile region
    typename F::Ctx c(psm);
    const f4 pv = sm.pk_ped[pair][c.lane];
    const i2 tg = sm.pk_tag[pair][c.lane];
    const f4 ea = sm.pk_env[pair][c.sub][0], ec = sm.pk_env[pair][c.sub][2];
    const i4 eb = __builtin_bit_cast(i4, sm.pk_env[pair][c.sub][1]);
    Ped q{pv.x, pv.y, pv.z, pv.w, tg.x};
    Env e;
    e.ax = ea.x; e.ay = ea.y; e.adx = ea.z; e.ady = ea.w;
    e.now = eb.x; e.n_resets = eb.y; e.total = (uint32_t)eb.z;
    e.acc_ret = ec.x; e.acc_intr = ec.y; e.acc_stat = ec.z;
    c.i = tg.y;
    c.esc_base = __builtin_bit_cast(int, ec.w);
    const bool valid = eb.w >= 0;
    c.env = valid ? eb.w : w.env;
    const bool active = c.i < N;
    const bool owner = valid && c.owner;
    const uint32_t gid = p.env_id_offset + (uint32_t)c.env;
    const size_t E = (size_t)p.n_envs;
    uint4 nzr = make_uint4(0, 0, 0, 0);
    bool have = false;
    float2 lane_act = make_float2(0.f, 0.f), lane_adir = make_float2(0.f, 0.f);
    // outputs are staged like the unpacked kernel's (rollout_body): 9 words per env-step in LDS, one 64-lane store per env
    // every kStageSteps steps.  Env A uses this wave's staging rows, env B the rows of the wave that handed it over.
    const int stage_slot = c.sub == 0 ? w.slot : sm.pk_slot[pair][1];
    const int slot_b = __builtin_amdgcn_readlane(stage_slot, 32);
    const int env_a = __builtin_amdgcn_readlane(c.env, 0), env_b = __builtin_amdgcn_readlane(c.env, 32);
    const bool valid_b = __builtin_amdgcn_readlane(valid ? 1 : 0, 32) != 0;
    const int fl_s = c.lane / kGravRow, fl_k = c.lane - fl_s * kGravRow;
    int prio_slot = 0;
    if constexpr (EVAC_PRIO && !FW::kPace) prio_slot = simd_wave_slot();
    int pace_seen = 0, pace_prio = 0;
    for (int t = 0; t < n_steps; ++t) {
        if constexpr (EVAC_PRIO && FW::kPace) pace_step(sm, w.slot & 3, w.slot >> 2, w.lane, t, pace_seen, pace_prio);
        else if constexpr (EVAC_PRIO != 0) {
            if (p.fair) set_wave_priority(t + prio_slot);
        }
        const int slot32 = t & 31;
        if (slot32 == 0) {                // actions of the next 32 steps, one step per lane of the half (random_agent.py:8-9)
            if (actions) {
                if (t + c.li < n_steps) lane_act = actions[(size_t)(t + c.li) * E + c.env];
            } else {
                lane_act = philox_action(p, gid, e.total + (uint32_t)c.li);
            }
            lane_adir = agent_direction(p, lane_act.x, lane_act.y);
        }
        float2 adir;
        adir.x = F::fetch(lane_adir.x, c.sub * 32, slot32);
        adir.y = F::fetch(lane_adir.y, c.sub * 32, slot32);
        const uint32_t sel = e.total & 3u;
        if (!have || sel == 0u) {
            nzr = philox4x32_10(make_uint4(gid, (uint32_t)c.i, e.total >> 2, kStreamNoise), p.seed_lo, p.seed_hi);
            have = true;
        }
        const uint32_t wsel = sel == 0 ? nzr.x : (sel == 1 ? nzr.y : (sel == 2 ? nzr.z : nzr.w));
        const float nz = (u01(wsel) - 0.5f) * p.noise_coef;
        StepOut o;
        step_env<F, GRAV>(p, c, active, q, e, adir, nz, o);
        // (no episode ends inside a packed launch: eligibility)
        const int staged = t % kStageSteps;
        if (owner) {                      // (an empty second half has no owner: it would stage into env A's rows)
            float* st = sm.stage[stage_slot][staged];
            *(f4*)(st + 0) = f4{e.ax, e.ay, o.ex, o.ey};
            *(f4*)(st + 4) = f4{o.gx, o.gy, o.reward, 0.0f};
            st[8] = 0.0f;
        }
        if (staged == kStageSteps - 1 || t == n_steps - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // this wave staged them: in-order LDS, no barrier needed
            if (fl_s <= staged) {
                const size_t row = (size_t)(t - staged + fl_s) * E;
                slab_out[(row + env_a) * kGravRow + fl_k] = sm.stage[w.slot][fl_s][fl_k];
                if (valid_b) slab_out[(row + env_b) * kGravRow + fl_k] = sm.stage[slot_b][fl_s][fl_k];
            }
        }
    }
    if (valid && active) {
        p.ped[(size_t)c.env * N + c.i] = make_float4(q.x, q.y, q.dx, q.dy);
        p.status[(size_t)c.env * N + c.i] = (uint8_t)q.st;
    }
    const unsigned long long m_row = ballot(needs_row(p, q.st)), m_now = ballot((unsigned)(q.st - kViscek) < 3u);
    if (owner) {
        p.agent[c.env] = make_float4(e.ax, e.ay, e.adx, e.ady);
        p.clock[c.env] = make_int4(e.now, e.n_resets, (int)e.total, 0);
        p.acc[c.env] = make_float4(e.acc_ret, e.acc_intr, e.acc_stat, 0.0f);
        if (moving_out) moving_out[c.env] = (m_row & c.gmask) != 0ull ? F::count(m_now, c.gmask) : 0;
        if (p.pack_stats) atomicAdd(p.pack_stats, 1u);
    }
    return true;
}

}  // namespace evac
