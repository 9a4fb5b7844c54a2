"""Direct oracle checks of the PRODUCTION kernel instantiations.

The kernels bench.py times -- k_rollout_default_config<Wave<1, 1024>> (CU-wide workgroups, pace keeping, load schedule,
compile-time default configuration) for BASELINE config 2 and k_rollout_default_config<Team<8>> for config 5 -- used to be
tied to the oracle only through bit-identity chains (production face == generic face == step kernel == oracle).  Here they
free-run in Philox mode and every env is compared with an oracle episode driven by oracle/philox.py's restatement of the
device streams: the packed slab (observation, reward, flags) step by step and the final state.

The GPU is f32, the oracle runs in the reference's precision (f64 pedestrians): a threshold comparison whose f64 margin is
below TIE may legitimately flip in f32, after which the trajectories separate (SURVEY.md 7 'Parity definition').  A
pedestrian is therefore compared only as long as no such tie can have reached it (oracle_episode tracks that per pedestrian,
from the ORACLE's own margins), and the tests assert how much was compared.

Also here: evac_rollout == step-by-step evac_step, bit for bit, for the cell-list family (N = 600, 1024), fed the actions
the rollout draws on device (restated by oracle/philox.py) -- the production face, not the diagnostic one that records them.
"""
import os

import numpy as np
import pytest

from oracle import evac_oracle as O
from oracle import philox as P
from tests.test_gpu_parity import cfg_from_params, flat_oracle_obs, grav_tolerance

pytestmark = pytest.mark.gpu

TIE = 2e-6


@pytest.fixture(scope="module")
def ea():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu tests need an MI355X (torch.cuda.is_available() is False)")
    import evacuation_amd
    return evacuation_amd


def _Env(**kv):
    """Create-time options (evac_options_t, by the names of their diagnostic switches) for the handles made inside the block."""
    from evacuation_amd.options import from_switches, kernel_options
    return kernel_options(from_switches(**kv))


def oracle_episode(p, wrap, seed, gid, T, start=None):
    """The oracle stepping env `gid` for T steps on the device's Philox draws (reset draws, RandomAgent actions, noise), with
    the same-step autoreset of evac_rollout.

    Ties are tracked PER PEDESTRIAN (at N = 1024 half a million pair distances per step make an env-level exclusion useless):
    `taint[i]` = pedestrian i's f32 trajectory may have separated from the f64 one -- one of its own comparisons (a pair
    distance against the radius, a status radius, a wall) had a margin below TIE, or a tainted pedestrian stood within its
    interaction radius (+ SPREAD for the tainted one's position error) when the neighbour sums were taken.  `counts` = a
    tainted pedestrian stood near a status radius, so the env's counts (rewards, flags, the gravity exit term) may differ.
    Returns per step (flat observation, reward, terminated, truncated, gravity tolerances, taint copy, counts) and the final
    state.  `start` = (OracleState, steps taken so far, resets so far): continue from a given state -- a batch's state LATE in an
    episode, taken from the device (f32 values are exact in f64: the start state is an input like the reset draws are) -- instead of
    from the reset; the Philox counters continue where that state's env stood."""
    n = p.number_of_pedestrians
    SPREAD = 2e-3
    if start is None:
        dr = P.reset_draws(seed, [gid], n, 0)[0]
        st = O.env_reset(p, dr[:, 0:2].astype(np.float64), dr[:, 2:4].astype(np.float64))
        n_resets, t0 = 1, 0
    else:
        st, t0, n_resets = start
    rows = []
    taint, counts = np.zeros(n, bool), False
    for t in range(t0, t0 + T):
        pre, pre_status = st.pos.copy(), st.status.copy()
        act = P.random_action(seed, [gid], t)[0]
        nz = P.step_noise(seed, [gid], n, t, p.noise_coef)[0].astype(np.float64)
        with np.errstate(all="ignore"):
            out = O.env_step(p, st, act, nz)
        # neighbour sums (pre-step positions): rows of FOLLOWER / VISCEK against the moving pedestrians (area.py:104-106)
        moving = pre_status != O.ESCAPED
        dm = O.pairwise_distance(pre, pre, np.float64)
        np.fill_diagonal(dm, np.inf)
        pair = moving[:, None] & moving[None, :]
        taint |= ((np.abs(dm - O.R_PEDESTRIAN) < TIE) & pair).any(axis=1)
        taint |= ((dm < O.R_PEDESTRIAN + SPREAD) & pair & taint[None, :]).any(axis=1)
        # statuses and walls (post-step positions)
        pos = np.asarray(st.pos, dtype=np.float64)
        for dest, rad in ((st.agent_pos, O.R_LEADER), (O.EXIT_POSITION, O.R_EXIT), (O.EXIT_POSITION, O.R_ESCAPE)):
            d = np.abs(O.pairwise_distance(pos, np.asarray(dest, dtype=np.float64)[None, :], np.float64)[:, 0] - rad)
            taint |= d < TIE
            counts = counts or bool((taint & (d < SPREAD)).any())
        free = st.status != O.ESCAPED
        taint |= free & ((np.abs(np.abs(pos[:, 0]) - p.width) < TIE) | (np.abs(np.abs(pos[:, 1]) - p.height) < TIE))
        if out["terminated"] or out["truncated"]:            # evac_rollout always autoresets: the slab holds the reset observation
            done_counts = counts
            dr = P.reset_draws(seed, [gid], n, n_resets)[0]
            n_resets += 1
            st = O.env_reset(p, dr[:, 0:2].astype(np.float64), dr[:, 2:4].astype(np.float64))
            taint, counts = np.zeros(n, bool), False       # a fresh state from the Philox draws
            tol = grav_tolerance(st, wrap.alpha, p.eps) if wrap.positions == "grav" else None
            rows.append((flat_oracle_obs(st, wrap, p.eps), out["reward"], out["terminated"], out["truncated"], tol, taint.copy(), done_counts))
            continue
        tol = grav_tolerance(st, wrap.alpha, p.eps) if wrap.positions == "grav" else None
        rows.append((flat_oracle_obs(st, wrap, p.eps), out["reward"], out["terminated"], out["truncated"], tol, taint.copy(), counts))
    return rows, st, taint, counts


FACE_LOG = []     # (kernel variant, pedestrian-steps compared, of, env-steps whose rewards / flags were compared, of): printed at the end of the session (conftest)


def check_against_oracle(ea, p, wrap, E, T, seed, offset, expect_variant, late=0, options=None):
    """Returns (pedestrian-steps compared, pedestrian-steps in all, env-steps whose rewards / flags were compared).  `late`: the batch
    is first stepped `late` steps by the same face (launches of 100); the oracle then starts from THAT state -- the rows and columns
    of the neighbour sums are compacted there, most pedestrians have escaped, followers outnumber VISCEK pedestrians -- and the
    T steps that follow are compared."""
    import torch
    env = ea.BatchedEvacuationEnv(cfg_from_params(ea, p), wrap, num_envs=E, seed=seed, env_id_offset=offset, options=options)
    name = env.kernel_variant("rollout")
    for part in expect_variant:
        assert part in name, name
    env.reset()
    starts = [None] * E
    if late:
        for _ in range(late // 100):
            env.rollout(100)
        torch.cuda.synchronize()
        s0 = {k: v.cpu().numpy() for k, v in env.get_state().items()}
        clock = env.clock.cpu().numpy()
        for e in range(E):
            st0 = O.OracleState(s0["pos"][e].astype(np.float64), s0["dir"][e].astype(np.float64), s0["status"][e].astype(np.int8),
                                s0["agent_pos"][e].astype(np.float32), s0["agent_dir"][e].astype(np.float32), int(s0["now"][e]))
            starts[e] = (st0, int(clock[e, 2]), int(clock[e, 1]))
        name += f" -- from t = {late}"
    ro = env.rollout(T)                                      # ONE launch of the production face (no capture, no recording)
    torch.cuda.synchronize()
    assert env.team_error() == 0
    slab = ro["slab"].cpu().numpy()
    fin = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    D, n = env.obs_dim, p.number_of_pedestrians
    ped_steps = scalar_steps = 0
    for e in range(E):
        rows, st, taint, counts = oracle_episode(p, wrap, seed, offset + e, T, start=starts[e])
        for t in range(T):
            obs, rew, term, trunc, tol, tnt, cnt = rows[t]
            got = slab[t, e]
            clean = ~tnt
            if not cnt:
                assert bool(got[D + 1]) == term and bool(got[D + 2]) == trunc, (e, t)
                np.testing.assert_allclose(got[D], rew, rtol=1e-5, atol=1e-4, err_msg=f"env {e} step {t} reward")
                scalar_steps += 1
            if wrap.positions == "grav":
                np.testing.assert_allclose(got[0:2], obs[0:2], rtol=0, atol=1e-6, err_msg=f"env {e} step {t} agent")
                if not cnt and clean.all():                  # the sums run over every pedestrian
                    tol_p, tol_e = tol
                    np.testing.assert_allclose(got[2:4], obs[2:4], rtol=5e-5, atol=tol_e, err_msg=f"env {e} step {t} grad_exit")
                    np.testing.assert_allclose(got[4:6], obs[4:6], rtol=5e-5, atol=tol_p, err_msg=f"env {e} step {t} grad_ped")
                    ped_steps += n
            else:                                            # Box: row 0 leader, row 1 exit, row 2 + i pedestrian i
                C = D // (n + 2)
                g2, o2 = got[:D].reshape(n + 2, C), obs.reshape(n + 2, C)
                np.testing.assert_allclose(g2[:2], o2[:2], rtol=0, atol=1e-6, err_msg=f"env {e} step {t} leader / exit rows")
                np.testing.assert_allclose(g2[2:][clean], o2[2:][clean], rtol=0, atol=1e-5, err_msg=f"env {e} step {t} pedestrian rows")
                ped_steps += int(clean.sum())
        clean = ~taint                                       # the final state of every pedestrian that never met a tie
        np.testing.assert_array_equal(fin["status"][e][clean], st.status[clean], err_msg=f"env {e} final status")
        np.testing.assert_allclose(fin["pos"][e][clean], st.pos[clean], rtol=0, atol=1e-5, err_msg=f"env {e} final pos")
        np.testing.assert_allclose(fin["dir"][e][clean], st.dir[clean], rtol=0, atol=1e-5, err_msg=f"env {e} final dir")
        np.testing.assert_allclose(fin["agent_pos"][e], st.agent_pos, rtol=0, atol=1e-6)
        assert fin["now"][e] == st.now
    env.close()
    FACE_LOG.append((name, ped_steps, E * T * n, scalar_steps, E * T))
    return ped_steps, E * T * n, scalar_steps


def test_cu_wide_default_config_rollout_vs_oracle(ea):
    """BASELINE config 2's kernel: N = 60, one wave per env, 16 envs per CU-wide workgroup, scheduled, specialised.
    50 free-running steps with a truncation + autoreset at step 30, 96 envs (global env ids from 1000: a shard of a larger job)."""
    p = O.OracleParams(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=30)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    with _Env(EVAC_CU_WIDE="1"):
        peds, all_peds, scalars = check_against_oracle(ea, p, wrap, E=96, T=50, seed=0x5EED0002, offset=1000,
                                                       expect_variant=("k_rollout_default_config", "CU-wide"))
    assert peds >= 0.92 * all_peds and scalars >= 0.97 * 96 * 50, (peds, all_peds, scalars)     # (observed: 97.6 % / 100 %, printed at the end of the session)


@pytest.mark.parametrize("face", ["c2", "c2_chained", "c2_persistent", "c3", "c5_team8"])
def test_production_faces_late_in_an_episode_vs_oracle(ea, face):
    """the same faces checked where an episode spends most of its time -- from t = 1200 (N = 60, 256) / 600
    (N = 1024) on, 40 free-running steps against the oracle started from the batch's own state at that moment: compacted rows and
    columns, row-less envs (no tile, no loop), the teams' transposed few-rows sweep.  `c2_chained` / `c2_persistent`: BASELINE config 2's kernel in
    the forms the bench times since round 6 -- chained launches, one persistent kernel per join."""
    if face in ("c2", "c2_chained", "c2_persistent"):
        p = O.OracleParams(number_of_pedestrians=60, is_new_exiting_reward=True, is_new_followers_reward=True, max_timesteps=2000)
        wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
        opts = ea.KernelOptions(cu_wide=1, chain={"c2": 0, "c2_chained": 1, "c2_persistent": 2}[face])
        peds, all_peds, scalars = check_against_oracle(ea, p, wrap, E=96, T=40, seed=0x5EED0002, offset=1000, late=1200, options=opts,
                                                       expect_variant=("k_rollout_default_config", "CU-wide") + {"c2": (), "c2_chained": ("chained",), "c2_persistent": ("persistent",)}[face])
        assert peds >= 0.85 * all_peds and scalars >= 0.9 * 96 * 40, (peds, all_peds, scalars)
    elif face == "c3":
        p = O.OracleParams(number_of_pedestrians=256, is_new_exiting_reward=True, max_timesteps=2000)
        wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
        peds, all_peds, scalars = check_against_oracle(ea, p, wrap, E=24, T=40, seed=0x5EED0003, offset=0, late=1200, options=ea.KernelOptions(cu_wide=1),
                                                       expect_variant=("k_rollout_default_config", "4 waves/env", "CU-wide"))
        assert peds >= 0.5 * all_peds and scalars >= 0.7 * 24 * 40, (peds, all_peds, scalars)
    else:
        p = O.OracleParams(number_of_pedestrians=1024, is_new_exiting_reward=True, max_timesteps=2000)
        wrap = ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box")
        peds, all_peds, scalars = check_against_oracle(ea, p, wrap, E=8, T=40, seed=0x5EED0005, offset=64, late=600, options=ea.KernelOptions(team=8),
                                                       expect_variant=("k_rollout_default_config", "<8 CUs/env"))
        assert peds >= 0.5 * all_peds and scalars >= 0.3 * 8 * 40, (peds, all_peds, scalars)


def test_cu_wide_four_wave_rollout_vs_oracle(ea):
    """BASELINE config 3's kernel: N = 256, four waves per env, 4 envs per CU-wide workgroup with per-env LDS barriers."""
    p = O.OracleParams(number_of_pedestrians=256, is_new_exiting_reward=True, max_timesteps=2000)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    with _Env(EVAC_CU_WIDE="1"):
        peds, all_peds, scalars = check_against_oracle(ea, p, wrap, E=24, T=20, seed=0x5EED0003, offset=0,
                                                       expect_variant=("k_rollout_default_config", "4 waves/env", "CU-wide"))
    assert peds >= 0.62 * all_peds and scalars >= 0.9 * 24 * 20, (peds, all_peds, scalars)      # (observed: 72.3 % / 97.1 %)


@pytest.mark.parametrize("team", ["8", "16"])
def test_team_default_config_rollout_vs_oracle(ea, team):
    """BASELINE config 5's kernel: N = 1024, 8 CUs per env (what a 32-env shard gets; forced here, a batch of 8 envs would take
    16) and the 16-CU form, rel + ohe Box observation, 8 envs x 10 free-running steps.  At
    N = 1024 half a million pair distances per step make near-ties common and their effect spreads through the crowd one
    interaction radius per step, so the comparison is per pedestrian (oracle_episode): every pedestrian row of the observation
    and of the final state that no tie can have reached."""
    p = O.OracleParams(number_of_pedestrians=1024, is_new_exiting_reward=True, max_timesteps=2000)
    wrap = ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box")
    with _Env(EVAC_TEAM=team):
        peds, all_peds, scalars = check_against_oracle(ea, p, wrap, E=8, T=10, seed=0x5EED0005, offset=64,
                                                       expect_variant=("k_rollout_default_config", f"<{team} CUs/env"))
    assert peds >= 0.8 * all_peds and scalars >= 48, (peds, all_peds, scalars)                   # (observed: 87.1 % / 56 of 80)


@pytest.mark.parametrize("n,E,wrap_kw", [(600, 6, dict(positions="grav", alpha=3)),
                                         (1024, 5, dict(positions="rel", statuses="ohe", type="Box"))])
def test_cell_list_rollout_equals_step_by_step(ea, n, E, wrap_kw):
    """evac_rollout through the production face of Cells<16> == T evac_step launches fed the actions the rollout drew on
    device, bit for bit: observations, rewards, flags, episode records and the final state (truncation + autoreset inside)."""
    import torch
    T, seed, off = 30, 0x5EED0007, 11
    p = O.OracleParams(number_of_pedestrians=n, is_new_exiting_reward=True, max_timesteps=17, intrinsic_reward_coef=0.5)
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    with _Env(EVAC_TEAM="0"):
        a = ea.BatchedEvacuationEnv(cfg_from_params(ea, p), wrap, num_envs=E, seed=seed, env_id_offset=off)
        b = ea.BatchedEvacuationEnv(cfg_from_params(ea, p), wrap, num_envs=E, seed=seed, env_id_offset=off)
    assert "cell list" in a.kernel_variant("rollout") and "cell list" in b.kernel_variant("step")
    a.reset(); b.reset()
    ro = a.rollout(T)
    n_done = 0
    for t in range(T):
        act = torch.as_tensor(P.random_action(seed, off + np.arange(E), t), device=b.device)
        obs, r, te, tr, info = b.step(act)
        assert torch.equal(obs.view(torch.int32), ro["obs"][t].view(torch.int32)), t
        assert torch.equal(r.view(torch.int32), ro["reward"][t].view(torch.int32)), t
        assert (te == ro["terminated"][t]).all() and (tr == ro["truncated"][t]).all(), t
        done = (te | tr).bool()
        n_done += int(done.sum())
        if done.any():
            assert torch.equal(info["episode_stats"][done].view(torch.int32), ro["episode_stats"][t][done].view(torch.int32))
    assert n_done >= E
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    a.close(); b.close()
