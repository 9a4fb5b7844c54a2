"""GPU tests of the SURVEY.md 8(f) rows either side of the step path: episode records and final_info (row 2),
the destination form of step() for the trainer's rollout storage (row 3) and trajectory capture in the reference's
frame format (row 4) -- the latter against the REFERENCE's recorded trajectories (tests/golden), not against itself."""
import os

import numpy as np
import pytest

from tests import helpers as H
from tests.test_gpu_parity import ATOL, cfg_from_params, ea  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["traj_n60_s1_noise05_ens05", "traj_n10_s4_long", "traj_n256_s5"])
def test_capture_reproduces_the_reference_trajectory(ea, name):
    """rollout(capture_envs=, actions=, noise=) fed with the reference episode's reset draws, actions and noise:
    the captured frames are the reference's recorded pos / status / leader trajectory (what Pedestrians.save and
    Agent.save stored, pedestrians.py:33-35, area.py:32-33), frame by frame."""
    from evacuation_amd.trajectory import capture_to_memory
    d = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    p = H.load_params(d["params_json"])
    n, T, E = p.number_of_pedestrians, min(40, len(d["action"])), 3
    env = ea.BatchedEvacuationEnv(cfg_from_params(ea, p), ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E)
    draws = np.concatenate([d["draw_pos"], d["draw_dir"]], axis=1).astype(np.float32)[None].repeat(E, 0)
    env.reset(draws=draws)
    initial = {k: v.clone() for k, v in env.get_state().items()}
    acts = d["action"][:T].astype(np.float32)[:, None, :].repeat(E, 1)
    noise = d["noise"][:T].astype(np.float32)[:, None, :].repeat(E, 1)
    ro = env.rollout(T, actions=acts, noise=noise, capture_envs=2)
    assert ro["trajectory"].shape == (T, 2, n + 1, 3)
    for k in range(2):                                         # both captured envs ran the same episode
        ped_mem, agent_mem = capture_to_memory(ro, k, initial=initial)
        assert len(ped_mem["positions"]) == T + 1 and len(ped_mem["statuses"]) == T + 1 and len(agent_mem["position"]) == T
        for t in range(T + 1):
            np.testing.assert_allclose(ped_mem["positions"][t], d["pos"][t], rtol=0, atol=ATOL, err_msg=f"frame {t}")
            assert ped_mem["positions"][t].dtype == np.float64 and ped_mem["positions"][t].shape == (n, 2)
            assert [s.value for s in ped_mem["statuses"][t]] == d["status"][t].tolist(), f"frame {t}"
            assert isinstance(ped_mem["statuses"][t][0], ea.Status)
        for t in range(T):
            np.testing.assert_allclose(agent_mem["position"][t], d["agent_pos"][t + 1], rtol=0, atol=1e-6)
            assert agent_mem["position"][t].dtype == np.float32
    # the packed record of the same launch agrees with the reference's rewards / flags
    np.testing.assert_allclose(ro["reward"][:, 0].cpu().numpy(), d["reward"][:T], rtol=1e-5, atol=1e-4)
    assert (ro["terminated"][:, 0].cpu().numpy() != 0).tolist() == d["terminated"][:T].tolist()
    env.close()


def test_step_writes_the_trainers_storage_directly(ea):
    """rpo_agent.py:158-163,182-196: obs[step] / rewards[step] / dones[step] rows are the step's destinations."""
    import torch
    n, E, T = 60, 32, 12
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=5)           # episodes end inside the window: autoreset rows too
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    a = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=3)
    b = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=3)
    D = a.obs_dim
    obs = torch.full((T + 1, E, D), float("nan"), device=a.device)
    rew = torch.full((T, E), float("nan"), device=a.device)
    term = torch.full((T, E), 7, dtype=torch.uint8, device=a.device)
    trunc = torch.full((T, E), 7, dtype=torch.uint8, device=a.device)
    first, _ = a.reset()
    obs[0].copy_(first)
    b.reset()
    acts = torch.rand((T, E, 2), device=a.device) * 2 - 1
    for t in range(T):
        o, r, te, tr, _ = a.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=term[t], out_truncated=trunc[t])
        assert o.data_ptr() == obs[t + 1].data_ptr() and r.data_ptr() == rew[t].data_ptr()
        ob, rb, teb, trb, _ = b.step(acts[t])
        assert (obs[t + 1] == ob).all() and (rew[t] == rb).all() and (term[t] == teb).all() and (trunc[t] == trb).all()
    assert trunc[4].all() and not trunc[3].any() and trunc[9].all()          # max_timesteps = 5
    assert not torch.isnan(obs).any() and not torch.isnan(rew).any() and (term < 2).all()
    with pytest.raises(ValueError):
        a.step(acts[0], out_obs=obs[:, 0])                                   # not a contiguous [E, D] row
    with pytest.raises(ValueError):
        a.step(acts[0], out_reward=rew[0].double())
    a.close(); b.close()


def test_step_launchers_equal_step(ea):
    """step_launcher: one pre-bound call per row of the trainer's storage (no per-call argument work) == step(out_*=), with
    autoreset rows, final observations and episode records included; a launcher re-used with its action tensor re-filled."""
    import torch
    n, E, T = 60, 48, 14
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=6, is_new_exiting_reward=True)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    a = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=5)
    b = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=5)
    D = a.obs_dim
    obs = torch.zeros((T + 1, E, D), device=a.device)
    rew = torch.zeros((T, E), device=a.device)
    term = torch.zeros((T, E), dtype=torch.uint8, device=a.device)
    trunc = torch.zeros((T, E), dtype=torch.uint8, device=a.device)
    acts = torch.rand((T, E, 2), device=a.device) * 2 - 1
    a.reset(); b.reset()
    go = [a.step_launcher(acts[t], out_obs=obs[t + 1], out_reward=rew[t], out_terminated=term[t], out_truncated=trunc[t]) for t in range(T)]
    for t in range(T):
        go[t]()
        ob, rb, teb, trb, info = b.step(acts[t])
        assert torch.equal(obs[t + 1].view(torch.int32), ob.view(torch.int32)) and torch.equal(rew[t].view(torch.int32), rb.view(torch.int32))
        assert torch.equal(term[t], teb) and torch.equal(trunc[t], trb)
        assert torch.equal(a.final_obs.view(torch.int32), b.final_obs.view(torch.int32))
        assert torch.equal(a.final_stats.view(torch.int32), b.final_stats.view(torch.int32))
    assert trunc[5].all() and trunc[11].all()
    act = torch.zeros((E, 2), device=a.device)
    one = a.step_launcher(act)                                # the env's own output tensors, one action tensor re-filled in place
    for t in range(3):
        act.copy_(acts[t])
        one()
        ob, rb, _, _, _ = b.step(acts[t])
        assert torch.equal(a.obs.view(torch.int32), ob.view(torch.int32)) and torch.equal(a.reward.view(torch.int32), rb.view(torch.int32))
    with pytest.raises(ValueError):
        a.step_launcher(acts[0].double())
    a.close(); b.close()


def test_normalized_env_accepts_destinations(ea):
    import torch
    E, T = 16, 6
    cfg = ea.EnvConfig(number_of_pedestrians=20, max_timesteps=4)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    a = ea.NormalizedVectorEnv.make(cfg, wrap, num_envs=E, seed=5)
    b = ea.NormalizedVectorEnv.make(cfg, wrap, num_envs=E, seed=5)
    obs = torch.zeros((T + 1, E, a.obs_dim), device=a.env.device)
    rew = torch.zeros((T, E), device=a.env.device)
    obs[0].copy_(a.reset()[0]); b.reset()
    acts = torch.rand((T, E, 2), device=a.env.device) * 4 - 2               # beyond the action box: ClipAction is in the chain
    for t in range(T):
        a.step(acts[t], out_obs=obs[t + 1], out_reward=rew[t])
        ob, rb, _, _, _ = b.step(acts[t])
        assert (obs[t + 1] == ob).all() and (rew[t] == rb).all()
    assert (obs.abs() <= 1).all()
    a.close(); b.close()


def test_final_info_needs_no_extra_call(ea):
    """What rpo_agent.py:198-203 asks of `infos` -- membership of "final_info", iteration over it, `info["episode"]` of the finished
    envs and None for the others -- answered without an extra call; the record carries the reference's nine logging keys (env.py:115-125)."""
    import torch
    n, E, L = 12, 6, 7
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=L)
    env = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=11)
    env.reset()
    acts = torch.zeros((E, 2), device=env.device); acts[:, 0] = 1.0
    returns = torch.zeros(E, device=env.device)
    seen = []
    for t in range(1, 2 * L + 1):
        obs, reward, terminations, truncations, infos = env.step(acts)
        returns += reward
        if "final_info" in infos:                                            # (the trainer's access pattern)
            for info in infos["final_info"]:
                if info and "episode" in info:
                    seen.append((t, info))
            assert infos["_final_info"].tolist() == (terminations | truncations).bool().cpu().tolist()
            assert len(infos["final_info"]) == E
        else:
            assert not (terminations | truncations).any()
            with pytest.raises(KeyError):
                infos["final_info"]
        if t in (L, 2 * L):
            assert "final_info" in infos
            if t == L:
                first_returns = returns.clone(); returns.zero_()
    assert len(seen) == 2 * E and {t for t, _ in seen} == {L, 2 * L}
    keys = {"episode_intrinsic_reward", "episode_status_reward", "episode_reward", "episode_length", "escaped_pedestrians",
            "exiting_pedestrians", "following_pedestrians", "viscek_pedestrians", "overall_timesteps"}   # env.py:115-125
    for t, info in seen:
        assert keys <= set(info), keys - set(info)
        assert info["episode_length"] == L and info["episode"]["l"] == L
        assert info["overall_timesteps"] == t and info["n_episodes"] == t // L
        assert info["escaped_pedestrians"] + info["exiting_pedestrians"] + info["following_pedestrians"] + info["viscek_pedestrians"] == n
    got = sorted(info["episode"]["r"] for t, info in seen if t == L)
    np.testing.assert_allclose(got, sorted(first_returns.cpu().tolist()), rtol=1e-5)
    # the rollout's record carries the same words
    env2 = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(positions="grav"), num_envs=E, seed=11)
    env2.reset()
    ro = env2.rollout(2 * L, actions=acts[None].repeat(2 * L, 1, 1))
    from evacuation_amd.vector_env import stats_int_view
    ints = stats_int_view(ro["episode_stats"]).cpu().numpy()
    assert (ints[L - 1, :, 0] == L).all() and (ints[2 * L - 1, :, 0] == 2 * L).all() and (ints[2 * L - 1, :, 1] == 2).all()
    env.close(); env2.close()


def _episode_record_files():
    from tests import helpers as H
    return H.episode_record_files()


@pytest.mark.parametrize("path", _episode_record_files(), ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize("face", ["step", "rollout"])
def test_episode_record_equals_the_references_log(ea, path, face):
    """VERDICT r05 item 5a.  The nine-key dict the REFERENCE logs at the reset after an episode (env.py:114-127; captured from the
    reference by tests/golden/make_golden.py: one episode run to truncation, one crafted episode that terminates with everybody
    escaped) against evac_episode_stats_t of the same episode on the GPU: the same start state, the reference's actions and the noise
    it drew.  Status counts, episode_length and overall_timesteps exactly; the three reward sums to 1e-5 relative (f32 sums of 40-52
    terms against the reference's f64).  `step`: teacher-forced (set to the reference's pre-state before every step -- the
    accumulators are not part of that state); `rollout`: free-running in ONE launch (evac_rollout with given actions and noise)."""
    import json
    import torch
    from tests import helpers as H
    d = np.load(path)
    p = H.load_params(d["params_json"])
    keys = json.loads(str(d["episode_record_keys"]))
    ref = dict(zip(keys, d["episode_record"]))
    T, n, E = len(d["action"]), int(p.number_of_pedestrians), 3
    cfg = cfg_from_params(ea, p)
    env = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(positions="grav", alpha=3), num_envs=E, seed=1)
    dev = env.device

    def rep(a, dtype=np.float32):                       # the same episode in every env of the batch
        return np.broadcast_to(np.asarray(a, dtype=dtype)[None], (E,) + np.shape(a)).copy()

    def put_state(k):
        env.set_state(pos=rep(d["pos"][k]), dir=rep(d["dir"][k]), status=rep(d["status"][k], np.uint8), agent_pos=rep(d["agent_pos"][k]),
                      agent_dir=rep(d["agent_dir"][k]), now=rep(d["now"][k], np.int32))
    env.reset()                                         # (episode sums zeroed, n_episodes = 1 like the reference's first reset)
    put_state(0)
    if face == "step":
        for k in range(T):
            put_state(k)
            obs, rew, term, trunc, infos = env.step(rep(d["action"][k]), noise=rep(d["noise"][k]))
        torch.cuda.synchronize()
        assert bool(term.any()) == bool(d["terminated"][-1]) and bool(trunc.any()) == bool(d["truncated"][-1])
        stats = infos["episode_stats"]
    else:
        ro = env.rollout(T, actions=np.broadcast_to(d["action"][:, None, :], (T, E, 2)).copy(),
                         noise=np.broadcast_to(d["noise"][:, None, :], (T, E, n)).astype(np.float32).copy())
        torch.cuda.synchronize()
        done = ((ro["terminated"] != 0) | (ro["truncated"] != 0)).cpu().numpy()
        assert done[-1].all() and not done[:-1].any(), "the free-running f32 episode ended at another step than the reference's"
        stats = ro["episode_stats"][-1]
    from evacuation_amd.vector_env import STATS_FIELDS, stats_int_view
    f = stats.cpu().numpy()
    ints = stats_int_view(stats).cpu().numpy()
    for e in range(E):
        got = {k: float(f[e, j]) for j, k in enumerate(STATS_FIELDS)}
        got["overall_timesteps"], got["n_episodes"] = int(ints[e, 0]), int(ints[e, 1])
        for k in ("escaped_pedestrians", "exiting_pedestrians", "following_pedestrians", "viscek_pedestrians", "episode_length", "overall_timesteps"):
            assert got[k] == ref[k], (face, e, k, got[k], ref[k])
        for k in ("episode_reward", "episode_intrinsic_reward", "episode_status_reward"):
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-5, atol=1e-4, err_msg=f"{face} env {e} {k}")
        assert got["n_episodes"] == 1
    env.close()


@pytest.mark.parametrize("alpha", [4, 6, 14, 30])
def test_gravity_powers_beyond_the_common_ones(ea, alpha):
    """alpha + 2 = 32 fell out of the binary powering; every integer power up to 63 now has a case."""
    from tests.test_gpu_parity import compare_step, gpu_step_batch
    from oracle import evac_oracle as O
    rng = np.random.default_rng(alpha)
    n, E = 40, 6
    p = O.OracleParams(number_of_pedestrians=n)
    pre = []
    for _ in range(E):
        st = O.env_reset(p, rng.uniform(-1, 1, (n, 2)), rng.uniform(-1, 1, (n, 2)))
        st.agent_pos = rng.uniform(-0.5, 0.5, 2).astype(np.float32)
        pre.append(st)
    acts = rng.uniform(-1, 1, (E, 2)).astype(np.float32)
    nzs = rng.uniform(-0.1, 0.1, (E, n)).astype(np.float32)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=alpha)
    got = gpu_step_batch(ea, p, wrap, pre, acts, nzs)
    compare_step(p, wrap, pre, acts, nzs, got, min_checked=E - 2)
    assert np.abs(got["obs"][:, 4:6]).max() > 0


@pytest.mark.parametrize("n", [65, 100, 128, 200, 256, 300, 512, 600, 1024])
def test_cell_list_and_all_pairs_kernels_agree(ea, n):
    """EVAC_CELLS=1 / 0 select the cell-list / all-pairs kernels for every N > 64.  Teacher-forced (the all-pairs env
    is set to the cell-list env's state before every step, same actions and noise): identical neighbour sets, so the
    same positions to summation rounding and exactly the same statuses and flags; then a short free-running rollout."""
    import torch
    E, T = 4, 24
    cfg = ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, noise_coef=0.4, max_timesteps=15)
    envs = {}
    for mode in ("1", "0"):
        for key, wrap_kw in (("grav", dict(positions="grav", alpha=3)), ("box", dict(positions="rel", statuses="ohe", type="Box"))):
            env = ea.BatchedEvacuationEnv(cfg, ea.EnvWrappersConfig(**wrap_kw), num_envs=E, seed=n, options=ea.KernelOptions(cells=int(mode)))
            assert ("cell list" in env.kernel_variant()) == (mode == "1") and ("cell list" in env.kernel_variant("step")) == (mode == "1")
            env.reset()
            envs[(mode, key)] = env
    torch.manual_seed(n)
    for key in ("grav", "box"):
        a, b = envs[("1", key)], envs[("0", key)]
        for t in range(T):
            st = a.get_state()
            b.set_state(**{k: v for k, v in st.items()})
            b.clock.copy_(a.clock); b.acc.copy_(a.acc)
            acts = torch.rand((E, 2), device=a.device) * 2 - 1
            noise = (torch.rand((E, n), device=a.device) - 0.5) * 0.4
            oa, ra, ta, tra, _ = a.step(acts, noise=noise)
            ob, rb, tb, trb, _ = b.step(acts, noise=noise)
            sa, sb = a.get_state(), b.get_state()
            assert (sa["status"] == sb["status"]).all() and (ta == tb).all() and (tra == trb).all(), (key, t)
            # a pedestrian whose neighbours' headings nearly cancel has an ill-conditioned mean heading: the rounding
            # of the sum (f32 there, 2^-21..2^-23 fixed point here) then shows; allow a handful of those
            for k in ("pos", "dir"):
                diff = (sa[k] - sb[k]).abs()
                assert diff.max() < 2e-4 and int((diff > 2e-6).sum()) <= 4, (key, t, k, float(diff.max()))
            torch.testing.assert_close(ra, rb, rtol=1e-5, atol=1e-5)
            if key == "box":
                assert (oa - ob).abs().max() < 2e-4 and int(((oa - ob).abs() > 2e-6).sum()) <= 4
            else:
                torch.testing.assert_close(oa[:, :2], ob[:, :2], rtol=0, atol=1e-7)
                torch.testing.assert_close(oa[:, 2:], ob[:, 2:], rtol=1e-4, atol=1e-2)      # sums of terms up to 1875
        assert int(tra.sum()) == 0 and (a.clock[:, 1] >= 2).all()                          # an autoreset happened on the way
        ra, rb = a.rollout(4), b.rollout(4)                                                 # same state, Philox noise, 4 steps
        assert (ra["slab"][..., -2:] == rb["slab"][..., -2:]).all()
        torch.testing.assert_close(a.get_state()["pos"], b.get_state()["pos"], rtol=0, atol=1e-5)
    for env in envs.values():
        env.close()


def test_facade_records_the_frames_the_reference_records(ea):
    """VERDICT r02 'missing' 4: ``setup_env(cfg(draw=True))`` -- the single-env facade keeps ``pedestrians.memory`` /
    ``agent.memory`` exactly as the reference's step() does (env.py:137,153-155; pedestrians.py:27,33-35; area.py:30-33):
    the reset frame always, one frame per step while ``draw`` is set; checked against a reference episode's recorded
    trajectory.  And the giff_freq rule of reset() (env.py:110-112, 322-324)."""
    d = np.load(os.path.join(H.GOLDEN, "traj_n60_s1_noise05_ens05.npz"))
    p = H.load_params(d["params_json"])
    T = 25
    env = ea.setup_env(cfg_from_params(ea, p, draw=True), ea.EnvWrappersConfig(positions="grav", alpha=3))
    draws = np.concatenate([d["draw_pos"], d["draw_dir"]], axis=1).astype(np.float32)
    env.reset(options={"draws": draws})
    assert len(env.pedestrians.memory["positions"]) == 1 and env.agent.memory["position"] == []
    for k in range(T):
        env.step(d["action"][k], noise=d["noise"][k])
    pm, am = env.pedestrians.memory, env.agent.memory
    assert len(pm["positions"]) == len(pm["statuses"]) == T + 1 and len(am["position"]) == T
    for t in range(T + 1):
        np.testing.assert_allclose(pm["positions"][t], d["pos"][t], rtol=0, atol=ATOL, err_msg=f"frame {t}")
        assert pm["positions"][t].dtype == np.float64 and [s.value for s in pm["statuses"][t]] == d["status"][t].tolist()
        assert isinstance(pm["statuses"][t][0], ea.Status)
    for t in range(T):
        np.testing.assert_allclose(am["position"][t], d["agent_pos"][t + 1], rtol=0, atol=1e-6)
        assert am["position"][t].dtype == np.float32
    env.close()
    # without draw only the reset frame is kept, until reset() switches drawing on for every giff_freq-th episode
    env = ea.setup_env(ea.EnvConfig(number_of_pedestrians=10, max_timesteps=3, giff_freq=2), ea.EnvWrappersConfig())
    env.reset()
    assert env.draw is False
    for _ in range(3):
        out = env.step(np.array([0.3, 0.1], dtype=np.float32))
    assert out[3] is True and len(env.pedestrians.memory["positions"]) == 1 and env.agent.memory["position"] == []
    env.reset()                                            # second episode: (n_episodes + 1) % giff_freq == 0
    assert env.draw is True and env.save_next_episode_anim is True
    for _ in range(3):
        out = env.step(np.array([0.3, 0.1], dtype=np.float32))
    assert out[3] is True and len(env.pedestrians.memory["positions"]) == 4 and len(env.agent.memory["position"]) == 3
    assert env.draw is False and env.save_next_episode_anim is False     # what save_animation() leaves behind (env.py:322-324)
    env.close()


def test_stale_infos_are_refused(ea):
    """infos["final_info"] is built lazily from buffers the env reuses every step; first asking an OLD infos object
    after the env has stepped again must fail loudly instead of describing the wrong step."""
    import torch
    E = 6
    env = ea.BatchedEvacuationEnv(ea.EnvConfig(number_of_pedestrians=20, max_timesteps=2), ea.EnvWrappersConfig(positions="grav"), num_envs=E)
    env.reset()
    act = torch.zeros((E, 2), device=env.device) + 0.5
    env.step(act)
    _, _, _, trunc, infos = env.step(act)                 # every env truncates here
    assert bool(trunc.all()) and "final_info" in infos and len(infos["final_info"]) == E     # read in time: fine, and cached
    _, _, _, _, old = env.step(act)
    env.step(act)                                          # the env moved on before `old` was looked at
    with pytest.raises(RuntimeError, match="stepped again"):
        _ = "final_info" in old
    assert len(infos["final_info"]) == E                  # what was built in time stays valid
    env.close()
