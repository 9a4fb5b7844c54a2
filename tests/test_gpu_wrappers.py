"""GPU: the device wrapper chain (evac_norm_* + clip_action) against oracle/gym_wrappers.py, the NumPy
restatement of the trainer's chain (rpo_agent.py:24-33).  Raw env outputs come from a twin env without
the chain, so only the wrapper arithmetic and its ordering around autoresets are under test."""
import numpy as np
import pytest

from oracle import gym_wrappers as G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wrap_kw", [dict(positions="grav", alpha=3), dict(positions="rel", statuses="ohe", type="Box")])
def test_normalized_vector_env_matches_the_restated_chain(wrap_kw):
    import torch
    import evacuation_amd as ea
    E, n, T, gamma, seed = 6, 20, 40, 0.97, 321
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=13, is_new_exiting_reward=True)   # several autoresets
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    nenv = ea.NormalizedVectorEnv.make(cfg, wrap, num_envs=E, gamma=gamma, seed=seed)
    assert nenv.env.env_config.clip_action
    import dataclasses
    raw = ea.BatchedEvacuationEnv(dataclasses.replace(cfg, clip_action=True), wrap, num_envs=E, seed=seed)
    D = raw.obs_dim
    stats = [G.WrappedEnvStats(D, gamma=gamma) for _ in range(E)]
    o_n, _ = nenv.reset()
    o_r, _ = raw.reset()
    want = np.stack([stats[e].observation(o_r[e].cpu().numpy().astype(np.float64)) for e in range(E)])
    np.testing.assert_allclose(o_n.cpu().numpy(), want, rtol=0, atol=2e-6)
    rng = np.random.default_rng(0)
    n_done = 0
    for t in range(T):
        act = torch.as_tensor(rng.uniform(-1.6, 1.6, (E, 2)).astype(np.float32)).cuda()      # exercises ClipAction
        on, rn, ten, trn, infn = nenv.step(act)
        orr, rr, ter, trr, infr = raw.step(act)
        assert (ten == ter).all() and (trn == trr).all()
        te, tr = ter.cpu().numpy(), trr.cpu().numpy()
        w_obs, w_fin, w_rew = G.vector_step(stats, orr.cpu().numpy().astype(np.float64),
                                            infr["final_observation"].cpu().numpy().astype(np.float64),
                                            rr.cpu().numpy().astype(np.float64), te, tr)
        np.testing.assert_allclose(on.cpu().numpy(), w_obs, rtol=0, atol=2e-6, err_msg=f"t={t} obs")
        np.testing.assert_allclose(rn.cpu().numpy(), w_rew, rtol=1e-5, atol=1e-6, err_msg=f"t={t} reward")
        done = (te | tr).astype(bool)
        n_done += int(done.sum())
        if done.any():
            np.testing.assert_allclose(infn["final_observation"].cpu().numpy()[done], w_fin[done], rtol=0, atol=2e-6)
    assert n_done >= 2 * E
    st = nenv.norm_state.cpu().numpy()
    for e in range(E):
        np.testing.assert_allclose(st[e, :D], stats[e].obs_rms.mean, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(st[e, D:2 * D], stats[e].obs_rms.var, rtol=1e-8, atol=1e-12)
        assert abs(st[e, 2 * D] - stats[e].obs_rms.count) < 1e-9
        np.testing.assert_allclose(st[e, 3 * D + 3], stats[e].returns[0], rtol=1e-6)
    assert (np.abs(on.cpu().numpy()) <= 1.0).all()
    nenv.close(); raw.close()


def test_clip_action_changes_only_out_of_range_actions():
    import torch
    import evacuation_amd as ea
    cfg = ea.EnvConfig(number_of_pedestrians=8)
    import dataclasses
    a = ea.BatchedEvacuationEnv(cfg, num_envs=2, seed=3, autoreset=False)
    b = ea.BatchedEvacuationEnv(dataclasses.replace(cfg, clip_action=True), num_envs=2, seed=3, autoreset=False)
    a.reset(); b.reset()
    act = torch.tensor([[0.5, -0.25], [3.0, 0.5]], dtype=torch.float32).cuda()
    a.step(act); b.step(act)
    sa, sb = a.get_state(), b.get_state()
    assert (sa["agent_dir"][0] == sb["agent_dir"][0]).all()                 # in range: untouched
    want = np.array([1.0, 0.5]) / np.linalg.norm([1.0, 0.5]) * 0.01          # (3, .5) clipped to (1, .5)
    np.testing.assert_allclose(sb["agent_dir"][1].cpu().numpy(), want, rtol=1e-6)
    assert not torch.allclose(sa["agent_dir"][1], sb["agent_dir"][1])


def test_normalized_masked_reset_counts_only_reset_envs():
    import torch
    import evacuation_amd as ea
    env = ea.NormalizedVectorEnv.make(ea.EnvConfig(number_of_pedestrians=10), ea.EnvWrappersConfig(positions="grav"), num_envs=4)
    env.reset()
    D = env.obs_dim
    c0 = env.norm_state[:, 2 * D].clone()
    env.reset(mask=np.array([1, 0, 0, 1], dtype=np.uint8))
    c1 = env.norm_state[:, 2 * D]
    assert torch.allclose(c1 - c0, torch.tensor([1.0, 0.0, 0.0, 1.0], dtype=torch.float64, device=c1.device))
    env.close()


@pytest.mark.parametrize("n,wrap_kw", [(10, dict(positions="grav", alpha=3)),            # sub-wave family, 6-feature obs
                                       (60, dict(positions="grav", alpha=3)),            # one wave per env
                                       (100, dict(positions="grav", alpha=5)),           # two waves per env
                                       (24, dict(positions="rel", statuses="ohe", type="Box")),     # generic obs, sub-wave
                                       (60, dict(positions="abs", statuses="ohe", type="Box")),
                                       (130, dict(positions="rel", statuses="no", type="Box"))])
def test_fused_normalised_step_equals_step_plus_chain_bit_for_bit(n, wrap_kw):
    """evac_step_normalized (one launch) against evac_step + evac_norm_step (two launches): identical outputs and
    identical running statistics, across autoresets (terminal observation counted before the reset observation)."""
    import torch
    import evacuation_amd as ea
    E, T, seed = 9, 45, 77
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=11, is_new_exiting_reward=True)
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    a = ea.NormalizedVectorEnv.make(cfg, wrap, num_envs=E, gamma=0.95, seed=seed)
    b = ea.NormalizedVectorEnv.make(cfg, wrap, num_envs=E, gamma=0.95, seed=seed)
    oa, _ = a.reset(); ob, _ = b.reset()
    assert torch.equal(oa, ob)
    g = torch.Generator(device="cpu").manual_seed(5)
    n_done = 0
    for t in range(T):
        act = (torch.rand(E, 2, generator=g) * 3.0 - 1.5).cuda()
        ra = a.step(act, fused=True)
        rb = b.step(act, fused=False)
        for k, (x, y) in enumerate(zip(ra[:4], rb[:4])):
            assert torch.equal(x, y), f"t={t} output {k}"
        done = (ra[2] | ra[3]).bool()
        n_done += int(done.sum())
        if done.any():
            assert torch.equal(ra[4]["final_observation"][done], rb[4]["final_observation"][done]), f"t={t} final obs"
        assert torch.equal(a.norm_state, b.norm_state), f"t={t} statistics"
    assert n_done >= 2 * E
    a.close(); b.close()


def test_fused_normalised_step_writes_into_caller_storage():
    import torch
    import evacuation_amd as ea
    E, n = 5, 60
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=7)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    a = ea.NormalizedVectorEnv.make(cfg, wrap, num_envs=E, seed=3)
    b = ea.NormalizedVectorEnv.make(cfg, wrap, num_envs=E, seed=3)
    a.reset(); b.reset()
    T = 12
    obs = torch.zeros(T, E, a.env.obs_dim, device="cuda"); rew = torch.zeros(T, E, device="cuda")
    te = torch.zeros(T, E, dtype=torch.uint8, device="cuda"); tr = torch.zeros(T, E, dtype=torch.uint8, device="cuda")
    for t in range(T):
        act = torch.full((E, 2), 0.3, device="cuda")
        a.step(act, out_obs=obs[t], out_reward=rew[t], out_terminated=te[t], out_truncated=tr[t])
        o, r, x, y, _ = b.step(act, fused=False)
        assert torch.equal(obs[t], o) and torch.equal(rew[t], r) and torch.equal(te[t], x.view(torch.uint8)) \
            and torch.equal(tr[t], y.view(torch.uint8))
    assert int(tr.sum()) >= E
    a.close(); b.close()
