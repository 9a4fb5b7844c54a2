"""BASELINE configs 4 and 5 at their FULL sizes on one MI355X (both fit in HBM): the 8-GPU partition is emulated by
eight handles with env_id_offset = r * E / 8 -- exactly what rank r of ShardedEvacuationEnv creates -- whose
concatenated results must equal one handle that owns all envs, bit for bit; plus the size-independent invariants on
the full batch.  (The multi-GPU run itself adds only the all-gather, covered by tests/test_distributed_cpu.py and
tests/test_bench_launcher_cpu.py.)"""
import numpy as np
import pytest

from tests.test_gpu_parity import ea  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def invariants(ea, cfg, wrap, env, ro, T, period):
    import torch
    n, E = cfg.number_of_pedestrians, env.num_envs
    st = env.get_state()
    pos, dr, status = st["pos"], st["dir"], st["status"]
    assert torch.isfinite(ro["slab"]).all() and torch.isfinite(pos).all()
    assert (pos.abs() <= 1.0).all() and set(torch.unique(status).tolist()) <= {1, 2, 3, 4}
    exit_xy = torch.tensor([0.0, -1.0], device=pos.device)
    d_lead = (pos - st["agent_pos"][:, None, :]).norm(dim=-1)
    d_exit = (pos - exit_xy).norm(dim=-1)
    want = torch.ones_like(status)
    want[d_lead < 0.2] = 2
    want[d_exit < 0.4] = 3
    want[d_exit < 0.01] = 4
    near_tie = ((d_lead - 0.2).abs() < 1e-6) | ((d_exit - 0.4).abs() < 1e-6) | ((d_exit - 0.01).abs() < 1e-6)
    assert ((want == status) | near_tie).all()                                   # status == classifier(position)
    assert (pos[status == 4] - exit_xy).norm(dim=-1).max() < 0.01 if (status == 4).any() else True
    v = status == 1
    np.testing.assert_allclose(dr[v].norm(dim=-1).cpu().numpy(), cfg.step_size, rtol=1e-5)   # Vicsek step length
    r = ro["reward"]
    assert r.min() >= -6.0 - 1e-4 and r.max() <= -1.0 + 25.0 * n
    tr = ro["truncated"] != 0
    k = T // period
    assert all(tr[period * j - 1].all() for j in range(1, k + 1)) and int(tr.sum()) == k * E
    stats = ro["episode_stats"][period - 1]
    assert (stats[:, 1] == period).all() and (stats[:, 4:8].sum(dim=1) == n).all()
    np.testing.assert_allclose(stats[:, 0].cpu().numpy(), r[:period].sum(dim=0).cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n,E,T,period,wrap_kw", [
    (60, 32768, 120, 50, dict(positions="grav", alpha=3)),                         # BASELINE config 4
    (1024, 256, 24, 10, dict(positions="rel", statuses="ohe", type="Box")),        # BASELINE config 5
], ids=["c4_n60x32768_grav", "c5_n1024x256_box_ohe"])
def test_full_size_as_eight_shards(ea, n, E, T, period, wrap_kw):
    import torch
    G, seed = 8, 0x5EED0004
    cfg = ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, max_timesteps=period)
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    whole = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=seed)
    whole.reset()
    acts = torch.rand((E, 2), device=whole.device) * 2 - 1
    o_w = whole.step(acts)[0].clone()                                            # the step API ...
    ro_w = whole.rollout(T)                                                      # ... and the rollout
    per = E // G
    st_w = whole.get_state()
    for r in range(G):
        shard = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=per, seed=seed, env_id_offset=r * per)
        shard.reset()
        lo, hi = r * per, (r + 1) * per
        o_s = shard.step(acts[lo:hi].contiguous())[0]
        assert (o_s == o_w[lo:hi]).all(), f"shard {r} step obs"
        ro_s = shard.rollout(T)
        assert (ro_s["slab"] == ro_w["slab"][:, lo:hi]).all(), f"shard {r} slab"
        assert (ro_s["episode_stats"].view(torch.int32) == ro_w["episode_stats"][:, lo:hi].view(torch.int32)).all()
        st_s = shard.get_state()
        for k in st_s:
            assert (st_s[k] == st_w[k][lo:hi]).all(), (r, k)
        shard.close()
        del shard, ro_s, st_s
    whole.close()
    # invariants on a fresh full-size batch whose truncations line up with the rollout
    env = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=seed + 1)
    env.reset()
    ro = env.rollout(T)                                                          # ends a few steps after an autoreset
    invariants(ea, cfg, wrap, env, ro, T, period)
    env.close()
