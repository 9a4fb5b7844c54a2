"""Packed rollouts (csrc/evac_packed.h): two late-episode one-wave envs share a wave.  Trajectories, statuses, flags and
rewards must be bit-identical to the unpacked kernel; the per-env float sums (gravity observation, intrinsic reward) agree to
f32 rounding; and whether an env is packed depends on the env and the launch alone."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ea():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import evacuation_amd
    return evacuation_amd


def _make(ea, cfg, wrap, E, seed, pack, cu_wide, env_id_offset=0):
    old = {k: os.environ.get(k) for k in ("EVAC_PACK", "EVAC_CU_WIDE")}
    try:
        os.environ["EVAC_PACK"] = "1" if pack else "0"
        os.environ["EVAC_CU_WIDE"] = "1" if cu_wide else "0"
        return ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=seed, env_id_offset=env_id_offset)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _late_state(env, torch, steps):
    """Run the episode forward so that most pedestrians have escaped."""
    env.reset()
    for _ in range(steps // 100):
        env.rollout(100)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in env.get_state().items()}


@pytest.mark.parametrize("n,E,cu_wide,alpha,coef", [(60, 203, True, 3, 0.0), (60, 64, False, 2, 0.5), (40, 37, True, 3, 0.0), (64, 50, False, 3, 0.0)])
def test_packed_rollout_matches_unpacked(ea, n, E, cu_wide, alpha, coef):
    import torch
    cfg = ea.EnvConfig(number_of_pedestrians=n, is_new_exiting_reward=True, intrinsic_reward_coef=coef)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=alpha)
    ref = _make(ea, cfg, wrap, E, 5, pack=False, cu_wide=cu_wide)
    pk = _make(ea, cfg, wrap, E, 5, pack=True, cu_wide=cu_wide)
    st = _late_state(ref, torch, 900)
    pk.reset()
    pk.set_state(**st)
    pk.clock.copy_(ref.clock); pk.acc.copy_(ref.acc)                       # step counters (Philox) and accumulators too
    moving = ((st["status"] >= 1) & (st["status"] <= 3)).sum(1)
    assert int((moving <= 32).sum()) > E // 2                              # most envs qualify by their load
    packed_before = int(pk.pack_stats[0])
    for T in (20, 7, 50, 20):
        a = ref.rollout(T)
        b = pk.rollout(T)
        torch.cuda.synchronize()
        assert torch.equal(a["terminated"], b["terminated"]) and torch.equal(a["truncated"], b["truncated"])
        if coef == 0.0:
            assert torch.equal(a["reward"].view(torch.int32), b["reward"].view(torch.int32))
        else:
            torch.testing.assert_close(a["reward"], b["reward"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(a["obs"], b["obs"], rtol=2e-6, atol=2e-6 * float(a["obs"].abs().max()))
        sa, sb = ref.get_state(), pk.get_state()
        for k in sa:                                                       # positions, directions, statuses, leader, clock: bit for bit
            assert torch.equal(sa[k], sb[k]), (T, k)
        assert torch.equal(ref.clock, pk.clock)
        torch.testing.assert_close(ref.acc, pk.acc, rtol=1e-5, atol=1e-5)
    assert int(pk.pack_stats[0]) - packed_before > E                       # packing did happen, launch after launch
    assert int(ref.pack_stats[0]) == 0
    ref.close(); pk.close()


def test_packing_depends_on_the_env_only(ea):
    """An env of a 96-env batch (packed with some partner, CU-wide workgroups) and the same env alone in a batch of one
    (global env id through env_id_offset; packed with an empty half): bit-identical outputs, observations included --
    whether an env is packed, and hence how its sums are rounded, does not depend on what else is in the batch."""
    import torch
    cfg = ea.EnvConfig(number_of_pedestrians=60, is_new_exiting_reward=True, intrinsic_reward_coef=0.25)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    big = _make(ea, cfg, wrap, 96, 9, pack=True, cu_wide=True)
    st = _late_state(big, torch, 1000)
    clock, acc = big.clock.clone(), big.acc.clone()
    T = 24
    a = big.rollout(T)
    torch.cuda.synchronize()
    assert int(big.pack_stats[0]) > 48
    for eidx in (3, 10, 40, 41, 77, 95):
        one = _make(ea, cfg, wrap, 1, 9, pack=True, cu_wide=False, env_id_offset=eidx)
        one.reset()
        one.set_state(**{k: v[eidx:eidx + 1].contiguous() for k, v in st.items()})
        one.clock.copy_(clock[eidx:eidx + 1]); one.acc.copy_(acc[eidx:eidx + 1])
        b = one.rollout(T)
        torch.cuda.synchronize()
        for key in ("obs", "reward", "terminated", "truncated"):
            assert torch.equal(a[key][:, eidx].contiguous().view(torch.int32), b[key][:, 0].contiguous().view(torch.int32)), (eidx, key)
        one.close()
    big.close()
