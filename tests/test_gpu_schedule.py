"""The CU-wide rollout workgroups and their load schedule (evac_bind_workspace; the deal of the next launch made inside
the rollout kernel, k_schedule for the first one and for evac_reschedule) are performance devices:
which wave carries which env, and with which issue priority, must not change a single bit of the results."""
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


def _make(ea, cfg, wrap, E, seed, cu_wide, schedule):
    return ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=seed,
                                   options=ea.KernelOptions(cu_wide=1 if cu_wide else 0, workspace=bool(schedule)))


@pytest.mark.parametrize("n,E,wrap_kw", [
    (60, 1000, dict(positions="grav", alpha=3)),                          # E not a multiple of 16: identity tail of the schedule
    (60, 512, dict(positions="rel", statuses="ohe", type="Box")),
    (33, 77, dict(positions="grav", alpha=2)),
    (64, 160, dict(positions="abs", statuses="cat", type="Dict")),        # the env fills its wave
    (200, 53, dict(positions="grav", alpha=3)),                           # four-wave envs: 4 per CU-wide workgroup, barriers per env in LDS
    (256, 40, dict(positions="rel", statuses="ohe", type="Box")),
])
def test_cu_wide_scheduled_rollout_is_bit_identical(ea, n, E, wrap_kw):
    import torch
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=70, is_new_exiting_reward=True, intrinsic_reward_coef=0.5)
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    ref = _make(ea, cfg, wrap, E, 7, cu_wide=False, schedule=False)       # four envs per 256-thread workgroup
    wide = _make(ea, cfg, wrap, E, 7, cu_wide=True, schedule=False)       # sixteen per CU-wide workgroup, identity order
    sched = _make(ea, cfg, wrap, E, 7, cu_wide=True, schedule=True)       # + envs dealt to SIMDs by load
    assert "CU-wide" in wide.kernel_variant("rollout") and "CU-wide" not in ref.kernel_variant("rollout")
    for env in (ref, wide, sched):
        env.reset()
    outs = [[], [], []]
    for chunk in (30, 60, 25, 60):                                        # every launch deals the next one's envs; episodes end inside
        for k, env in enumerate((ref, wide, sched)):
            r = env.rollout(chunk)
            outs[k].append({key: r[key].clone() for key in ("obs", "reward", "terminated", "truncated", "episode_stats")})
    torch.cuda.synchronize()
    for c in range(len(outs[0])):
        for key in outs[0][c]:
            a = outs[0][c][key]
            for k in (1, 2):
                b = outs[k][c][key]
                assert torch.equal(a.view(torch.uint8) if a.dtype != torch.uint8 else a,
                                   b.view(torch.uint8) if b.dtype != torch.uint8 else b), (c, key, k)
    sa = ref.get_state()
    for env in (wide, sched):
        sb = env.get_state()
        for key in sa:
            assert torch.equal(sa[key], sb[key]), key
    # the schedule is a permutation of the envs (both buffers: the one the last launch ran under, the one it dealt)
    assert sched.schedule_generation() == 4
    for buf in (2, 3):
        assert sorted(sched.schedule[buf].cpu().numpy().tolist()) == list(range(E))
    moving = sched.schedule_loads().cpu().numpy()
    st = sa["status"].cpu().numpy()
    if n <= 64:     # (one-wave envs: the length of the pair loop -- the moving pedestrians, or 0 without a row to evaluate)
        rows = ((st == 1) | ((st == 2) & (cfg.enslaving_degree != 1.0))).any(1)
        assert (moving == np.where(rows, ((st >= 1) & (st <= 3)).sum(1), 0)).all()       # what the last launch left behind
    else:
        assert (moving == ((st >= 1) & (st <= 3)).sum(1)).all()
    for env in (ref, wide, sched):
        env.close()


def test_schedule_balances_simd_groups(ea):
    """The deal (made inside the rollout kernel, and by k_schedule) on the loads a real episode produces: every group of four SIMD-mates (waves w, w+4, w+8, w+12 of a
    workgroup) gets one env of each load quartile, and the heaviest SIMD is lighter than with random placement."""
    import torch
    E = 4096
    cfg = ea.EnvConfig(number_of_pedestrians=60)
    env = _make(ea, cfg, ea.EnvWrappersConfig(positions="grav"), E, 1, cu_wide=True, schedule=True)
    env.reset()
    for _ in range(15):                                                    # launches of < 50 steps deal the next launch's envs themselves
        env.rollout(40)
    torch.cuda.synchronize()
    assert env.schedule_generation() == 15
    load = env.schedule_loads().cpu().numpy().copy()                       # pedestrians still moving, per env, at t = 600
    prev = env.schedule[(15 - 2) & 1].cpu().numpy().copy()                 # ... and at t = 560: what launch 14 dealt launch 15's envs by
    assert load.min() >= 0 and load.max() <= 60 and load.std() > 3
    dealt_in_kernel = env.schedule_perm().cpu().numpy().copy()             # the deal launch 14 made for launch 15 (workgroup 0, at its start)
    _check_deal(dealt_in_kernel, prev, E)
    env.rebind_workspace()                                                 # evac_reschedule: the same deal from the latest loads, now
    torch.cuda.synchronize()
    perm = env.schedule_perm().cpu().numpy()
    _check_deal(perm, load, E)
    env.close()


def _check_deal(perm, load, E):
    assert sorted(perm.tolist()) == list(range(E))
    by_slot = load[perm]
    q = np.sort(load)
    assert by_slot[:16].max() <= q[15]                                     # workgroup 0 (it deals the next launch first): the 16 lightest envs
    slot_load = by_slot[16:].reshape(E // 16 - 1, 4, 4)                    # the others: [workgroup][k-th wave of the SIMD][SIMD]
    sums = slot_load.sum(1)                                                # per SIMD
    rng = np.random.default_rng(0)
    rand = load[rng.permutation(E)].reshape(-1, 4).sum(1)
    assert sums.max() < rand.max() and sums.max() - sums.min() < 0.5 * (rand.max() - rand.min())
    q, n = q[16:], E - 16
    for k in range(4):                                                     # one env per quartile in every SIMD ...
        lo, hi = q[k * n // 4], q[(k + 1) * n // 4 - 1]
        assert ((slot_load[:, 3 - k, :] >= lo) & (slot_load[:, 3 - k, :] <= hi)).all()   # ... the heaviest in the SIMD's first (oldest) wave


@pytest.mark.parametrize("n,E,cu_wide,box", [(60, 333, False, False), (60, 333, True, False), (10, 500, False, False), (30, 200, False, True),
                                              (256, 37, False, False), (256, 37, True, False), (100, 50, False, True),
                                              (1024, 8, False, True), (1024, 40, False, True), (700, 12, False, False)])
def test_default_config_kernel_is_bit_identical(ea, n, E, cu_wide, box):
    """The rollout kernels specialised for the reference's default configuration (k_rollout_default_config: gravity
    observation with alpha = 3, or the Box of relative positions + one-hot statuses) against the generic ones
    (EVAC_SPECIALIZE=0), for every kernel family: sub-wave, one wave, multi-wave, CU-wide, teams, cell list."""
    import torch
    # (the status rewards and ClipAction stay run-time options of the specialised kernels: vary them with the case)
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=90, is_new_exiting_reward=bool(E % 2), is_new_followers_reward=bool(n % 3),
                       clip_action=bool(E % 3 == 0))
    wrap = ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box") if box else ea.EnvWrappersConfig(positions="grav", alpha=3)
    envs = []
    for spec in (0, 1):
        envs.append(ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=21,
                                            options=ea.KernelOptions(specialize=spec, cu_wide=1 if cu_wide else 0)))
    gen, spec = envs
    gen.reset(); spec.reset()
    for T in (40, 100, 3, 64):
        a, b = gen.rollout(T), spec.rollout(T)
        torch.cuda.synchronize()
        assert spec.team_error() == 0
        assert torch.equal(a["slab"].view(torch.int32), b["slab"].view(torch.int32))
        assert torch.equal(a["episode_stats"].view(torch.int32), b["episode_stats"].view(torch.int32))
    for _ in range(5):                                                     # and the step API (k_step_default_config)
        act = torch.rand((E, 2), device=gen.device) * 2 - 1
        o1, r1, t1, u1, _ = gen.step(act)
        o2, r2, t2, u2, _ = spec.step(act)
        assert torch.equal(o1.view(torch.int32), o2.view(torch.int32)) and torch.equal(r1.view(torch.int32), r2.view(torch.int32))
        assert torch.equal(t1, t2) and torch.equal(u1, u2)
    sa, sb = gen.get_state(), spec.get_state()
    assert all(torch.equal(sa[k], sb[k]) for k in sa) and torch.equal(gen.acc, spec.acc) and torch.equal(gen.clock, spec.clock)
    gen.close(); spec.close()
