"""Randomised sweep: every scheduling / decomposition device switched on (CU-wide workgroups, load schedule, teams) against
the plain kernels (everything off), over random room sizes, batch sizes, observation modes, enslaving degrees, launch
lengths, with and without caller-provided actions.  States, flags, episode records, observations and rewards must be
bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SWITCHES = ("EVAC_CU_WIDE", "EVAC_TEAM", "EVAC_WORKSPACE")


@pytest.fixture(scope="module")
def ea():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import evacuation_amd
    return evacuation_amd


def _make(ea, cfg, wrap, E, seed, **env):
    from evacuation_amd.options import from_switches
    assert set(env) <= set(SWITCHES)
    return ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=seed, options=from_switches(**env))


CASES = []
_rng = np.random.default_rng(20261003)
for _ in range(14):
    n = int(_rng.choice([33, 48, 60, 64, 130, 200, 256, 600, 777, 1024]))
    E = int(_rng.integers(3, 70)) if n > 64 else int(_rng.integers(20, 300))
    if n > 512:
        E = int(_rng.integers(2, 40))
    mode = _rng.choice(["grav", "grav", "relbox", "absdict"])
    ens = float(_rng.choice([1.0, 1.0, 0.5, 0.1]))
    CASES.append((n, E, str(mode), ens, int(_rng.integers(0, 1 << 30))))


@pytest.mark.parametrize("n,E,mode,ens,seed", CASES)
def test_all_devices_on_equals_all_off(ea, n, E, mode, ens, seed):
    import torch
    wrap_kw = {"grav": dict(positions="grav", alpha=3), "relbox": dict(positions="rel", statuses="ohe", type="Box"),
               "absdict": dict(positions="abs", statuses="cat", type="Dict")}[mode]
    rng = np.random.default_rng(seed)
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=int(rng.integers(40, 400)), is_new_exiting_reward=True,
                       is_new_followers_reward=bool(rng.integers(0, 2)), enslaving_degree=ens, noise_coef=float(rng.choice([0.2, 0.5])))
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    off = _make(ea, cfg, wrap, E, seed % 1000, EVAC_CU_WIDE=0, EVAC_TEAM=0, EVAC_WORKSPACE=0)
    on = _make(ea, cfg, wrap, E, seed % 1000, EVAC_CU_WIDE=1)                          # teams by default where they apply
    off.reset(); on.reset()
    # start some envs late in their episode so that the viscek-only rows and the transposed sweeps come into play
    st = off.get_state()
    status = st["status"].clone()
    esc = torch.from_numpy(rng.random((E, n)) < rng.uniform(0.0, 0.9)).to(status.device)
    status[esc] = 4
    for env in (off, on):
        env.set_state(status=status)
    for T in [int(x) for x in rng.integers(1, 60, size=4)]:
        acts = None
        if rng.integers(0, 2):
            acts = torch.from_numpy(rng.uniform(-1, 1, (T, E, 2)).astype(np.float32)).to(off.device)
        a = off.rollout(T, actions=acts)
        b = on.rollout(T, actions=acts)
        torch.cuda.synchronize()
        assert on.team_error() == 0
        assert torch.equal(a["terminated"], b["terminated"]) and torch.equal(a["truncated"], b["truncated"])
        assert torch.equal(a["obs"].view(torch.int32), b["obs"].view(torch.int32))
        assert torch.equal(a["reward"].view(torch.int32), b["reward"].view(torch.int32))
        assert torch.equal(a["episode_stats"].view(torch.int32), b["episode_stats"].view(torch.int32))
        sa, sb = off.get_state(), on.get_state()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (T, k)
        assert torch.equal(off.clock, on.clock)
    off.close(); on.close()


# ---- chained launches and the persistent kernel (evac_options_t.chain = 1 / 2) over random shapes: runs of launches of RANDOM lengths in
# ---- flight without a join, then a join, against one plain launch after the other
CHAIN_CASES = []
for _ in range(12):
    if _rng.integers(0, 3) == 0:
        n, E = int(_rng.choice([130, 200, 256])), 4 * int(_rng.integers(2, 40))          # four-wave envs, four per CU-wide workgroup
    else:
        n, E = int(_rng.choice([33, 48, 60, 64])), 16 * int(_rng.integers(2, 40))        # one-wave envs, sixteen per workgroup
    CHAIN_CASES.append((n, E, str(_rng.choice(["grav", "grav", "relbox", "absdict"])), float(_rng.choice([1.0, 1.0, 0.5])), int(_rng.integers(1, 3)),
                        int(_rng.integers(0, 1 << 30))))


@pytest.mark.parametrize("n,E,mode,ens,form,seed", CHAIN_CASES)
def test_launches_in_flight_equal_one_launch_after_the_other(ea, n, E, mode, ens, form, seed):
    import torch
    wrap_kw = {"grav": dict(positions="grav", alpha=3), "relbox": dict(positions="rel", statuses="ohe", type="Box"),
               "absdict": dict(positions="abs", statuses="cat", type="Dict")}[mode]
    rng = np.random.default_rng(seed)
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=int(rng.integers(30, 200)), is_new_exiting_reward=True,
                       is_new_followers_reward=bool(rng.integers(0, 2)), enslaving_degree=ens, noise_coef=float(rng.choice([0.2, 0.5])))
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    plain = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=seed % 1000, options=ea.KernelOptions(cu_wide=0, workspace=False))
    flying = ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=seed % 1000, options=ea.KernelOptions(cu_wide=1, chain=form))
    assert ("chained" if form == 1 else "persistent") in flying.kernel_variant(), flying.kernel_variant()
    plain.reset(); flying.reset()
    D = plain.obs_dim
    for burst in range(3):
        lengths = [int(x) for x in rng.integers(1, 40, size=int(rng.integers(2, 9)))]
        outs = [{"slab": torch.empty((T, E, D + 3), device=plain.device), "episode_stats": torch.zeros((T, E, plain.stats_words), device=plain.device)} for T in lengths]
        refs = [plain.rollout(T) for T in lengths]
        for T, o in zip(lengths, outs):
            flying.rollout_launcher(T, o)()                  # nothing waits between them
        flying.join()
        torch.cuda.current_stream().synchronize()
        assert flying.team_error(sync=False) == 0
        for j, (o, r) in enumerate(zip(outs, refs)):
            assert torch.equal(o["slab"].view(torch.int32), r["slab"].view(torch.int32)), (burst, j, lengths)
            done = (r["terminated"] != 0) | (r["truncated"] != 0)
            assert torch.equal(o["episode_stats"].view(torch.int32)[done], r["episode_stats"].view(torch.int32)[done]), (burst, j)
        if burst == 1:                                       # something that is not a rollout in between: both forms start afresh behind it
            act = torch.from_numpy(rng.uniform(-1, 1, (E, 2)).astype(np.float32)).to(plain.device)
            for x, y in zip(plain.step(act)[:4], flying.step(act)[:4]):
                assert torch.equal(x, y)
    sa, sb = plain.get_state(), flying.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(plain.clock, flying.clock)
    plain.close(); flying.close()
