"""Team kernels (csrc/evac_team.h): one env of 513..1024 pedestrians on 2 / 4 / 8 CUs.  They must reproduce the
one-workgroup-per-env cell-list kernels bit for bit -- integer heading sums, the same reduction tree -- whatever the team
size, and never leave a barrier hanging."""
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


def _make(ea, cfg, wrap, E, seed, team):
    from evacuation_amd.options import current_default          # (inside a `with _env_var(...)` block: that block's options + the team size)
    return ea.BatchedEvacuationEnv(cfg, wrap, num_envs=E, seed=seed, options=current_default().replace(team=int(team)))


@pytest.mark.parametrize("n,E,team,wrap_kw,ens", [
    (1024, 8, 8, dict(positions="rel", statuses="ohe", type="Box"), 1.0),     # BASELINE config 5's observation
    (1024, 5, 8, dict(positions="grav", alpha=3), 1.0),                       # E not a multiple of 8: idle team slots
    (1000, 32, 8, dict(positions="grav", alpha=2), 0.5),                      # follower rows needed (enslaving_degree < 1)
    (600, 16, 4, dict(positions="abs", statuses="cat", type="Dict"), 1.0),    # members without pedestrians (600 < 1024)
    (1024, 16, 2, dict(positions="rel", statuses="no", type="Box"), 1.0),
    (777, 40, 4, dict(positions="grav", alpha=3), 0.1),
    (1024, 16, 16, dict(positions="rel", statuses="ohe", type="Box"), 1.0),   # teams of 16 (batches of <= 16 envs per GPU): one ped wave per member
    (700, 9, 16, dict(positions="grav", alpha=3), 0.5),                       # ... members without pedestrians, idle team slots, follower rows
])
def test_team_rollout_equals_one_workgroup_per_env(ea, n, E, team, wrap_kw, ens):
    import torch
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=35, is_new_exiting_reward=True, intrinsic_reward_coef=0.5,
                       enslaving_degree=ens)
    wrap = ea.EnvWrappersConfig(**wrap_kw)
    ref = _make(ea, cfg, wrap, E, 11, team=0)
    tm = _make(ea, cfg, wrap, E, 11, team=team)
    assert "CUs/env" in tm.kernel_variant("rollout") and "CUs/env" not in ref.kernel_variant("rollout")
    ref.reset(); tm.reset()
    for chunk in (20, 30, 1, 25):                        # episodes end (and envs reset) inside the launches
        a = ref.rollout(chunk)
        b = tm.rollout(chunk)
        torch.cuda.synchronize()
        assert tm.team_error() == 0
        for key in ("obs", "reward", "terminated", "truncated", "episode_stats"):
            assert torch.equal(a[key].view(torch.int32), b[key].view(torch.int32)), (chunk, key)
    sa, sb = ref.get_state(), tm.get_state()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), key
    # and the step API of the team handle (one workgroup per env) continues the same trajectory
    act = torch.rand((E, 2), device=ref.device) * 2 - 1
    o1, r1, t1, u1, _ = ref.step(act)
    o2, r2, t2, u2, _ = tm.step(act)
    assert torch.equal(o1.view(torch.int32), o2.view(torch.int32)) and torch.equal(r1.view(torch.int32), r2.view(torch.int32))
    ref.close(); tm.close()


@pytest.mark.parametrize("n,E,team,ens", [(1024, 8, 8, 1.0), (1024, 8, 2, 1.0), (900, 8, 4, 1.0), (1024, 4, 8, 0.9), (1024, 8, 16, 1.0)])
def test_team_rollout_late_in_an_episode(ea, n, E, team, ens):
    """Late in an episode a member has few rows to evaluate (only its VISCEK pedestrians under enslaving_degree 1) and the
    pair sweep runs transposed -- rows dealt to the waves, lanes over the columns (Team::neighbour_sum); with
    enslaving_degree < 1 the followers keep their rows and both forms alternate.  Same bits as the cell-list kernels all the
    way through an episode."""
    import torch
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=1500, is_new_exiting_reward=True, enslaving_degree=ens)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    ref = _make(ea, cfg, wrap, E, 5, team=0)
    tm = _make(ea, cfg, wrap, E, 5, team=team)
    ref.reset(); tm.reset()
    for chunk in (300, 300, 300, 300, 301, 100):          # the truncation at 1500 and the reset fall inside the fifth launch
        a = ref.rollout(chunk)
        b = tm.rollout(chunk)
        torch.cuda.synchronize()
        assert tm.team_error() == 0
        for key in ("obs", "reward", "terminated", "truncated"):
            assert torch.equal(a[key].view(torch.int32), b[key].view(torch.int32)), (chunk, key)
    sa, sb = ref.get_state(), tm.get_state()
    for key in sa:
        assert torch.equal(sa[key], sb[key]), key
    st = sa["status"] if "status" in sa else None
    ref.close(); tm.close()


def test_team_nan_poisoning_reaches_every_member(ea):
    """A zero direction (0/0 heading, area.py:101) poisons every FOLLOWER / VISCEK pedestrian of the env in the reference
    (area.py:118-119): the flag has to cross the team."""
    import torch
    n, E = 1024, 8
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=1000)
    wrap = ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box")
    ref = _make(ea, cfg, wrap, E, 3, team=0)
    tm = _make(ea, cfg, wrap, E, 3, team=8)
    for env in (ref, tm):
        env.reset()
        st = env.get_state()
        d = st["dir"].clone()
        who = 640 + int((st["status"][2, 640:] == 1).nonzero()[0])    # env 2, a VISCEK pedestrian of the sixth member or later
        d[2, who] = 0.0
        env.set_state(dir=d)
    a = ref.rollout(3); b = tm.rollout(3)
    torch.cuda.synchronize()
    assert tm.team_error() == 0
    sa, sb = ref.get_state(), tm.get_state()
    assert torch.isnan(sa["pos"][2]).any() and not torch.isnan(sa["pos"][1]).any()
    for key in sa:
        assert torch.equal(sa[key].view(torch.uint8) if sa[key].dtype == torch.uint8 else sa[key].view(torch.int32),
                           sb[key].view(torch.uint8) if sb[key].dtype == torch.uint8 else sb[key].view(torch.int32)), key
    assert torch.equal(a["obs"].view(torch.int32), b["obs"].view(torch.int32))
    ref.close(); tm.close()


def test_workspace_binding_errors_and_sizes(ea):
    """evac_workspace_bytes / evac_bind_workspace through the C ABI: too small or misaligned workspaces are refused, NULL
    unbinds (rollouts then run the one-workgroup kernels), and the handle reports which kernels it will launch."""
    import ctypes as C
    import torch
    from evacuation_amd import _lib
    lib = _lib.load()
    cfg = ea.EnvConfig(number_of_pedestrians=1024, max_timesteps=50)
    env = _make(ea, cfg, ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box"), 8, 1, team=8)
    need = int(lib.evac_workspace_bytes(env._h))
    assert need >= 2 * 8 * 4 + 2 * 8 * 1024 * 16                         # schedule + at least the teams' double-buffered tile
    buf = torch.zeros(need + 512, dtype=torch.uint8, device=env.device)
    base = buf.data_ptr()
    aligned = (base + 255) // 256 * 256
    assert lib.evac_bind_workspace(env._h, C.c_void_p(aligned), C.c_int64(need - 1)) == _lib.ERR_INVALID_ARGUMENT
    assert lib.evac_bind_workspace(env._h, C.c_void_p(aligned + 8), C.c_int64(need)) == _lib.ERR_INVALID_ARGUMENT
    assert lib.evac_bind_workspace(env._h, None, C.c_int64(0)) == _lib.EVAC_OK          # unbound: plain kernels
    env.reset()
    a = env.rollout(5)["obs"].clone()
    assert lib.evac_bind_workspace(env._h, C.c_void_p(aligned), C.c_int64(need)) == _lib.EVAC_OK
    ref = _make(ea, cfg, ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box"), 8, 1, team=0)
    ref.reset()
    b = ref.rollout(5)["obs"]
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    c2 = env.rollout(5)["obs"]; d2 = ref.rollout(5)["obs"]                   # now through the teams
    torch.cuda.synchronize()
    assert env.team_error() == 0 and torch.equal(c2.view(torch.int32), d2.view(torch.int32))
    env.close(); ref.close()


def _env_var(name, value):
    """The create-time option behind the diagnostic switch `name`, for the handles made inside the block."""
    from evacuation_amd.options import from_switches, kernel_options
    return kernel_options(from_switches(**{name: value}))


def test_team_that_loses_a_member_is_reported_and_keeps_its_state(ea):
    """a team whose members are not all resident times out.  Fault injection (EVAC_TEAM_FAULT=1: the grid
    is launched one workgroup short, so the last env's team never completes a barrier): the launch still ends, that env's state
    is NOT written back, the error word is raised, every later call on the handle returns EVAC_ERR_TEAM_ABORTED without a
    sync until it is cleared, and the handle then runs one workgroup per env."""
    import torch
    from evacuation_amd import _lib
    n, E = 1024, 8
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=100)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    with _env_var("EVAC_TEAM_FAULT", "1"):
        tm = _make(ea, cfg, wrap, E, 9, team=8)
    ref = _make(ea, cfg, wrap, E, 9, team=0)
    tm.reset(); ref.reset()
    assert "CUs/env" in tm.kernel_variant("rollout")
    before = {k: v.clone() for k, v in tm.get_state().items()}
    tm.rollout(3)                                        # returns at once: the error shows up on the device later
    a = ref.rollout(3)
    torch.cuda.synchronize()
    assert tm.team_error() == 1
    with pytest.raises(_lib.EvacError) as ei:
        tm.get_state()
    assert ei.value.code == _lib.ERR_TEAM_ABORTED and "lost a member" in str(ei.value)
    with pytest.raises(_lib.EvacError):
        tm.rollout(3)
    tm.team_clear_error()
    assert tm.team_error() == 0 and "CUs/env" not in tm.kernel_variant("rollout")     # one workgroup per env from now on
    after = tm.get_state()
    for k in before:                                     # the env whose team broke (the last one) kept its pre-launch state ...
        assert torch.equal(before[k][E - 1], after[k][E - 1]), k
    sr = ref.get_state()
    for k in sr:                                         # ... the complete teams stepped as the reference kernels do
        assert torch.equal(sr[k][: E - 1], after[k][: E - 1]), k
    tm.reset(); ref.reset()                              # restore the batch, then the handle works again (cell-list kernels)
    b = tm.rollout(4); a = ref.rollout(4)
    torch.cuda.synchronize()
    # (the env that was not stepped is three steps behind in its Philox noise counter: compare the others)
    assert torch.equal(a["obs"][:, : E - 1].view(torch.int32), b["obs"][:, : E - 1].view(torch.int32))
    tm.close(); ref.close()


@pytest.mark.parametrize("coop", ["1", "0"])
def test_team_rollout_with_another_stream_busy(ea, coop):
    """the sharded env overlaps the all-gather of one chunk with the next rollout, so the team kernels
    must stay correct while a second stream keeps kernels resident on the device (a plain launch whose bounded waits simply
    outlast the intruder -- the default --, or the cooperative launch of EVAC_TEAM_COOP=1): no team error, same bits as the
    one-workgroup kernels."""
    import torch
    n, E = 1024, 32
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=60, is_new_exiting_reward=True)
    wrap = ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box")
    ref = _make(ea, cfg, wrap, E, 21, team=0)
    with _env_var("EVAC_TEAM_COOP", coop):
        tm = _make(ea, cfg, wrap, E, 21, team=8)
    ref.reset(); tm.reset()
    outs = [ref.rollout(25)["slab"].clone() for _ in range(4)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.zeros(64 * 1024 * 1024, device=tm.device)          # 256 MB: every pass keeps all CUs busy for ~0.1 ms
    got = []
    for k in range(4):
        with torch.cuda.stream(side):
            for _ in range(40):
                big.mul_(1.0001).add_(1.0)
        got.append(tm.rollout(25)["slab"].clone())
    torch.cuda.synchronize()
    assert tm.team_error() == 0
    for a, b in zip(outs, got):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    ref.close(); tm.close()


def test_two_team_grids_on_two_streams_take_turns(ea):
    """two team-capable handles on one device, each on a stream of its own (train and eval envs) -- each
    grid alone fits the device (16 envs x 16 CUs = every CU), both at once do not, and members polling on the CUs the other
    grid's missing members need would time out on both sides.  evac_rollout chains the team launches of a device (a device-side
    wait on the previous team launch's event): no team error on either handle, same bits as the one-workgroup kernels."""
    import torch
    n, E = 1024, 16
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=45, is_new_exiting_reward=True)
    wrap = ea.EnvWrappersConfig(positions="rel", statuses="ohe", type="Box")
    refs = [_make(ea, cfg, wrap, E, 31 + k, team=0) for k in range(2)]
    tms = [_make(ea, cfg, wrap, E, 31 + k, team=16) for k in range(2)]
    assert all("16 CUs/env" in t.kernel_variant("rollout") for t in tms)        # (known at bind time: evac_bind_workspace checks the fit)
    for e in refs + tms:
        e.reset()
    want = [[r.rollout(20)["slab"].clone() for _ in range(5)] for r in refs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [[], []]
    for k in range(5):                                  # launches of the two handles alternate, nothing waited for in between
        for j in (0, 1):
            with torch.cuda.stream(streams[j]):
                got[j].append(tms[j].rollout(20)["slab"].clone())
    torch.cuda.synchronize()
    for j in (0, 1):
        assert tms[j].team_error(sync=False) == 0 and tms[j].team_error() == 0
        for a, b in zip(want[j], got[j]):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    for e in refs + tms:
        e.close()


@pytest.mark.parametrize("n,team", [(1024, 8), (1024, 16), (1024, 2), (513, 8), (640, 4)])
def test_packed_f32_heading_sums_are_exact_at_their_limit(ea, n, team):
    """The team sweeps add the integer headings eight at a time in packed f32 before the partial sum goes to an integer
    accumulator (evac_team.h, kTeamExactBatch): exact only below 2^24.  The worst case -- every pedestrian of the room inside one
    interaction radius, all headings along one axis, so that every partial sum is eight times the scale (2^21 - 17 for N = 1024,
    capped at 2^21 - 16 for smaller rooms) -- must still reproduce the cell-list kernels' integer sums bit for bit."""
    import torch
    E = 8 if team != 16 else 4
    cfg = ea.EnvConfig(number_of_pedestrians=n, max_timesteps=1000, is_new_exiting_reward=True, enslaving_degree=1.0, noise_coef=0.0)
    wrap = ea.EnvWrappersConfig(positions="grav", alpha=3)
    ref = _make(ea, cfg, wrap, E, 3, team=0)
    tm = _make(ea, cfg, wrap, E, 3, team=team)
    assert "CUs/env" in tm.kernel_variant("rollout") and "CUs/env" not in ref.kernel_variant("rollout")
    ref.reset(); tm.reset()
    g = torch.Generator().manual_seed(n + team)
    for axis in range(8):                                       # +x, -x, +y, -y; then the same with few rows (the transposed sweep)
        few_rows = axis >= 4
        axis %= 4
        d = torch.zeros((E, n, 2))
        d[..., axis // 2] = 0.01 * (1.0 if axis % 2 == 0 else -1.0)
        pos = (torch.rand((E, n, 2), generator=g) - 0.5) * 0.05 + torch.tensor([0.3, 0.4])        # one blob, well inside the radius of 0.1
        status = torch.ones((E, n), dtype=torch.uint8)                                            # all VISCEK: every row is evaluated
        if few_rows:                                            # three of four pedestrians FOLLOWERS: columns, but no rows (enslaving_degree 1)
            status[:, torch.arange(n) % 4 != 0] = 2
        st = dict(pos=pos, dir=d, status=status,
                  agent_pos=torch.tensor([[-0.8, -0.8]]).repeat(E, 1), agent_dir=torch.zeros((E, 2)), now=torch.zeros((E,), dtype=torch.int32))
        ref.set_state(**st); tm.set_state(**st)
        a, b = ref.rollout(3), tm.rollout(3)
        torch.cuda.synchronize()
        assert tm.team_error() == 0
        for key in ("obs", "reward", "terminated", "truncated"):
            assert torch.equal(a[key].view(torch.int32), b[key].view(torch.int32)), (axis, key)
        sa, sb = ref.get_state(), tm.get_state()
        for key in sa:
            assert torch.equal(sa[key], sb[key]), (axis, few_rows, key)
    ref.close(); tm.close()
