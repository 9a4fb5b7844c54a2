"""Run the oracle side by side with the real reference (only where /root/reference exists; the GPU
box and any other machine skip this file -- the committed fixtures cover them).

Route 1 (stub-free, tests/golden/_reference_loader.load_core) executes the reference's own
area/pedestrians/statuses/distances/reward modules with nothing replaced."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import _reference_loader as L  # noqa: E402
from oracle import evac_oracle as O  # noqa: E402

pytestmark = pytest.mark.skipif(not L.reference_available(), reason="reference checkout not present")


@pytest.mark.parametrize("n,seed,noise,ens", [(60, 11, 0.2, 1.0), (60, 12, 0.8, 0.5), (17, 13, 0.05, 0.1), (256, 14, 0.2, 1.0)])
def test_core_dynamics_free_running_same_global_rng(n, seed, noise, ens):
    core = L.load_core()
    rw = core.reward.Reward(True, True, False, -1.0)
    area = core.area.Area(rw, 1.0, 1.0, 0.01, noise, 1e-8)
    agent = core.area.Agent(ens)
    agent.reset()
    peds = core.pedestrians.Pedestrians(n)
    np.random.seed(seed)
    peds.reset(agent.position, area.exit.position)

    p = O.OracleParams(number_of_pedestrians=n, noise_coef=noise, enslaving_degree=ens,
                       is_new_exiting_reward=True, is_new_followers_reward=True)
    np.random.seed(seed)
    st = O.env_reset(p, *O.draw_reset(n))
    np.testing.assert_array_equal(st.pos, peds.positions)
    act = np.random.Generator(np.random.PCG64(seed))
    steps = 150 if n <= 60 else 30
    for t in range(steps):
        a = act.uniform(-1, 1, 2).astype(np.float32)
        rng_state = np.random.get_state()
        agent, term_a, r_a = area.agent_step(a.copy(), agent)
        peds, term_p, r_p, r_i = area.pedestrians_step(peds, agent, t + 1)
        np.random.set_state(rng_state)                 # the oracle consumes the same global stream
        nz = O.draw_step_noise(p, st)
        out = O.env_step(p, st, a, nz)
        np.testing.assert_allclose(st.pos, peds.positions, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(st.dir, peds.directions, rtol=1e-9, atol=1e-9)
        np.testing.assert_array_equal(st.status, L.status_codes(peds.statuses))
        np.testing.assert_array_equal(st.agent_pos, agent.position)
        np.testing.assert_allclose(out["reward_ped"], r_p, rtol=1e-12)
        np.testing.assert_allclose(out["intrinsic"], r_i, rtol=1e-9)
        assert out["reward_agent"] == r_a


def test_full_env_with_shells_matches_stub_free_core():
    """The gymnasium/wandb shells used for fixture generation do not change the dynamics."""
    full = L.load_full()
    cfg = full.EnvConfig(number_of_pedestrians=30, wandb_enabled=False, path_logs=L.log_dir(),
                         is_new_exiting_reward=True)
    env = full.setup_env(cfg, full.EnvWrappersConfig(positions="grav", alpha=3))
    np.random.seed(5)
    env.reset()
    acts = np.random.Generator(np.random.PCG64(5)).uniform(-1, 1, (40, 2)).astype(np.float32)
    traj = []
    for a in acts:
        obs, r, te, tr, _ = env.step(a)
        traj.append((env.unwrapped.pedestrians.positions.copy(), r))

    core = L.load_core()
    rw = core.reward.Reward(True, True, False, -1.0)
    area = core.area.Area(rw, 1.0, 1.0, 0.01, 0.2, 1e-8)
    agent = core.area.Agent(1.0); agent.reset()
    peds = core.pedestrians.Pedestrians(30)
    np.random.seed(5)
    peds.reset(agent.position, area.exit.position)
    for t, a in enumerate(acts):
        agent, _, r_a = area.agent_step(a.copy(), agent)
        peds, _, r_p, r_i = area.pedestrians_step(peds, agent, t + 1)
        np.testing.assert_array_equal(peds.positions, traj[t][0])
        assert r_a + r_p + 0.0 * r_i == traj[t][1]
