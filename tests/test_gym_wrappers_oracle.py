"""CPU checks of the wrapper-chain restatement (oracle/gym_wrappers.py).  gymnasium is absent, so these
pin the restatement to first principles: the running statistics must equal the batch statistics of
everything seen so far (up to the epsilon-count prior), for any chunking."""
import numpy as np

from oracle import gym_wrappers as G


def test_running_mean_std_equals_batch_statistics():
    rng = np.random.default_rng(0)
    x = rng.normal(3.0, 2.0, size=(500, 4))
    r = G.RunningMeanStd(shape=(4,))
    for row in x:
        r.update(row[None])
    # prior: count 1e-4 at mean 0 / var 1 -> negligible after 500 samples
    np.testing.assert_allclose(r.mean, x.mean(axis=0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r.var, x.var(axis=0), rtol=1e-4)
    r2 = G.RunningMeanStd(shape=(4,))
    r2.update(x[:123]); r2.update(x[123:])
    np.testing.assert_allclose(r2.mean, r.mean, rtol=1e-12)
    np.testing.assert_allclose(r2.var, r.var, rtol=1e-10)
    assert abs(r.count - (500 + 1e-4)) < 1e-9


def test_wrapped_env_semantics():
    s = G.WrappedEnvStats(obs_dim=2, gamma=0.9)
    o1 = s.observation(np.array([1.0, -2.0]))
    assert np.all(np.abs(o1) <= 1.0)
    # first sample: mean ~= x (prior count 1e-4), so the normalised value is ~0
    assert np.all(np.abs(o1) < 0.05)
    r1 = s.reward(2.0, False)
    assert s.returns[0] == 2.0
    r2 = s.reward(1.0, False)
    assert abs(s.returns[0] - (2.0 * 0.9 + 1.0)) < 1e-12
    s.reward(5.0, True)                       # terminated: the return restarts from the reward
    assert s.returns[0] == 5.0
    assert -100.0 <= r1 <= 100.0 and -100.0 <= r2 <= 100.0
    np.testing.assert_array_equal(G.clip_action(np.array([1.5, -0.2])), [1.0, -0.2])


def test_vector_step_updates_twice_on_done():
    st = [G.WrappedEnvStats(3), G.WrappedEnvStats(3)]
    raw = np.array([[0.1, 0.2, 0.3], [0.5, 0.5, 0.5]])
    fin = np.array([[9.0, 9.0, 9.0], [7.0, 7.0, 7.0]])
    obs, f, rew = G.vector_step(st, raw, fin, np.array([-1.0, -1.0]), [False, False], [False, True])
    assert abs(st[0].obs_rms.count - 1.0001) < 1e-12 and abs(st[1].obs_rms.count - 2.0001) < 1e-12
    assert np.all(f[0] == 0) and np.any(f[1] != 0)
