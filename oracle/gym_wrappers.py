"""The trainer's per-env gymnasium wrapper chain restated in NumPy -- TEST INFRASTRUCTURE.

Reference call site: ``wrapping(env, gamma)`` /root/reference/src/agents/rpo_agent.py:24-33, applied
to every sub-env of ``gym.vector.SyncVectorEnv`` (rpo_agent.py:35-39,123-126):

    FlattenObservation -> RecordEpisodeStatistics -> ClipAction -> NormalizeObservation
    -> TransformObservation(clip(obs, -1, 1)) -> NormalizeReward(gamma) -> TransformReward(clip(r, -100, 100))

PARITY UNPINNED: gymnasium is a third-party dependency that the reference does not pin
(requirements.txt:6 says just ``gymnasium``) and that is absent from this image (no network), so
nothing here could be checked against the real package.  What is restated is the published
algorithm of gymnasium 0.29.1 (the last release with the ``infos["final_info"]`` API the trainer uses,
rpo_agent.py:198-203):
  * ``gymnasium/wrappers/normalize.py``: ``RunningMeanStd`` (parallel-variance update of Chan et al.
    with count initialised to epsilon = 1e-4, mean 0, var 1, all float64), ``NormalizeObservation``
    (update with the observation, then ``(obs - mean) / sqrt(var + 1e-8)``; also on ``reset``),
    ``NormalizeReward`` (``returns = returns * gamma * (1 - terminated) + reward``, update the return
    statistics with ``returns``, ``reward / sqrt(var + 1e-8)``);
  * ``gymnasium/wrappers/clip_action.py``: ``np.clip(action, low, high)``;
  * ``gymnasium/vector/sync_vector_env.py``: when a sub-env is done its (wrapped) ``reset`` is called in
    the same step, the wrapped terminal observation goes to ``infos["final_observation"]``.
Each sub-env has its OWN statistics because the wrappers are applied inside the thunk.
"""
from __future__ import annotations

import numpy as np


class RunningMeanStd:
    """gymnasium 0.29.1 wrappers/normalize.py RunningMeanStd."""

    def __init__(self, epsilon: float = 1e-4, shape=()):
        self.mean = np.zeros(shape, "float64")
        self.var = np.ones(shape, "float64")
        self.count = epsilon

    def update(self, x):
        x = np.asarray(x, dtype=np.float64)
        batch_mean, batch_var, batch_count = np.mean(x, axis=0), np.var(x, axis=0), x.shape[0]
        delta = batch_mean - self.mean
        tot = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot
        m2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / tot
        self.mean, self.var, self.count = new_mean, m2 / tot, tot


class WrappedEnvStats:
    """Statistics of ONE wrapped sub-env (NormalizeObservation + NormalizeReward state)."""

    def __init__(self, obs_dim: int, gamma: float = 0.99, epsilon: float = 1e-8, obs_clip: float = 1.0,
                 reward_clip: float = 100.0):
        self.obs_rms = RunningMeanStd(shape=(obs_dim,))
        self.return_rms = RunningMeanStd(shape=())
        self.returns = np.zeros(1)
        self.gamma, self.epsilon, self.obs_clip, self.reward_clip = gamma, epsilon, obs_clip, reward_clip

    def observation(self, obs):
        """NormalizeObservation.normalize + TransformObservation(clip)."""
        self.obs_rms.update(np.asarray(obs, dtype=np.float64)[None])
        out = (obs - self.obs_rms.mean) / np.sqrt(self.obs_rms.var + self.epsilon)
        return np.clip(out, -self.obs_clip, self.obs_clip)

    def reward(self, rew, terminated):
        """NormalizeReward.step + TransformReward(clip)."""
        self.returns = self.returns * self.gamma * (1 - float(terminated)) + rew
        self.return_rms.update(self.returns)
        out = rew / np.sqrt(self.return_rms.var + self.epsilon)
        return float(np.clip(out, -self.reward_clip, self.reward_clip))


def clip_action(action, low=-1.0, high=1.0):
    """ClipAction.action."""
    return np.clip(action, low, high)


def vector_step(stats, raw_obs, raw_final_obs, raw_reward, terminated, truncated):
    """What SyncVectorEnv returns for one step given the raw (unwrapped, flattened) outputs of every
    sub-env: ``raw_obs`` is the reset observation for sub-envs that finished, ``raw_final_obs`` their
    terminal observation.  Returns (obs, final_obs, reward); statistics are updated in place."""
    E = len(stats)
    obs = np.zeros_like(np.asarray(raw_obs, dtype=np.float64))
    fin = np.zeros_like(obs)
    rew = np.zeros(E)
    for e in range(E):
        done = bool(terminated[e]) or bool(truncated[e])
        if done:
            fin[e] = stats[e].observation(raw_final_obs[e])     # wrapped env.step(): terminal observation
            rew[e] = stats[e].reward(raw_reward[e], terminated[e])
            obs[e] = stats[e].observation(raw_obs[e])           # wrapped env.reset() in the same step
        else:
            obs[e] = stats[e].observation(raw_obs[e])
            rew[e] = stats[e].reward(raw_reward[e], terminated[e])
    return obs, fin, rew
