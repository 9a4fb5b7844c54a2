"""HostVectorEnv: the batched env behind the surface ``gym.vector.SyncVectorEnv`` gives the reference trainer.

The reference's rollout loop (/root/reference/src/agents/rpo_agent.py:168,193-203) hands the vector env a NumPy action
batch and does NumPy / ``torch.Tensor(...)`` work on what comes back::

    next_obs, reward, terminations, truncations, infos = self.envs.step(action.cpu().numpy())
    done = np.logical_or(terminations, truncations)
    rewards[step] = torch.tensor(reward).to(self.device).view(-1)
    next_obs, next_done = torch.Tensor(next_obs).to(self.device), torch.Tensor(done).to(self.device)

``BatchedEvacuationEnv`` / ``NormalizedVectorEnv`` return device tensors (the zero-copy form a GPU trainer wants); this
adapter is the drop-in for the UNMODIFIED trainer: NumPy actions in; ``obs [E, D]`` float32, ``reward [E]`` float64,
``terminations`` / ``truncations [E]`` bool NumPy arrays out (the dtypes of SyncVectorEnv's own buffers); ``infos`` as the
device env builds them (``"final_info" in infos`` / ``infos["final_info"]`` with the reference's episode record and
``episode = {"r", "l"}``; the lazily built list reuses the flags this adapter has already brought to the host).

One pinned staging buffer each way: the actions go up and the packed step outputs come down with one asynchronous copy
each and ONE stream synchronisation per step.  Like SyncVectorEnv the returned arrays are buffers of the env that the next
``step`` overwrites (``copy=True``, SyncVectorEnv's default, hands out copies instead)."""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np
import torch

from .vector_env import BatchedEvacuationEnv
from .wrappers import NormalizedVectorEnv


class HostVectorEnv:
    """``gym.vector.SyncVectorEnv``-shaped host face of a batched device env (see the module docstring).

    ``env``: a ``BatchedEvacuationEnv`` or a ``NormalizedVectorEnv`` (the trainer's wrapper chain).  ``make`` builds what
    ``SyncVectorEnv([make_env(env_config, wrap_config, gamma)] * num_envs)`` builds (rpo_agent.py:35-39,123-126)."""

    def __init__(self, env, copy: bool = True):
        self.env = env
        base = env.env if isinstance(env, NormalizedVectorEnv) else env
        if not isinstance(base, BatchedEvacuationEnv):
            raise TypeError("HostVectorEnv wraps a BatchedEvacuationEnv or a NormalizedVectorEnv")
        self._base = base
        self.copy = bool(copy)
        self.num_envs, self.obs_dim, self.device = base.num_envs, base.obs_dim, base.device
        self.single_action_space, self.single_observation_space = base.single_action_space, base.single_observation_space
        self.action_space, self.observation_space = base.action_space, base.observation_space
        E, D = self.num_envs, self.obs_dim
        # device-side packing of the step outputs: [obs(D) | reward | terminated | truncated] per env, one message down
        self._d_out = torch.empty((E, D + 3), dtype=torch.float32, device=self.device)
        self._h_out = torch.empty((E, D + 3), dtype=torch.float32, pin_memory=True)
        self._h_act = torch.empty((E, 2), dtype=torch.float32, pin_memory=True)
        self._d_act = torch.empty((E, 2), dtype=torch.float32, device=self.device)
        self._np_out, self._np_act = self._h_out.numpy(), self._h_act.numpy()
        self._reward = np.zeros((E,), dtype=np.float64)            # SyncVectorEnv's buffer dtypes
        self._term = np.zeros((E,), dtype=np.bool_)
        self._trunc = np.zeros((E,), dtype=np.bool_)
        self._obs = np.zeros((E, D), dtype=np.float32)

    @classmethod
    def make(cls, env_config, wrap_config=None, num_envs: int = 1, gamma: float = 0.99, normalize: bool = True, copy: bool = True, **kw):
        """The env + the trainer's wrapper chain (``wrapping(env, gamma)``, rpo_agent.py:24-33) for ``num_envs`` envs."""
        if normalize:
            return cls(NormalizedVectorEnv.make(env_config, wrap_config, num_envs=num_envs, gamma=gamma, **kw), copy=copy)
        cfg = dataclasses.replace(env_config, clip_action=True)
        return cls(BatchedEvacuationEnv(cfg, wrap_config, num_envs=num_envs, autoreset=True, **kw), copy=copy)

    # ------------------------------------------------------------------------------------------
    def _download(self, obs, reward=None, term=None, trunc=None):
        D = self.obs_dim
        out = self._d_out
        out[:, :D].copy_(obs)
        if reward is not None:
            out[:, D].copy_(reward)
            out[:, D + 1].copy_(term)
            out[:, D + 2].copy_(trunc)
        self._h_out.copy_(out, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        np.copyto(self._obs, self._np_out[:, :D])
        if reward is not None:
            np.copyto(self._reward, self._np_out[:, D])               # float32 -> float64, exact
            np.not_equal(self._np_out[:, D + 1], 0.0, out=self._term)
            np.not_equal(self._np_out[:, D + 2], 0.0, out=self._trunc)

    def reset(self, seed: Optional[int] = None, options=None):
        """``SyncVectorEnv.reset(seed=)`` -> ``(obs [E, D] float32, {})``."""
        obs, infos = self.env.reset(seed=seed, options=options)
        self._download(obs)
        return (self._obs.copy() if self.copy else self._obs), infos

    def step(self, actions):
        """``SyncVectorEnv.step(actions [E, 2])`` -> ``(obs, reward, terminations, truncations, infos)`` as NumPy arrays."""
        a = np.asarray(actions, dtype=np.float32)
        if a.shape != (self.num_envs, 2):
            raise ValueError(f"actions: expected shape {(self.num_envs, 2)}, got {a.shape}")
        np.copyto(self._np_act, a)
        self._d_act.copy_(self._h_act, non_blocking=True)
        obs, reward, term, trunc, infos = self.env.step(self._d_act)
        self._download(obs, reward, term, trunc)
        if hasattr(infos, "_done"):                                   # the lazy final_info list: the flags are on the host already
            infos._done = np.logical_or(self._term, self._trunc)
        if self.copy:
            return self._obs.copy(), self._reward.copy(), self._term.copy(), self._trunc.copy(), infos
        return self._obs, self._reward, self._term, self._trunc, infos

    def close(self):
        self.env.close()
