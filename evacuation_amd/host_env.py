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

One pinned buffer each way and ONE stream synchronisation per step.  Round 5: the step kernel reads the actions FROM and writes its
outputs TO those pinned host buffers itself (pinned host memory is mapped into the device's address space at the same address on
ROCm; checked with hipHostGetDevicePointer, with the staged form of round 4 -- an asynchronous copy each way -- as the fallback):
no copy calls, one pre-bound launch.  Like SyncVectorEnv the returned arrays are buffers of the env that the next ``step`` overwrites
(``copy=True``, SyncVectorEnv's default, hands out copies instead)."""
from __future__ import annotations

import dataclasses
import sys
from typing import Optional

import numpy as np
import torch

from .vector_env import BatchedEvacuationEnv, StepInfos
from .wrappers import NormalizedVectorEnv


class HostVectorEnv:
    """``gym.vector.SyncVectorEnv``-shaped host face of a batched device env (see the module docstring).

    ``env``: a ``BatchedEvacuationEnv`` or a ``NormalizedVectorEnv`` (the trainer's wrapper chain).  ``make`` builds what
    ``SyncVectorEnv([make_env(env_config, wrap_config, gamma)] * num_envs)`` builds (rpo_agent.py:35-39,123-126)."""

    def __init__(self, env, copy: bool = True, zero_copy: bool = True):
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
        # the step kernel writes its outputs straight into the planes of ONE device buffer -- obs [E, D] f32 | reward [E] f32 |
        # terminated [E] u8 | truncated [E] u8 (step(out_*=)) -- which comes down as one message (round 4: four strided device
        # copies into a packed [E, D + 3] buffer cost 30 us of host time per step)
        n_f32 = E * D + E
        self._d_out = torch.empty((n_f32 * 4 + 2 * E,), dtype=torch.uint8, device=self.device)
        self._h_out = torch.empty((n_f32 * 4 + 2 * E,), dtype=torch.uint8, pin_memory=True)
        f32 = self._d_out[:n_f32 * 4].view(torch.float32)
        self._d_obs, self._d_reward = f32[:E * D].view(E, D), f32[E * D:]
        self._d_term, self._d_trunc = self._d_out[n_f32 * 4:n_f32 * 4 + E], self._d_out[n_f32 * 4 + E:]
        h = self._h_out.numpy()
        hf = h[:n_f32 * 4].view(np.float32)
        self._np_obs, self._np_reward = hf[:E * D].reshape(E, D), hf[E * D:]
        self._np_term, self._np_trunc = h[n_f32 * 4:n_f32 * 4 + E], h[n_f32 * 4 + E:]
        self._h_act = torch.empty((E, 2), dtype=torch.float32, pin_memory=True)
        self._d_act = torch.empty((E, 2), dtype=torch.float32, device=self.device)
        self._np_act = self._h_act.numpy()
        # zero-copy: the kernel's own loads / stores go to the pinned buffers (one launcher, bound once)
        self.zero_copy = bool(zero_copy) and self._device_sees(self._h_out) and self._device_sees(self._h_act)
        self._launch, self._closed = None, False
        if self.zero_copy:
            hb = self._h_out
            hf32 = hb[:n_f32 * 4].view(torch.float32)
            self._p_obs, self._p_reward = hf32[:E * D].view(E, D), hf32[E * D:]
            self._p_term, self._p_trunc = hb[n_f32 * 4:n_f32 * 4 + E], hb[n_f32 * 4 + E:]
            norm = None
            if isinstance(env, NormalizedVectorEnv):
                norm = (env.norm_state, env.gamma, env.obs_clip, env.reward_clip, env.epsilon)
            self._launch = base.step_launcher(self._h_act, out_obs=self._p_obs, out_reward=self._p_reward, out_terminated=self._p_term,
                                              out_truncated=self._p_trunc, _norm=norm, _allow_host=True)
        self._pools, self._pool_next = [[None] * self._POOL for _ in range(4)], [0, 0, 0, 0]
        self._reward = np.zeros((E,), dtype=np.float64)            # SyncVectorEnv's buffer dtypes
        self._term = np.zeros((E,), dtype=np.bool_)
        self._trunc = np.zeros((E,), dtype=np.bool_)
        self._obs = np.zeros((E, D), dtype=np.float32)

    @classmethod
    def make(cls, env_config, wrap_config=None, num_envs: int = 1, gamma: float = 0.99, normalize: bool = True, copy: bool = True,
             zero_copy: bool = True, **kw):
        """The env + the trainer's wrapper chain (``wrapping(env, gamma)``, rpo_agent.py:24-33) for ``num_envs`` envs."""
        if normalize:
            return cls(NormalizedVectorEnv.make(env_config, wrap_config, num_envs=num_envs, gamma=gamma, **kw), copy=copy, zero_copy=zero_copy)
        cfg = dataclasses.replace(env_config, clip_action=True)
        return cls(BatchedEvacuationEnv(cfg, wrap_config, num_envs=num_envs, autoreset=True, **kw), copy=copy, zero_copy=zero_copy)

    # ``copy=True`` hands out arrays the caller may keep.  A FRESH 147 KB observation array per step (4096 envs) is an mmap, three
    # dozen page faults and a munmap whenever the C library's heap is not in the mood to recycle the block -- per process, by luck:
    # whole runs at 34 us per step and whole runs at 70-100 (profiles/r05_i_host_vector_env_zero_copy.txt).  So the arrays come from
    # a small pool, and an array goes out again only when NOBODY holds it any more (its reference count is back to the pool's own:
    # views, ``torch.from_numpy`` tensors and slices all hold the base array); otherwise a new one takes its slot.
    _POOL = 4
    _FREE_REFCOUNT = None      # what sys.getrefcount says of a pooled array nobody else holds: measured once, not assumed

    @classmethod
    def _free_refcount(cls) -> int:
        """The reference count `_fresh` sees for an array that only the pool holds -- the pool's slot, the local name and
        getrefcount's own argument on CPython 3.10, but an interpreter that borrows stack references counts differently: measured with
        the very access pattern `_fresh` uses.  -1: no usable reference counts (not CPython): every array is then made fresh."""
        if cls._FREE_REFCOUNT is None:
            if not hasattr(sys, "getrefcount"):
                cls._FREE_REFCOUNT = -1
            else:
                pool = [np.empty(1, dtype=np.float32)]
                a = pool[0]
                free = sys.getrefcount(a)
                held = a                                   # one more holder must be visible, or the count tells nothing
                cls._FREE_REFCOUNT = free if sys.getrefcount(a) == free + 1 else -1
                del held
        return cls._FREE_REFCOUNT

    def _fresh(self, kind: int, shape, dtype):
        pool = self._pools[kind]
        i = self._pool_next[kind] = (self._pool_next[kind] + 1) % self._POOL
        a = pool[i]
        free = self._free_refcount()
        if a is None or free < 0 or sys.getrefcount(a) > free:      # (somebody still holds it -- or the counts cannot be trusted: a new one)
            a = pool[i] = np.empty(shape, dtype=dtype)
        return a

    @staticmethod
    def _device_sees(t: torch.Tensor) -> bool:
        """Is this pinned host tensor mapped into the device's address space at its own address?"""
        import ctypes as C
        try:
            hip = C.CDLL("libamdhip64.so")
            hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
            dp = C.c_void_p()
            rc = hip.hipHostGetDevicePointer(C.byref(dp), C.c_void_p(t.data_ptr()), 0)
            return rc == 0 and dp.value == t.data_ptr()
        except Exception:  # noqa: BLE001
            return False

    # ------------------------------------------------------------------------------------------
    def _download(self, stepped: bool):
        self._h_out.copy_(self._d_out, non_blocking=True)
        self._sync()
        np.copyto(self._obs, self._np_obs)
        if stepped:
            np.copyto(self._reward, self._np_reward)                  # float32 -> float64, exact
            np.not_equal(self._np_term, 0, out=self._term)
            np.not_equal(self._np_trunc, 0, out=self._trunc)

    def _sync(self):
        torch.cuda.current_stream(self.device).synchronize()

    def reset(self, seed: Optional[int] = None, options=None):
        """``SyncVectorEnv.reset(seed=)`` -> ``(obs [E, D] float32, {})``."""
        obs, infos = self.env.reset(seed=seed, options=options)
        if self._launch is not None:
            self._p_obs.copy_(obs, non_blocking=True)
            self._sync()
            return (self._np_obs.copy() if self.copy else self._np_obs), infos
        self._d_obs.copy_(obs)
        self._download(False)
        return (self._obs.copy() if self.copy else self._obs), infos

    def step(self, actions):
        """``SyncVectorEnv.step(actions [E, 2])`` -> ``(obs, reward, terminations, truncations, infos)`` as NumPy arrays."""
        a = np.asarray(actions, dtype=np.float32)
        if a.shape != (self.num_envs, 2):
            raise ValueError(f"actions: expected shape {(self.num_envs, 2)}, got {a.shape}")
        if self._closed:
            raise RuntimeError("HostVectorEnv.step() after close()")
        np.copyto(self._np_act, a)
        if self._launch is not None:                                  # zero-copy: the kernel reads the pinned actions and writes the pinned outputs
            self._launch()
            self._sync()
            b = self._base
            infos = StepInfos(b, self._p_term, self._p_trunc, final_observation=b.final_obs, episode_stats=b.final_stats) if b.autoreset else {}
            if self.copy:                                             # ONE copy of each output: pinned buffer -> the array handed out
                E, D = self.num_envs, self.obs_dim
                obs, rew = self._fresh(0, (E, D), np.float32), self._fresh(1, (E,), np.float64)
                term, trunc = self._fresh(2, (E,), np.bool_), self._fresh(3, (E,), np.bool_)
                np.copyto(obs, self._np_obs)
                np.copyto(rew, self._np_reward)                       # float32 -> float64, exact
                np.not_equal(self._np_term, 0, out=term)
                np.not_equal(self._np_trunc, 0, out=trunc)
                if b.autoreset:
                    infos._done = term | trunc
                return obs, rew, term, trunc, infos
            np.copyto(self._reward, self._np_reward)                  # float32 -> float64, exact
            np.not_equal(self._np_term, 0, out=self._term)
            np.not_equal(self._np_trunc, 0, out=self._trunc)
            if b.autoreset:
                infos._done = self._term | self._trunc
            return self._np_obs, self._reward, self._term, self._trunc, infos      # (obs: the pinned plane itself, rewritten by the next step)
        self._d_act.copy_(self._h_act, non_blocking=True)
        _, _, _, _, infos = self.env.step(self._d_act, out_obs=self._d_obs, out_reward=self._d_reward, out_terminated=self._d_term,
                                          out_truncated=self._d_trunc)
        self._download(True)
        if hasattr(infos, "_done"):                                   # the lazy final_info list: the flags are on the host already
            infos._done = np.logical_or(self._term, self._trunc)
        if self.copy:
            E, D = self.num_envs, self.obs_dim
            obs, rew = self._fresh(0, (E, D), np.float32), self._fresh(1, (E,), np.float64)
            term, trunc = self._fresh(2, (E,), np.bool_), self._fresh(3, (E,), np.bool_)
            np.copyto(obs, self._obs); np.copyto(rew, self._reward); np.copyto(term, self._term); np.copyto(trunc, self._trunc)
            return obs, rew, term, trunc, infos
        return self._obs, self._reward, self._term, self._trunc, infos

    def close(self):
        self._closed, self._launch = True, None       # (the launcher holds the env's handle: it goes before the env does)
        self.env.close()
