"""BatchedEvacuationEnv: E independent evacuation envs stepped by ONE fused HIP kernel per step.

Replaces ``gym.vector.SyncVectorEnv([make_env]*num_envs)`` of the reference trainer
(/root/reference/src/agents/rpo_agent.py:123-128,168,193-203) for the env + observation-wrapper
part of the chain: same ``reset(seed=) -> (obs, infos)`` / ``step(actions) -> (obs, reward,
terminations, truncations, infos)`` surface, same-step autoreset (gymnasium 0.29 semantics the
trainer assumes), but observations, rewards and flags are device tensors that never leave HBM.

PyTorch is used only as the device-buffer container (``tensor.data_ptr()``) and for the current
HIP stream; all compute is in libevac.so (csrc/), reached through ctypes (include/evac.h).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict as TDict, Optional

import numpy as np
import torch

from . import _lib
from .config import EnvConfig, EnvWrappersConfig, obs_dim, to_c_config
from .options import KernelOptions, current_default
from .spaces import Box, Dict

# evac_episode_stats_t (include/evac.h): the nine keys of the reference's per-episode logging dict (env.py:115-125)
# + Time.n_episodes.  Words 0-7 are floats, words 8-9 int32 (read them through ``stats_int_view``).
STATS_FIELDS = ("episode_reward", "episode_length", "episode_intrinsic_reward", "episode_status_reward",
                "escaped_pedestrians", "exiting_pedestrians", "following_pedestrians", "viscek_pedestrians")
STATS_INT_FIELDS = ("overall_timesteps", "n_episodes")
STATS_WORDS = _lib.EPISODE_STATS_WORDS


def stats_int_view(stats: torch.Tensor) -> torch.Tensor:
    """The integer words (overall_timesteps, n_episodes) of an episode-stats tensor [..., 10] as int32 [..., 2]."""
    return stats.view(torch.int32)[..., len(STATS_FIELDS):]


class StepInfos(dict):
    """``infos`` of a vector step.  Holds the device tensors (``final_observation``, ``episode_stats``; rows are
    meaningful where terminated | truncated) and, like gymnasium 0.29's SyncVectorEnv, answers ``"final_info" in
    infos`` / ``infos["final_info"]`` / ``infos["_final_info"]`` -- what the reference trainer iterates
    (rpo_agent.py:198-203) -- WITHOUT an extra call: the list is built on first access (one small device-to-host
    read of the flags; nothing is read if the trainer never asks)."""

    _LAZY = ("final_info", "_final_info")

    def __init__(self, env, terminated, truncated, **kw):
        super().__init__(**kw)
        self._env, self._term, self._trunc = env, terminated, truncated
        self._done = None
        self._step_id = env._steps_taken

    def _check_fresh(self):
        # The flags, the terminal observations and the episode records live in buffers the env REUSES every step: an
        # infos object first asked for "final_info" after the env has stepped again would silently describe the wrong
        # step (a trainer that stores infos and inspects them later; asynchronous loggers).
        if self._env._steps_taken != self._step_id:
            raise RuntimeError("infos['final_info'] of an earlier step was first read after the env had stepped again: its buffers "
                               "are reused every step (like the reference's live-reference observations, env.py:98-104) -- read it "
                               "before the next step(), or clone infos['final_observation'] / infos['episode_stats'] at step time")

    def _done_mask(self):
        if self._done is None:             # (HostVectorEnv supplies the mask: it has the flags on the host already)
            self._check_fresh()
            self._done = ((self._term != 0) | (self._trunc != 0)).cpu().numpy()
        return self._done

    def __contains__(self, key):
        if key in self._LAZY and not dict.__contains__(self, key):
            return bool(self._done_mask().any())
        return dict.__contains__(self, key)

    def __missing__(self, key):
        if key not in self._LAZY or not self._done_mask().any():
            raise KeyError(key)
        done = self._done_mask()
        self._check_fresh()                # (the episode records are read from the env's buffers now)
        self["final_info"] = np.array(self._env.final_info_list(self, done=done), dtype=object)
        self["_final_info"] = done.copy()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def observation_space_for(env_config: EnvConfig, wrap: EnvWrappersConfig):
    """Spaces as declared by the reference (env.py:86-96, gravity_encoding.py:52-57,
    wrappers.py:32-45,61-75)."""
    n = env_config.number_of_pedestrians
    f32 = np.float32
    if wrap.positions == "grav":
        return Dict({"agent_position": Box(-1, 1, (2,), f32),
                     "grad_potential_pedestrians": Box(-1, 1, (2,), f32),
                     "grad_potential_exit": Box(-1, 1, (2,), f32)})
    if wrap.type == "Box":
        c = {"no": 2, "ohe": 6, "cat": 3}[wrap.statuses]
        return Box(-1, 1, (n + 2, c), f32)
    d = {"agent_position": Box(-1, 1, (2,), f32),
         "pedestrians_positions": Box(-1, 1, (n, 2), f32),
         "exit_position": Box(-1, 1, (2,), f32)}
    if wrap.statuses == "ohe":
        d["pedestrians_statuses"] = Box(0, 1, (n, 4), f32)
    elif wrap.statuses == "cat":
        d["pedestrians_statuses"] = Box(0, 1, (n,), f32)
    return Dict(d)


def split_observation(flat, env_config: EnvConfig, wrap: EnvWrappersConfig):
    """Views of a flat ``[..., D]`` observation in the reference's structure (Dict keys / Box shape).
    Works on torch tensors and numpy arrays alike (pure slicing + reshape)."""
    n = env_config.number_of_pedestrians
    lead = tuple(flat.shape[:-1])
    if wrap.positions == "grav":
        return {"agent_position": flat[..., 0:2], "grad_potential_exit": flat[..., 2:4],
                "grad_potential_pedestrians": flat[..., 4:6]}
    if wrap.type == "Box":
        c = {"no": 2, "ohe": 6, "cat": 3}[wrap.statuses]
        return flat.reshape(lead + (n + 2, c))
    out = {"agent_position": flat[..., 0:2], "exit_position": flat[..., 2:4],
           "pedestrians_positions": flat[..., 4:4 + 2 * n].reshape(lead + (n, 2))}
    if wrap.statuses == "ohe":
        out["pedestrians_statuses"] = flat[..., 4 + 2 * n:4 + 6 * n].reshape(lead + (n, 4))
    elif wrap.statuses == "cat":
        out["pedestrians_statuses"] = flat[..., 4 + 2 * n:4 + 3 * n]
    return out


class BatchedEvacuationEnv:
    """``num_envs`` evacuation envs on one GPU.

    Parameters mirror ``setup_env(env_config, wrap_config)`` (src/env/__init__.py:18-21) plus the
    batch size, the device, the Philox ``seed`` and ``env_id_offset`` (global id of env 0, so that
    a sharded run draws the same random numbers as a single-GPU run).  ``options`` (``KernelOptions`` = ``evac_options_t``)
    selects which kernels the handle launches -- results do not depend on it; ``options.parts = 2`` (or -1: where it pays) makes
    ``rollout`` two concurrent half-batch kernels on streams the handle owns: see ``join``."""

    def __init__(self, env_config: EnvConfig, wrap_config: Optional[EnvWrappersConfig] = None, num_envs: int = 1,
                 device="cuda:0", seed: int = 0, env_id_offset: int = 0, autoreset: bool = True,
                 options: Optional[KernelOptions] = None):
        self.lib = _lib.load()                       # raises if the HIP library is missing
        self.env_config = env_config
        self.wrap_config = wrap_config or EnvWrappersConfig()
        self.num_envs = int(num_envs)
        self.n_ped = int(env_config.number_of_pedestrians)
        self.autoreset = bool(autoreset)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedEvacuationEnv needs an MI355X device ('cuda:N' under PyTorch-ROCm); "
                               "there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to PyTorch; evacuation_amd has no CPU fallback")
        self.seed_value = int(seed)
        self.env_id_offset = int(env_id_offset)
        self._cfg = to_c_config(env_config, self.wrap_config)
        self._h = C.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._dev_index = dev_index
        self.options = options if options is not None else current_default()
        c_opt = self.options.to_c()
        _lib.check(self.lib.evac_create_ex(C.byref(self._cfg), self.num_envs, dev_index, C.c_uint64(self.seed_value),
                                           C.c_uint64(self.env_id_offset), C.byref(c_opt), C.byref(self._h)))
        self.num_parts = int(self.lib.evac_num_parts(self._h))
        self.own_streams = int(self.lib.evac_own_streams(self._h))    # 2: parts = 2 or chained launches (kernels in flight per rollout round)
        self.obs_dim = int(self.lib.evac_obs_dim(self._h))
        assert self.obs_dim == obs_dim(env_config, self.wrap_config)
        E, N, dev = self.num_envs, self.n_ped, self.device
        # caller-owned state (include/evac.h 'Device layouts')
        self.ped = torch.zeros((E, N, 4), dtype=torch.float32, device=dev)
        self.status = torch.zeros((E, N), dtype=torch.uint8, device=dev)
        self.agent = torch.zeros((E, 4), dtype=torch.float32, device=dev)
        self.clock = torch.zeros((E, 4), dtype=torch.int32, device=dev)
        self.acc = torch.zeros((E, 4), dtype=torch.float32, device=dev)
        _lib.check(self.lib.evac_bind_state(self._h, _ptr(self.ped), _ptr(self.status), _ptr(self.agent),
                                            _ptr(self.clock), _ptr(self.acc)), self._h)
        # workspace of evac_rollout (include/evac.h): the load schedule of large batches of one-wave envs and the exchange
        # areas of the team kernels; performance devices, results do not depend on them (options.workspace = False leaves it
        # unbound, for A/B runs)
        self.workspace = None
        self.schedule = None
        if self.options.workspace:
            nbytes = int(self.lib.evac_workspace_bytes(self._h))
            self.workspace = torch.zeros((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=dev)
            assert self.workspace.data_ptr() % 256 == 0
            _lib.check(self.lib.evac_bind_workspace(self._h, _ptr(self.workspace), C.c_int64(nbytes)), self._h)
            if nbytes >= 16 * E and self.own_streams == 0:   # (two parts: each schedules itself inside a slice of its own; chained: four deep)
                self.schedule = self.workspace[:16 * E].view(torch.int32).view(4, E)    # moving[2][E] | perm[2][E] (include/evac.h)
        # step outputs (reused every step; callers that keep them must clone, like the reference's
        # live-reference observations, env.py:98-104)
        self.obs = torch.zeros((E, self.obs_dim), dtype=torch.float32, device=dev)
        self.reward = torch.zeros((E,), dtype=torch.float32, device=dev)
        self.terminated = torch.zeros((E,), dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros((E,), dtype=torch.uint8, device=dev)
        self.final_obs = torch.zeros((E, self.obs_dim), dtype=torch.float32, device=dev)
        self.final_stats = torch.zeros((E, STATS_WORDS), dtype=torch.float32, device=dev)
        self.stats_words = STATS_WORDS
        self.single_action_space = Box(-1.0, 1.0, (2,), np.float32)            # env.py:69
        self.single_observation_space = observation_space_for(env_config, self.wrap_config)
        self.action_space = Box(-1.0, 1.0, (E, 2), np.float32)
        self.observation_space = Box(-np.inf, np.inf, (E, self.obs_dim), np.float32)
        self.algorithmic_bytes_per_env_step = int(self.lib.evac_algorithmic_bytes_per_env_step(self._h))
        self._was_reset = False
        self._steps_taken = 0          # step() calls so far (StepInfos: a lazily built final_info must be read before the next one)
        self._step_cache = OrderedDict()   # step(): bound ctypes calls by buffer addresses, oldest first (see step)
        self._step_misses = 0          # consecutive step() calls that missed the cache (see _remember_step)
        self._stream_args = {}         # raw stream handle -> its ctypes argument

    # ------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_tensor(self, t: torch.Tensor, shape, dtype, name, allow_host: bool = False):
        # (allow_host: PINNED host memory is device-accessible at the same address on ROCm -- HostVectorEnv lets the step kernel read
        # its actions from and write its outputs to such buffers directly instead of staging them through device copies)
        if allow_host and t.device.type == "cpu" and t.is_pinned() and tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.is_contiguous():
            return t
        if tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != self.device or not t.is_contiguous():
            raise ValueError(f"{name}: expected contiguous {dtype} tensor of shape {tuple(shape)} on {self.device}, "
                             f"got {t.dtype} {tuple(t.shape)} on {t.device}")
        return t

    def _as_device(self, x, shape, dtype, name):
        if x is None:
            return None
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x))
        x = x.to(device=self.device, dtype=dtype).contiguous()
        return self._check_tensor(x, shape, dtype, name)

    def rebind_workspace(self) -> None:
        """Re-deal the envs to the SIMDs now, from the loads the workspace holds (``evac_reschedule``).  bench.py restores a
        snapshot of state + workspace before its kernel-time replays, so that they run under the deal the timed blocks had."""
        if self.workspace is not None:
            _lib.check(self.lib.evac_reschedule(self._h, self._stream()), self._h)

    def schedule_generation(self) -> int:
        """Rollout launches made under the load schedule so far (-1: none): launch g reads ``schedule[2 + (g & 1)]``, leaves its
        loads in ``schedule[g & 1]`` and deals ``schedule[2 + ((g + 1) & 1)]`` for the next one."""
        return int(self.lib.evac_schedule_generation(self._h))

    def schedule_loads(self) -> Optional[torch.Tensor]:
        """``moving[E]`` as the most recent rollout launch left it (pedestrians still moving per env), or None."""
        g = self.schedule_generation()
        return None if self.schedule is None or g < 1 else self.schedule[(g - 1) & 1]

    def schedule_perm(self) -> Optional[torch.Tensor]:
        """``perm[slot] = env`` that the NEXT rollout launch runs under, or None before the first deal."""
        g = self.schedule_generation()
        return None if self.schedule is None or g < 0 else self.schedule[2 + (g & 1)]

    def resolved_options(self) -> KernelOptions:
        """The options the handle ended up with: automatic choices resolved, diagnostic ``EVAC_*`` overrides applied."""
        o = _lib.EvacOptions()
        _lib.check(self.lib.evac_get_options(self._h, C.byref(o)), self._h)
        return KernelOptions(workspace=self.workspace is not None, **{f: int(getattr(o, f)) for f, _ in _lib.EvacOptions._fields_})

    def join(self, stream=None) -> None:
        """``options.parts = 2`` / ``options.chain = 1``: make ``stream`` (default: the current stream) wait for everything the handle's own two streams
        have been given so far (``evac_join``).  ``rollout_launcher`` launches do NOT do this by themselves -- consecutive launches
        must not meet at a common point, or the two halves would run in lock-step again -- so call it before anything consumes a
        slab.  ``rollout()`` and every other method join by themselves.  A no-op for ``parts = 1``."""
        if self.own_streams:
            st = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
            _lib.check(self.lib.evac_join(self._h, st), self._h)

    def order_next_rollout(self) -> None:
        """The next ``rollout_launcher`` launch follows what the launching stream holds at that moment (``evac_order_next_rollout``)."""
        if self.own_streams:
            self.lib.evac_order_next_rollout(self._h)

    def part_streams(self):
        """The handle's own streams (``options.parts = 2``) as ``torch.cuda.ExternalStream`` objects, for timing events; else ``[]``."""
        return [torch.cuda.ExternalStream(int(self.lib.evac_part_stream(self._h, k)), device=self.device) for k in range(self.own_streams)]

    def team_error(self, sync: bool = True) -> int:
        """Non-zero if a barrier of a team rollout (N > 512, few envs) timed out: the outputs of that launch -- and of the
        launches queued behind it -- are void.  ``sync=True`` synchronises the device first; ``sync=False`` reads the word as it
        stands, for a caller that has just waited for its own stream (the pipelined consumer of ``rollout_launcher`` slabs, a
        hipGraph replay: the enqueueing calls cannot know -- poll this BEFORE trusting a slab)."""
        v = C.c_int32(0)
        fn = self.lib.evac_team_error if sync else self.lib.evac_team_error_nosync
        _lib.check(fn(self._h, C.byref(v)), self._h)
        return int(v.value)

    def team_clear_error(self) -> None:
        """Acknowledge a team error (``EvacError`` with code ERR_TEAM_ABORTED from any call): the handle then runs one
        workgroup per env; reset or restore the batch afterwards (the aborted launch left some envs un-stepped)."""
        _lib.check(self.lib.evac_team_clear_error(self._h), self._h)

    def close(self):
        # (the bound calls of step() hold the handle: they go first, so that a step() after close() takes the checked path and
        # gets EVAC_ERR_INVALID_ARGUMENT for the NULL handle instead of calling into a destroyed one)
        if getattr(self, "_step_cache", None) is not None:
            self._step_cache.clear()
            self._stream_args.clear()
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.evac_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # ------------------------------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options=None, *, mask=None, draws=None):
        """EvacuationEnv.reset for every env (env.py:106-139).  ``seed`` is accepted for API
        compatibility; like the reference (whose dynamics use the global NumPy RNG, not
        ``self.np_random``) it does NOT reseed the dynamics -- the Philox key is fixed at
        construction.  ``draws`` [E,N,4] injects the U(-1,1) reset draws (parity tests)."""
        E, N = self.num_envs, self.n_ped
        mask_t = self._as_device(mask, (E,), torch.uint8, "mask")
        draws_t = self._as_device(draws, (E, N, 4), torch.float32, "draws")
        _lib.check(self.lib.evac_reset(self._h, _ptr(mask_t), _ptr(draws_t), _ptr(self.obs), self._stream()), self._h)
        self._was_reset = True
        return self.obs, {}

    def step(self, actions, noise=None, *, out_obs=None, out_reward=None, out_terminated=None, out_truncated=None, _norm=None):
        """EvacuationEnv.step + wrappers for the batch (env.py:141-171).  ``actions`` [E,2] f32 on
        the device; ``noise`` [E,N] injects the per-pedestrian angular noise (parity tests).

        Destination form (the trainer's rollout storage, rpo_agent.py:158-163,182-196): ``out_obs`` [E,D] f32,
        ``out_reward`` [E] f32, ``out_terminated`` / ``out_truncated`` [E] uint8 are written by the kernel itself
        -- e.g. ``out_obs=obs_buf[t + 1]``, ``out_reward=rewards[t]`` -- so the step needs no copy into the
        buffers afterwards.  Contiguous rows of the caller's tensors; the returned tensors are those."""
        # The plain call of a policy loop -- the same device tensors every step, or the same rows of a rollout storage every
        # update -- goes through a cache of bound ctypes calls keyed by the buffers' addresses: the argument checks and the
        # ctypes objects of a step are built once per distinct set of buffers (what step_launcher makes explicit), and the host
        # side of step() stays below the kernel's own ~6 us.
        key = None
        if noise is None and type(actions) is torch.Tensor:
            key = (actions.data_ptr(), None if out_obs is None else out_obs.data_ptr(), None if out_reward is None else out_reward.data_ptr(),
                   None if out_terminated is None else out_terminated.data_ptr(), None if out_truncated is None else out_truncated.data_ptr(),
                   None if _norm is None else (_norm[0].data_ptr(),) + tuple(_norm[1:]))
            ent = self._step_cache.get(key)
            if ent is not None and ent[0](actions, out_obs, out_reward, out_terminated, out_truncated):
                raw = torch._C._cuda_getCurrentRawStream(self._dev_index)
                st = self._stream_args.get(raw)
                if st is None:
                    st = self._stream_args[raw] = C.c_void_p(raw)
                rc = ent[1](st)
                if rc != 0:
                    _lib.check(rc, self._h)
                self._steps_taken += 1
                self._step_misses = 0
                # (the caller's own objects back, as the uncached path returns them; the env's buffers otherwise)
                obs = self.obs if out_obs is None else out_obs
                rew = self.reward if out_reward is None else out_reward
                term = self.terminated if out_terminated is None else out_terminated
                trunc = self.truncated if out_truncated is None else out_truncated
                infos = StepInfos(self, term, trunc, final_observation=self.final_obs, episode_stats=self.final_stats) if self.autoreset else {}
                return obs, rew, term, trunc, infos
        E, N = self.num_envs, self.n_ped
        if isinstance(actions, torch.Tensor) and actions.device == self.device and actions.dtype == torch.float32 \
                and actions.is_contiguous() and tuple(actions.shape) == (E, 2):
            act = actions
        else:
            act = self._as_device(actions, (E, 2), torch.float32, "actions")
        nz = self._as_device(noise, (E, N), torch.float32, "noise")
        obs = self.obs if out_obs is None else self._check_tensor(out_obs, (E, self.obs_dim), torch.float32, "out_obs")
        rew = self.reward if out_reward is None else self._check_tensor(out_reward, (E,), torch.float32, "out_reward")
        term = self.terminated if out_terminated is None else self._check_tensor(out_terminated, (E,), torch.uint8, "out_terminated")
        trunc = self.truncated if out_truncated is None else self._check_tensor(out_truncated, (E,), torch.uint8, "out_truncated")
        fo = _ptr(self.final_obs) if self.autoreset else None
        fs = _ptr(self.final_stats) if self.autoreset else None
        if _norm is None:
            _lib.check(self.lib.evac_step(self._h, _ptr(act), _ptr(nz), _ptr(obs), _ptr(rew), _ptr(term), _ptr(trunc),
                                          int(self.autoreset), fo, fs, self._stream()), self._h)
        else:       # NormalizedVectorEnv: the trainer's normalisation chain fused into the same launch
            state, gamma, obs_clip, reward_clip, eps = _norm
            _lib.check(self.lib.evac_step_normalized(self._h, _ptr(act), _ptr(nz), _ptr(obs), _ptr(rew), _ptr(term), _ptr(trunc),
                                                     int(self.autoreset), fo, fs, _ptr(state), gamma, obs_clip, reward_clip,
                                                     eps, self._stream()), self._h)
        self._steps_taken += 1
        if key is not None and act is actions:
            self._remember_step(key, act, out_obs, out_reward, out_terminated, out_truncated, (obs, rew, term, trunc), fo, fs, _norm)
        infos = {}
        if self.autoreset:
            # device tensors; rows are meaningful where terminated | truncated (no host sync unless "final_info" is asked for)
            infos = StepInfos(self, term, trunc, final_observation=self.final_obs, episode_stats=self.final_stats)
        return obs, rew, term, trunc, infos

    _STEP_CACHE_ENTRIES = 4096       # >= the reference trainer's num_steps storage rows (rpo_agent.py:60: 2048)

    def _remember_step(self, key, act, out_obs, out_reward, out_terminated, out_truncated, outs, fo, fs, _norm):
        """Bind the ctypes call of a step that just passed every argument check (see ``step``).  An entry holds ADDRESSES, not
        tensors: a hit is keyed by the ``data_ptr`` of the tensors being passed NOW -- alive by definition -- and re-validates
        what an address alone does not pin (shape, dtype, contiguity, device: a different view or a tensor of another device can
        start at the same address).  Nothing the caller owns is kept alive, so a policy loop that makes a fresh ``actions``
        tensor every step pins no memory (a kept tensor per miss held up to 4096 x E x 8 bytes); only the env's own
        buffers and the normalisation state -- which live as long as the env anyway -- are referenced.  Oldest entries are
        evicted one at a time; after 256 misses in a row (fresh tensors at fresh addresses every step: nothing to hit) calls
        without any ``out_*`` storage row are no longer bound, so such a loop stops paying for entries it never uses."""
        if self._step_misses >= 256 and out_obs is None and out_reward is None and out_terminated is None and out_truncated is None:
            return
        self._step_misses += 1
        E, D = self.num_envs, self.obs_dim
        f32, u8, idx = torch.float32, torch.uint8, self._dev_index

        def ok(t, shape, dtype):
            return t.shape == shape and t.dtype is dtype and t.is_cuda and t.get_device() == idx and t.is_contiguous()

        def same(a, o, r, t, u, _s=((E, 2), (E, D), (E,), (E,), (E,))):
            return (ok(a, _s[0], f32) and (o is None or ok(o, _s[1], f32)) and (r is None or ok(r, _s[2], f32))
                    and (t is None or ok(t, _s[3], u8)) and (u is None or ok(u, _s[4], u8)))
        obs, rew, term, trunc = outs
        ar = int(self.autoreset)
        a = (_ptr(act), _ptr(obs), _ptr(rew), _ptr(term), _ptr(trunc))
        if _norm is None:
            fn = self.lib.evac_step

            def call(st):
                return fn(self._h, a[0], None, a[1], a[2], a[3], a[4], ar, fo, fs, st)   # (the handle as it is at call time)
        else:
            fn = self.lib.evac_step_normalized
            state, gamma, obs_clip, reward_clip, eps = _norm
            a_state = _ptr(state)

            def call(st, _keep=state):      # (the wrapper's own statistics buffer)
                return fn(self._h, a[0], None, a[1], a[2], a[3], a[4], ar, fo, fs, a_state, gamma, obs_clip, reward_clip, eps, st)
        cache = self._step_cache
        cache.pop(key, None)
        while len(cache) >= self._STEP_CACHE_ENTRIES:
            cache.popitem(last=False)
        cache[key] = (same, call)

    def step_launcher(self, actions, *, out_obs=None, out_reward=None, out_terminated=None, out_truncated=None, stream=None,
                      _norm=None, _allow_host: bool = False):
        """A zero-argument callable that enqueues ``step(actions, out_*=...)`` with every ctypes argument prepared once -- for a
        trainer that steps through preallocated storage (one launcher per row: ``actions[t]``, ``out_obs=obs[t + 1]``, ...) or
        re-fills one ``actions`` tensor in place before every call.  The call only enqueues the kernel (~3.5 us of host time
        against ~8 for ``step()``, whose argument checks and ``infos`` object it skips); outputs land in the given tensors (or
        ``self.obs`` / ``reward`` / ``terminated`` / ``truncated``), ``final_obs`` / ``final_stats`` are filled as by ``step``.
        On the stream that is current at each call, or always on ``stream`` if one is given."""
        E = self.num_envs
        ah = bool(_allow_host)
        act = self._check_tensor(actions, (E, 2), torch.float32, "actions", ah)
        obs = self.obs if out_obs is None else self._check_tensor(out_obs, (E, self.obs_dim), torch.float32, "out_obs", ah)
        rew = self.reward if out_reward is None else self._check_tensor(out_reward, (E,), torch.float32, "out_reward", ah)
        term = self.terminated if out_terminated is None else self._check_tensor(out_terminated, (E,), torch.uint8, "out_terminated", ah)
        trunc = self.truncated if out_truncated is None else self._check_tensor(out_truncated, (E,), torch.uint8, "out_truncated", ah)
        h, ar = self._h, int(self.autoreset)
        a = (_ptr(act), None, _ptr(obs), _ptr(rew), _ptr(term), _ptr(trunc))
        fo = _ptr(self.final_obs) if self.autoreset else None
        fs = _ptr(self.final_stats) if self.autoreset else None
        keep = (act, obs, rew, term, trunc)                      # the tensors stay alive as long as the launcher does
        cur, dev = torch.cuda.current_stream, self.device
        a_stream = C.c_void_p(stream.cuda_stream) if stream is not None else None
        if _norm is None:
            fn = self.lib.evac_step

            def launch(_keep=keep):
                rc = fn(h, a[0], a[1], a[2], a[3], a[4], a[5], ar, fo, fs, a_stream if a_stream is not None else C.c_void_p(cur(dev).cuda_stream))
                if rc != 0:
                    _lib.check(rc, h)
                self._steps_taken += 1
            return launch
        # the trainer's normalisation chain fused into the same launch (NormalizedVectorEnv; HostVectorEnv binds it once)
        fn = self.lib.evac_step_normalized
        state, gamma, obs_clip, reward_clip, eps = _norm
        a_state = _ptr(state)

        def launch_norm(_keep=(keep, state)):
            rc = fn(h, a[0], a[1], a[2], a[3], a[4], a[5], ar, fo, fs, a_state, gamma, obs_clip, reward_clip, eps,
                    a_stream if a_stream is not None else C.c_void_p(cur(dev).cuda_stream))
            if rc != 0:
                _lib.check(rc, h)
            self._steps_taken += 1
        return launch_norm

    def kernel_variant(self, mode: str = "rollout") -> str:
        """Name of the kernel instantiation behind ``step`` ("step") or ``rollout`` ("rollout")."""
        return self.lib.evac_kernel_variant(self._h, 1 if mode == "rollout" else 0).decode()

    def rollout_launcher(self, n_steps: int, out: TDict, stream=None):
        """A zero-argument callable that enqueues ``rollout(n_steps, out=out)`` (RandomAgent actions) with all ctypes
        arguments prepared once: for loops that launch the same shape many times.  On the stream that is current at
        each call, or always on ``stream`` (a torch stream) if one is given -- which saves the lookup, ~1.5 us per call.
        With ``options.parts = 2`` / ``options.chain = 1`` the kernels go to the handle's own streams, which are put behind what the
        launching stream holds ONCE per ``join()`` (at the first launch after it), and NOTHING waits for them until ``join()``: keep
        ``out`` alive and untouched on your stream until then (``order_next_rollout()`` if you did touch it).
        The call only ENQUEUES: for rooms of more than 512 pedestrians (team kernels) poll ``team_error(sync=False)`` after waiting
        for the launch and before consuming its slab -- a launch that lost a team member still returns success here."""
        T, E, D = int(n_steps), self.num_envs, self.obs_dim
        slab = self._check_tensor(out["slab"], (T, E, D + 3), torch.float32, "slab")
        stats = out.get("episode_stats")
        if stats is not None:
            self._check_tensor(stats, (T, E, STATS_WORDS), torch.float32, "episode_stats")
        fn, h = self.lib.evac_rollout, self._h
        a_slab, a_stats, dev = _ptr(slab), _ptr(stats), self.device
        cur = torch.cuda.current_stream
        if stream is not None:
            a_stream = C.c_void_p(stream.cuda_stream)

            def launch_on_stream():
                rc = fn(h, T, None, None, a_slab, a_stats, 0, None, None, a_stream)
                if rc != 0:
                    _lib.check(rc, h)
            return launch_on_stream

        def launch():
            rc = fn(h, T, None, None, a_slab, a_stats, 0, None, None, C.c_void_p(cur(dev).cuda_stream))
            if rc != 0:
                _lib.check(rc, h)
        return launch

    def rollout(self, n_steps: int, actions=None, record_actions: bool = False, out: Optional[TDict] = None,
                capture_envs: int = 0, noise=None):
        """``n_steps`` env steps in ONE kernel launch with the state held in registers (the
        rollout loop rpo_agent.py:180-203 with given or RandomAgent actions).  The kernel writes one
        packed time-major slab ``[T,E,D+3] = [obs | reward | terminated | truncated]`` (f32); the
        returned dict holds it as ``slab`` plus zero-copy views ``obs [T,E,D]``, ``reward [T,E]``,
        ``terminated`` / ``truncated [T,E]`` (f32 0/1), and ``episode_stats [T,E,8]`` (rows valid where an
        episode ended), ``actions [T,E,2]`` if recorded.  Pass a previous result as ``out`` to reuse it.
        ``capture_envs=K`` additionally records the trajectories of the first K envs for rendering
        (what ``Pedestrians.save`` / ``Agent.save`` keep in the reference): ``trajectory [T,K,N+1,3]`` with
        views ``positions [T,K,N,2]``, ``statuses [T,K,N]`` and ``agent_positions [T,K,2]``.
        ``noise`` [T,E,N] injects the per-pedestrian angular noise (parity tests; diagnostic kernel face)."""
        T, E, D = int(n_steps), self.num_envs, self.obs_dim
        dev = self.device
        if out is None:
            out = {"slab": torch.empty((T, E, D + 3), dtype=torch.float32, device=dev),
                   "episode_stats": torch.zeros((T, E, STATS_WORDS), dtype=torch.float32, device=dev)}
            if record_actions:
                out["actions"] = torch.empty((T, E, 2), dtype=torch.float32, device=dev)
            if capture_envs:
                out["trajectory"] = torch.empty((T, int(capture_envs), self.n_ped + 1, 3), dtype=torch.float32, device=dev)
        slab = self._check_tensor(out["slab"], (T, E, D + 3), torch.float32, "slab")
        if "obs" not in out:
            out["obs"], out["reward"] = slab[..., :D], slab[..., D]
            out["terminated"], out["truncated"] = slab[..., D + 1], slab[..., D + 2]
        act = self._as_device(actions, (T, E, 2), torch.float32, "actions")
        traj = out.get("trajectory")
        k_cap = 0
        if traj is not None:
            k_cap = int(traj.shape[1])
            self._check_tensor(traj, (T, k_cap, self.n_ped + 1, 3), torch.float32, "trajectory")
            out["positions"], out["statuses"] = traj[:, :, :self.n_ped, 0:2], traj[:, :, :self.n_ped, 2]
            out["agent_positions"] = traj[:, :, self.n_ped, 0:2]
        nz = self._as_device(noise, (T, E, self.n_ped), torch.float32, "noise")
        if self.own_streams:
            # (the outputs may be fresh from torch's stream-ordered allocator -- memory that kernels still queued on the current stream
            # may be using -- and `episode_stats` is zero-filled on the current stream: the handle's own streams must follow all that)
            self.lib.evac_order_next_rollout(self._h)
        _lib.check(self.lib.evac_rollout(self._h, T, _ptr(act), _ptr(out.get("actions")), _ptr(slab),
                                         _ptr(out.get("episode_stats")), k_cap, _ptr(traj), _ptr(nz), self._stream()), self._h)
        self.join()                   # (two parts: the returned tensors are ordered behind the launch on the current stream, as ever)
        return out

    def observe(self, out: Optional[torch.Tensor] = None):
        """Observation of the current state without stepping (env.py:98-104 through the wrappers)."""
        out = self.obs if out is None else self._check_tensor(out, (self.num_envs, self.obs_dim), torch.float32, "out")
        _lib.check(self.lib.evac_observe(self._h, _ptr(out), self._stream()), self._h)
        return out

    def split_observation(self, flat):
        return split_observation(flat, self.env_config, self.wrap_config)

    # ------------------------------------------------------------------------------------------
    def get_state(self) -> TDict[str, torch.Tensor]:
        """State in the reference's shapes (pedestrians.py:6-27, area.py:12-59): pos/dir [E,N,2],
        status [E,N], agent_pos/agent_dir [E,2], now [E]."""
        E, N, dev = self.num_envs, self.n_ped, self.device
        st = {"pos": torch.empty((E, N, 2), dtype=torch.float32, device=dev),
              "dir": torch.empty((E, N, 2), dtype=torch.float32, device=dev),
              "status": torch.empty((E, N), dtype=torch.uint8, device=dev),
              "agent_pos": torch.empty((E, 2), dtype=torch.float32, device=dev),
              "agent_dir": torch.empty((E, 2), dtype=torch.float32, device=dev),
              "now": torch.empty((E,), dtype=torch.int32, device=dev)}
        _lib.check(self.lib.evac_get_state(self._h, _ptr(st["pos"]), _ptr(st["dir"]), _ptr(st["status"]),
                                           _ptr(st["agent_pos"]), _ptr(st["agent_dir"]), _ptr(st["now"]),
                                           self._stream()), self._h)
        return st

    def set_state(self, pos=None, dir=None, status=None, agent_pos=None, agent_dir=None, now=None):  # noqa: A002
        E, N = self.num_envs, self.n_ped
        a = [self._as_device(pos, (E, N, 2), torch.float32, "pos"), self._as_device(dir, (E, N, 2), torch.float32, "dir"),
             self._as_device(status, (E, N), torch.uint8, "status"),
             self._as_device(agent_pos, (E, 2), torch.float32, "agent_pos"),
             self._as_device(agent_dir, (E, 2), torch.float32, "agent_dir"),
             self._as_device(now, (E,), torch.int32, "now")]
        _lib.check(self.lib.evac_set_state(self._h, *[_ptr(t) for t in a], self._stream()), self._h)
        self._keepalive = a   # the copy kernel runs asynchronously on the stream

    # ------------------------------------------------------------------------------------------
    def final_info_list(self, infos, terminated=None, truncated=None, done=None):
        """gymnasium-0.29 style ``infos['final_info']`` (list with one dict or None per env), as
        consumed by rpo_agent.py:198-203: per finished env the reference's nine-key episode record (env.py:115-125)
        plus ``episode = {"r", "l"}`` (RecordEpisodeStatistics).  Synchronises with the device; ``infos["final_info"]``
        calls this on first access."""
        if done is None:
            term = (self.terminated if terminated is None else terminated).bool()
            trunc = (self.truncated if truncated is None else truncated).bool()
            done = (term | trunc).cpu().numpy()
        stats_t = dict.__getitem__(infos, "episode_stats")
        stats = stats_t.cpu().numpy()
        ints = stats_int_view(stats_t).cpu().numpy()
        out = [None] * self.num_envs
        for k in np.nonzero(done)[0]:
            rec = {f: float(stats[k, j]) for j, f in enumerate(STATS_FIELDS)}
            for j, f in enumerate(STATS_INT_FIELDS):
                rec[f] = int(ints[k, j])
            rec["episode"] = {"r": rec["episode_reward"], "l": int(rec["episode_length"])}
            out[k] = rec
        return out
