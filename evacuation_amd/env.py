"""Single-env facade with the reference's own surface: ``setup_env(env_config, wrap_config)`` ->
env with ``reset(seed, options) -> (obs, {})`` and ``step(action) -> (obs, reward, terminated,
truncated, {})`` (/root/reference/src/env/__init__.py:18-21, src/env/env/env.py:34-171).

It is a batch of one on the GPU (``BatchedEvacuationEnv(num_envs=1)``), with NumPy in/out, so the
reference's scripted agents (README.md:69-91, random_agent.py, baseline_wacuum_cleaner.py:14-28)
run against it unchanged.  For throughput use ``BatchedEvacuationEnv`` directly.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .config import EnvConfig, EnvWrappersConfig
from .spaces import Box
from .statuses import Status
from .vector_env import BatchedEvacuationEnv, observation_space_for, split_observation

_STATUS_BY_CODE = {s.value: s for s in Status}


class _PedestriansView:
    """``env.pedestrians`` of the reference (pedestrians.py:6-44), read from device state."""

    def __init__(self, env: "EvacuationEnv"):
        self._env = env
        self.num = env._batched.n_ped
        self.memory = {"positions": [], "statuses": []}       # pedestrians.py:10,27

    def save(self) -> None:
        """Pedestrians.save (pedestrians.py:33-35): one frame of (N, 2) float64 positions and (N,) Status members."""
        self.memory["positions"].append(self.positions.astype(np.float64))
        self.memory["statuses"].append(self.statuses.copy())

    @property
    def positions(self) -> np.ndarray:
        return self._env._state()["pos"][0]

    @property
    def directions(self) -> np.ndarray:
        return self._env._state()["dir"][0]

    @property
    def status_codes(self) -> np.ndarray:
        return self._env._state()["status"][0]

    @property
    def statuses(self) -> np.ndarray:
        return np.array([_STATUS_BY_CODE[int(c)] for c in self.status_codes])

    @property
    def status_stats(self):
        c = self.status_codes
        return {"escaped": int((c == 4).sum()), "exiting": int((c == 3).sum()),
                "following": int((c == 2).sum()), "viscek": int((c == 1).sum())}


class _AgentView:
    """``env.agent`` (area.py:12-33)."""

    def __init__(self, env: "EvacuationEnv"):
        self._env = env
        self.enslaving_degree = env.env_config.enslaving_degree
        self.start_position = np.zeros(2, dtype=np.float32)
        self.start_direction = np.zeros(2, dtype=np.float32)
        self.memory = {"position": []}                         # area.py:17,30

    def save(self) -> None:
        """Agent.save (area.py:32-33): one (2,) float32 leader position per step."""
        self.memory["position"].append(self.position.astype(np.float32))

    @property
    def position(self) -> np.ndarray:
        return self._env._state()["agent_pos"][0]

    @property
    def direction(self) -> np.ndarray:
        return self._env._state()["agent_dir"][0]


class _Exit:
    position = np.array([0, -1], dtype=np.float32)        # area.py:36-39


class _AreaView:
    """``env.area`` attributes the reference's agents and wrappers read
    (baseline_wacuum_cleaner.py:14-28, gravity_encoding.py:50)."""

    def __init__(self, cfg: EnvConfig):
        self.width, self.height = cfg.width, cfg.height
        self.step_size, self.noise_coef, self.eps = cfg.step_size, cfg.noise_coef, cfg.eps
        self.exit = _Exit()


class _TimeView:
    """``env.time`` (area.py:42-59)."""

    def __init__(self, env: "EvacuationEnv"):
        self._env = env
        self.max_timesteps = env.env_config.max_timesteps

    @property
    def now(self) -> int:
        return int(self._env._batched.clock[0, 0].item())

    @property
    def n_episodes(self) -> int:
        return int(self._env._batched.clock[0, 1].item())

    @property
    def overall_timesteps(self) -> int:
        return int(self._env._batched.clock[0, 2].item())


class EvacuationEnv:
    """Evacuation env, one instance (env.py:34-171).  No gymnasium dependency; when gymnasium is
    installed the spaces are gymnasium spaces, so its wrappers can be stacked on top."""

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, cfg: EnvConfig, wrap_config: Optional[EnvWrappersConfig] = None, device="cuda:0", seed: int = 0):
        self.env_config = cfg
        self.wrap_config = wrap_config or EnvWrappersConfig()
        self._device, self._seed = device, seed
        self._batched = BatchedEvacuationEnv(cfg, self.wrap_config, num_envs=1, device=device, seed=seed,
                                             autoreset=False)
        self.action_space = Box(-1.0, 1.0, (2,), np.float32)                    # env.py:69
        self.observation_space = observation_space_for(cfg, self.wrap_config)  # env.py:86-96 + wrappers
        self.pedestrians = _PedestriansView(self)
        self.agent = _AgentView(self)
        self.area = _AreaView(cfg)
        self.time = _TimeView(self)
        self.intrinsic_reward_coef = cfg.intrinsic_reward_coef
        self.render_mode = cfg.render_mode
        self.experiment_name = cfg.experiment_name
        self._act = torch.zeros((1, 2), dtype=torch.float32, device=self._batched.device)
        self._cache = None
        # the feed of the reference's animation code (env.py:81-83): frames are recorded while `draw` is set -- from the
        # config, or switched on by reset() for every giff_freq-th episode -- exactly as the reference records them; drawing
        # them (save_animation) stays out of scope
        self.draw = bool(cfg.draw)
        self.giff_freq = int(cfg.giff_freq)
        self.save_next_episode_anim = False

    def with_wrappers(self, wrap_config: EnvWrappersConfig) -> "EvacuationEnv":
        """EnvWrappersConfig.wrap_env(env): the wrappers are the kernel's observation epilogue."""
        if (wrap_config.positions, wrap_config.statuses, wrap_config.type, wrap_config.alpha) == \
                (self.wrap_config.positions, self.wrap_config.statuses, self.wrap_config.type, self.wrap_config.alpha):
            return self
        self.close()
        return EvacuationEnv(self.env_config, wrap_config, self._device, self._seed)

    @property
    def unwrapped(self) -> "EvacuationEnv":
        return self

    # episode accumulators (env.py:65-67)
    @property
    def episode_reward(self) -> float:
        return float(self._batched.acc[0, 0].item())

    @property
    def episode_intrinsic_reward(self) -> float:
        return float(self._batched.acc[0, 1].item())

    @property
    def episode_status_reward(self) -> float:
        return float(self._batched.acc[0, 2].item())

    def _state(self):
        if self._cache is None:
            self._cache = {k: v.cpu().numpy() for k, v in self._batched.get_state().items()}
        return self._cache

    def _obs_to_numpy(self, flat: torch.Tensor):
        o = split_observation(flat[0].cpu().numpy(), self.env_config, self.wrap_config)
        return o

    def reset(self, seed=None, options=None):
        """env.py:106-139.  As in the reference, ``seed`` does not reseed the dynamics."""
        self._cache = None
        if self.save_next_episode_anim or (self.time.n_episodes + 1) % self.giff_freq == 0:      # env.py:110-112
            self.draw = True
            self.save_next_episode_anim = True
        draws = None if options is None else options.get("draws")
        if draws is not None:
            draws = np.asarray(draws, dtype=np.float32)[None]
        obs, _ = self._batched.reset(seed=seed, draws=draws)
        self._cache = None
        self.pedestrians.memory = {"positions": [], "statuses": []}                             # pedestrians.py:27
        self.agent.memory = {"position": []}                                                    # area.py:30
        self.pedestrians.save()                                                                 # env.py:137: always, drawing or not
        return self._obs_to_numpy(obs), {}

    def step(self, action, noise=None):
        """env.py:141-171.  ``action`` must be a float array-like (the reference divides it in
        place, area.py:189-190, and raises for integer input)."""
        a = np.asarray(action)
        if not np.issubdtype(a.dtype, np.floating):
            raise TypeError("action must be a float array (reference area.py:190 raises UFuncTypeError for ints)")
        if a.shape != (2,):
            raise ValueError(f"action must have shape (2,), got {a.shape}")
        self._cache = None
        self._act.copy_(torch.from_numpy(a.astype(np.float32)).reshape(1, 2))
        if noise is not None:
            noise = np.asarray(noise, dtype=np.float32)[None]
        obs, reward, term, trunc, _ = self._batched.step(self._act, noise=noise)
        terminated, truncated = bool(term[0].item()), bool(trunc[0].item())
        if self.draw:                                                                           # env.py:153-155
            self.pedestrians.save()
            self.agent.save()
            if terminated or truncated:
                # env.py:164-165 calls save_animation() here; drawing is out of scope, its bookkeeping is not (env.py:322-324):
                # the frames stay in pedestrians.memory / agent.memory until the next reset, for the caller to render
                if self.save_next_episode_anim:
                    self.save_next_episode_anim = False
                    self.draw = False
        return (self._obs_to_numpy(obs), float(reward[0].item()), terminated, truncated, {})

    def render(self):
        raise NotImplementedError("rendering is out of scope of the MI355X hot path (see DESIGN.md)")

    def save_animation(self):
        """The reference draws ``pedestrians.memory`` / ``agent.memory`` with matplotlib here (env.py:241-324).  Out of scope:
        the frames are recorded (``draw``), hand them to the reference's own method with
        ``evacuation_amd.trajectory.feed_reference_env(ref_env, env.pedestrians.memory, env.agent.memory)``."""
        raise NotImplementedError("rendering is out of scope of the MI355X hot path (see DESIGN.md)")

    def seed(self, seed=None):   # env.py:326-328 (a no-op there too, apart from gym bookkeeping)
        return [seed]

    def close(self):
        self._batched.close()


def setup_env(env_config: EnvConfig, wrap_config: Optional[EnvWrappersConfig] = None, device="cuda:0", seed: int = 0):
    """src/env/__init__.py:18-21."""
    wrap_config = wrap_config or EnvWrappersConfig()
    wrap_config.check()
    return EvacuationEnv(env_config, wrap_config, device=device, seed=seed)
