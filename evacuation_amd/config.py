"""EnvConfig / EnvWrappersConfig: the public configuration of the reference, field for field.

Mirrors /root/reference/src/env/env/config.py:3-100 and src/env/wrappers/config.py:8-93 (same names,
defaults, meaning and error behaviour) so a user of the reference can pass the same settings.
Logging fields are accepted for compatibility and ignored; ``draw`` / ``giff_freq`` steer the recording of frames by the
single-env facade (evacuation_amd/env.py), whose drawing stays out of scope."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal, Optional

from . import _lib


@dataclass
class EnvConfig:
    experiment_name: str = "test"
    # geometry
    number_of_pedestrians: int = 10
    width: float = 1.0
    height: float = 1.0
    step_size: float = 0.01
    noise_coef: float = 0.2
    eps: float = 1e-8
    # leader
    enslaving_degree: float = 1.0
    # reward
    is_new_exiting_reward: bool = False
    is_new_followers_reward: bool = True
    intrinsic_reward_coef: float = 0.0
    is_termination_agent_wall_collision: bool = False
    init_reward_each_step: float = -1.0
    # timing
    max_timesteps: int = 2_000
    n_episodes: int = 0
    n_timesteps: int = 0
    # logging (accepted, unused here)
    render_mode: Optional[str] = None
    draw: bool = False
    verbose: bool = False
    giff_freq: int = 500
    wandb_enabled: bool = True
    path_giff: str = "saved_data/giff"
    path_png: str = "saved_data/png"
    path_logs: str = "saved_data/logs"
    # extension (not in the reference): see include/evac.h evac_config_t.nan_guard
    nan_guard: bool = False
    # extension: gym.wrappers.ClipAction of the trainer's wrapper chain (rpo_agent.py:27) fused into the step
    clip_action: bool = False

    def __post_init__(self):
        # config.py:97-100
        assert self.n_episodes == 0, NotImplementedError
        assert self.n_timesteps == 0, NotImplementedError


@dataclass
class EnvWrappersConfig:
    num_obs_stacks: int = 1
    positions: Literal["abs", "rel", "grav"] = "abs"
    statuses: Literal["no", "ohe", "cat"] = "no"
    type: Literal["Dict", "Box"] = "Dict"
    alpha: float = 3

    def __post_init__(self):
        # wrappers/config.py:42-44
        assert self.num_obs_stacks == 1, NotImplementedError

    def check(self):
        """The error behaviour of wrap_env (wrappers/config.py:76-93)."""
        if self.positions == "grav":
            if self.type == "Dict":
                return
            if self.type == "Box":
                raise NotImplementedError
            raise ValueError
        if self.positions not in ("abs", "rel"):
            raise ValueError(f"Invalid value of `positions`='{self.positions}'.")
        if self.type not in ("Dict", "Box"):
            raise ValueError(f"Invalid value of `type`='{self.type}'.")
        if self.statuses not in ("no", "ohe", "cat"):
            raise ValueError(f"Invalid value of `type`='{self.statuses}'. Must be 'no', 'ohe' or 'cat'.")

    def wrap_env(self, env):
        """wrappers/config.py:46-93.  The observation wrappers are fused into the step kernel, so
        'wrapping' selects the kernel's observation epilogue."""
        self.check()
        return env.with_wrappers(self)


def to_c_config(env: EnvConfig, wrap: EnvWrappersConfig) -> "_lib.EvacConfig":
    wrap.check()
    c = _lib.EvacConfig()
    c.number_of_pedestrians = int(env.number_of_pedestrians)
    c.width, c.height = float(env.width), float(env.height)
    c.step_size, c.noise_coef, c.eps = float(env.step_size), float(env.noise_coef), float(env.eps)
    c.enslaving_degree = float(env.enslaving_degree)
    c.is_new_exiting_reward = int(bool(env.is_new_exiting_reward))
    c.is_new_followers_reward = int(bool(env.is_new_followers_reward))
    c.intrinsic_reward_coef = float(env.intrinsic_reward_coef)
    c.is_termination_agent_wall_collision = int(bool(env.is_termination_agent_wall_collision))
    c.init_reward_each_step = float(env.init_reward_each_step)
    c.max_timesteps = int(env.max_timesteps)
    c.positions = _lib.POS[wrap.positions]
    c.statuses = _lib.STAT[wrap.statuses]
    c.type = _lib.TYPE[wrap.type]
    c.alpha = float(wrap.alpha)
    c.nan_guard = int(bool(env.nan_guard))
    c.clip_action = int(bool(env.clip_action))
    return c


def obs_dim(env: EnvConfig, wrap: EnvWrappersConfig) -> int:
    """Floats per env of the flattened observation (include/evac.h 'Observation layouts')."""
    wrap.check()
    n = env.number_of_pedestrians
    if wrap.positions == "grav":
        return 6
    sc = {"no": 0, "ohe": 4, "cat": 1}[wrap.statuses]
    if wrap.type == "Box":
        return (n + 2) * (2 + sc)
    return 4 + 2 * n + sc * n
