"""The trainer's per-env gym wrapper chain on the device.

Mirrors ``wrapping(env, gamma)`` of /root/reference/src/agents/rpo_agent.py:24-33 applied inside every
thunk of ``gym.vector.SyncVectorEnv`` (rpo_agent.py:35-39,123-126): FlattenObservation (our
observations are already flat, in gymnasium's key order), RecordEpisodeStatistics (episode records of
the step kernel), ClipAction (``EnvConfig.clip_action``, fused into the step kernel),
NormalizeObservation + clip(-1,1), NormalizeReward(gamma) + clip(-100,100) -- one set of running
statistics per env, updated inside the step kernel itself (``evac_step_normalized``; ``evac_norm_step`` is the same
chain as a separate launch), so observations never leave the GPU."""
from __future__ import annotations

import dataclasses

import torch

from . import _lib
from .vector_env import BatchedEvacuationEnv, _ptr


class NormalizedVectorEnv:
    """``BatchedEvacuationEnv`` + the reference trainer's wrapper chain.  Same ``reset`` / ``step`` surface."""

    def __init__(self, env: BatchedEvacuationEnv, gamma: float = 0.99, obs_clip: float = 1.0,
                 reward_clip: float = 100.0, epsilon: float = 1e-8):
        if not env.autoreset:
            raise ValueError("the trainer's vector env autoresets (SyncVectorEnv); construct the env with autoreset=True")
        self.env = env
        self.lib = env.lib
        self.gamma, self.obs_clip, self.reward_clip, self.epsilon = float(gamma), float(obs_clip), float(reward_clip), float(epsilon)
        self.num_envs, self.obs_dim = env.num_envs, env.obs_dim
        self.single_action_space, self.single_observation_space = env.single_action_space, env.single_observation_space
        w = int(self.lib.evac_norm_state_doubles(env._h))
        self.norm_state = torch.empty((env.num_envs, w), dtype=torch.float64, device=env.device)
        _lib.check(self.lib.evac_norm_init(env._h, _ptr(self.norm_state), env._stream()), env._h)

    @classmethod
    def make(cls, env_config, wrap_config=None, num_envs: int = 1, gamma: float = 0.99, **kw):
        """``SyncVectorEnv([make_env(env_config, wrap_config, gamma)] * num_envs)`` (rpo_agent.py:123-126)."""
        cfg = dataclasses.replace(env_config, clip_action=True)
        return cls(BatchedEvacuationEnv(cfg, wrap_config, num_envs=num_envs, autoreset=True, **kw), gamma=gamma)

    def reset(self, seed=None, options=None, *, mask=None, draws=None):
        """Wrapped reset: NormalizeObservation also normalises (and counts) the reset observation."""
        mask_t = self.env._as_device(mask, (self.num_envs,), torch.uint8, "mask")
        obs, info = self.env.reset(seed=seed, options=options, mask=mask_t, draws=draws)
        _lib.check(self.lib.evac_norm_reset(self.env._h, _ptr(mask_t), _ptr(obs), _ptr(self.norm_state), self.obs_clip,
                                            self.epsilon, self.env._stream()), self.env._h)
        return obs, info

    def step(self, actions, noise=None, *, fused: bool = True, **out):
        """Wrapped step: ONE kernel launch (``evac_step_normalized``: the step and the normalisation chain fused).
        ``out_obs= / out_reward= / out_terminated= / out_truncated=`` (see ``BatchedEvacuationEnv.step``) make the
        kernel write -- normalised -- straight into the caller's rollout storage.  ``fused=False`` runs the step and the
        chain as two launches (``evac_step`` + ``evac_norm_step``): same results bit for bit, kept as a cross-check."""
        if fused:
            return self.env.step(actions, noise=noise, _norm=(self.norm_state, self.gamma, self.obs_clip, self.reward_clip,
                                                              self.epsilon), **out)
        obs, reward, term, trunc, infos = self.env.step(actions, noise=noise, **out)
        _lib.check(self.lib.evac_norm_step(self.env._h, _ptr(obs), _ptr(infos["final_observation"]), _ptr(reward),
                                           _ptr(term), _ptr(trunc), _ptr(self.norm_state), self.gamma, self.obs_clip,
                                           self.reward_clip, self.epsilon, self.env._stream()), self.env._h)
        return obs, reward, term, trunc, infos

    def final_info_list(self, infos):
        return self.env.final_info_list(infos)

    def close(self):
        self.env.close()
