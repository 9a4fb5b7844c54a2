"""Scripted agents of the reference that drive the hot path in its README loop.

``RandomAgent`` mirrors /root/reference/src/agents/random_agent.py:4-9 (``act`` = action_space.sample())."""
from __future__ import annotations


class BaseAgent:
    def __init__(self, action_space):
        self.action_space = action_space

    def act(self, obs):
        raise NotImplementedError


class RandomAgent(BaseAgent):
    def act(self, obs):
        return self.action_space.sample()
