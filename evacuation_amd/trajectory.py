"""Trajectory capture in the reference's own frame format.

The reference keeps, for rendering, ``pedestrians.memory = {'positions': [...], 'statuses': [...]}`` (one (N,2)
float array and one (N,) array of ``Status`` members per frame; ``Pedestrians.save``,
/root/reference/src/env/env/pedestrians.py:33-35, first called at reset, env.py:137) and
``agent.memory = {'position': [...]}`` (one (2,) float32 array per step; ``Agent.save``, area.py:32-33), and
``EvacuationEnv.save_animation`` (env.py:241-324) draws frame ``i`` from index ``i`` of both lists.

``evac_rollout``'s capture buffer (``BatchedEvacuationEnv.rollout(capture_envs=K)``) holds the same data as one device
tensor ``[T, K, N+1, 3]``; ``capture_to_memory`` turns one env of it into those two dicts, so that the reference's
unchanged matplotlib code can be fed from the GPU env (``feed_reference_env``).  Rendering itself stays on the host
and out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

from .statuses import Status

_STATUS_BY_CODE = {s.value: s for s in Status}


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def capture_to_memory(rollout: dict, env_index: int = 0, initial: Optional[dict] = None, status_cls=None
                      ) -> Tuple[Dict[str, List[np.ndarray]], Dict[str, List[np.ndarray]]]:
    """``(pedestrians.memory, agent.memory)`` of env ``env_index`` of a ``rollout(capture_envs=K)`` result.

    ``initial``: ``get_state()`` taken before the rollout; its frame is put in front of the pedestrians' lists like
    the ``save()`` the reference does inside ``reset`` (pedestrians.py:27 / env.py:137) -- the leader's list has no such
    frame in the reference either.  ``status_cls``: the enum to build the status arrays from (default: this
    package's ``Status``; pass the reference's own class when feeding its ``save_animation``)."""
    by_code = _STATUS_BY_CODE if status_cls is None else {s.value: s for s in status_cls}
    pos = _np(rollout["positions"])[:, env_index].astype(np.float64)          # reference positions are float64
    st = _np(rollout["statuses"])[:, env_index].astype(np.int64)
    agent = _np(rollout["agent_positions"])[:, env_index].astype(np.float32)  # the leader is float32 (area.py:22-24)
    ped_mem: Dict[str, List[np.ndarray]] = {"positions": [], "statuses": []}
    if initial is not None:
        ped_mem["positions"].append(_np(initial["pos"])[env_index].astype(np.float64))
        ped_mem["statuses"].append(np.array([by_code[int(c)] for c in _np(initial["status"])[env_index]]))
    for t in range(pos.shape[0]):
        ped_mem["positions"].append(pos[t].copy())
        ped_mem["statuses"].append(np.array([by_code[int(c)] for c in st[t]]))
    agent_mem = {"position": [agent[t].copy() for t in range(agent.shape[0])]}
    return ped_mem, agent_mem


def feed_reference_env(ref_env, ped_memory, agent_memory) -> None:
    """Load captured frames into an instance of the REFERENCE's EvacuationEnv so that its own
    ``save_animation()`` (env.py:241-324) renders them: that method reads ``pedestrians.memory``, ``agent.memory``
    and ``time.now`` (the number of frames) only."""
    ref_env.pedestrians.memory = ped_memory
    ref_env.agent.memory = agent_memory
    ref_env.time.now = min(len(agent_memory["position"]), len(ped_memory["positions"]))
