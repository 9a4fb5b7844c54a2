"""ctypes binding of libevac.so (include/evac.h).  There is NO CPU fallback: if the HIP library is
missing or cannot be loaded this module raises, and every op needs a visible MI355X."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# EVAC_LIB overrides the library path (profiling builds made by tools/ablate.sh only)
LIB_PATH = os.environ.get("EVAC_LIB") or os.path.join(HERE, "libevac.so")

EVAC_OK = 0
ERR_INVALID_ARGUMENT, ERR_NOT_BOUND, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_TEAM_ABORTED = -1, -2, -3, -4, -5, -6
POS = {"abs": 0, "rel": 1, "grav": 2}
STAT = {"no": 0, "ohe": 1, "cat": 2}
TYPE = {"Dict": 0, "Box": 1}
MAX_PEDESTRIANS = 1024
VERSION = 150
EPISODE_STATS_WORDS = 10       # evac_episode_stats_t: 8 floats + 2 int32


class EvacConfig(C.Structure):
    """evac_config_t"""
    _fields_ = [
        ("number_of_pedestrians", C.c_int32), ("width", C.c_float), ("height", C.c_float),
        ("step_size", C.c_float), ("noise_coef", C.c_float), ("eps", C.c_float),
        ("enslaving_degree", C.c_float), ("is_new_exiting_reward", C.c_int32),
        ("is_new_followers_reward", C.c_int32), ("intrinsic_reward_coef", C.c_float),
        ("is_termination_agent_wall_collision", C.c_int32), ("init_reward_each_step", C.c_float),
        ("max_timesteps", C.c_int32), ("positions", C.c_int32), ("statuses", C.c_int32),
        ("type", C.c_int32), ("alpha", C.c_float), ("nan_guard", C.c_int32), ("clip_action", C.c_int32),
    ]


class EvacOptions(C.Structure):
    """evac_options_t: which kernels a handle launches (never what they compute); -1 = automatic"""
    _fields_ = [(f, C.c_int32) for f in ("subwave", "cells", "cu_wide", "team", "specialize", "parts", "team_coop", "team_fault", "chain")]


class EvacError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libevac error {code}: {msg}")
        self.code = code


# Every symbol include/evac.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SIGNATURES = {
    "evac_version": (C.c_int, []),
    "evac_status_string": (C.c_char_p, [C.c_int]),
    "evac_last_error": (C.c_char_p, [_P]),
    "evac_config_validate": (C.c_int, [C.POINTER(EvacConfig)]),
    "evac_config_obs_dim": (C.c_int64, [C.POINTER(EvacConfig)]),
    "evac_create": (C.c_int, [C.POINTER(EvacConfig), C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(_P)]),
    "evac_create_ex": (C.c_int, [C.POINTER(EvacConfig), C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(EvacOptions), C.POINTER(_P)]),
    "evac_get_options": (C.c_int, [_P, C.POINTER(EvacOptions)]),
    "evac_join": (C.c_int, [_P, _P]),
    "evac_order_next_rollout": (C.c_int, [_P]),
    "evac_num_parts": (C.c_int32, [_P]),
    "evac_own_streams": (C.c_int32, [_P]),
    "evac_part_stream": (_P, [_P, C.c_int32]),
    "evac_destroy": (C.c_int, [_P]),
    "evac_obs_dim": (C.c_int64, [_P]),
    "evac_num_envs": (C.c_int32, [_P]),
    "evac_kernel_variant": (C.c_char_p, [_P, C.c_int32]),
    "evac_bind_state": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "evac_workspace_bytes": (C.c_int64, [_P]),
    "evac_bind_workspace": (C.c_int, [_P, _P, C.c_int64]),
    "evac_reschedule": (C.c_int, [_P, _P]),
    "evac_schedule_generation": (C.c_int32, [_P]),
    "evac_team_error": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "evac_team_error_nosync": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "evac_team_clear_error": (C.c_int, [_P]),
    "evac_peer_gather": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "evac_reset": (C.c_int, [_P, _P, _P, _P, _P]),
    "evac_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P, _P]),
    "evac_rollout": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, C.c_int32, _P, _P, _P]),
    "evac_get_state": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "evac_set_state": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "evac_observe": (C.c_int, [_P, _P, _P]),
    "evac_algorithmic_bytes_per_env_step": (C.c_int64, [_P]),
    "evac_norm_state_doubles": (C.c_int64, [_P]),
    "evac_step_normalized": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "evac_norm_init": (C.c_int, [_P, _P, _P]),
    "evac_norm_reset": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, _P]),
    "evac_norm_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load libevac.so (built in-tree by evacuation_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m evacuation_amd.build` (needs hipcc). "
            "evacuation_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            if os.environ.get("EVAC_LIB"):   # a profiling build of an older tree (A/B runs): entry points added since are absent
                continue
            raise AttributeError(f"{LIB_PATH} does not export {name}: stale build? (python -m evacuation_amd.build --force)")
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, handle=None):
    if rc != EVAC_OK:
        lib = load()
        msg = lib.evac_last_error(handle)
        text = msg.decode() if msg else ""
        raise EvacError(rc, text or lib.evac_status_string(rc).decode())
