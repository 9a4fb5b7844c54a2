"""KernelOptions: the create-time options of a batched env (``evac_options_t`` of include/evac.h).

They select WHICH kernels a handle launches -- sub-wave / one-wave / cell-list / team families, CU-wide workgroups, the
default-configuration instantiations, one kernel or two concurrent half-batch kernels per rollout -- never what they compute:
every combination gives bit-identical results (tests/test_gpu_variants_sweep.py).  The reference has no analogue (it steps its
envs one after another in Python, /root/reference/src/agents/rpo_agent.py:123-126).  Every field defaults to -1 = automatic.

``kernel_options(...)`` is a context manager that sets the default for the envs created inside the block BY THIS THREAD (A/B
tests, ``tools/``); product code passes ``options=`` explicitly.  Nothing here reads or writes process-wide variables: the
``EVAC_*`` variables the library still reads are diagnostic overrides for unmodified callers only."""
from __future__ import annotations

import contextlib
import dataclasses
import threading
from typing import Optional

from . import _lib

AUTO = -1


@dataclasses.dataclass(frozen=True)
class KernelOptions:
    subwave: int = AUTO      # 0: one wave per env also for N <= 32
    cells: int = AUTO        # 1 / 0: the cell-list kernels for every N > 64 / never (automatic: N > 512)
    cu_wide: int = AUTO      # 1 / 0: CU-wide rollout workgroups always / never (automatic: batches that fill the CUs)
    team: int = AUTO         # 0 / 2 / 4 / 8 / 16 workgroups per env for N > 512 (automatic: what the batch leaves free)
    specialize: int = AUTO   # 0: never the default-configuration instantiations
    parts: int = 1           # 1: one kernel per rollout on the caller's stream; 2: two half-batch kernels on the handle's own streams
                             # (``BatchedEvacuationEnv.join``); -1: 2 where it pays.  (1, not -1: the stream contract of existing callers)
    team_coop: int = AUTO    # 1: cooperative launches of the team grids
    team_fault: int = AUTO   # 1: fault injection (tests)
    chain: int = 0           # 2: ONE PERSISTENT rollout kernel per ``join`` -- every launch a command of its ring, the state stays in registers
                             #    (include/evac.h: the kernel holds its CUs while it has commands and leaves by itself after ~150 us without one);
                             # 1: chained rollout launches on the handle's own two streams (include/evac.h; ``join`` as for parts = 2);
                             # -1: where it pays.  (0, not -1: the stream contract of existing callers)
    workspace: bool = True   # bind the rollout workspace (load schedule, team exchange areas); False: A/B runs without it

    def replace(self, **kw) -> "KernelOptions":
        return dataclasses.replace(self, **{k: (int(v) if k != "workspace" else bool(v)) for k, v in kw.items()})

    def to_c(self) -> "_lib.EvacOptions":
        return _lib.EvacOptions(*(int(getattr(self, f)) for f, _ in _lib.EvacOptions._fields_))


# the diagnostic switch of each option (include/evac.h) -- so that tools written against those names can say
# from_switches(EVAC_CU_WIDE=1, EVAC_WORKSPACE=0) and get plain options
SWITCH_FIELDS = {"EVAC_SUBWAVE": "subwave", "EVAC_CELLS": "cells", "EVAC_CU_WIDE": "cu_wide", "EVAC_TEAM": "team",
                 "EVAC_SPECIALIZE": "specialize", "EVAC_PARTS": "parts", "EVAC_TEAM_COOP": "team_coop", "EVAC_TEAM_FAULT": "team_fault", "EVAC_CHAIN": "chain",
                 "EVAC_WORKSPACE": "workspace"}


def from_switches(**switches) -> KernelOptions:
    return KernelOptions().replace(**{SWITCH_FIELDS[k]: int(v) for k, v in switches.items()})


_local = threading.local()


def current_default() -> KernelOptions:
    return getattr(_local, "options", None) or KernelOptions()


@contextlib.contextmanager
def kernel_options(options: Optional[KernelOptions] = None, **fields):
    """Default ``KernelOptions`` of the envs this thread creates inside the block: ``with kernel_options(cu_wide=1, parts=2): ...``"""
    old = getattr(_local, "options", None)
    _local.options = (options or current_default()).replace(**fields)
    try:
        yield _local.options
    finally:
        _local.options = old
