"""Build libevac.so (HIP, gfx950) in-tree with hipcc.  `python -m evacuation_amd.build [--force]`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libevac.so")
SOURCES = [os.path.join(CSRC, "evac_api.hip")]
DEPENDS = SOURCES + [os.path.join(CSRC, f) for f in ("evac_common.h", "evac_families.h", "evac_device.h", "evac_subwave.h", "evac_team.h", "evac_gather.h")] + [
    os.path.join(os.path.dirname(HERE), "include", "evac.h")]
ARCH = "gfx950"
# -ffp-contract=off: fused multiply-adds are written explicitly in the kernels, so the f32 arithmetic
# is the same IEEE operation sequence as the NumPy f32 oracle wherever both use the same formula.
# -fno-slp-vectorize: the packed-f32 arithmetic of the kernels is written explicitly (v_pk_* through vector types and inline
# asm); what the SLP vectoriser adds on top are pairs built with v_mov shuffles around scalar code (-1 % at C2, A/B in
# profiles/r04_a_c2_ab_tile_layout_salu_diet_driver_args.txt: lib_v1 / lib_v1n).
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-shared"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in DEPENDS)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [hipcc_path()] + FLAGS + SOURCES + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
