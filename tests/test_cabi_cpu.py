"""CPU checks of the C-ABI library and the host logic (no compute calls: there is no GPU here).

* libevac.so builds for gfx950, loads, and exports every symbol include/evac.h declares;
* config validation mirrors the reference's errors (wrappers/config.py:76-82, config.py:97-100);
* observation dims / splitting match the reference's shapes on the golden fixtures;
* the product fails loudly without a device and never imports the oracle."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import evacuation_amd as ea
from evacuation_amd import _lib, build
from evacuation_amd.config import obs_dim, to_c_config
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "evac.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evac_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    syms = declared_symbols()
    assert len(syms) >= 21
    for s in syms:
        assert s in _lib.SIGNATURES, f"{s} declared in evac.h but not bound in _lib.py"
        getattr(lib, s)
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.evac_version() == _lib.VERSION == 140


def test_code_object_targets_gfx950(lib):
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in data
    assert b"gfx942" not in data and b"sm_" not in data      # one target, no multi-arch fat binary


def test_config_struct_layout_and_validation(lib):
    assert C.sizeof(_lib.EvacConfig) == 19 * 4
    c = to_c_config(ea.EnvConfig(number_of_pedestrians=60), ea.EnvWrappersConfig(positions="grav"))
    assert lib.evac_config_validate(C.byref(c)) == 0 and lib.evac_config_obs_dim(C.byref(c)) == 6
    c.number_of_pedestrians = 0
    assert lib.evac_config_validate(C.byref(c)) == _lib.ERR_INVALID_ARGUMENT
    c.number_of_pedestrians = 1025
    assert lib.evac_config_validate(C.byref(c)) == _lib.ERR_INVALID_ARGUMENT
    c.number_of_pedestrians = 60
    c.type = 1                                  # grav + Box
    assert lib.evac_config_validate(C.byref(c)) == _lib.ERR_UNSUPPORTED
    assert b"NotImplementedError" in lib.evac_last_error(None)
    c.type = 7
    assert lib.evac_config_validate(C.byref(c)) == _lib.ERR_INVALID_ARGUMENT


def test_reference_error_behaviour_of_the_configs():
    with pytest.raises(NotImplementedError):
        ea.EnvWrappersConfig(positions="grav", type="Box").check()       # wrappers/config.py:79-80
    with pytest.raises(ValueError):
        ea.EnvWrappersConfig(positions="grav", type="Tuple").check()     # wrappers/config.py:81-82
    with pytest.raises(AssertionError):
        ea.EnvWrappersConfig(num_obs_stacks=2)                            # wrappers/config.py:44
    with pytest.raises(AssertionError):
        ea.EnvConfig(n_episodes=1)                                        # config.py:99
    d = ea.EnvConfig()
    assert (d.number_of_pedestrians, d.step_size, d.noise_coef, d.enslaving_degree, d.max_timesteps) == (10, 0.01, 0.2, 1.0, 2000)
    assert (d.is_new_exiting_reward, d.is_new_followers_reward, d.intrinsic_reward_coef, d.init_reward_each_step) == (False, True, 0.0, -1.0)
    w = ea.EnvWrappersConfig()
    assert (w.positions, w.statuses, w.type, w.alpha) == ("abs", "no", "Dict", 3)
    assert [s.value for s in ea.Status] == [1, 2, 3, 4] and ea.Status.ESCAPED.value == 4


@pytest.mark.parametrize("pos,stat,typ", [("grav", "no", "Dict")] + H.OBS_VARIANTS)
def test_obs_dims_and_split_match_reference_shapes(lib, pos, stat, typ):
    from evacuation_amd.vector_env import observation_space_for, split_observation
    d = np.load(os.path.join(H.GOLDEN, "traj_n60_s0.npz"))
    cfg, wrap = ea.EnvConfig(number_of_pedestrians=60), ea.EnvWrappersConfig(positions=pos, statuses=stat, type=typ)
    dim = obs_dim(cfg, wrap)
    c = to_c_config(cfg, wrap)
    assert lib.evac_config_obs_dim(C.byref(c)) == dim
    parts = split_observation(np.arange(2 * dim, dtype=np.float32).reshape(2, dim), cfg, wrap)
    space = observation_space_for(cfg, wrap)
    if pos == "grav":
        assert dim == 6 and set(parts) == {"agent_position", "grad_potential_exit", "grad_potential_pedestrians"}
        assert list(space.keys()) == sorted(space.keys())
    elif typ == "Box":
        ref = d[f"obs_{pos}_{stat}_box"][0]
        assert parts.shape == (2,) + ref.shape and space.shape == ref.shape
        assert parts[0, 1, 0] == ref.shape[1]            # row-major [(N+2), C]
    else:
        for k, v in parts.items():
            ref = d[f"obs_{pos}_{stat}_dict__{k}"][0] if not (pos == "abs" and stat == "no") else None
            if ref is not None:
                assert v.shape == (2,) + ref.shape, k
            assert space[k].shape == v.shape[1:]
        assert sum(int(np.prod(v.shape[1:])) for v in parts.values()) == dim


def test_create_fails_loudly_without_a_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    c = to_c_config(ea.EnvConfig(), ea.EnvWrappersConfig())
    h = C.c_void_p()
    assert lib.evac_create(C.byref(c), 4, 0, 0, 0, C.byref(h)) == _lib.ERR_NO_DEVICE and not h.value
    with pytest.raises(RuntimeError, match="no CPU"):
        ea.BatchedEvacuationEnv(ea.EnvConfig(), num_envs=2, device="cpu")
    with pytest.raises(RuntimeError):
        ea.setup_env(ea.EnvConfig(), ea.EnvWrappersConfig())


def test_product_never_imports_the_oracle():
    """No file of the product package, bench's GPU leg excepted by construction, references oracle/."""
    pkg = os.path.join(ROOT, "evacuation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "evac_oracle" not in text or f == "evac_device.h" and "oracle/philox.py" in text, f
    code = "import sys, evacuation_amd, evacuation_amd.vector_env, evacuation_amd.env; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_kernel_resource_budgets():
    """Compile the device code to gfx950 assembly and check the register budgets the design relies on:
    every step / rollout kernel fits 4 waves per SIMD (<= 128 VGPRs; C3 lost a quarter of its occupancy when one
    grew to 135), and the headline kernels (1 wave per env, sub-wave) neither spill VGPRs nor use scratch."""
    import tempfile
    src = os.path.join(ROOT, "evacuation_amd", "csrc", "evac_api.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "evac.s")
        flags = [f for f in build.FLAGS if f not in ("-fPIC", "-shared")]       # the product's own flags
        subprocess.run([build.hipcc_path()] + flags + ["-S", "--cuda-device-only", src, "-o", out], check=True, capture_output=True)
        text = open(out).read()
    meta = text[text.index("amdhsa.kernels:"):]
    kernels = {}
    for block in meta.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        kernels[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1))
                         for k in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "private_segment_fixed_size")}
    assert len(kernels) >= 40
    names = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.splitlines()
    for mangled, name in zip(kernels, names):
        k = kernels[mangled]
        if any(t in name for t in ("k_step", "k_rollout", "k_reset", "k_observe")):
            assert k["vgpr_count"] <= 128, (name, k)
        # the default faces of the production families: one wave per env (4- and 16-wave workgroups), sub-wave, cell list, teams
        headline = ("k_step" in name or "k_rollout" in name) and "diag" not in name and \
                   ("Wave<1," in name or "_sub<" in name or "Cells<" in name or "Team<" in name)
        if headline:
            assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, (name, k)
        if "Wave<4, 1024>" in name:        # the CU-wide form of the four-wave envs: a few spilled scalars at most
            assert k["private_segment_fixed_size"] <= 16, (name, k)
    assert sum("Cells<" in n and "k_rollout<" in n for n in names) == 8      # 4 sizes x 2 observation faces
    assert sum("Team<" in n and "k_rollout<" in n for n in names) == 8       # 4 team sizes x 2 observation faces
    for fam in ("Wave<1, 1024>", "Wave<4, 1024>"):                           # generic + default-configuration faces, 2 observation faces each
        assert sum(fam in n and "k_rollout<" in n for n in names) == 2 and sum(fam in n and "k_rollout_default_config<" in n for n in names) == 2
    assert sum("k_rollout_default_config" in n for n in names) >= 30 and sum("k_step_default_config" in n for n in names) >= 20
