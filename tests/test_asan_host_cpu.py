"""SURVEY.md 5 (sanitizers): the HOST side of libevac built with AddressSanitizer + UBSan and driven through every C-ABI
entry point that needs no GPU (tools/asan_host.sh, tools/asan/host_driver.c).  GPU ASan / xnack+ is not available on the
pool; the device code is covered by the parity tests."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_side_is_clean_under_asan_and_ubsan(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "asan host driver: ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
