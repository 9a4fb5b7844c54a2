import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter):
    """The near-tie exclusions of the teacher-forced parity test, per fixture (a silent growth must be visible)."""
    import sys
    mod = sys.modules.get("test_gpu_parity") or sys.modules.get("tests.test_gpu_parity")
    log = getattr(mod, "TIE_LOG", None)
    if log:
        terminalreporter.write_sep("-", "near-tie exclusions (steps of a fixture whose env-level outputs were skipped; pedestrians left out)")
        for label, n, ties, peds in log:
            terminalreporter.write_line(f"  {label:34s} steps {n:4d}  with a near-tie {ties:3d}  pedestrians excluded {peds:3d}")

