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
        terminalreporter.write_sep("-", "near-ties (steps of a fixture with a comparison within 1e-6 of its threshold; of them env-level outputs checked "
                                        "EITHER WAY; the rest skipped them; pedestrians left out of the element-wise check)")
        for label, n, ties, resolved, peds in log:
            terminalreporter.write_line(f"  {label:34s} steps {n:4d}  with a near-tie {ties:3d}  checked either way {resolved:3d}  skipped {ties - resolved:3d}  "
                                        f"pedestrians excluded {peds:3d}")

    mod = sys.modules.get("test_gpu_production_faces") or sys.modules.get("tests.test_gpu_production_faces")
    log = getattr(mod, "FACE_LOG", None)
    if log:
        terminalreporter.write_sep("-", "production faces against the oracle: what was compared (the rest: pedestrians a near-tie may have reached)")
        for name, peds, all_peds, scalars, all_scalars in log:
            terminalreporter.write_line(f"  {name[:96]:96s} pedestrian-steps {peds:7d} of {all_peds:7d} ({100.0 * peds / max(all_peds, 1):5.1f} %)  "
                                        f"rewards / flags {scalars:5d} of {all_scalars:5d} env-steps ({100.0 * scalars / max(all_scalars, 1):5.1f} %)")
