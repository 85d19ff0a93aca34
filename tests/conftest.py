import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _n_devices():
    try:
        from rtk_visual_inertial_navigation_amd import solver
        return solver.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests/` on a box without a GPU skips the gpu-marked tests instead of failing them.  When the
    marker is asked for explicitly (-m gpu, the GPU box) nothing is skipped: a missing device must fail loudly there."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _n_devices() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device in this process (gpu-marked tests run on the GPU box with -m gpu)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
