import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture
def debug_knobs():
    """Engine diagnostics knobs (include/mjpc_hip_debug.h) for one test: debug_knobs(name, value) sets, value None clears; everything
    set through the fixture is cleared again afterwards (the knobs are process-wide and read when an engine is created)."""
    from mujoco_mpc_amd import capi
    touched = set()

    def setter(name, value):
        touched.add(name)
        capi.debug_set(name, value)
    yield setter
    for name in touched:
        capi.debug_set(name, None)
