import os
import sys

import pytest

# The CPU oracle (and torch) use OpenMP.  On a many-core GPU host shared with other jobs a 256-thread team per small
# parallel loop turns every barrier into a scheduling lottery (observed: the same GPU suite taking 6x longer on a busy
# box), so the checker runs on a small team with sleeping waits.  Set before anything loads an OpenMP runtime.
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, os.cpu_count() or 1))))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure only)"""
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle
