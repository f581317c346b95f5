import os
import sys

import pytest

# The CPU oracle (and torch) use OpenMP.  On a many-core GPU host shared with other jobs a 256-thread team per small
# parallel loop turns every barrier into a scheduling lottery (observed: the same GPU suite taking 6x longer on a busy
# box), so the checker runs on a small team with sleeping waits.  Set before anything loads an OpenMP runtime.
def _cpu_budget() -> int:
    """CPUs this container may actually use: the cgroup CPU quota when there is one (a 256-core box whose container holds a
    16-CPU quota is throttled for the rest of every 100 ms period once 16 CPUs' worth of time is burnt — and a throttled
    host thread stalls every GPU round that waits for its challenge), else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return n


# half of the budget for the checker's OpenMP team, the rest stays free for the host threads of the product under test
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(8, _cpu_budget() // 2))))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the long tail of a sweep whose default subset already covers every code path "
                                       "(CENO_RUN_SLOW=1 runs it; keeps the default -m gpu run far below the driver's step limit)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("CENO_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow sweep: set CENO_RUN_SLOW=1 to run it")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    """the CPU oracle (test infrastructure only)"""
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle
