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
# every eq declaration the tests hand to ceno_hip_sumcheck_begin_eq is spot-checked against its table (csrc/sumcheck.hip)
os.environ.setdefault("CENO_HIP_EQ_VERIFY", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _prefetch_torch_libs():
    """On a box whose image is cold the first `import torch` of a GPU run has taken 6-10 minutes here (5.3 GB of shared objects paged in by
    random 4 KB faults; the test bodies themselves take ~15 s).  The first GPU tests do not need torch: meanwhile a background thread reads
    torch's core libraries front to back, so that the import finds them in the page cache.  Harmless when they are cached already."""
    import importlib.util
    import threading

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    lib = os.path.join(list(spec.submodule_search_locations)[0], "lib")

    def work():
        try:
            # only what `import torch` really walks through (~1 GB): the multi-GB solver libraries it merely links against (libmagma, MIOpen,
            # rocsolver, ...) are mapped, hardly read — reading all 5.3 GB on a slow image would compete with the tests for the same storage
            for f in ("libtorch_cpu.so", "libtorch_hip.so", "libtorch_python.so", "libc10.so", "libc10_hip.so", "libamdhip64.so", "librocblas.so"):
                path = os.path.join(lib, f)
                if not os.path.exists(path):
                    continue
                fd = os.open(path, os.O_RDONLY)
                try:
                    if hasattr(os, "posix_fadvise"):
                        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_SEQUENTIAL)
                    while os.read(fd, 16 << 20):
                        pass
                finally:
                    os.close(fd)
        except OSError:
            pass

    threading.Thread(target=work, name="torch-lib-prefetch", daemon=True).start()


def pytest_configure(config):
    markexpr = getattr(config.option, "markexpr", "") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr and os.path.exists("/dev/kfd"):
        _prefetch_torch_libs()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: the long tail of a sweep whose default subset already covers every code path "
                                       "(CENO_RUN_SLOW=1 runs it; keeps the default -m gpu run far below the driver's step limit)")


def _needs_torch(item) -> bool:
    """does the test body import the real torch (directly, or through the torch.distributed helpers)?  Device buffers alone come from tests/hipbuf.py"""
    import inspect
    import re

    try:
        src = inspect.getsource(item.function)
    except (OSError, TypeError):
        return True
    return bool(re.search(r"^\s*import torch\b", src, flags=re.M)) or "dist_worker" in src or "cdist" in src or "torch.distributed" in src


def pytest_collection_modifyitems(config, items):
    # torch-free tests first (stable): on a cold image they run while the prefetch thread above warms torch's libraries
    items.sort(key=lambda it: 1 if ("gpu" in it.keywords and _needs_torch(it)) else 0)
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


# ---- where a GPU run spends its wall time (a cold box has taken 400-1000 s for a suite that runs in 30 s on a warm one): the five slowest
# phases and the time before the first test go to the terminal summary and, when the directory exists, to gpurun_out/pytest_gpu_timing.txt ----
import time as _time

_T0 = _time.time()
_PHASES = []
_FIRST_TEST = [None]
_GUARD_SKIPPED = []


def pytest_runtest_setup(item):
    """OPT-IN guard (CENO_GPU_TEST_GUARD=1).  A GPU run on a box whose image is cold has taken up to 1000 s here (30 s warm).  With the guard
    on, the few torch.distributed tests at the end (whose first `import torch` is the expensive part on such a box) are skipped — and named in
    the summary — when the torch-free tests already took minutes.  By DEFAULT nothing is skipped: the multi-rank tests run or fail, so a green
    run always includes them."""
    if os.environ.get("CENO_GPU_TEST_GUARD") != "1" or "gpu" not in item.keywords or not _needs_torch(item):
        return
    elapsed = _time.time() - _T0
    if "torch" in sys.modules:
        if elapsed > 900:
            _GUARD_SKIPPED.append(item.nodeid)
            pytest.skip(f"GPU run at {elapsed:.0f} s: remaining torch.distributed tests skipped to stay inside the step limit")
    elif elapsed > 240:
        _GUARD_SKIPPED.append(item.nodeid)
        pytest.skip(f"torch-free GPU tests took {elapsed:.0f} s (cold image): torch.distributed tests skipped to stay inside the step limit")


def pytest_runtest_logreport(report):
    if _FIRST_TEST[0] is None and report.when == "setup":
        _FIRST_TEST[0] = _time.time() - _T0
    if report.duration >= 1.0:
        _PHASES.append((report.duration, f"{report.nodeid} [{report.when}]"))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    markexpr = getattr(config.option, "markexpr", "") or ""
    if "gpu" not in markexpr or "not gpu" in markexpr:
        return
    lines = [f"wall {_time.time() - _T0:.1f} s, first test after {(_FIRST_TEST[0] or 0):.1f} s; phases of 1 s and more:"]
    lines += [f"  {d:8.1f} s  {name}" for d, name in sorted(_PHASES, reverse=True)[:8]]
    if _GUARD_SKIPPED:
        # a skip is silent in a -q run: the multi-rank tests that did NOT run on this (cold) box are named in the summary and in the artifact
        lines.append(f"NOT RUN on this box (cold-image guard CENO_GPU_TEST_GUARD=1 was on): {len(_GUARD_SKIPPED)} torch.distributed tests")
        lines += [f"  not run: {n}" for n in _GUARD_SKIPPED]
    for ln in lines:
        terminalreporter.write_line(ln)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        try:
            with open(os.path.join(out_dir, "pytest_gpu_timing.txt"), "a") as f:
                f.write("\n".join(lines) + "\n")
        except OSError:
            pass
