"""CPU sanitizer job (round-4 verdict item 7), behind CENO_RUN_SANITIZERS=1: `python -m ceno_amd.build --sanitize` builds
  * the oracle (C) and the C++ host layer under AddressSanitizer + UndefinedBehaviorSanitizer,
  * a driver of the shared-memory exchange (ranks as threads) and of the pool's spin lock under ThreadSanitizer and under ASan,
and this module runs the CPU test-suite subset that exercises those libraries against the sanitized builds (LD_PRELOAD of the ASan runtime,
CENO_PROVER_LIB / CENO_ORACLE_LIB) plus the drivers.  CPU only: nothing here touches the GPU build, and no sanitizer ever runs on the GPU box.
A clean run is committed as profiles/r05_sanitizers.log."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(os.environ.get("CENO_RUN_SANITIZERS") != "1", reason="sanitizer job: set CENO_RUN_SANITIZERS=1 (builds take ~3 minutes)")

SUITES = ["tests/test_host_cpu.py", "tests/test_oracle_field.py", "tests/test_oracle_golden.py", "tests/test_oracle_basefold.py",
          "tests/test_oracle_witgen.py", "tests/test_ref_goldens.py"]
BAD = ("ERROR: AddressSanitizer", "runtime error:", "WARNING: ThreadSanitizer", "ERROR: LeakSanitizer")


@pytest.fixture(scope="module")
def san():
    sys.path.insert(0, ROOT)
    from ceno_amd import build

    return build.build_sanitized(verbose=False)


def test_cpu_suites_under_asan_and_ubsan(san):
    env = dict(os.environ, LD_PRELOAD=san["asan_runtime"], ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               CENO_PROVER_LIB=san["prover_asan"], CENO_ORACLE_LIB=san["oracle_asan"], OMP_NUM_THREADS="4", CENO_RUN_SANITIZERS="0")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + SUITES, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=3000)
    print(out.stdout[-3000:])
    assert out.returncode == 0, out.stdout[-3000:]
    assert not any(b in out.stdout for b in BAD), out.stdout[-3000:]


@pytest.mark.parametrize("which", ["exchange_tsan", "exchange_asan"])
def test_exchange_and_pool_lock_drivers(san, which):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1")
    out = subprocess.run([san[which], "4", "20000", "8"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
    print(out.stdout[-2000:])
    assert out.returncode == 0, out.stdout[-2000:]
    assert "shm exchange: 4 ranks x 20000 gathers: ok" in out.stdout and "PoolMutex: 8 threads x 200000 sections: ok" in out.stdout
    assert not any(b in out.stdout for b in BAD), out.stdout[-2000:]
