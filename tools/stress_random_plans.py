#!/usr/bin/env python3
"""One-off stress: the seeded random-plan differential tests of tests/test_gpu_parity.py over many more seeds than the
suite runs (usage: python tools/stress_random_plans.py [first_seed] [n_seeds]).  Exits non-zero on the first mismatch."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_gpu_parity as T
from ceno_amd import Device, prover as pv

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = Device(0)
pv.plib()


class MP:  # the two methods of pytest's monkeypatch the tests use
    def setenv(self, k, v):
        os.environ[k] = v

    def delenv(self, k, raising=False):
        os.environ.pop(k, None)


bad = 0
for seed in range(first, first + n):
    for name in ("test_sumcheck_random_plans_differential", "test_sumcheck_random_plans_lds_blocked_kernel", "test_tower_random_specs_differential"):
        f = getattr(T, name)
        try:
            if "monkeypatch" in f.__code__.co_varnames[: f.__code__.co_argcount]:
                f(dev, pv, seed, MP())
            else:
                f(dev, pv, seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", name, seed, repr(e)[:200])
    for k in ("CENO_HIP_GEN_MIN_LOG", "CENO_HIP_GEN_STAGE_KB", "CENO_HIP_GEN_PIPE_MIN_LOG", "CENO_HIP_GEN_ROUND0", "CENO_HIP_NO_GEN"):
        os.environ.pop(k, None)
print("seeds", first, "..", first + n - 1, "failures", bad)
sys.exit(1 if bad else 0)
