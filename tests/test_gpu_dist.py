"""Multi-rank runs of the C++ sharded driver on ONE GPU: with the host shared-memory exchange there is no device
collective on the round path, so `world` processes can share device 0 — the real HIP engine, the real cross-process
exchange, the real replicated tail.  The result must equal the single-prover proof of the unsharded tables."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,n_local", [(2, 9), (4, 7), (8, 5)])
def test_cpp_sharded_driver_shared_memory_exchange(world, n_local):
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + world), WORLD_SIZE=str(world))
        procs = []
        for rank in range(world):
            e = dict(env, RANK=str(rank))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), tmp, str(n_local), "shm_gpu"], env=e))
        for p in procs:
            assert p.wait(timeout=600) == 0
        res = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    k = 3
    n_total = n_local + world.bit_length() - 1
    full = [po.fill_splitmix(2 << n_total, 0xCE10 + j, 0).reshape(-1, 2) for j in range(k)]
    omsgs, ochal, ofin = po.sumcheck_prove(full, po.ext([1]), [list(range(k))], n_total, k, po.StubTranscript(0xF5))
    for r in range(world):
        assert np.array_equal(res[r]["msgs"], omsgs)
        assert np.array_equal(res[r]["chal"], ochal)
        assert np.array_equal(res[r]["fin"], ofin)


@pytest.mark.parametrize("world,n_total", [(1, 7), (2, 8), (4, 9)])
def test_cpp_batched_mixed_size_sharded_driver(world, n_total):
    """SURVEY section 8(e) mixed-size batches through the C++ driver (ceno_dist_batched_sumcheck_prove): classes sharded
    along their own top bits or replicated, `world` processes on one GPU, against the single-prover oracle proof"""
    from tests.dist_worker import batched_case

    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29950 + world), WORLD_SIZE=str(world))
        procs = []
        for rank in range(world):
            e = dict(env, RANK=str(rank))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), tmp, str(n_total), "shm_gpu_batched"], env=e))
        for p in procs:
            assert p.wait(timeout=600) == 0
        res = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    tables, coeffs, terms, off = [], [], [], 0
    for c in batched_case(n_total):
        tables += c["tables"]
        coeffs.append(c["coeffs"])
        terms += [[off + j for j in t] for t in c["terms"]]
        off += len(c["tables"])
    omsgs, ochal, ofin = po.sumcheck_prove(tables, np.concatenate(coeffs), terms, n_total, 3, po.StubTranscript(0xF5))
    for r in range(world):
        assert np.array_equal(res[r]["msgs"], omsgs)
        assert np.array_equal(res[r]["chal"], ochal)
        assert np.array_equal(res[r]["fin"], ofin)
