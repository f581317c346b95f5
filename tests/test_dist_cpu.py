"""world_size 2 and 4 gloo runs of the sharded sumcheck driver on CPU: the sharded proof must be
identical, on every rank, to the single-prover proof of the unsharded tables."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sum_partials_is_modular():
    from ceno_amd import dist as cdist

    P = po.P
    parts = np.array([[[P - 1, 5]], [[P - 1, P - 3]], [[7, 1]]], dtype=np.uint64)
    out = cdist.sum_partials(parts)
    assert int(out[0, 0]) == (2 * (P - 1) + 7) % P and int(out[0, 1]) == (5 + P - 3 + 1) % P


@pytest.mark.parametrize("world,n_local", [(2, 5), (4, 3)])
def test_sharded_sumcheck_matches_unsharded(world, n_local):
    from ceno_amd import build

    build.build_all()
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + world * 7 + n_local), WORLD_SIZE=str(world))
        procs = []
        for rank in range(world):
            e = dict(env, RANK=str(rank))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), tmp, str(n_local)], env=e))
        for p in procs:
            assert p.wait(timeout=300) == 0
        res = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    k = 3
    n_total = n_local + world.bit_length() - 1
    full = [po.fill_splitmix(2 << n_total, 0xCE10 + j, 0).reshape(-1, 2) for j in range(k)]
    omsgs, ochal, ofin = po.sumcheck_prove(full, po.ext([1]), [list(range(k))], n_total, k, po.StubTranscript(0xF5))
    for r in range(world):
        assert np.array_equal(res[r]["msgs"], omsgs)
        assert np.array_equal(res[r]["chal"], ochal)
        assert np.array_equal(res[r]["fin"], ofin)


@pytest.mark.parametrize("world,n_total", [(2, 6), (4, 5)])
def test_sharded_batched_mixed_size_sumcheck_matches_unsharded(world, n_total):
    """front-loaded classes of different sizes, sharded along their own top bits or replicated"""
    from ceno_amd import build
    from tests.dist_worker import batched_case

    build.build_all()
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29650 + world * 7 + n_total), WORLD_SIZE=str(world))
        procs = []
        for rank in range(world):
            e = dict(env, RANK=str(rank))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), tmp, str(n_total), "batched"], env=e))
        for p in procs:
            assert p.wait(timeout=300) == 0
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


@pytest.mark.parametrize("world", [2, 4])
def test_shared_memory_exchange_between_processes(world):
    """the per-round exchange of the C++ sharded driver: 20000 back-to-back gathers of varying size between `world`
    processes, every payload word checked on every rank (two-slot sequence protocol, ceno_amd/host/dist.cpp)"""
    from ceno_amd import build

    build.build_all()
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29800 + world), WORLD_SIZE=str(world))
        procs = []
        for rank in range(world):
            e = dict(env, RANK=str(rank))
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), tmp, "20000", "shm"], env=e))
        for p in procs:
            assert p.wait(timeout=300) == 0
        for r in range(world):
            assert open(os.path.join(tmp, f"rank{r}.txt")).read() == "0"
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("ceno_dist_")]
