"""world_size 2 and 4 gloo runs of the sharded sumcheck driver on CPU: the sharded proof must be
identical, on every rank, to the single-prover proof of the unsharded tables."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from tests.ranks import run_ranks

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
        run_ranks(world, [tmp, str(n_local)], deadline_s=300)
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
        run_ranks(world, [tmp, str(n_total), "batched"], deadline_s=300)
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
        run_ranks(world, [tmp, "20000", "shm"], deadline_s=300)
        for r in range(world):
            assert open(os.path.join(tmp, f"rank{r}.txt")).read() == "0"
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("ceno_dist_")]


@pytest.mark.parametrize("world,log2_n", [(2, 7), (4, 8)])
def test_row_sharded_tower_proof_matches_unsharded(world, log2_n):
    """the GKR half of a chip over ROW-SHARDED columns (a5-a10) a second time, independent of ceno_amd/host/dist_gkr.cpp: gloo
    ranks holding block-cyclic row shards run record inference and tower building locally (the oracle's primitives), gather the tower tops,
    prove the large layers with exchanged partial sums and an interleaved gather — and must end, on every rank, with the tower proof the
    oracle's prover produces from the whole columns (CpuTowerProver::create_proof, ceno_zkvm/src/scheme/cpu/mod.rs:346-554)"""
    from tests.dist_worker import chip_case

    with tempfile.TemporaryDirectory() as tmp:
        run_ranks(world, [tmp, str(log2_n), "chip_gloo"], extra_env={"CENO_TEST_ROW_BLOCK_LOG": "2", "OMP_NUM_THREADS": "2"}, deadline_s=600)
        res = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    cols, coeffs, terms, out_terms, (alpha, beta), shape = chip_case(log2_n, w=6, shape=(2, 3, 0, 4))
    rows = 1 << log2_n
    recs = [po.wit_infer(cols, coeffs[ts[0]: ts[-1] + 1], [terms[t] for t in ts], log2_n) for ts in out_terms]
    prod_specs, out_evals = [], []
    for group in (recs[:2], recs[2:5]):
        limbs = po.interleaving_mles_to_mles(group, rows, 2, (1, 0))
        layers = po.infer_tower_product_witness(int(limbs[0].shape[0]).bit_length(), limbs)
        prod_specs.append(layers)
        out_evals += [layers[0][0][0], layers[0][1][0]]
    layers = po.infer_tower_logup_witness(None, po.interleaving_mles_to_mles(recs[5:], rows, 2, alpha))
    out_evals += [layers[0][k][0] for k in range(4)]
    tr = po.StubTranscript(21)
    for e in out_evals:
        tr.append_ext((int(e[0]), int(e[1])))
    oproof = po.tower_prove(prod_specs, [layers], tr)
    for r in range(world):
        assert np.array_equal(res[r]["msgs"], oproof.msgs)
        assert np.array_equal(res[r]["point"], oproof.point[: res[r]["point"].shape[0]])
        assert np.array_equal(res[r]["prod"], oproof.prod_evals) and np.array_equal(res[r]["logup"], oproof.logup_evals)


@pytest.mark.parametrize("world,n,q,log2", [(2, 8, 5, 5), (4, 9, 6, 5), (2, 8, 6, 6)])
def test_row_sharded_rotation_argument_matches_unsharded(world, n, q, log2):
    """the rotation argument of a keccak-style chip over ROW-SHARDED columns a second time, independent of ceno_amd/host/prover.cpp
    prover_prove_rotation_sharded: gloo ranks with block-cyclic row shards rotate locally, run q local rounds with exchanged partial sums, gather
    the folded tables and finish replicated — and must end, on every rank, with the messages, points and evaluations the oracle's prove_rotation
    produces from the whole columns (gkr_iop/src/gkr/layer/cpu/mod.rs:249-389)"""
    from tests.dist_worker import rotation_case

    with tempfile.TemporaryDirectory() as tmp:
        run_ranks(world, [tmp, str(n), "rotation_gloo"],
                  extra_env={"CENO_TEST_ROW_BLOCK_LOG": str(q), "CENO_TEST_ROT_LOG": str(log2), "OMP_NUM_THREADS": "2"}, deadline_s=600)
        res = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    cols, pairs, subgroup, rt = rotation_case(n, log2)
    tr = po.StubTranscript(8)
    msgs, evals, origin, left, right = po.prove_rotation(cols, pairs, subgroup, log2, np.ascontiguousarray(rt), tr)
    for r in range(world):
        assert np.array_equal(res[r]["msgs"], msgs), f"rank {r}: messages"
        assert np.array_equal(res[r]["origin"], origin) and np.array_equal(res[r]["left"], left) and np.array_equal(res[r]["right"], right), f"rank {r}: points"
        assert np.array_equal(res[r]["evals"], evals), f"rank {r}: evaluations"


@pytest.mark.parametrize("world,q", [(2, 3), (4, 3)])
def test_row_sharded_main_constraint_sumcheck_matches_unsharded(world, q):
    """the batched main-constraint sumcheck over ROW-SHARDED tables a second time, independent of ceno_amd/host/main_constraints.cpp
    prover_main_constraints_sharded: gloo ranks with block-cyclic row shards of two chips of different sizes (Prefix selectors that start and end
    anywhere) and the whole tables of a third that is too small to shard (its part of a message added once) run q local rounds with exchanged
    partial sums, gather every sharded table and finish replicated — and must end, on every rank, with the
    messages, point and evaluations of the oracle's sumcheck prover on the whole tables (the sumcheck of prove_batched_main_constraints,
    ceno_zkvm/src/scheme/cpu/mod.rs:1255-1337)"""
    from tests.dist_worker import main_case

    with tempfile.TemporaryDirectory() as tmp:
        run_ranks(world, [tmp, "0", "main_gloo"], extra_env={"CENO_TEST_ROW_BLOCK_LOG": str(q), "OMP_NUM_THREADS": "2"}, deadline_s=600)
        res = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    chips = main_case(world)
    tabs, terms, coeffs = [], [], []
    for ch in chips:
        start = len(tabs)
        tabs += ch["cols"] + [po.selector_compute(po.SEL_PREFIX, ch["point"], ch["off"], ch["n"])]
        terms += [[start + j for j in t] for t in ch["terms"]]
        coeffs += ch["coeffs"]
    max_nv = max(ch["nv"] for ch in chips)
    omsgs, ochal, ofin = po.sumcheck_prove(tabs, po.ext(coeffs), terms, max_nv, 4, po.StubTranscript(5))
    for r in range(world):
        assert np.array_equal(res[r]["msgs"], omsgs), f"rank {r}: messages"
        assert np.array_equal(res[r]["rt"], ochal) and np.array_equal(res[r]["evals"], ofin), f"rank {r}: point / evaluations"
