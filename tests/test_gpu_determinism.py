"""Repeat-run determinism of every kernel family UNDER LOAD (round-5 verdict item 7).

The inline-assembly store hazard of round 5 was a bug of the kind no parity test finds in one run: the words were right
most of the time and wrong under load.  tests/test_isa_lint.py guards the cause; these tests guard the symptom, family by family — dense rounds
(k_dense), the persistent small-round kernels (k_mid / k_tail / k_tile), the fused tower rounds (k_tower), the Poseidon2 leaf hash + tree
(commit_traces), the Basefold fold / commit rounds + query gathers (batch open), on-device witness generation with the per-XCD lookup counters —
each repeated while a second host thread keeps the device busy with large sumchecks on another stream.  Every repetition must produce the words of
the first (and the first is checked against the oracle elsewhere: tests/test_gpu_parity.py, test_gpu_flows.py, test_gpu_commit.py,
test_gpu_basefold.py, test_gpu_shard_wide.py).  The last test repeats the wide shard's whole flow with the chip proofs in cohorts."""
import threading

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
P = po.P


@pytest.fixture(scope="module")
def dev():
    from ceno_amd import Device

    d = Device(0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def prover():
    from ceno_amd import prover as p

    return p


class Load:
    """a host thread that keeps the device oversubscribed: nv = 21 dense sumchecks back to back on its own stream"""

    def __init__(self, dev, prover):
        self.dev, self.prover = dev, prover
        self.stop = threading.Event()
        self.count = 0
        self.err = None

    def __enter__(self):
        self.tabs = [self.dev.synthetic(21, True, 0x10AD + j) for j in range(3)]
        self.stream = self.dev.stream_create()
        self.th = threading.Thread(target=self.run)
        self.th.start()
        return self

    def run(self):
        one = np.array([[1, 0]], dtype=np.uint64)
        try:
            while not self.stop.is_set():
                self.prover.sumcheck_prove(self.dev, self.tabs, one, [[0, 1, 2]], 21, 3, self.prover.Transcript.stub(1), stream=self.stream)
                self.count += 1
        except Exception as e:  # noqa: BLE001
            self.err = e

    def __exit__(self, *a):
        self.stop.set()
        self.th.join()
        for t in self.tabs:
            t.free()
        self.dev.stream_destroy(self.stream)
        assert self.err is None, self.err
        assert self.count >= 1


def _same(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return np.array_equal(a, b)
    return a == b


def _repeat(reps, f):
    first = f()
    for i in range(1, reps):
        assert _same(f(), first), f"repetition {i} differs from the first"
    return first


def test_dense_rounds_are_the_same_in_every_run(dev, prover):
    tabs = [dev.synthetic(19, True, 0xD0 + j) for j in range(3)]
    one = np.array([[1, 0]], dtype=np.uint64)
    st = dev.stream_create()
    with Load(dev, prover):
        _repeat(200, lambda: prover.sumcheck_prove(dev, tabs, one, [[0, 1, 2]], 19, 3, prover.Transcript.stub(7), stream=st))
    for t in tabs:
        t.free()
    dev.stream_destroy(st)


@pytest.mark.parametrize("nv", [9, 14])
def test_small_round_kernels_are_the_same_in_every_run(dev, prover, nv):
    """a generic plan (two terms, degree 3, base and extension tables) at sizes whose rounds all run on the persistent mid / tail kernels and the
    host tail"""
    tabs = [dev.synthetic(nv, j % 2 == 0, 0x5A11 + j) for j in range(4)]
    coeffs = np.array([[3, 5], [7, 11]], dtype=np.uint64)
    terms = [[0, 1, 2], [1, 3]]
    st = dev.stream_create()
    with Load(dev, prover):
        _repeat(400, lambda: prover.sumcheck_prove(dev, tabs, coeffs, terms, nv, 3, prover.Transcript.stub(9), stream=st))
    for t in tabs:
        t.free()
    dev.stream_destroy(st)


def test_tower_rounds_are_the_same_in_every_run(dev, prover, monkeypatch):
    """two product towers and a LogUp tower of 2^16 / 2^17 entries, the fused k_tower rounds forced on from 2^6 pairs"""
    monkeypatch.setenv("CENO_HIP_TOWER_FAST_MIN_LOG", "6")
    rows = 1 << 12
    recs = [dev.synthetic(12, True, 0x70 + j) for j in range(8 + 5)]
    st = dev.stream_create()

    def once():
        p1 = prover.Tower.build_prod(dev, recs[:4], rows, stream=st)
        p2 = prover.Tower.build_prod(dev, recs[4:8], rows, stream=st)
        lk = prover.Tower.build_logup(dev, None, recs[8:], rows, (5, 6), stream=st)
        proof = prover.tower_create_proof(dev, [p1, p2], [lk], prover.Transcript.stub(3), stream=st)
        out = (proof.msgs.copy(), proof.prod_evals.copy(), proof.logup_evals.copy(), proof.point.copy())
        for t in (p1, p2, lk):
            t.free()
        return out

    with Load(dev, prover):
        _repeat(100, once)
    for t in recs:
        t.free()
    dev.stream_destroy(st)


def test_commitment_is_the_same_in_every_run(dev, prover):
    """transpose, RS encoding, the Poseidon2 leaf hash over three height classes and the tree: one root, every time; one opening of it too"""
    traces = [dev.synthetic((rows * w - 1).bit_length(), False, 0xC0 + i) for i, (rows, w) in enumerate(((1 << 14, 22), (1 << 11, 40), (1 << 14, 13)))]
    ptrs = [(t.device_ptr, rows, w) for t, (rows, w) in zip(traces, ((1 << 14, 22), (1 << 11, 40), (1 << 14, 13)))]
    st = dev.stream_create()

    def once():
        pcs = prover.PcsData(dev, None, 1, st, device_ptrs=ptrs)
        out = (pcs.root().copy(), pcs.open(12345)[1].copy())
        pcs.free()
        return out

    with Load(dev, prover):
        _repeat(150, once)
    for t in traces:
        t.free()
    dev.stream_destroy(st)


def test_opening_is_the_same_in_every_run(dev, prover):
    """batch open of a two-class commitment: batching, the degree-2 sumcheck, fold + commit rounds, proof of work, query gathers"""
    shapes = ((1 << 13, 9), (1 << 10, 5))
    traces = [dev.synthetic((rows * w - 1).bit_length(), False, 0x0B + i) for i, (rows, w) in enumerate(shapes)]
    st = dev.stream_create()
    pcs = prover.PcsData(dev, None, 1, st, device_ptrs=[(t.device_ptr, rows, w) for t, (rows, w) in zip(traces, shapes)])
    points = [po.rand_ext(13, 1), po.rand_ext(10, 2)]
    evals = [np.array([pcs.witness_mle(m, c).evaluate(points[m]) for c in range(w)], dtype=np.uint64) for m, (_, w) in enumerate(shapes)]
    with Load(dev, prover):
        _repeat(80, lambda: pcs.basefold_open(points, evals, 24, 8, prover.Transcript.poseidon2(b"open")))
    pcs.free()
    for t in traces:
        t.free()
    dev.stream_destroy(st)


def test_shard_witness_generation_is_the_same_in_every_run(dev, prover):
    """45 witness-generation kernels counting into one session's per-XCD lookup counters, the tables' mlt columns, the commitment over all of it"""
    from ceno_amd import synthetic

    flow = synthetic.ShardFlowWide(dev, prover, log_cycles=13, n_queries=8, pow_bits=4)

    def once():
        pcs = flow.generate_witness()
        pcs.finish()
        out = pcs.root().copy()
        pcs.free()
        return out

    with Load(dev, prover):
        _repeat(60, once)
    flow.close()


def test_chip_proofs_in_cohorts_are_the_same_in_every_run(dev, prover):
    """the whole wide-shard flow with the middle tower layers of all 54 chips proved in cohorts (host/cohort.cpp: records and towers of all chips
    in shared launches, one cohort launch per layer with the sub-cubes' messages added up by whichever workgroup arrives last, sixteen host
    threads answering) — every chip proof, fork sample, main-constraint message and opening word, every time"""
    from ceno_amd import synthetic

    flow = synthetic.ShardFlowWide(dev, prover, log_cycles=13, n_queries=8, pow_bits=4)

    def once():
        flow.run(lambda: prover.Transcript.stub(0x5A), lambda: prover.Transcript.stub(0xF0), lanes=8)
        a = flow.artifacts
        out = ([(p.tower_msgs.copy(), p.tower_point.copy(), p.tower_logup_evals.copy()) for p in a["chip_proofs"]], a["fork_samples"], a["msgs"].copy(),
               a["evals"].copy(), a["open_proof"].copy())
        flow.free_last()
        return out

    with Load(dev, prover):
        _repeat(40, once)
    flow.close()
