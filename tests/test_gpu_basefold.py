"""GPU parity for the Basefold batch open: the HIP path must emit, word for word, the proof the
oracle's restated prover emits under the same transcript — stub and Poseidon2 duplex — and the oracle's restated verifier
(ceno_recursion_v2/src/pcs/mod.rs:1111-1316,7494-7781,8125-8204) must accept it.  One mixed-height commitment per
commit_traces, p3 grinding, one-base-sample query indices.  PARITY UNPINNED vs the reference: constants, label packing."""
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


def make_case(seed, shapes):
    traces = [po.rand_base((1 << nv) * w, seed + 7 * i).reshape(1 << nv, w) for i, (nv, w) in enumerate(shapes)]
    points = [po.rand_ext(nv, seed + 100 + i) for i, (nv, _) in enumerate(shapes)]
    evals = [np.array([po.mle_evaluate(t[:, c].copy(), p) for c in range(t.shape[1])], dtype=np.uint64) for t, p in zip(traces, points)]
    return traces, points, evals


@pytest.mark.parametrize("shapes,rate_log,nq,pow_bits", [
    ([(5, 3)], 1, 5, 4),
    ([(6, 4), (6, 2)], 1, 7, 0),
    ([(8, 3), (4, 5), (8, 1), (1, 2), (3, 7)], 1, 9, 8),
    ([(1, 1)], 1, 3, 2),
    ([(7, 6), (5, 2)], 2, 4, 5),
])
def test_open_matches_oracle_word_for_word_and_verifies(dev, shapes, rate_log, nq, pow_bits):
    from ceno_amd import prover

    stream = dev.stream_create()
    traces, points, evals = make_case(3, shapes)
    pcs = prover.PcsData(dev, traces, rate_log, stream)
    proof = pcs.basefold_open(points, evals, nq, pow_bits, prover.Transcript.stub(0xBF))
    expect = po.basefold_open(traces, points, evals, rate_log, nq, pow_bits, po.StubTranscript(0xBF))
    assert proof.shape == expect.shape
    bad = np.nonzero(proof != expect)[0]
    assert bad.size == 0, f"first mismatch at word {bad[:5]} of {proof.size}"
    roots = pcs.root().reshape(1, 4)   # ONE commitment for all matrices
    assert np.array_equal(roots, po.basefold_commit_roots(traces, rate_log))
    assert po.basefold_verify(shapes, roots, points, evals, rate_log, nq, pow_bits, po.StubTranscript(0xBF), proof) == 0
    pcs.free()
    dev.stream_destroy(stream)


@pytest.mark.parametrize("shapes_w,shapes_f,nq,pow_bits", [
    ([(6, 3), (4, 2), (6, 1)], [(5, 4), (3, 2)], 6, 5),
    ([(3, 2)], [(7, 3), (7, 1), (2, 5)], 5, 0),
])
def test_open_of_witness_and_fixed_commitments_poseidon2_transcript(dev, shapes_w, shapes_f, nq, pow_bits):
    """PCS::batch_open over TWO commitments (`rounds` = witness, fixed: cpu/mod.rs:1418-1457) under the Poseidon2 duplex
    transcript: the product's host challenger + device grinding against the oracle's own challenger, word for word"""
    from ceno_amd import prover

    stream = dev.stream_create()
    shapes = shapes_w + shapes_f
    traces, points, evals = make_case(11, shapes)
    pw = prover.PcsData(dev, traces[:len(shapes_w)], 1, stream)
    pf = prover.PcsData(dev, traces[len(shapes_w):], 1, stream)
    sizes = [len(shapes_w), len(shapes_f)]
    proof = pw.basefold_open(points, evals, nq, pow_bits, prover.Transcript.poseidon2(b"open"), more_commits=[pf])
    expect = po.basefold_open(traces, points, evals, 1, nq, pow_bits, po.DuplexTranscript(b"open"), commit_sizes=sizes)
    assert proof.shape == expect.shape
    bad = np.nonzero(proof != expect)[0]
    assert bad.size == 0, f"first mismatch at word {bad[:5]} of {proof.size}"
    roots = np.stack([pw.root(), pf.root()])
    assert np.array_equal(roots, po.basefold_commit_roots(traces, 1, commit_sizes=sizes))
    assert po.basefold_verify(shapes, roots, points, evals, 1, nq, pow_bits, po.DuplexTranscript(b"open"), proof, commit_sizes=sizes) == 0
    pw.free()
    pf.free()
    dev.stream_destroy(stream)


def test_transcript_base_operations_and_grinding_match_the_oracle(dev):
    """sample_base / sample_bits / check_witness / clone / export-import of the host challenger against the oracle's duplex
    challenger; the device proof-of-work search finds the oracle's (least) witness for every number of pending inputs"""
    from ceno_amd import prover

    for n_pre in range(6):
        t, o = prover.Transcript.poseidon2(b"pow"), po.DuplexTranscript(b"pow")
        for k in range(n_pre):
            t.append_base(1000 + k)
            o.append_base(1000 + k)
        if n_pre == 5:  # outputs left in the buffer: grinding must clear them
            assert t.sample_base() == o.sample_base()
        kind, st = t.export_state()
        assert kind == 1 and np.array_equal(st, o.export_state())
        c = t.clone()
        w = t.grind(dev, 9)
        assert w == o.grind(9)
        assert c.check_witness(9, w) and c.sample_ext() == t.sample_ext() == o.sample_ext()
        assert t.sample_bits(20) == o.sample_bits(20)
        t2 = prover.Transcript.poseidon2(b"")
        t2.import_state(t.export_state()[1])
        assert t2.sample_base() == t.sample_base() == o.sample_base()
    t, o = prover.Transcript.stub(4), po.StubTranscript(4)   # no exportable state: the host searches through clones
    assert t.export_state()[0] == 0
    assert t.grind(dev, 7) == o.grind(7) and t.sample_base() == o.sample_base()


def test_open_at_chip_size_verifies(dev):
    """ADD-chip shaped commitment (2^16 x 22 here) plus two smaller tables; too big for the oracle prover's
    quadratic pieces to be pleasant, so only the verifier (linear in the proof) is run."""
    from ceno_amd import prover

    stream = dev.stream_create()
    shapes = [(16, 22), (12, 9), (16, 3)]
    rng = np.random.default_rng(9)
    traces = [(rng.integers(0, 1 << 62, size=(1 << nv, w), dtype=np.uint64)) % np.uint64(P) for nv, w in shapes]
    points = [po.rand_ext(nv, 500 + i) for i, (nv, _) in enumerate(shapes)]
    pcs = prover.PcsData(dev, traces, 1, stream)
    evals = []
    for i, (nv, w) in enumerate(shapes):
        ev = np.zeros((w, 2), dtype=np.uint64)
        for c in range(w):
            ev[c] = pcs.witness_mle(i, c).evaluate(points[i])
        evals.append(ev)
    tr = prover.Transcript.poseidon2(b"open")
    proof = pcs.basefold_open(points, evals, 20, 10, tr)
    roots = pcs.root().reshape(1, 4)

    class P2(object):  # the oracle verifier driven by the host library's Poseidon2 transcript through its C table
        def __init__(self):
            self.t = prover.Transcript.poseidon2(b"open")

        def ptr(self):
            return self.t.h

    assert po.basefold_verify(shapes, roots, points, evals, 1, 20, 10, P2(), proof) == 0
    tampered = proof.copy()
    tampered[4 * 16 + 2] ^= np.uint64(1)  # a commit-round root
    assert po.basefold_verify(shapes, roots, points, evals, 1, 20, 10, P2(), tampered) != 0
    pcs.free()
    dev.stream_destroy(stream)


def test_open_of_witness_and_fixed_commitments_at_chip_size_verifies(dev):
    """`rounds` = (witness commitment: an ADD-shaped trace and two smaller tables, fixed commitment: one table) at 2^16 rows under the
    Poseidon2 duplex transcript, 16-bit proof of work found on the device: the ORACLE's own challenger and verifier (nothing of the
    product) accept the proof and reject tampering in every region"""
    from ceno_amd import prover

    stream = dev.stream_create()
    shapes_w, shapes_f = [(16, 22), (12, 9), (16, 3)], [(14, 4), (9, 2)]
    shapes = shapes_w + shapes_f
    rng = np.random.default_rng(10)
    traces = [(rng.integers(0, 1 << 62, size=(1 << nv, w), dtype=np.uint64)) % np.uint64(P) for nv, w in shapes]
    points = [po.rand_ext(nv, 600 + i) for i, (nv, _) in enumerate(shapes)]
    pw = prover.PcsData(dev, traces[:3], 1, stream)
    pf = prover.PcsData(dev, traces[3:], 1, stream)
    evals = []
    for i, (nv, w) in enumerate(shapes):
        pcs, m = (pw, i) if i < 3 else (pf, i - 3)
        evals.append(np.array([pcs.witness_mle(m, c).evaluate(points[i]) for c in range(w)], dtype=np.uint64))
    nq, pow_bits, sizes = 30, 16, [3, 2]
    proof = pw.basefold_open(points, evals, nq, pow_bits, prover.Transcript.poseidon2(b"open"), more_commits=[pf])
    roots = np.stack([pw.root(), pf.root()])
    assert po.basefold_verify(shapes, roots, points, evals, 1, nq, pow_bits, po.DuplexTranscript(b"open"), proof, commit_sizes=sizes) == 0
    n = 16
    for pos in (1, 4 * n + 2, 8 * n + 3, 8 * n + 2 * len(shapes), 8 * n + 2 * len(shapes) + 5, len(proof) - 3):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert po.basefold_verify(shapes, roots, points, evals, 1, nq, pow_bits, po.DuplexTranscript(b"open"), bad, commit_sizes=sizes) != 0, pos
    assert po.basefold_verify(shapes, roots[::-1].copy(), points, evals, 1, nq, pow_bits, po.DuplexTranscript(b"open"), proof, commit_sizes=sizes) != 0
    pw.free()
    pf.free()
    dev.stream_destroy(stream)


def test_batch_columns_and_fold_commit_primitives(dev):
    """kernel-level parity: column batching with unreduced accumulators, and one fused fold+commit round"""
    import ctypes as C
    from tests import hipbuf as torch

    from ceno_amd import _lib

    L = _lib.lib()
    n, width = 1 << 9, 37
    cols = po.rand_base(n * width, 4).reshape(width, n)
    cols[0, :4] = P - 1
    coeffs = po.rand_ext(width, 8)
    coeffs[0] = (P - 1, P - 1)
    d_cols = torch.from_numpy(cols.view(np.int64)).to("cuda:0")
    d_acc = torch.zeros(2 * n, dtype=torch.int64, device="cuda:0")
    c = np.ascontiguousarray(coeffs)
    for accumulate in (0, 1):
        rc = L.ceno_hip_batch_columns(dev.h, d_cols.data_ptr(), n, width, c.ctypes.data_as(C.POINTER(C.c_uint64)), d_acc.data_ptr(), accumulate, None)
        assert rc == 0
    got = d_acc.cpu().numpy().view(np.uint64).reshape(n, 2)
    for i in (0, 1, 3, 17, n - 1):
        want = (0, 0)
        for k in range(width):
            want = po.e2_add(want, po.e2_mul((int(coeffs[k, 0]), int(coeffs[k, 1])), (int(cols[k, i]), 0)))
        want = po.e2_add(want, want)
        assert (int(got[i, 0]), int(got[i, 1])) == want


@pytest.mark.parametrize("seed", [s_ if s_ < 6 else pytest.param(s_, marks=pytest.mark.slow) for s_ in range(10)])
def test_open_random_shapes_differential(dev, seed):
    """seeded random commitments (1-5 matrices, 1-9 variables, 1-9 columns, blow-up 2 or 4, ragged row counts that are
    zero padded) opened at random points: proof equal to the oracle's word for word, verifier accepts"""
    import random

    from ceno_amd import prover

    rng = random.Random(77 + seed)
    rate_log = rng.choice([1, 1, 2])
    shapes, traces = [], []
    for i in range(rng.randint(1, 5)):
        nv, w = rng.randint(1, 9), rng.randint(1, 9)
        n_inst = rng.randint(max(1, (1 << nv) // 2 + 1), 1 << nv) if nv > 1 else rng.choice([1, 2])
        t = po.rand_base(n_inst * w, 1000 * seed + i).reshape(n_inst, w)
        traces.append(t)
        shapes.append((nv, w))
    padded = []
    for t, (nv, w) in zip(traces, shapes):
        p = np.zeros((1 << nv, w), dtype=np.uint64)
        p[: t.shape[0]] = t
        padded.append(p)
    points = [po.rand_ext(nv, 31 * seed + i) for i, (nv, _) in enumerate(shapes)]
    evals = [np.array([po.mle_evaluate(p[:, c].copy(), pt) for c in range(p.shape[1])], dtype=np.uint64) for p, pt in zip(padded, points)]
    nq, pow_bits = rng.randint(1, 12), rng.choice([0, 3, 6])
    stream = dev.stream_create()
    pcs = prover.PcsData(dev, traces, rate_log, stream)
    proof = pcs.basefold_open(points, evals, nq, pow_bits, prover.Transcript.stub(seed))
    expect = po.basefold_open(padded, points, evals, rate_log, nq, pow_bits, po.StubTranscript(seed))
    assert np.array_equal(proof, expect)
    roots = pcs.root().reshape(1, 4)
    assert po.basefold_verify(shapes, roots, points, evals, rate_log, nq, pow_bits, po.StubTranscript(seed), proof) == 0
    pcs.free()
    dev.stream_destroy(stream)


_HOST_TOP_SCRIPT = r"""
import sys, zlib
import numpy as np
sys.path.insert(0, sys.argv[1])
from oracle import pyoracle as po
from ceno_amd import Device, prover
dev = Device(0)
stream = dev.stream_create()
out = []
for shapes, nq in (([(9, 3), (4, 5), (9, 1), (1, 2), (3, 7)], 9), ([(12, 2)], 6), ([(2, 3), (1, 1)], 4)):
    traces = [po.rand_base((1 << nv) * w, 3 + 7 * i).reshape(1 << nv, w) for i, (nv, w) in enumerate(shapes)]
    points = [po.rand_ext(nv, 103 + i) for i, (nv, _) in enumerate(shapes)]
    evals = [np.array([po.mle_evaluate(t[:, c].copy(), p) for c in range(t.shape[1])], dtype=np.uint64) for t, p in zip(traces, points)]
    pcs = prover.PcsData(dev, traces, 1, stream)
    proof = pcs.basefold_open(points, evals, nq, 3, prover.Transcript.stub(0xBF))
    rows, path = pcs.open(5 % (1 << max(nv for nv, _ in shapes)))
    out.append("%08x %08x %08x" % (zlib.crc32(pcs.root().tobytes()), zlib.crc32(proof.tobytes()), zlib.crc32(path.tobytes() + b"".join(r.tobytes() for r in rows))))
    pcs.free()
print("RESULT " + " | ".join(out))
"""


@pytest.mark.parametrize("levels", ["0", "3", "9"])
def test_host_finished_tree_tops_change_nothing(dev, levels):
    """CENO_HIP_HOST_TOP = how many top levels of every Merkle tree the HOST computes (default 5).  Roots, Basefold proofs and row
    openings must be identical for every split: all on the device (0), a split inside the tree (3), and a host half taller than
    most trees (9: whole trees above their leaf digests on the host; mixed heights keep their injections on the device).  The
    default split is what every other test of this file compares with the oracle."""
    import os, subprocess, sys, zlib
    from ceno_amd import prover

    root_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CENO_HIP_HOST_TOP=levels)
    r = subprocess.run([sys.executable, "-c", _HOST_TOP_SCRIPT, root_dir], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):]
    # the same cases in THIS process (default split)
    stream = dev.stream_create()
    want = []
    for shapes, nq in (([(9, 3), (4, 5), (9, 1), (1, 2), (3, 7)], 9), ([(12, 2)], 6), ([(2, 3), (1, 1)], 4)):
        traces, points, evals = make_case(3, shapes)
        pcs = prover.PcsData(dev, traces, 1, stream)
        proof = pcs.basefold_open(points, evals, nq, 3, prover.Transcript.stub(0xBF))
        expect = po.basefold_open(traces, points, evals, 1, nq, 3, po.StubTranscript(0xBF))
        assert np.array_equal(proof, expect)
        rows, path = pcs.open(5 % (1 << max(nv for nv, _ in shapes)))
        want.append("%08x %08x %08x" % (zlib.crc32(pcs.root().tobytes()), zlib.crc32(proof.tobytes()),
                                        zlib.crc32(path.tobytes() + b"".join(r.tobytes() for r in rows))))
        pcs.free()
    dev.stream_destroy(stream)
    assert got == " | ".join(want)


def _fold(t, r):
    return [po.e2_add(t[2 * j], po.e2_mul(r, po.e2_sub(t[2 * j + 1], t[2 * j]))) for j in range(len(t) // 2)]


@pytest.mark.parametrize("nvs", [[11, 9, 11, 6, 3, 1, 0, 9], [4], [1, 0, 1], [14, 14, 12, 13]])
def test_open_rounds_all_matrices_in_one_launch(dev, nvs):
    """ceno_hip_open_rounds_*: the opening's degree-2 sumcheck over matrices of mixed heights (suffix alignment: a matrix of v variables joins in
    round n - v), one launch per round — every message and the final evaluations against a pure-python restatement of the rounds
    (ceno_recursion_v2/src/pcs/mod.rs:1111-1316): fold the live tables with the previous challenge, p(1) = sum eq(1) f(1), p(2) = sum eq(2) f(2)."""
    from ceno_amd import api

    stream = dev.stream_create()
    n = max(nvs)
    eq = [po.rand_ext(1 << nv, 500 + i) for i, nv in enumerate(nvs)]
    f = [po.rand_ext(1 << nv, 900 + i) for i, nv in enumerate(nvs)]
    d_eq, d_f = [dev.upload(t) for t in eq], [dev.upload(t) for t in f]
    h = api.OpenRounds(dev, d_eq, d_f, nvs, stream)
    cur = [([(int(a), int(b)) for a, b in e], [(int(a), int(b)) for a, b in t]) for e, t in zip(eq, f)]
    ch = None
    for r in range(n):
        p1, p2 = (0, 0), (0, 0)
        for m, nv in enumerate(nvs):
            if nv == 0 or nv < n - r:
                continue  # not live yet (or never: a single entry has no variable)
            if nv > n - r:  # was live in the round before: bind that round's variable
                cur[m] = (_fold(cur[m][0], ch), _fold(cur[m][1], ch))
            e, t = cur[m]
            assert len(e) == 1 << (n - r)
            for j in range(len(e) // 2):
                p1 = po.e2_add(p1, po.e2_mul(e[2 * j + 1], t[2 * j + 1]))
                e2 = po.e2_sub(po.e2_add(e[2 * j + 1], e[2 * j + 1]), e[2 * j])
                t2 = po.e2_sub(po.e2_add(t[2 * j + 1], t[2 * j + 1]), t[2 * j])
                p2 = po.e2_add(p2, po.e2_mul(e2, t2))
        got = h.round(ch)
        assert (int(got[0, 0]), int(got[0, 1])) == p1 and (int(got[1, 0]), int(got[1, 1])) == p2, f"round {r}"
        c = po.rand_ext(1, 7000 + r)[0]
        ch = (int(c[0]) % P, int(c[1]) % P)
    fin = h.finish(ch if n else None)
    for m, nv in enumerate(nvs):
        want = _fold(cur[m][1], ch)[0] if nv >= 1 else cur[m][1][0]
        assert (int(fin[m, 0]), int(fin[m, 1])) == want, f"final evaluation of matrix {m}"
    # the inputs are left as they were
    for m in range(len(nvs)):
        assert np.array_equal(d_eq[m].download(stream), eq[m]) and np.array_equal(d_f[m].download(stream), f[m])
    with pytest.raises(Exception):
        h.round(ch)  # all rounds are done
    h.free()
    dev.stream_destroy(stream)


def test_one_launch_per_round_writes_the_proof_of_the_handles_per_height_group(dev, monkeypatch):
    """CENO_BASEFOLD_OPEN_ROUNDS=0 (one generic sumcheck handle per height group, the path before round 6) and the default (all matrices in one
    launch per round): the same proof, which is the oracle's"""
    from ceno_amd import prover

    shapes = [(10, 3), (4, 5), (10, 1), (1, 2), (3, 7), (9, 2), (7, 1)]
    stream = dev.stream_create()
    traces, points, evals = make_case(11, shapes)
    pcs = prover.PcsData(dev, traces, 1, stream)
    proof = pcs.basefold_open(points, evals, 8, 3, prover.Transcript.stub(0xBF))
    monkeypatch.setenv("CENO_BASEFOLD_OPEN_ROUNDS", "0")
    proof_handles = pcs.basefold_open(points, evals, 8, 3, prover.Transcript.stub(0xBF))
    monkeypatch.delenv("CENO_BASEFOLD_OPEN_ROUNDS")
    assert np.array_equal(proof, proof_handles)
    assert np.array_equal(proof, po.basefold_open(traces, points, evals, 1, 8, 3, po.StubTranscript(0xBF)))
    pcs.free()
    dev.stream_destroy(stream)
