"""Oracle self-consistency for the Basefold batch open: the restated verifier
(ceno_recursion_v2/src/pcs/mod.rs:1111-1316,7494-7781) accepts what the restated prover emits and
rejects any tampering.  PARITY UNPINNED — no reference vectors exist for this path (oracle/basefold.c)."""
import numpy as np
import pytest

from oracle import pyoracle as po

P = po.P


def rand_base(rng, shape):
    return (rng.integers(0, 1 << 63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)) % np.uint64(P)


def make_case(seed, shapes):
    rng = np.random.default_rng(seed)
    traces = [rand_base(rng, (1 << nv, w)) for nv, w in shapes]
    points = [rand_base(rng, (nv, 2)) for nv, _ in shapes]
    evals = [np.array([po.mle_evaluate(t[:, c].copy(), p) for c in range(t.shape[1])], dtype=np.uint64) for t, p in zip(traces, points)]
    return traces, points, evals


def test_fft_matches_dft_by_definition():
    rng = np.random.default_rng(3)
    for log_n in (1, 2, 5, 8):
        col = rand_base(rng, (1 << log_n,))
        assert np.array_equal(po.fft_bitrev(col), po.dft_bitrev(col))


@pytest.mark.parametrize("shapes", [[(5, 3)], [(6, 4), (6, 2)], [(7, 3), (4, 5), (7, 1), (1, 2)], [(1, 1)]])
def test_open_then_verify_accepts_and_tampering_rejects(shapes):
    rate_log, nq, pow_bits = 1, 6, 4
    traces, points, evals = make_case(11, shapes)
    proof = po.basefold_open(traces, points, evals, rate_log, nq, pow_bits, po.StubTranscript(0xBF))
    roots = po.basefold_commit_roots(traces, rate_log)
    ok = po.basefold_verify(shapes, roots, points, evals, rate_log, nq, pow_bits, po.StubTranscript(0xBF), proof)
    assert ok == 0
    n = max(nv for nv, _ in shapes)
    # flip one word in each region of the proof: sumcheck message, commit root, final message, query payload
    for pos in (0, 4 * n + 1, 8 * n, 8 * n + 2 * len(shapes) + 3, len(proof) - 1):
        bad = proof.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert po.basefold_verify(shapes, roots, points, evals, rate_log, nq, pow_bits, po.StubTranscript(0xBF), bad) != 0, pos
    # a wrong claimed evaluation breaks the initial claim
    bad_evals = [e.copy() for e in evals]
    bad_evals[0][0, 0] = (int(bad_evals[0][0, 0]) + 1) % P
    assert po.basefold_verify(shapes, roots, points, bad_evals, rate_log, nq, pow_bits, po.StubTranscript(0xBF), proof) != 0
    # a different commitment is rejected by the input openings
    bad_roots = roots.copy()
    bad_roots[-1, 2] ^= np.uint64(1)
    assert po.basefold_verify(shapes, bad_roots, points, evals, rate_log, nq, pow_bits, po.StubTranscript(0xBF), proof) != 0


def test_pow_and_rate_variants():
    shapes = [(5, 2), (3, 3)]
    traces, points, evals = make_case(5, shapes)
    for rate_log, pow_bits in ((2, 0), (1, 8)):
        proof = po.basefold_open(traces, points, evals, rate_log, 4, pow_bits, po.StubTranscript(1))
        roots = po.basefold_commit_roots(traces, rate_log)
        assert po.basefold_verify(shapes, roots, points, evals, rate_log, 4, pow_bits, po.StubTranscript(1), proof) == 0
        if pow_bits:
            bad = proof.copy()
            bad[8 * 5 + 2 * 2] += np.uint64(1)  # pow witness
            assert po.basefold_verify(shapes, roots, points, evals, rate_log, 4, pow_bits, po.StubTranscript(1), bad) != 0
