"""Oracle self-consistency for the Basefold batch open: the restated verifier
(ceno_recursion_v2/src/pcs/mod.rs:1111-1316,7494-7781) accepts what the restated prover emits and
rejects any tampering; the mixed-height commitment (p3 MerkleTreeMmcs) against a pure-Python model of the published
algorithm; the base-field challenger operations (sample_bits / check_witness / grind, pcs/mod.rs:8125-8204) against a
pure-Python model of the duplex rules.  PARITY UNPINNED beyond that — no reference vectors exist for this path."""
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
    assert roots.shape == (1, 4)  # ONE commitment for all matrices (scheme/cpu/mod.rs:559-584)
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


def test_two_commitments_witness_and_fixed():
    """PCS::batch_open takes `rounds` = [(witness commitment, openings), (fixed commitment, openings)] (cpu/mod.rs:1418-1457):
    the shorter commitment is opened at query >> bits_reduced (pcs/mod.rs:7553-7554)"""
    shapes = [(6, 3), (4, 2), (6, 1), (5, 4), (3, 2)]
    sizes = [3, 2]
    traces, points, evals = make_case(21, shapes)
    for tr in (lambda: po.StubTranscript(9), lambda: po.DuplexTranscript(b"open")):
        proof = po.basefold_open(traces, points, evals, 1, 5, 3, tr(), commit_sizes=sizes)
        roots = po.basefold_commit_roots(traces, 1, commit_sizes=sizes)
        assert roots.shape == (2, 4)
        assert np.array_equal(roots[0], po.basefold_commit_roots(traces[:3], 1)[0])
        assert np.array_equal(roots[1], po.basefold_commit_roots(traces[3:], 1)[0])
        assert po.basefold_verify(shapes, roots, points, evals, 1, 5, 3, tr(), proof, commit_sizes=sizes) == 0
        assert po.basefold_verify(shapes, roots[::-1].copy(), points, evals, 1, 5, 3, tr(), proof, commit_sizes=sizes) != 0
        for pos in range(len(proof) - 60, len(proof), 7):
            bad = proof.copy()
            bad[pos] = (int(bad[pos]) + 1) % P
            assert po.basefold_verify(shapes, roots, points, evals, 1, 5, 3, tr(), bad, commit_sizes=sizes) != 0, pos


# ---- mixed-height MMCS against a pure-Python model of p3-merkle-tree 0.4.3 ----
def _sponge(vals):
    st = np.zeros(8, dtype=np.uint64)
    k = 0
    for v in vals:
        st[k] = v
        k += 1
        if k == 4:
            st = po.poseidon2_permute(st)
            k = 0
    if k:
        st = po.poseidon2_permute(st)
    return st[:4].copy()


def _compress(a, b):
    return po.poseidon2_permute(np.concatenate([a, b]))[:4].copy()


def _model_mmcs(mats):
    """mats: (width, rows) column-major.  MerkleTree::new: tallest first (stable), inject shorter matrices at their layer"""
    order = sorted(range(len(mats)), key=lambda i: -mats[i].shape[1])
    H = mats[order[0]].shape[1]
    layer = [_sponge([v for i in order if mats[i].shape[1] == H for v in mats[i][:, r]]) for r in range(H)]
    layers = [layer]
    while len(layer) > 1:
        n = len(layer) // 2
        inj = [i for i in order if mats[i].shape[1] == n]
        nxt = []
        for j in range(n):
            d = _compress(layer[2 * j], layer[2 * j + 1])
            if inj:
                d = _compress(d, _sponge([v for i in inj for v in mats[i][:, j]]))
            nxt.append(d)
        layer = nxt
        layers.append(layer)
    return layers


@pytest.mark.parametrize("shapes", [[(3, 2)], [(3, 5), (3, 1)], [(2, 3), (4, 2), (2, 6), (4, 5), (1, 1)], [(0, 3), (2, 2)], [(4, 1), (0, 2), (3, 9)]])
def test_mmcs_matches_python_model_and_opens(shapes):
    rng = np.random.default_rng(5)
    mats = [rand_base(rng, (w, 1 << lr)) for lr, w in shapes]
    levels = po.mmcs_commit(mats)
    model = _model_mmcs(mats)
    assert len(levels) == len(model)
    for a, b in zip(levels, model):
        assert np.array_equal(a, np.stack(b))
    H = max(lr for lr, _ in shapes)
    root = levels[-1][0]
    for index in range(1 << H):
        rows, path = po.mmcs_open(mats, levels, index)
        want = np.concatenate([m[:, index >> (H - lr)] for m, (lr, _) in zip(mats, shapes)])
        assert np.array_equal(rows, want)
        for l in range(H):
            assert np.array_equal(path[l], levels[l][(index >> l) ^ 1])
        assert po.mmcs_verify(shapes, root, index, rows, path) == 0
        bad = rows.copy()
        bad[-1] ^= np.uint64(1)
        assert po.mmcs_verify(shapes, root, index, bad, path) != 0
        if H:
            assert po.mmcs_verify(shapes, root, index ^ 1, rows, path) != 0
    if len(set(lr for lr, _ in shapes)) == 1 and len(shapes) == 1:  # one matrix: the plain tree
        lr, w = shapes[0]
        plain = po.merkle_commit(mats[0], lr, w)
        assert all(np.array_equal(a, b) for a, b in zip(levels, plain))


def test_duplex_challenger_rules_and_grinding():
    """p3-challenger 0.4.3 DuplexChallenger<_, _, 8, 4>: model the buffers in Python; sample_bits = low bits of one base sample
    (pcs/mod.rs:8164-8204); check_witness = observe + sample_bits == 0 (pcs/mod.rs:8125-8155); grind finds such a witness"""
    class Model:
        def __init__(self):
            self.state = np.zeros(8, dtype=np.uint64)
            self.inp, self.out = [], []

        def duplex(self):
            for i, v in enumerate(self.inp):
                self.state[i] = v
            self.inp = []
            self.state = po.poseidon2_permute(self.state)
            self.out = [int(x) for x in self.state[:4]]

        def observe(self, v):
            self.out = []
            self.inp.append(v)
            if len(self.inp) == 4:
                self.duplex()

        def sample(self):
            if self.inp or not self.out:
                self.duplex()
            return self.out.pop()

    rng = np.random.default_rng(1)
    for n_pre in range(6):
        m, t = Model(), po.DuplexTranscript(b"")
        for _ in range(n_pre):
            v = int(rand_base(rng, (1,))[0])
            m.observe(v)
            t.append_base(v)
        assert t.sample_base() == m.sample()
        e = t.sample_ext()
        assert e == (m.sample(), m.sample())
        m.observe(5), m.observe(6)
        t.append_ext((5, 6))
        assert t.sample_bits(13) == m.sample() & ((1 << 13) - 1)
        # grinding: the witness passes check_witness on a clone, all smaller ones fail, and the transcript has advanced
        import copy

        bits = 6
        w = t.grind(bits)
        for cand in range(w + 1):
            c = copy.deepcopy(m)
            c.observe(cand)
            assert ((c.sample() & ((1 << bits) - 1)) == 0) == (cand == w)
        m.observe(w)
        m.sample()
        assert t.sample_base() == m.sample()
    # the stub transcript follows the same generic rules
    t = po.StubTranscript(3)
    t2 = po.StubTranscript(3)
    w = t.grind(7)
    assert t2.check_witness(7, w)
    assert t.sample_base() == t2.sample_base()
