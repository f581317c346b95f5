"""Pin the CPU oracle against the reference's own known-answer tests and identity tests.

golden:  ceno_zkvm/src/scheme/utils.rs:934-1194  -> tests/golden/tower_witness.json
identities: gkr_iop/src/utils.rs:356-441 (succinct evaluators == MLE.evaluate),
            gkr_iop/src/selector.rs:396-435 (quark selector),
            ceno_zkvm/src/scheme/tests.rs:447-500 (tower prove -> verify, leaf sizes 2..512)
"""
import json
import os
import random

import numpy as np
import pytest

from oracle import pyoracle as po

P = po.P
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tower_witness.json")))


def E(vals):
    return po.ext(vals)


def as_ints(a):
    return [int(x[0]) for x in a], [int(x[1]) for x in a]


# ------------------------------------------------------------------------------------------
# literal known-answer vectors
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", GOLD["interleaving_mles_to_mles"], ids=lambda c: c["ref"])
def test_interleaving_golden(case):
    mles = [E(m) for m in case["mles"]]
    res = po.interleaving_mles_to_mles(mles, case["num_instances"], case["num_limbs"], case["default"])
    for got, exp in zip(res, case["expected"]):
        c0, c1 = as_ints(got)
        assert c0 == exp and all(x == 0 for x in c1)


def test_infer_tower_product_witness_golden():
    case = GOLD["infer_tower_product_witness"][0]
    last = [E(l) for l in case["last_layer"]]
    layers = po.infer_tower_product_witness(case["num_vars"], last)
    assert len(layers) == case["num_layers"]
    assert all(len(l) == case["limbs_per_layer"] for l in layers)
    left, right = layers[0]
    assert left.shape[0] == 1 and right.shape[0] == 1
    assert po.e2_mul(tuple(map(int, left[0])), tuple(map(int, right[0]))) == (case["final_product"], 0)
    # layer l has 2^l evaluations per limb (cpu/mod.rs:678-685)
    for l, lay in enumerate(layers):
        assert lay[0].shape[0] == 1 << l and lay[1].shape[0] == 1 << l


def test_infer_tower_logup_witness_golden():
    case = GOLD["infer_tower_logup_witness"][0]
    q = [E(x) for x in case["q"]]
    layers = po.infer_tower_logup_witness(None, q)
    assert len(layers) == case["num_layers"]
    for got_layer, exp_layer in zip(layers, case["layers"]):
        for got, exp in zip(got_layer, exp_layer):
            c0, c1 = as_ints(got)
            assert c0 == exp and all(x == 0 for x in c1)


def test_logup_with_numerators_matches_formula():
    rng = random.Random(5)
    n = 3
    p = [po.rand_ext(1 << n, 11), po.rand_ext(1 << n, 12)]
    q = [po.rand_ext(1 << n, 13), po.rand_ext(1 << n, 14)]
    layers = po.infer_tower_logup_witness(p, q)
    # one layer up: (p, q) <- (q1*p2 + q2*p1, q1*q2) over each half (utils.rs:464-479)
    up = layers[n - 1]
    half = 1 << (n - 1)
    for index in range(2):
        for j in range(half):
            i = index * half + j
            p1, p2, q1, q2 = (tuple(map(int, a[i])) for a in (p[0], p[1], q[0], q[1]))
            assert tuple(map(int, up[index][j])) == po.e2_add(po.e2_mul(q1, p2), po.e2_mul(q2, p1))
            assert tuple(map(int, up[2 + index][j])) == po.e2_mul(q1, q2)


# ------------------------------------------------------------------------------------------
# identity tests restated from gkr_iop/src/utils.rs
# ------------------------------------------------------------------------------------------
R5 = E([123, 456, 789, 3210, 9876])


def test_eq_table_is_lsb_first_and_matches_eq_eval():
    pt = po.rand_ext(5, 77)
    eq = po.build_eq(pt)
    # entry i = prod_k (bit_k(i) ? r_k : 1 - r_k)
    for i in (0, 1, 6, 19, 31):
        acc = (1, 0)
        for k in range(5):
            r = tuple(map(int, pt[k]))
            acc = po.e2_mul(acc, r if (i >> k) & 1 else po.e2_sub((1, 0), r))
        assert tuple(map(int, eq[i])) == acc
    # sum_x eq(x,r) f(x) == f(r)
    f = po.rand_ext(32, 78)
    s = (0, 0)
    for i in range(32):
        s = po.e2_add(s, po.e2_mul(tuple(map(int, eq[i])), tuple(map(int, f[i]))))
    assert s == po.mle_evaluate(f, pt)
    other = po.rand_ext(5, 79)
    assert po.mle_evaluate(eq, other) == po.eq_eval(pt, other)


def test_eval_wellform_address_vec():
    # M'(r) = r0 + 2 r1 + ...  (utils.rs:215-232)
    for n in range(1, 6):
        v = E(list(range(1 << n)))
        assert po.eval_wellform_address_vec(0, 1, R5[:n]) == po.mle_evaluate(v, R5[:n])
        v2 = E([(7 - 3 * i) % P for i in range(1 << n)])
        assert po.eval_wellform_address_vec(7, 3, R5[:n], descending=True) == po.mle_evaluate(v2, R5[:n])


def test_eval_inner_and_outer_repeated_incremental_vec():  # utils.rs:398-441, definitions utils.rs:310-327
    """eval_inner_repeated_incremental_vec(k, r) = wellform(r[k:]), eval_outer_...(k, r) = wellform(r[:k]); the
    reference checks them against the explicit tables [each value repeated 2^k times] / [0..2^k repeated]."""
    r = po.ext([123, 456, 789, 3210, 9876])  # the reference's own point
    for n in range(1, 6):
        for k in range(n + 1):
            inner = E([i for i in range(1 << (n - k)) for _ in range(1 << k)])
            outer = E([j for _ in range(1 << (n - k)) for j in range(1 << k)])
            want_in, want_out = po.mle_evaluate(inner, r[:n]), po.mle_evaluate(outer, r[:n])
            got_in = po.eval_wellform_address_vec(0, 1, r[k:n]) if k < n else (0, 0)
            got_out = po.eval_wellform_address_vec(0, 1, r[:k]) if k > 0 else (0, 0)
            assert got_in == want_in and got_out == want_out, (n, k)


def test_eval_stacked_wellform_address_vec():  # utils.rs:355-374
    for n in range(5):
        v = [0] + [j for i in range(n + 1) for j in range(1 << i)]
        assert po.eval_stacked_wellform_address_vec(R5[: n + 1]) == po.mle_evaluate(E(v), R5[: n + 1])


def test_eval_stacked_constant_vec():  # utils.rs:376-395
    for n in range(5):
        v = [0] + [i for i in range(n + 1) for _ in range(1 << i)]
        assert po.eval_stacked_constant_vec(R5[: n + 1]) == po.mle_evaluate(E(v), R5[: n + 1])


def test_eq_eval_less_or_equal_than():
    a, b = po.rand_ext(4, 3), po.rand_ext(4, 4)
    eqa, eqb = po.build_eq(a), po.build_eq(b)
    for max_idx in range(16):
        s = (0, 0)
        for i in range(max_idx + 1):
            s = po.e2_add(s, po.e2_mul(tuple(map(int, eqa[i])), tuple(map(int, eqb[i]))))
        assert po.eq_eval_less_or_equal_than(max_idx, a, b) == s


# ------------------------------------------------------------------------------------------
# selectors: compute() table evaluated at in_point == evaluate()   (selector.rs:396-435)
# ------------------------------------------------------------------------------------------
def test_quark_lt_selector_reference_case():
    n_points, n_vars = 5, 3
    out_rt = po.rand_ext(n_vars, 21)
    sel = po.selector_compute(po.SEL_QUARK_LT, out_rt, 0, n_points)
    eq = po.build_eq(out_rt)
    zero = (0, 0)
    got = [tuple(map(int, x)) for x in sel]
    exp = [tuple(map(int, x)) for x in eq]
    assert got[0] == exp[0] and got[1] == exp[1] and got[2] == zero and got[3] == zero
    assert got[4] == exp[4] and got[5] == zero and got[6] == exp[6] and got[7] == zero
    in_rt = po.rand_ext(n_vars, 22)
    assert po.mle_evaluate(sel, in_rt) == po.selector_evaluate(po.SEL_QUARK_LT, out_rt, in_rt, 0, n_points)


@pytest.mark.parametrize("n_inst", [1, 2, 3, 7, 8, 13, 16])
def test_quark_lt_selector_sizes(n_inst):
    nv = 4
    o, i = po.rand_ext(nv, 31 + n_inst), po.rand_ext(nv, 41 + n_inst)
    sel = po.selector_compute(po.SEL_QUARK_LT, o, 0, n_inst)
    assert po.mle_evaluate(sel, i) == po.selector_evaluate(po.SEL_QUARK_LT, o, i, 0, n_inst)


@pytest.mark.parametrize("offset,n_inst", [(0, 16), (0, 5), (3, 7), (15, 1), (0, 0), (4, 0)])
def test_prefix_selector(offset, n_inst):
    nv = 4
    o, i = po.rand_ext(nv, 51), po.rand_ext(nv, 52)
    sel = po.selector_compute(po.SEL_PREFIX, o, offset, n_inst)
    eq = po.build_eq(o)
    for x in range(16):
        exp = tuple(map(int, eq[x])) if offset <= x < offset + n_inst else (0, 0)
        assert tuple(map(int, sel[x])) == exp
    if offset + n_inst > 0:
        assert po.mle_evaluate(sel, i) == po.selector_evaluate(po.SEL_PREFIX, o, i, offset, n_inst)


def test_whole_and_ordered_sparse_selector():
    nv, snv = 6, 3
    o, i = po.rand_ext(nv, 61), po.rand_ext(nv, 62)
    assert po.mle_evaluate(po.selector_compute(po.SEL_WHOLE, o), i) == po.selector_evaluate(po.SEL_WHOLE, o, i)
    idx = [1, 4, 6]
    for n_inst in (1, 5, 8):
        sel = po.selector_compute(po.SEL_ORDERED_SPARSE, o, 0, n_inst, idx, snv)
        eq = po.build_eq(o)
        for x in range(64):
            keep = (x >> snv) < n_inst and (x & 7) in idx
            assert tuple(map(int, sel[x])) == (tuple(map(int, eq[x])) if keep else (0, 0))
        assert po.mle_evaluate(sel, i) == po.selector_evaluate(po.SEL_ORDERED_SPARSE, o, i, 0, n_inst, idx, snv)


# ------------------------------------------------------------------------------------------
# sumcheck: prover messages satisfy the restated verifier; front-load rule
# ------------------------------------------------------------------------------------------
def _claimed_sum(mles, coeffs, terms, max_nv):
    tot = (0, 0)
    for c, t in zip(coeffs, terms):
        nv = int(mles[t[0]].shape[0]).bit_length() - 1
        s = (0, 0)
        for x in range(1 << nv):
            v = (1, 0)
            for j in t:
                e = mles[j][x]
                v = po.e2_mul(v, (int(e[0]), int(e[1])) if mles[j].ndim == 2 else (int(e), 0))
            s = po.e2_add(s, v)
        tot = po.e2_add(tot, po.e2_mul(tuple(map(int, c)), s))
    return tot


@pytest.mark.parametrize("nv,k", [(1, 1), (3, 2), (5, 3), (6, 4)])
def test_sumcheck_dense_roundtrip(nv, k):
    mles = [po.rand_ext(1 << nv, 100 + j) for j in range(k)]
    coeffs = po.rand_ext(1, 99)
    terms = [list(range(k))]
    msgs, chal, fin = po.sumcheck_prove(mles, coeffs, terms, nv, k, po.StubTranscript(1))
    claim = _claimed_sum(mles, coeffs, terms, nv)
    point, expected = po.sumcheck_verify(claim, msgs, po.StubTranscript(1))
    assert np.array_equal(point, chal)
    for j in range(k):
        assert tuple(map(int, fin[j])) == po.mle_evaluate(mles[j], chal)
    assert po.sumcheck_expected_from_evals([nv] * k, coeffs, terms, nv, chal, fin) == expected
    assert po.recover_claim_from_final(expected, msgs, chal) == claim
    # first message: p(0) + p(1) == claim, so p(1) alone is what is sent
    m2, f2 = po.sumcheck_dense_mt(mles, chal, threads=2)
    # the dense fused path has coefficient 1
    msgs1, chal1, fin1 = po.sumcheck_prove(mles, po.ext([1]), terms, nv, k, po.StubTranscript(1))
    m3, f3 = po.sumcheck_dense_mt(mles, chal1, threads=3)
    assert np.array_equal(m3, msgs1) and np.array_equal(f3, fin1)


def test_sumcheck_mixed_sizes_frontload_and_base_inputs():
    # three "chips" of 5, 3 and 2 variables inside one 5-variable sumcheck; base and ext tables
    big = [po.rand_base(32, 1), po.rand_ext(32, 2), po.rand_ext(32, 3)]
    mid = [po.rand_ext(8, 4), po.rand_base(8, 5)]
    small = [po.rand_ext(4, 6)]
    mles = big + mid + small
    terms = [[0, 1, 2], [1, 2], [3, 4], [3, 3, 4], [5], [5, 5]]
    coeffs = po.rand_ext(len(terms), 7)
    nv, d = 5, 3
    msgs, chal, fin = po.sumcheck_prove(mles, coeffs, terms, nv, d, po.StubTranscript(9))
    claim = _claimed_sum(mles, coeffs, terms, nv)
    point, expected = po.sumcheck_verify(claim, msgs, po.StubTranscript(9))
    assert np.array_equal(point, chal)
    nvs = [5, 5, 5, 3, 3, 2]
    for j, m in enumerate(mles):
        assert tuple(map(int, fin[j])) == po.mle_evaluate(m, chal[: nvs[j]])
    assert po.sumcheck_expected_from_evals(nvs, coeffs, terms, nv, chal, fin) == expected
    assert po.recover_claim_from_final(expected, msgs, chal) == claim


def test_sumcheck_rejects_bad_plans():
    m = [po.rand_ext(4, 1), po.rand_ext(8, 2)]
    with pytest.raises(ValueError):
        po.sumcheck_prove(m, po.ext([1]), [[0, 1]], 3, 2, po.StubTranscript(1))  # mixed sizes in a term
    with pytest.raises(ValueError):
        po.sumcheck_prove(m, po.ext([1]), [[]], 3, 2, po.StubTranscript(1))  # empty term
    with pytest.raises(ValueError):
        po.sumcheck_prove(m, po.ext([1]), [[0, 0, 0]], 3, 2, po.StubTranscript(1))  # degree > max_degree


def test_extrapolate_uni_poly():
    rng = random.Random(3)
    coef = [(rng.randrange(P), rng.randrange(P)) for _ in range(4)]

    def ev(x):
        acc = (0, 0)
        for c in reversed(coef):
            acc = po.e2_add(po.e2_mul(acc, x), c)
        return acc

    pts = [ev((i, 0)) for i in range(4)]
    x = (rng.randrange(P), rng.randrange(P))
    assert po.extrapolate_uni_poly(pts[0], po.ext(pts[1:]), x) == ev(x)


# ------------------------------------------------------------------------------------------
# tower: prove -> verify, leaf.evaluate(rt) == claimed eval  (scheme/tests.rs:447-500)
# ------------------------------------------------------------------------------------------
def _prod_spec(nv, seed):
    last = [po.rand_ext(1 << (nv - 1), seed), po.rand_ext(1 << (nv - 1), seed + 1)]
    return po.infer_tower_product_witness(nv, last)


def _logup_spec(nv, seed, with_p=True):
    q = [po.rand_ext(1 << (nv - 1), seed), po.rand_ext(1 << (nv - 1), seed + 1)]
    p = [po.rand_ext(1 << (nv - 1), seed + 2), po.rand_ext(1 << (nv - 1), seed + 3)] if with_p else None
    return po.infer_tower_logup_witness(p, q)


@pytest.mark.parametrize("leaf_log", range(1, 10))  # leaf layer sizes 2..512
def test_tower_product_roundtrip(leaf_log):
    nv = leaf_log + 1  # two limbs of 2^leaf_log
    spec = _prod_spec(nv, 1000 + leaf_log)
    proof = po.tower_prove([spec], [], po.StubTranscript(5))
    out_evals = np.stack([spec[0][0][0], spec[0][1][0]])[None]
    rc, pt, pc, lp, lq = po.tower_verify(out_evals, None, [nv], proof, po.StubTranscript(5))
    assert rc == 0
    assert np.array_equal(pt, proof.point[:nv])
    # the final claim is the evaluation of the interleaved leaf layer at rt (point = rt || r_merge)
    leaf = np.concatenate([spec[nv - 1][0], spec[nv - 1][1]])
    assert po.mle_evaluate(leaf, pt) == tuple(map(int, pc[0]))


def test_tower_mixed_heights_prod_and_logup():
    # NB: a 1-layer spec mixed with taller ones is rejected by the reference's own verifier (its
    # initial claim includes every spec, verifier.rs:1466-1474, while round 0 only folds specs with
    # max_round > 1, :1587) — so heights here are all >= 2, as in real chips.
    specs_p = [_prod_spec(6, 1), _prod_spec(4, 3), _prod_spec(2, 5)]
    specs_l = [_logup_spec(5, 7, True), _logup_spec(6, 11, False)]
    proof = po.tower_prove(specs_p, specs_l, po.StubTranscript(8))
    po_ev = np.stack([np.stack([s[0][0][0], s[0][1][0]]) for s in specs_p])
    lo_ev = np.stack([np.stack([s[0][k][0] for k in range(4)]) for s in specs_l])
    nvs = [6, 4, 2, 5, 6]
    rc, pt, pc, lp, lq = po.tower_verify(po_ev, lo_ev, nvs, proof, po.StubTranscript(8))
    assert rc == 0
    # every tower round runs a fresh sumcheck, so only the tallest specs end at the returned point
    for i, s in enumerate(specs_p):
        nv = len(s)
        if nv != 6:
            continue
        leaf = np.concatenate([s[nv - 1][0], s[nv - 1][1]])
        assert po.mle_evaluate(leaf, pt[:nv]) == tuple(map(int, pc[i]))
    for i, s in enumerate(specs_l):
        nv = len(s)
        if nv != 6:
            continue
        pl = np.concatenate([s[nv - 1][0], s[nv - 1][1]])
        ql = np.concatenate([s[nv - 1][2], s[nv - 1][3]])
        assert po.mle_evaluate(pl, pt[:nv]) == tuple(map(int, lp[i]))
        assert po.mle_evaluate(ql, pt[:nv]) == tuple(map(int, lq[i]))
    # tampering with one message must be rejected
    proof.msgs[4] ^= np.uint64(1)
    rc, *_ = po.tower_verify(po_ev, lo_ev, nvs, proof, po.StubTranscript(8))
    assert rc != 0


def test_wit_infer_matches_definition():
    nv = 4
    mles = [po.rand_base(16, 1), po.rand_base(16, 2), po.rand_ext(16, 3)]
    terms = [[0], [0, 1], [1, 2], [0, 0, 2]]
    coeffs = po.rand_ext(4, 9)
    out = po.wit_infer(mles, coeffs, terms, nv)
    for x in range(16):
        acc = (0, 0)
        for c, t in zip(coeffs, terms):
            v = tuple(map(int, c))
            for j in t:
                e = mles[j][x]
                v = po.e2_mul(v, (int(e[0]), int(e[1])) if mles[j].ndim == 2 else (int(e), 0))
            acc = po.e2_add(acc, v)
        assert tuple(map(int, out[x])) == acc


# ------------------------------------------------------------------------------------------
# rotation (a11): cyclic tables and the identity of test_rotation_next_base_mle_eval (gkr_iop/src/utils.rs:332-353)
# ------------------------------------------------------------------------------------------
def test_cyclic_tables_match_reference_prefix_and_are_cyclic():
    t5, t6 = po.cyclic_table(5), po.cyclic_table(6)
    # first entries as listed in gkr_iop/src/gkr/booleanhypercube.rs:10-31 and the wrap-around entry
    assert list(t5[:20]) == [1, 2, 4, 8, 16, 5, 10, 20, 13, 26, 17, 7, 14, 28, 29, 31, 27, 19, 3, 6]
    assert t5[31] == 1 and len(set(t5[:31].tolist())) == 31
    assert t6[63] == 1 and len(set(t6[:63].tolist())) == 63
    assert list(t6[37:48]) == [44, 27, 54, 47, 29, 58, 55, 45, 25, 50, 39]  # booleanhypercube.rs:86-96


@pytest.mark.parametrize("log2", [5, 6])
def test_rotation_next_base_mle_eval_identity(log2):
    nv = log2 + 2
    poly = np.arange(1 << nv, dtype=np.uint64)
    rotated = po.rotation_next_base_mle(poly, log2)
    point = po.rand_ext(nv, 9 + log2)
    left, right = po.rotation_points(point, log2)
    rot_eval = po.mle_evaluate(rotated, point)
    le, re = po.mle_evaluate(poly, left), po.mle_evaluate(poly, right)
    rk = tuple(map(int, point[log2 - 1]))
    exp = po.e2_add(po.e2_mul(po.e2_sub((1, 0), rk), le), po.e2_mul(rk, re))
    assert rot_eval == exp


def test_prove_rotation_messages_verify():
    log2, nv = 5, 8
    # a witness where target = rotated(source) on the selected subgroup makes the claimed sum zero
    src = po.rand_base(1 << nv, 5)
    tgt = po.rotation_next_base_mle(src, log2)
    rt = po.rand_ext(nv, 6)
    msgs, evals, origin, left, right = po.prove_rotation([src, tgt], [(0, 1)], 23, log2, rt, po.StubTranscript(3))
    # verifier side: claimed sum 0, degree 2
    tr = po.StubTranscript(3)
    tr.append_label(b"combine subset evals")
    tr.sample_ext()
    point, expected = po.sumcheck_verify((0, 0), msgs, tr)
    assert np.array_equal(point, origin)
    # left/right evals are evaluations of the source at the rotation points, target at the origin
    assert tuple(map(int, evals[0])) == po.mle_evaluate(src, left)
    assert tuple(map(int, evals[1])) == po.mle_evaluate(src, right)
    assert tuple(map(int, evals[2])) == po.mle_evaluate(tgt, origin)
    # expected evaluation = sel(origin) * (rotated(origin) - target(origin)) with rotated(origin) from left/right
    eq = po.build_eq(rt)
    sel = po.rotation_selector(eq, 23, log2)
    rk = tuple(map(int, origin[log2 - 1]))
    rot = po.e2_add(po.e2_mul(po.e2_sub((1, 0), rk), tuple(map(int, evals[0]))), po.e2_mul(rk, tuple(map(int, evals[1]))))
    assert po.e2_mul(po.mle_evaluate(sel, origin), po.e2_sub(rot, tuple(map(int, evals[2])))) == expected


@pytest.mark.parametrize("n,k", [(1, 3), (2, 1), (3, 3), (4, 2), (6, 3), (11, 3), (13, 4)])
def test_dense_avx512_equals_scalar(n, k):
    """bench.py's cpu_baseline times the AVX-512 form of the fused dense sumcheck (oracle/dense_avx512.c): its messages, folded tables' effect and
    final evaluations must equal the scalar restatement's word for word (rounds of fewer than eight pairs run its scalar remainder)"""
    if not po.have_avx512():
        pytest.skip("this CPU has no AVX-512 F + DQ")
    tabs = [po.rand_ext(1 << n, 900 + 7 * n + j) for j in range(k)]
    ch = po.rand_ext(n, 31 + n)
    for threads in (1, 3):
        a = po.sumcheck_dense_mt(tabs, ch, threads=threads)
        b = po.sumcheck_dense_mt(tabs, ch, threads=threads, avx512=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # edge values: all zero, all p - 1 (the largest canonical residue in both limbs)
    for fill in (0, po.P - 1):
        tabs = [np.full((1 << n, 2), fill, dtype=np.uint64) for _ in range(k)]
        a = po.sumcheck_dense_mt(tabs, ch, threads=2)
        b = po.sumcheck_dense_mt(tabs, ch, threads=2, avx512=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
