"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Integer / field work: equality is exact (canonical uint64 limbs); no tolerance anywhere.
"""
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
    from ceno_amd import prover as pv

    pv.plib()
    return pv


def tup(a):
    return int(a[0]), int(a[1])


# ------------------------------------------------------------------------------------------
# MLE primitives
# ------------------------------------------------------------------------------------------
def test_upload_download_roundtrip_and_views(dev):
    t = po.rand_ext(64, 1)
    m = dev.upload(t)
    assert np.array_equal(m.download(), t)
    b = po.rand_base(32, 2)
    mb = dev.upload(b)
    assert np.array_equal(mb.download(), b) and not mb.is_ext and mb.num_vars == 5
    v = m.view_chunk(4, 3)
    assert np.array_equal(v.download(), t[48:64])


def test_synthetic_fill_matches_oracle_generator(dev):
    for nv, is_ext, seed, off in [(0, True, 7, 0), (5, True, 0xCE10, 0), (7, False, 0xCE11, 13), (12, True, 3, 1 << 20)]:
        m = dev.synthetic(nv, is_ext, seed, off)
        n_words = (1 << nv) * (2 if is_ext else 1)
        exp = po.fill_splitmix(n_words, seed, off)
        assert np.array_equal(m.download().reshape(-1), exp)


@pytest.mark.parametrize("nv", [0, 1, 2, 5, 9, 14])
def test_eq_build(dev, nv):
    pt = po.rand_ext(nv, 40 + nv) if nv else np.zeros((0, 2), dtype=np.uint64)
    got = dev.eq_build(pt).download()
    assert np.array_equal(got, po.build_eq(pt))
    sc = (123456789, 987654321)
    got = dev.eq_build(pt, scalar=sc).download()
    exp = po.build_eq(pt)
    for i in (0, (1 << nv) - 1, (1 << nv) // 3):
        assert tup(got[i]) == po.e2_mul(tup(exp[i]), sc)


def test_selectors(dev):
    from ceno_amd.api import Device  # noqa: F401

    nv = 7
    pt = po.rand_ext(nv, 77)
    for off, n in [(0, 128), (0, 77), (5, 100), (127, 1), (0, 0)]:
        got = dev.selector_build(po.SEL_PREFIX, pt, off, n).download()
        assert np.array_equal(got, po.selector_compute(po.SEL_PREFIX, pt, off, n))
    for n in (1, 7, 16):
        idx = [0, 3, 4, 7]
        got = dev.selector_build(po.SEL_ORDERED_SPARSE, pt, 0, n, idx, 3).download()
        assert np.array_equal(got, po.selector_compute(po.SEL_ORDERED_SPARSE, pt, 0, n, idx, 3))
    for n in (1, 2, 5, 64, 100, 128):
        got = dev.selector_build(po.SEL_QUARK_LT, pt, 0, n).download()
        assert np.array_equal(got, po.selector_compute(po.SEL_QUARK_LT, pt, 0, n))
    assert np.array_equal(dev.selector_build(po.SEL_WHOLE, pt).download(), po.build_eq(pt))
    from ceno_amd import CenoHipError

    with pytest.raises(CenoHipError):
        dev.selector_build(po.SEL_PREFIX, pt, 100, 100)  # end > 2^nv (selector.rs:144-150)


@pytest.mark.parametrize("nv,is_ext", [(0, True), (1, True), (6, True), (6, False), (13, True), (13, False)])
def test_evaluate_and_fix_variables(dev, nv, is_ext):
    t = po.rand_ext(1 << nv, 5) if is_ext else po.rand_base(1 << nv, 5)
    pt = po.rand_ext(nv, 6) if nv else np.zeros((0, 2), dtype=np.uint64)
    m = dev.upload(t)
    assert m.evaluate(pt) == po.mle_evaluate(t, pt)
    for k in range(1, min(nv, 4) + 1):
        got = m.fix_variables(pt[:k]).download()
        exp = t
        for j in range(k):
            exp = po.mle_fix_variable(exp, tup(pt[j]))
        assert np.array_equal(got, exp)


def test_evaluate_prefix_batch_matches_single_evaluations(dev):
    """base-field tables of 2^0 .. 2^17 rows at prefixes of one point, one pass (the final evaluations of the columns a batched main
    sumcheck read only linearly): every value equals the oracle's MultilinearExtension::evaluate and the single-table kernel's"""
    nvs = [0, 1, 3, 9, 10, 10, 11, 13, 16, 17, 5]
    pt = po.rand_ext(17, 4242)
    tabs = [po.rand_base(1 << nv, 300 + k) for k, nv in enumerate(nvs)]
    mles = [dev.upload(t) for t in tabs]
    got = dev.evaluate_prefix_batch(mles, pt)
    for k, nv in enumerate(nvs):
        exp = po.mle_evaluate(tabs[k], pt[:nv]) if nv else (int(tabs[k][0]), 0)
        assert tup(got[k]) == exp, (k, nv)
        if nv:
            assert mles[k].evaluate(pt[:nv]) == exp
    assert dev.evaluate_prefix_batch([], pt).shape == (0, 2)
    with pytest.raises(Exception):
        dev.evaluate_prefix_batch([dev.upload(po.rand_base(1 << 18, 1))], pt)  # more variables than the point has


def test_lincomb_base_batch_matches_integer_arithmetic(dev):
    """sum_j c_j col_j over base-field columns with extension-field coefficients = two base-field tables (c0 part, c1 part); groups of
    1 .. 9 columns and 2^0 .. 2^12 rows in one launch, coefficients at the edges of the field"""
    shapes = [(0, 1), (0, 5), (1, 3), (5, 4), (9, 7), (10, 9), (12, 2), (11, 8)]
    groups, coeffs, exp = [], [], []
    for g, (nv, w) in enumerate(shapes):
        tabs = [po.rand_base(1 << nv, 900 + 17 * g + j) for j in range(w)]
        c = po.rand_ext(w, 70 + g)
        c[0] = (P - 1, P - 1)
        if w > 1:
            c[1] = (0, 1)
        e0 = [sum(int(c[j][0]) * int(tabs[j][x]) for j in range(w)) % P for x in range(1 << nv)]
        e1 = [sum(int(c[j][1]) * int(tabs[j][x]) for j in range(w)) % P for x in range(1 << nv)]
        groups.append([dev.upload(t) for t in tabs])
        coeffs.append(c)
        exp.append((np.array(e0, dtype=np.uint64), np.array(e1, dtype=np.uint64)))
    outs = dev.lincomb_base_batch(groups, coeffs)
    for g, (a, b) in enumerate(outs):
        assert not a.is_ext and not b.is_ext and a.num_vars == shapes[g][0]
        assert np.array_equal(a.download(), exp[g][0]) and np.array_equal(b.download(), exp[g][1]), g
    with pytest.raises(Exception):
        dev.lincomb_base_batch([[groups[0][0], groups[3][0]]], [po.rand_ext(2, 1)])  # two sizes in one group


# ------------------------------------------------------------------------------------------
# sumcheck: every round message, every challenge, every final evaluation
# ------------------------------------------------------------------------------------------
def _run_both(dev, prover, tables, coeffs, terms, nv, d, groups=None, seed=1):
    mles = [dev.upload(t) for t in tables]
    msgs, chal, fin = prover.sumcheck_prove(dev, mles, coeffs, terms, nv, d, prover.Transcript.stub(seed), groups=groups)
    # oracle: groups expanded to full terms
    full_terms = [list(t) for t in terms]
    if groups:
        for common, members in groups:
            for t in members:
                full_terms[t] = list(common) + full_terms[t]
    omsgs, ochal, ofin = po.sumcheck_prove(tables, coeffs, full_terms, nv, d, po.StubTranscript(seed))
    assert np.array_equal(msgs, omsgs)
    assert np.array_equal(chal, ochal)
    assert np.array_equal(fin, ofin)
    return msgs, chal, fin


@pytest.mark.parametrize("nv", [1, 2, 3, 8, 12, 16])
@pytest.mark.parametrize("k", [1, 2, 3, 4])
def test_sumcheck_dense_ext(dev, prover, nv, k):
    tables = [po.rand_ext(1 << nv, 10 * k + j) for j in range(k)]
    _run_both(dev, prover, tables, po.rand_ext(1, 99), [list(range(k))], nv, k)


@pytest.mark.parametrize("nv", [1, 4, 11])
@pytest.mark.parametrize("k", [2, 3])
def test_sumcheck_dense_base_start(dev, prover, nv, k):
    tables = [po.rand_base(1 << nv, 50 + j) for j in range(k)]
    _run_both(dev, prover, tables, po.ext([1]), [list(range(k))], nv, k)


def test_sumcheck_generic_terms_and_degrees(dev, prover):
    nv = 9
    tables = [po.rand_ext(1 << nv, 1), po.rand_base(1 << nv, 2), po.rand_ext(1 << nv, 3), po.rand_base(1 << nv, 4),
              po.rand_ext(1 << nv, 5)]
    terms = [[0, 1, 2], [1, 3], [4], [0, 0, 4, 4], [2, 3, 4, 1, 0]]
    coeffs = po.rand_ext(len(terms), 6)
    _run_both(dev, prover, tables, coeffs, terms, nv, 5)
    # max_degree larger than every term
    _run_both(dev, prover, tables, coeffs[:3], terms[:3], nv, 4)


def test_sumcheck_mixed_sizes_frontload(dev, prover):
    big = [po.rand_base(1 << 10, 1), po.rand_ext(1 << 10, 2), po.rand_ext(1 << 10, 3)]
    mid = [po.rand_ext(1 << 6, 4), po.rand_base(1 << 6, 5)]
    small = [po.rand_ext(4, 6)]
    tiny = [po.rand_ext(1, 7)]  # zero variables
    tables = big + mid + small + tiny
    terms = [[0, 1, 2], [1, 2], [3, 4], [3, 3, 4], [5], [5, 5], [6, 6]]
    coeffs = po.rand_ext(len(terms), 8)
    msgs, chal, fin = _run_both(dev, prover, tables, coeffs, terms, 10, 3)
    # restated verifier accepts, final check uses the front-load rule
    claim = (0, 0)
    for c, t in zip(coeffs, terms):
        nvt = int(tables[t[0]].shape[0]).bit_length() - 1
        s = (0, 0)
        for x in range(1 << nvt):
            v = (1, 0)
            for j in t:
                e = tables[j][x]
                v = po.e2_mul(v, tup(e) if tables[j].ndim == 2 else (int(e), 0))
            s = po.e2_add(s, v)
        claim = po.e2_add(claim, po.e2_mul(tup(c), s))
    point, expected = po.sumcheck_verify(claim, msgs, po.StubTranscript(1))
    nvs = [10, 10, 10, 6, 6, 2, 0]
    assert po.sumcheck_expected_from_evals(nvs, coeffs, terms, 10, chal, fin) == expected


def test_sumcheck_common_factor_groups(dev, prover):
    nv = 8
    tables = [po.rand_ext(1 << nv, 20 + j) for j in range(7)]
    # eq-like common factor 0 over terms 0..2, common factors (1,2) over term 3, term 4 ungrouped
    terms = [[3, 4], [5], [4, 6], [6], [1, 5, 6]]
    groups = [([0], [0, 1, 2]), ([1, 2], [3])]
    coeffs = po.rand_ext(len(terms), 30)
    _run_both(dev, prover, tables, coeffs, terms, nv, 3, groups=groups)


def test_sumcheck_error_behaviour(dev, prover):
    from ceno_amd import CenoHipError, Sumcheck

    a, b = dev.upload(po.rand_ext(4, 1)), dev.upload(po.rand_ext(8, 2))
    with pytest.raises(CenoHipError):
        Sumcheck(dev, [a, b], po.ext([1]), [[0, 1]], 3, 2)  # mixed sizes inside one term
    with pytest.raises(CenoHipError):
        Sumcheck(dev, [a], po.ext([1]), [[]], 2, 2)  # empty product
    with pytest.raises(CenoHipError):
        Sumcheck(dev, [a], po.ext([1]), [[0, 0, 0]], 2, 2)  # degree > max_degree
    with pytest.raises(CenoHipError):
        Sumcheck(dev, [b], po.ext([1]), [[0]], 2, 2)  # mle larger than max_num_vars
    sc = Sumcheck(dev, [a], po.ext([1]), [[0]], 2, 1)
    with pytest.raises(CenoHipError):
        sc.round((1, 2))  # round 0 takes no challenge
    sc.round()
    with pytest.raises(CenoHipError):
        sc.round()  # round 1 needs a challenge
    sc.round((3, 4))
    with pytest.raises(CenoHipError):
        sc.round((5, 6))  # all rounds done
    fin = sc.finish((5, 6))
    assert tup(fin[0]) == po.mle_evaluate(a.download(), po.ext([(3, 4), (5, 6)]))
    with pytest.raises(CenoHipError):
        sc.finish((5, 6))


def test_sumcheck_abandoned_midway_does_not_hang(dev, prover):
    # pipelined mode has every round kernel queued behind a host mailbox: freeing the handle early must
    # release them (abort flag) instead of leaving kernels spinning
    import time

    from ceno_amd import Sumcheck

    mles = [dev.synthetic(14, True, 70 + j) for j in range(3)]
    for rounds_before_free in (1, 3):
        sc = Sumcheck(dev, mles, po.ext([1]), [[0, 1, 2]], 14, 3)
        sc.set_pipelined(True)
        ch = None
        for _ in range(rounds_before_free):
            sc.round(ch)
            ch = (5, 6)
        t0 = time.time()
        sc.free()
        dev.sync()
        assert time.time() - t0 < 2.0
    # a generic plan small enough to run inside the persistent tail kernel, abandoned after ALL its rounds but before finish:
    # the kernel is then waiting for the challenge it takes the final evaluations at, and the free must release it
    small = [dev.synthetic(6, True, 90 + j) for j in range(3)]
    sc = Sumcheck(dev, small, po.ext([1, 2]), [[0, 1], [1, 2]], 6, 2)
    sc.set_pipelined(True)
    ch = None
    for _ in range(6):
        sc.round(ch)
        ch = (7, 8)
    t0 = time.time()
    sc.free()
    dev.sync()
    assert time.time() - t0 < 2.0
    # the device is still healthy and a fresh sumcheck on the same tables is still bit-exact
    tables = [m.download() for m in mles]
    msgs, chal, fin = prover.sumcheck_prove(dev, mles, po.ext([1]), [[0, 1, 2]], 14, 3, prover.Transcript.stub(4))
    omsgs, ochal, ofin = po.sumcheck_prove(tables, po.ext([1]), [[0, 1, 2]], 14, 3, po.StubTranscript(4))
    assert np.array_equal(msgs, omsgs) and np.array_equal(fin, ofin)


def test_sumcheck_inputs_are_not_modified(dev, prover):
    t = [po.rand_ext(1 << 10, j) for j in range(3)]
    mles = [dev.upload(x) for x in t]
    prover.sumcheck_prove(dev, mles, po.ext([1]), [[0, 1, 2]], 10, 3, prover.Transcript.stub(3))
    for m, x in zip(mles, t):
        assert np.array_equal(m.download(), x)


# ------------------------------------------------------------------------------------------
# full-size property checks (BASELINE config #2 size: 3 x 2^22 ext)
# ------------------------------------------------------------------------------------------
def test_sumcheck_nv22_properties(dev, prover):
    nv, k = 22, 3
    mles = [dev.synthetic(nv, True, 0xCE10 + j) for j in range(k)]
    msgs, chal, fin = prover.sumcheck_prove(dev, mles, po.ext([1]), [[0, 1, 2]], nv, k, prover.Transcript.stub(0xF5))
    # (1) final evaluations equal independent MLE evaluations at the challenge point (device evaluate kernel
    #     shares no code with the sumcheck kernels)
    for j in range(k):
        assert mles[j].evaluate(chal) == tup(fin[j])
    # (2) the transcript of messages satisfies the restated verifier and the final product check
    expected = (1, 0)
    for j in range(k):
        expected = po.e2_mul(expected, tup(fin[j]))
    claim = po.recover_claim_from_final(expected, msgs, chal)
    point, exp2 = po.sumcheck_verify(claim, msgs, po.StubTranscript(0xF5))
    assert np.array_equal(point, chal) and exp2 == expected
    # (3) folding 8 variables on the device then finishing with the oracle reproduces messages 8..21
    folded = [m.fix_variables(chal[:8]).download() for m in mles]
    omsgs, _ = po.sumcheck_dense_mt(folded, chal[8:], threads=4)
    assert np.array_equal(omsgs, msgs[8:])


def test_sumcheck_nv26_bench_workload_verifies(dev, prover):
    """the bench.py workload itself (3 ext MLEs x 2^26, the size the metric is quoted on), checked through
    size-independent properties: (1) the claimed sum comes from an independent kernel path (pointwise product by
    wit_infer, then sum = 2^n * MLE(1/2, ..., 1/2) by the evaluate kernel), (2) the restated verifier accepts the 26
    messages for that claim and lands on the product of the final evaluations, (3) the final evaluations equal
    independent MLE evaluations, (4) the last 12 rounds replay bit for bit on the oracle from device-folded tables."""
    nv, k = 26, 3
    mles = [dev.synthetic(nv, True, 0xCE10 + j) for j in range(k)]
    msgs, chal, fin = prover.sumcheck_prove(dev, mles, po.ext([1]), [[0, 1, 2]], nv, k, prover.Transcript.stub(0xF5))
    prod = dev.wit_infer(mles, po.ext([1]), [[0, 1, 2]], [[0]], nv)[0]
    half = np.tile(np.array([[(P + 1) // 2, 0]], dtype=np.uint64), (nv, 1))
    s_half = prod.evaluate(half)
    claim = po.e2_mul(s_half, (pow(2, nv, P), 0))
    prod.free()
    point, expected = po.sumcheck_verify(claim, msgs, po.StubTranscript(0xF5))
    assert np.array_equal(point, chal)
    want = (1, 0)
    for j in range(k):
        assert mles[j].evaluate(chal) == tup(fin[j])
        want = po.e2_mul(want, tup(fin[j]))
    assert expected == want
    folded = [m.fix_variables(chal[:14]).download() for m in mles]
    omsgs, _ = po.sumcheck_dense_mt(folded, chal[14:], threads=4)
    assert np.array_equal(omsgs, msgs[14:])


# ------------------------------------------------------------------------------------------
# tower
# ------------------------------------------------------------------------------------------
def test_tower_build_matches_oracle(dev, prover):
    for k, rows in [(4, 8), (3, 8), (1, 4), (5, 16), (2, 2)]:
        recs = [po.rand_ext(rows, 100 + j) for j in range(k)]
        limbs = po.interleaving_mles_to_mles(recs, rows, 2, (1, 0))
        nv = int(limbs[0].shape[0]).bit_length()  # limb vars + 1
        layers = po.infer_tower_product_witness(nv, limbs)
        t = prover.Tower.build_prod(dev, [dev.upload(r) for r in recs], rows, (1, 0))
        assert t.num_vars == nv and t.num_limbs == 2
        for l in range(nv):
            for s in range(2):
                assert np.array_equal(t.layer(l, s), layers[l][s])
        assert np.array_equal(t.out_evals(), np.stack([layers[0][0][0], layers[0][1][0]]))


def test_tower_interleave_golden_vectors(dev, prover):
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tower_witness.json")))
    for case in gold["interleaving_mles_to_mles"]:
        recs = [dev.upload(po.ext(m)) for m in case["mles"]]
        t = prover.Tower.build_prod(dev, recs, case["num_instances"], (case["default"], 0))
        last = t.num_vars - 1
        for s in range(2):
            got = t.layer(last, s)
            assert [int(x[0]) for x in got] == case["expected"][s] and all(int(x[1]) == 0 for x in got)
    case = gold["infer_tower_logup_witness"][0]
    q = [dev.upload(po.ext(x)) for x in case["q"]]
    t = prover.Tower.from_last_layer(dev, [None, None, q[0], q[1]])
    assert t.num_vars == case["num_layers"]
    for l, exp_layer in enumerate(case["layers"]):
        for s in range(4):
            got = t.layer(l, s)
            assert [int(x[0]) for x in got] == exp_layer[s]


def test_tower_logup_build_matches_oracle(dev, prover):
    k, rows = 3, 16
    alpha = (777, 888)
    qs = [po.rand_ext(rows, 200 + j) for j in range(k)]
    ps = [po.rand_ext(rows, 300 + j) for j in range(k)]
    for with_p in (True, False):
        ql = po.interleaving_mles_to_mles(qs, rows, 2, alpha)
        pl = po.interleaving_mles_to_mles(ps, rows, 2, alpha) if with_p else None
        layers = po.infer_tower_logup_witness(pl, ql)
        t = prover.Tower.build_logup(dev, [dev.upload(p) for p in ps] if with_p else None, [dev.upload(q) for q in qs], rows, alpha)
        assert t.num_vars == len(layers) and t.num_limbs == 4
        for l in range(len(layers)):
            for s in range(4):
                assert np.array_equal(t.layer(l, s), layers[l][s])


@pytest.mark.parametrize("leaf_log", [1, 2, 5, 9])
def test_tower_proof_matches_oracle_and_verifies(dev, prover, leaf_log):
    nv = leaf_log + 1
    last = [po.rand_ext(1 << leaf_log, 1000 + leaf_log), po.rand_ext(1 << leaf_log, 2000 + leaf_log)]
    spec = po.infer_tower_product_witness(nv, last)
    oproof = po.tower_prove([spec], [], po.StubTranscript(5))
    t = prover.Tower.from_last_layer(dev, [dev.upload(x) for x in last])
    proof = prover.tower_create_proof(dev, [t], [], prover.Transcript.stub(5))
    assert np.array_equal(proof.msgs, oproof.msgs)
    assert np.array_equal(proof.prod_evals, oproof.prod_evals)
    assert np.array_equal(proof.point[:nv], oproof.point[:nv])


def test_tower_top_block_layout(dev, prover):
    """ceno_hip_tower_download_top: layers 0..n-1 in one copy, layer l limb b at element offset n_limbs (2^l - 1) + b 2^l —
    equal to the per-layer views (ceno_hip_tower_layer), for product and LogUp towers, shorter and taller than the block"""
    import ctypes as C

    for n_limbs, leaf_log in ((2, 13), (4, 12), (2, 4)):
        last = [po.rand_ext(1 << leaf_log, 700 + 10 * n_limbs + j) for j in range(n_limbs)]
        t = prover.Tower.from_last_layer(dev, [dev.upload(x) for x in last])
        top = dev.L.ceno_hip_tower_top_layers(t.h)
        assert top == min(t.num_vars, 11) and t.num_limbs == n_limbs
        for n_layers in sorted({1, min(3, top), top}):
            buf = np.zeros((n_limbs * ((1 << n_layers) - 1), 2), dtype=np.uint64)
            dev.check(dev.L.ceno_hip_tower_download_top(dev.h, t.h, n_layers, buf.ctypes.data_as(C.POINTER(C.c_uint64)), None))
            for l in range(n_layers):
                for b in range(n_limbs):
                    off = n_limbs * ((1 << l) - 1) + (b << l)
                    assert np.array_equal(buf[off: off + (1 << l)], t.layer(l, b)), (n_limbs, leaf_log, l, b)
        assert dev.L.ceno_hip_tower_download_top(dev.h, t.h, top + 1, buf.ctypes.data_as(C.POINTER(C.c_uint64)), None) != 0
        t.free()


@pytest.mark.parametrize("host_layers", [0, 3, 8, 10])
def test_tower_proof_is_the_same_wherever_the_small_layers_are_proved(dev, prover, monkeypatch, host_layers):
    """layers 1..CENO_TOWER_HOST_LAYERS of a tower proof run on the host from one copy of each tower's top block
    (ceno_hip_tower_download_top), the rest on the device: product and LogUp specs of different heights, every split point,
    bit-identical to the oracle's CpuTowerProver::create_proof (scheme/cpu/mod.rs:366-541)"""
    monkeypatch.setenv("CENO_TOWER_HOST_LAYERS", str(host_layers))
    lasts = [[po.rand_ext(1 << 11, 31), po.rand_ext(1 << 11, 32)], [po.rand_ext(1 << 6, 33), po.rand_ext(1 << 6, 34)]]
    lk = [po.rand_ext(1 << 9, 40 + j) for j in range(4)]
    pspecs = [po.infer_tower_product_witness(12, lasts[0]), po.infer_tower_product_witness(7, lasts[1])]
    lspec = po.infer_tower_logup_witness(lk[:2], lk[2:])
    oproof = po.tower_prove(pspecs, [lspec], po.StubTranscript(9))
    pt = [prover.Tower.from_last_layer(dev, [dev.upload(x) for x in l]) for l in lasts]
    lt = prover.Tower.from_last_layer(dev, [dev.upload(x) for x in lk])
    proof = prover.tower_create_proof(dev, pt, [lt], prover.Transcript.stub(9))
    assert np.array_equal(proof.msgs, oproof.msgs)
    assert np.array_equal(proof.prod_evals, oproof.prod_evals)
    assert np.array_equal(proof.logup_evals, oproof.logup_evals)
    assert np.array_equal(proof.point[:12], oproof.point[:12])


def test_tower_relation_mixed_specs(dev, prover):
    # read tower (6 layers), write tower (4 layers), two lookup towers (with and without numerators)
    def prod_last(nv, seed):
        return [po.rand_ext(1 << (nv - 1), seed), po.rand_ext(1 << (nv - 1), seed + 1)]

    pl = [prod_last(6, 1), prod_last(4, 3), prod_last(2, 5)]
    ql = [(prod_last(5, 7), prod_last(5, 9)), (None, prod_last(6, 11))]
    specs_p = [po.infer_tower_product_witness(len(l[0]).bit_length(), l) for l in pl]
    specs_l = [po.infer_tower_logup_witness(p, q) for p, q in ql]
    towers_p = [prover.Tower.from_last_layer(dev, [dev.upload(x) for x in l]) for l in pl]
    towers_l = [prover.Tower.from_last_layer(dev, [dev.upload(p[0]) if p else None, dev.upload(p[1]) if p else None,
                                                   dev.upload(q[0]), dev.upload(q[1])]) for p, q in ql]
    out_evals, proof = prover.prove_tower_relation(dev, towers_p, towers_l, prover.Transcript.stub(8))
    # oracle: same script (append out evals, then create_proof)
    tr = po.StubTranscript(8)
    po_ev = np.stack([np.stack([s[0][0][0], s[0][1][0]]) for s in specs_p])
    lo_ev = np.stack([np.stack([s[0][k][0] for k in range(4)]) for s in specs_l])
    for e in list(po_ev.reshape(-1, 2)) + list(lo_ev.reshape(-1, 2)):
        tr.append_ext(tup(e))
    oproof = po.tower_prove(specs_p, specs_l, tr)
    assert np.array_equal(out_evals, np.concatenate([po_ev.reshape(-1, 2), lo_ev.reshape(-1, 2)]))
    assert np.array_equal(proof.msgs, oproof.msgs)
    assert np.array_equal(proof.prod_evals, oproof.prod_evals)
    assert np.array_equal(proof.logup_evals, oproof.logup_evals)
    assert np.array_equal(proof.point[:6], oproof.point[:6])
    # and the restated verifier accepts the GPU proof
    vt = po.StubTranscript(8)
    for e in list(po_ev.reshape(-1, 2)) + list(lo_ev.reshape(-1, 2)):
        vt.append_ext(tup(e))
    oproof.msgs[:] = proof.msgs
    rc, pt, pc, lp, lq = po.tower_verify(po_ev, lo_ev, [6, 4, 2, 5, 6], oproof, vt)
    assert rc == 0


# ------------------------------------------------------------------------------------------
# sharded driver, C++ / RCCL path at world size 1 (the >1 logic is covered under gloo in test_dist_cpu.py)
# ------------------------------------------------------------------------------------------
def test_native_dist_path_world1_matches_oracle(dev, prover):
    nv, k = 12, 3
    tables = [po.rand_ext(1 << nv, 0xCE10 + j) for j in range(k)]
    mles = [dev.upload(t) for t in tables]
    comm = prover.RcclComm(1, 0, None)
    stream = dev.stream_create()
    msgs, chal, fin = prover.dist_sumcheck_prove(dev, comm, mles, po.ext([1]), [list(range(k))], nv, k, prover.Transcript.stub(0xF5), stream)
    omsgs, ochal, ofin = po.sumcheck_prove(tables, po.ext([1]), [list(range(k))], nv, k, po.StubTranscript(0xF5))
    assert np.array_equal(msgs, omsgs) and np.array_equal(chal, ochal) and np.array_equal(fin, ofin)
    # the python/torch engine gives the same proof (this is what bench.py cross-checks at N > 1)
    from ceno_amd import dist as cdist

    eng = cdist.HipShardEngine(dev, mles)
    m2, c2, f2 = cdist.sharded_sumcheck_prove(eng, nv, k, prover.Transcript.stub(0xF5), dist=None, world=1, rank=0)
    assert np.array_equal(m2, omsgs) and np.array_equal(f2, ofin)
    # the early-gather branch (taken at world > 1 once shards are small) run at world size 1
    import os

    for nv2, seed in ((15, 3), (9, 4)):
        t2 = [po.rand_ext(1 << nv2, 40 + j) for j in range(k)]
        m2_ = [dev.upload(t) for t in t2]
        os.environ["CENO_DIST_FORCE_GATHER"] = "1"
        try:
            got = prover.dist_sumcheck_prove(dev, comm, m2_, po.ext([1]), [list(range(k))], nv2, k, prover.Transcript.stub(seed), stream)
        finally:
            del os.environ["CENO_DIST_FORCE_GATHER"]
        exp = po.sumcheck_prove(t2, po.ext([1]), [list(range(k))], nv2, k, po.StubTranscript(seed))
        for g, e in zip(got, exp):
            assert np.array_equal(g, e)
    comm.close()
    dev.stream_destroy(stream)


# ------------------------------------------------------------------------------------------
# rotation argument (a11)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log2,sub", [(5, 23), (5, 32), (6, 24), (6, 1)])
def test_rotation_kernels_and_proof(dev, prover, log2, sub):
    nv = log2 + 4
    src = po.rand_base(1 << nv, 5 + log2)
    other = po.rand_base(1 << nv, 6 + log2)
    d_src, d_other = dev.upload(src), dev.upload(other)
    rot = dev.rotation_next_base_mle(d_src, log2)
    assert np.array_equal(rot.download(), po.rotation_next_base_mle(src, log2))
    rt = po.rand_ext(nv, 7)
    sel = dev.rotation_selector_build(rt, sub, log2)
    assert np.array_equal(sel.download(), po.rotation_selector(po.build_eq(rt), sub, log2))
    tgt = po.rotation_next_base_mle(src, log2)
    wit = [src, tgt, other]
    d_wit = [d_src, dev.upload(tgt), d_other]
    pairs = [(0, 1), (2, 0)]  # the second pair does not satisfy the relation: the proof is still a function of the inputs
    got = prover.prove_rotation(dev, d_wit, pairs, sub, log2, rt, prover.Transcript.stub(3))
    exp = po.prove_rotation(wit, pairs, sub, log2, rt, po.StubTranscript(3))
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    from ceno_amd import CenoHipError

    with pytest.raises(CenoHipError):
        dev.rotation_next_base_mle(dev.upload(po.rand_ext(64, 1)), 5)  # ext source is rejected like the reference


# ------------------------------------------------------------------------------------------
# batched main-constraint sumcheck (a12): three synthetic chips of different sizes in ONE sumcheck
# ------------------------------------------------------------------------------------------
def test_batched_main_constraints_matches_oracle(dev, prover):
    gch = [(11, 22), (33, 44)]
    chips = []
    rng_seed = 500
    for c, (nv, n_w, n_f) in enumerate([(10, 5, 1), (7, 3, 0), (5, 2, 1)]):
        wit = [po.rand_base(1 << nv, rng_seed + 10 * c + j) for j in range(n_w)]
        fixed = [po.rand_base(1 << nv, rng_seed + 10 * c + 7 + j) for j in range(n_f)]
        point = po.rand_ext(nv, 900 + c)
        n_inst = (1 << nv) - 3 - c
        # one structural witness = the Prefix selector of the chip
        sel = (po.SEL_PREFIX, 0, n_inst, 0, (), 0, point)
        s_id = n_w + n_f
        n_exprs = 2
        # sel * (alpha_0 * w0*w1 + alpha_1 * (w1 - beta*w2 ...)): a few monomial terms with polynomial scalars
        terms = [[s_id, 0, 1], [s_id, 1], [s_id, min(2, n_w - 1), 0, 0][: 3 + (nv > 6)]]
        scalars = [[((1, 0), [2])], [((5, 0), [3, 0]), ((7, 1), [1, 1])], [((2, 3), [3]), ((P - 1, 0), [2, 0, 1])]]
        chips.append(dict(nv=nv, wit=wit, fixed=fixed, sel=sel, n_exprs=n_exprs, terms=terms, scalars=scalars,
                          max_degree=max(len(t) for t in terms)))
    jobs = []
    for ch in chips:
        mles = [dev.upload(t) for t in ch["wit"] + ch["fixed"]] + [None]
        jobs.append(dict(num_vars=ch["nv"], mles=mles, n_witin=len(ch["wit"]), n_fixed=len(ch["fixed"]), n_structural=1,
                         selectors=[ch["sel"]], n_exprs=ch["n_exprs"], max_degree=ch["max_degree"], terms=ch["terms"],
                         scalars=ch["scalars"]))
    claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(77))
    # ---- expected, assembled from oracle pieces exactly as scheme/cpu/mod.rs:1052-1390 does ----
    tr = po.StubTranscript(77)
    tr.append_label(b"combine subset evals")
    alpha = tr.sample_ext()
    total_exprs = sum(ch["n_exprs"] for ch in chips)
    pows = [(1, 0)]
    for _ in range(total_exprs - 1):
        pows.append(po.e2_mul(pows[-1], alpha))
    tables, coeffs, terms, nvs = [], [], [], []
    a0 = 0
    for ch in chips:
        start = len(tables)
        sel_tab = po.selector_compute(ch["sel"][0], ch["sel"][6], ch["sel"][1], ch["sel"][2])
        tables += ch["wit"] + ch["fixed"] + [sel_tab]
        nvs += [ch["nv"]] * (len(ch["wit"]) + len(ch["fixed"]) + 1)
        chal = gch + pows[a0: a0 + ch["n_exprs"]]
        for t, monos in zip(ch["terms"], ch["scalars"]):
            sc = (0, 0)
            for coeff, ids in monos:
                v = coeff
                for i in ids:
                    v = po.e2_mul(v, chal[i])
                sc = po.e2_add(sc, v)
            coeffs.append(sc)
            terms.append([start + j for j in t])
        a0 += ch["n_exprs"]
    max_nv, max_deg = 10, max(ch["max_degree"] for ch in chips)
    omsgs, ochal, ofin = po.sumcheck_prove(tables, po.ext(coeffs), terms, max_nv, max_deg, tr)
    assert np.array_equal(msgs, omsgs) and np.array_equal(rt, ochal) and np.array_equal(evals, ofin)
    final_claim = po.sumcheck_expected_from_evals(nvs, po.ext(coeffs), terms, max_nv, ochal, ofin)
    assert claimed == po.recover_claim_from_final(final_claim, omsgs, ochal)
    # and the restated verifier accepts: claimed sum -> expected evaluation == front-load evaluation
    vt = po.StubTranscript(77)
    vt.append_label(b"combine subset evals")
    vt.sample_ext()
    _, expected = po.sumcheck_verify(claimed, msgs, vt)
    assert expected == final_claim


# ------------------------------------------------------------------------------------------
# limits and ragged plans
# ------------------------------------------------------------------------------------------
def test_sumcheck_limits_degree8_many_mles_and_unused_tables(dev, prover):
    from ceno_amd import CenoHipError, Sumcheck

    nv = 7
    tables = [po.rand_ext(1 << nv, 300 + j) if j % 3 else po.rand_base(1 << nv, 300 + j) for j in range(40)]
    # 60 terms of ragged degree 1..8 over 40 tables (two-kernel generic path: LDS staging does not fit),
    # table 39 is referenced by no term (it must still be folded and evaluated)
    terms = [[(7 * t + 3 * k) % 39 for k in range(1 + t % 8)] for t in range(60)]
    coeffs = po.rand_ext(len(terms), 11)
    _run_both(dev, prover, tables, coeffs, terms, nv, 8)
    # a plan that fits the fused LDS-staged kernel (<= 7 tables), degree 8
    _run_both(dev, prover, tables[:6], coeffs[:5], [[0, 1, 2, 3, 4, 5, 0, 1], [2], [3, 3], [4, 5, 1], [0, 0, 0, 0, 0]], nv, 8)
    with pytest.raises(CenoHipError):
        Sumcheck(dev, [dev.upload(tables[1])], po.ext([1]), [[0] * 9], nv, 9)  # degree 9 is outside the supported range


def test_sumcheck_dense_k4_base_and_single_variable(dev, prover):
    tables = [po.rand_base(1 << 9, 400 + j) for j in range(4)]
    _run_both(dev, prover, tables, po.rand_ext(1, 5), [[0, 1, 2, 3]], 9, 4)
    one = [po.rand_ext(2, 410 + j) for j in range(2)]
    _run_both(dev, prover, one, po.ext([3]), [[0, 1]], 1, 2)


def test_large_eq_and_evaluate_consistency(dev):
    # 2^24-entry eq table (256 MB): eq(r, r') evaluated through the table equals the closed form
    nv = 24
    r, rp = po.rand_ext(nv, 1), po.rand_ext(nv, 2)
    eq = dev.eq_build(r)
    assert eq.evaluate(rp) == po.eq_eval(r, rp)
    eq.free()


def test_sharded_batched_sumcheck_hip_engine_virtual_ranks(dev):
    """SURVEY §8(e) mixed-size sharding with the REAL device engine: `world` virtual ranks = threads of this process,
    each with its own stream and sumcheck handles, exchanging partials through an in-process all-gather.  The
    result must equal the single-prover proof of the unsharded plan (oracle)."""
    import threading

    from ceno_amd import dist as cdist
    from ceno_amd import prover
    from tests.dist_worker import batched_case

    world, n_total = 4, 9
    log_w = 2

    class ThreadDist:
        def __init__(self, world):
            self.world, self.slots, self.bar = world, [None] * world, threading.Barrier(world)

        def for_rank(self, rank):
            outer = self

            class D:
                def get_backend(self):
                    return "threads"

                def all_gather(self, outs, t):
                    outer.slots[rank] = t.clone()
                    outer.bar.wait()
                    for g in range(outer.world):
                        outs[g].copy_(outer.slots[g])
                    outer.bar.wait()

            return D()

    td = ThreadDist(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            stream = dev.stream_create()
            classes = []
            for c in batched_case(n_total):
                sharded = c["num_vars"] != 2
                m = 1 << (c["num_vars"] - log_w) if sharded else None
                tabs = [t[rank * m:(rank + 1) * m] for t in c["tables"]] if sharded else c["tables"]
                classes.append(dict(c, tables=tabs, sharded=sharded))
            results[rank] = cdist.sharded_batched_sumcheck_prove(cdist.hip_engine_factory(dev, stream), classes, n_total, 3,
                                                                 prover.Transcript.stub(0xF5), dist=td.for_rank(rank), world=world, rank=rank)
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            td.bar.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    tables, coeffs, terms, off = [], [], [], 0
    for c in batched_case(n_total):
        tables += c["tables"]
        coeffs.append(c["coeffs"])
        terms += [[off + j for j in t] for t in c["terms"]]
        off += len(c["tables"])
    omsgs, ochal, ofin = po.sumcheck_prove(tables, np.concatenate(coeffs), terms, n_total, 3, po.StubTranscript(0xF5))
    for r in range(world):
        msgs, chal, fins = results[r]
        assert np.array_equal(msgs, omsgs)
        assert np.array_equal(chal, ochal)
        assert np.array_equal(np.concatenate(fins), ofin)


def test_virtual_polynomials_builder_matches_oracle(dev):
    """VirtualPolynomialsBuilder mirror: lift (dedup, borrowed / owned), monomial terms, prove"""
    from ceno_amd import prover
    from ceno_amd.api import CenoHipError

    nv = 9
    tabs = [po.rand_ext(1 << nv, 31), po.rand_base(1 << nv, 32), po.rand_ext(1 << (nv - 3), 33), po.rand_ext(1 << (nv - 3), 34)]
    b = prover.VirtualPolynomialsBuilder(dev, nv)
    m0, m1 = dev.upload(tabs[0]), dev.upload(tabs[1])
    i0 = b.lift(m0)
    i1 = b.lift(m1)
    assert b.lift(m0) == i0  # the same MLE lifts to the same expression
    i2 = b.lift(dev.upload(tabs[2]), owned=True)
    i3 = b.lift(dev.upload(tabs[3]), owned=True)
    terms = [((3, 1), [i0, i1, i0]), ((5, 0), [i2, i3]), ((7, 9), [i1])]
    for sc, prod in terms:
        b.add_term(sc, prod)
    with pytest.raises(CenoHipError):
        b.add_term((1, 0), [i0, i2])  # factors of different sizes in one term
    msgs, chal, fin = b.prove(prover.Transcript.stub(0xF5))
    omsgs, ochal, ofin = po.sumcheck_prove(tabs, po.ext([t[0] for t in terms]), [t[1] for t in terms], nv, 3, po.StubTranscript(0xF5))
    assert np.array_equal(msgs, omsgs) and np.array_equal(chal, ochal) and np.array_equal(fin, ofin)
    b.free()


def test_memory_booking_for_a_scheduler():
    """mem_pool booking (ceno_zkvm/src/scheme/scheduler.rs:622-652): bookings count against the pool capacity together
    with live allocations; a refused booking allocates nothing and leaves the total unchanged"""
    from ceno_amd import Device
    from ceno_amd.api import CenoHipError

    d = Device(0, pool_bytes=1 << 26)  # 64 MiB pool
    L = d.L
    assert L.ceno_hip_mem_booked(d.h) == 0
    assert L.ceno_hip_mem_book(d.h, 40 << 20) == 0
    m = d.alloc(20, True)  # 16 MiB live
    assert L.ceno_hip_mem_book(d.h, 16 << 20) != 0  # 40 + 16 + 16 > 64
    assert L.ceno_hip_mem_booked(d.h) == 40 << 20
    assert L.ceno_hip_mem_book(d.h, 8 << 20) == 0
    m.free()
    assert L.ceno_hip_mem_unbook(d.h, 40 << 20) == 0
    assert L.ceno_hip_mem_booked(d.h) == 8 << 20
    assert L.ceno_hip_mem_unbook(d.h, 1 << 40) == 0  # over-unbooking clamps at zero
    assert L.ceno_hip_mem_booked(d.h) == 0
    d.close()


def test_pool_orders_a_reused_block_behind_the_stream_that_used_it_last():
    """A handle may be freed while kernels that read it are still queued; the pool hands the block to ANOTHER stream only
    once the old stream has drained (include/ceno_hip.h "Memory"; reference: the CUDA pool's stream-ordered frees behind
    `get_thread_stream`, gkr_iop/src/gpu/mod.rs:87-154)"""
    from ceno_amd import Device

    d = Device(0)
    s1, s2 = d.stream_create(), d.stream_create()
    nv = 23
    r = po.rand_ext(1, 99)
    for trial in range(3):
        # a queue of work on s1, then the fold that reads `a`; `a` is freed at once and its block re-filled through s2
        filler = [d.synthetic(nv, True, 1000 + k, stream=s1) for k in range(6)]
        a = d.synthetic(nv, True, 7 + trial, stream=s1)
        folded = a.fix_variables(r, stream=s1)
        ptr = a.device_ptr
        a.free()
        b = d.synthetic(nv, True, 5000 + trial, stream=s2)  # same size class: the freed block is the first candidate
        c = d.synthetic(nv, True, 6000 + trial, stream=s1)  # the stream that used it last may take it at once
        assert c.device_ptr == ptr or b.device_ptr == ptr
        d.sync(s1)
        d.sync(s2)
        want = d.synthetic(nv, True, 7 + trial, stream=s1).fix_variables(r, stream=s1)
        d.sync(s1)
        assert np.array_equal(folded.download(s1), want.download(s1)), "the re-filled block was overwritten before its reader ran"
        for m in filler + [b, c, folded, want]:
            m.free()
    d.stream_destroy(s1)
    d.stream_destroy(s2)
    d.close()


def test_lane_scheduler_runs_every_task_once_with_booking(dev, prover):
    """ceno_prover_lanes_run (scheduler.rs:231-336 + booking :622-652): every task runs exactly once on some lane's
    stream, results are bit-exact whatever the interleaving, and estimates are booked / unbooked around each task"""
    import ctypes as C

    L = prover.plib()
    TASK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)

    class LaneTask(C.Structure):
        _fields_ = [("fn", TASK), ("arg", C.c_void_p), ("estimated_bytes", C.c_size_t)]

    L.ceno_prover_lanes_run.restype = C.c_int
    L.ceno_prover_lanes_run.argtypes = [C.c_void_p, C.c_int, C.POINTER(LaneTask), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    n_tasks, nv, k = 7, 11, 3
    tables = [[po.rand_ext(1 << nv, 300 + 10 * t + j) for j in range(k)] for t in range(n_tasks)]
    mles = [[dev.upload(x) for x in tabs] for tabs in tables]
    results, seen_booked = [None] * n_tasks, []

    def body(arg, lane, stream):
        t = int(arg or 0)
        try:
            seen_booked.append(int(dev.L.ceno_hip_mem_booked(dev.h)))
            results[t] = prover.sumcheck_prove(dev, mles[t], po.ext([1]), [list(range(k))], nv, k, prover.Transcript.stub(50 + t),
                                               stream=C.c_void_p(stream))
            return 0
        except Exception:  # noqa: BLE001
            return -1

    cb = TASK(body)
    arr = (LaneTask * n_tasks)()
    for t in range(n_tasks):
        arr[t].fn, arr[t].arg, arr[t].estimated_bytes = cb, C.c_void_p(t) if t else None, (t + 1) << 20
    status = (C.c_int * n_tasks)(*([-99] * n_tasks))
    lanes = (C.c_int * n_tasks)(*([-1] * n_tasks))
    assert L.ceno_prover_lanes_run(dev.h, 3, arr, n_tasks, status, lanes) == 0
    assert list(status) == [0] * n_tasks and all(0 <= x < 3 for x in lanes)
    assert min(seen_booked) >= 1 << 20 and dev.L.ceno_hip_mem_booked(dev.h) == 0
    for t in range(n_tasks):
        exp = po.sumcheck_prove(tables[t], po.ext([1]), [list(range(k))], nv, k, po.StubTranscript(50 + t))
        for g, e in zip(results[t], exp):
            assert np.array_equal(g, e)


@pytest.mark.parametrize("seed", [s_ if s_ < 12 else pytest.param(s_, marks=pytest.mark.slow) for s_ in range(24)])
def test_sumcheck_random_plans_differential(dev, prover, seed):
    """seeded random plans against the oracle: 1-3 size classes (front-loading), base / ext tables, 1-9 terms of degree
    1-5, optional common-factor groups, sizes on both sides of the tile / two-kernel / dense thresholds, pipelined and
    round-by-round drivers"""
    import random

    from ceno_amd import Sumcheck

    rng = random.Random(1000 + seed)
    max_nv = rng.choice([3, 6, 9, 12, 15, 17, 18])
    n_classes = rng.choice([1, 1, 2, 3])
    sizes = sorted({max_nv} | {rng.randint(1, max_nv) for _ in range(n_classes - 1)}, reverse=True)
    tables, cls_of = [], []
    for ci, nv in enumerate(sizes):
        for j in range(rng.randint(1, 4)):
            is_ext = rng.random() < 0.6
            tables.append(po.rand_ext(1 << nv, 97 * seed + len(tables)) if is_ext else po.rand_base(1 << nv, 97 * seed + len(tables)))
            cls_of.append(ci)
    d = rng.randint(2, 5)
    terms = []
    for _ in range(rng.randint(1, 9)):
        ci = rng.randrange(len(sizes))
        members = [j for j, c in enumerate(cls_of) if c == ci]
        terms.append([rng.choice(members) for _ in range(rng.randint(1, d))])
    groups = None
    if rng.random() < 0.4:  # a common factor in front of all terms of the largest class that leave room for it
        big = [j for j, c in enumerate(cls_of) if c == 0]
        members = [t for t, fac in enumerate(terms) if cls_of[fac[0]] == 0 and len(fac) < d]
        if members:
            groups = [([rng.choice(big)], members)]
    coeffs = po.rand_ext(len(terms), 5000 + seed)
    msgs, chal, fin = _run_both(dev, prover, tables, coeffs, terms, max_nv, d, groups=groups, seed=seed + 7)
    # the same plan driven round by round without pipelining must give the same transcript
    mles = [dev.upload(t) for t in tables]
    sc = Sumcheck(dev, mles, coeffs, terms, max_nv, d, groups=groups)
    ch = None
    for i in range(max_nv):
        m = sc.round(ch)
        assert np.array_equal(m, msgs[i]), i
        ch = tup(chal[i])
    assert np.array_equal(sc.finish(ch), fin)
    sc.free()


@pytest.mark.parametrize("seed", [s_ if s_ < 14 else pytest.param(s_, marks=pytest.mark.slow) for s_ in range(28)])
def test_sumcheck_random_plans_lds_blocked_kernel(dev, prover, seed, monkeypatch):
    """the LDS-blocked generic kernel (k_gen, csrc/sumcheck_gen.hip) forced onto small plans: 1-3 size classes, several
    chips (connected components) per class with their own selectors, first rounds over base-field columns (the base-field
    phase 2) or extension tables, terms without a group, groups with two common factors, tables no term reads (fold-only
    components), small LDS budgets (tiles of 16 .. 256 pairs, 1 / 2 / 4 waves sharing a group's terms); pipelined host loop
    and round-by-round driver against the oracle"""
    import random

    from ceno_amd import Sumcheck

    rng = random.Random(4242 + seed)
    monkeypatch.setenv("CENO_HIP_GEN_MIN_LOG", "1")
    monkeypatch.setenv("CENO_HIP_GEN_PIPE_MIN_LOG", "0")
    monkeypatch.setenv("CENO_HIP_GEN_STAGE_KB", str(rng.choice([2, 6, 16, 48])))
    max_nv = rng.choice([4, 7, 9, 11, 13, 15])
    n_classes = rng.choice([1, 1, 2, 3])
    sizes = sorted({max_nv} | {rng.randint(2, max_nv) for _ in range(n_classes - 1)}, reverse=True)
    d = rng.randint(2, 5)
    all_base = rng.random() < 0.5   # witness columns in the base field: round 0 runs the base-field phase 2
    tables, nvs, terms, groups = [], [], [], []
    for nv in sizes:
        for chip in range(rng.randint(1, 3)):
            first = len(tables)
            n_cols = rng.randint(1, 6)
            for j in range(n_cols):
                is_ext = (not all_base) and rng.random() < 0.5
                tables.append(po.rand_ext(1 << nv, 31 * seed + len(tables)) if is_ext else po.rand_base(1 << nv, 31 * seed + len(tables)))
                nvs.append(nv)
            cols = list(range(first, first + n_cols))
            n_sel = rng.choice([0, 1, 1, 2])
            sels = []
            for _ in range(n_sel):
                sels.append(len(tables))
                tables.append(po.rand_ext(1 << nv, 31 * seed + len(tables)))
                nvs.append(nv)
            if rng.random() < 0.3:   # a table nothing reads
                tables.append(po.rand_base(1 << nv, 31 * seed + len(tables)))
                nvs.append(nv)
            for gi in range(max(1, n_sel)):
                common = sels[: gi + 1] if (n_sel and rng.random() < 0.3) else (sels[gi: gi + 1] if n_sel else [])
                members = []
                for _ in range(rng.randint(1, 7)):
                    room = d - len(common)
                    if room < 1:
                        break
                    members.append(len(terms))
                    terms.append([rng.choice(cols) for _ in range(rng.randint(1, room))])
                if common and members:
                    groups.append((common, members))
    if not terms:
        terms.append([0])
    coeffs = po.rand_ext(len(terms), 7000 + seed)
    msgs, chal, fin = _run_both(dev, prover, tables, coeffs, terms, max_nv, d, groups=groups or None, seed=seed + 3)
    mles = [dev.upload(t) for t in tables]
    sc = Sumcheck(dev, mles, coeffs, terms, max_nv, d, groups=groups or None)
    ch = None
    for i in range(max_nv):
        m = sc.round(ch)
        assert np.array_equal(m, msgs[i]), i
        ch = tup(chal[i])
    assert np.array_equal(sc.finish(ch), fin)
    sc.free()


@pytest.mark.parametrize("w,s0,relay", [(256, 128, 0), (64, 32, 0), (16, 8, 0), (4, 2, 0), (256, 128, 1), (8, 64, 1)])
@pytest.mark.parametrize("nv,n_mles,terms", [(13, 4, [[0, 1, 2], [1, 2, 3]]), (15, 6, [[0, 1], [2, 3, 4, 5], [1, 5]]), (10, 3, [[0, 1, 2]])])
def test_sumcheck_persistent_mid_rounds_geometries(dev, prover, monkeypatch, w, s0, relay, nv, n_mles, terms):
    """k_mid (rounds between the streaming kernels and the single-workgroup tail in one launch of resident workgroups): every
    geometry — many small slices, few large ones, the relay path used when the mailbox is in host memory — gives the oracle's
    proof bit for bit; base-field inputs start the ladder in round 0 (nv = 10: the whole sumcheck is k_mid + k_tail)"""
    monkeypatch.setenv("CENO_HIP_MID_W", str(w))
    monkeypatch.setenv("CENO_HIP_MID_S0", str(s0))
    monkeypatch.setenv("CENO_HIP_MID_RELAY", str(relay))
    is_ext = nv != 10
    tabs = [po.rand_ext(1 << nv, 500 + j) if is_ext else po.rand_base(1 << nv, 500 + j) for j in range(n_mles)]
    mles = [dev.upload(t) for t in tabs]
    coeffs = po.ext([(3 + 2 * i, 7 * i + 1) for i in range(len(terms))])
    deg = max(len(t) for t in terms)
    msgs, chal, fin = prover.sumcheck_prove(dev, mles, coeffs, terms, nv, deg, prover.Transcript.stub(0xA1))
    omsgs, ochal, ofin = po.sumcheck_prove(tabs, coeffs, terms, nv, deg, po.StubTranscript(0xA1))
    assert np.array_equal(msgs, omsgs) and np.array_equal(chal, ochal) and np.array_equal(fin, ofin)
    for m in mles:
        m.free()


@pytest.mark.parametrize("cap,budget_ns", [("0", None), ("1", None), ("3", None), ("12", "1e9"), ("12", "1"), (None, None)])
@pytest.mark.parametrize("nv,n_mles,terms,is_ext", [
    (13, 4, [[0, 1, 2], [1, 2, 3]], True),                      # k_mid + k_tail + host
    (9, 3, [[0, 1, 2]], False),                                 # base-field inputs, the tail kernel is the first kernel
    (4, 2, [[0, 1]], True),                                     # fewer rounds than the host would take: the device keeps round 0
    (2, 3, [[0, 1, 2], [2]], True),
    (11, 9, [[0, 1, 2], [3, 4], [5, 6, 7, 8], [0, 8], [2, 4, 6], [1], [3, 5, 7]], True),  # an expensive plan: few host rounds
])
def test_sumcheck_host_finished_tail_every_split(dev, prover, monkeypatch, cap, budget_ns, nv, n_mles, terms, is_ext):
    """The last rounds of a pipelined sumcheck are computed by the HOST on tables the tail kernel ships with its last message
    (CENO_HIP_HOST_TAIL caps the rounds, CENO_HIP_HOST_TAIL_NS is the per-round budget the plan's cost is held against).  Every
    split — none, one round, a few, as many as the tables allow (budget 1e9), the default — gives the oracle's messages, challenges
    and final evaluations word for word."""
    if cap is None:
        monkeypatch.delenv("CENO_HIP_HOST_TAIL", raising=False)
    else:
        monkeypatch.setenv("CENO_HIP_HOST_TAIL", cap)
    if budget_ns is None:
        monkeypatch.delenv("CENO_HIP_HOST_TAIL_NS", raising=False)
    else:
        monkeypatch.setenv("CENO_HIP_HOST_TAIL_NS", budget_ns)
    tabs = [po.rand_ext(1 << nv, 900 + j) if is_ext else po.rand_base(1 << nv, 900 + j) for j in range(n_mles)]
    mles = [dev.upload(t) for t in tabs]
    coeffs = po.ext([(5 + 3 * i, 11 * i + 2) for i in range(len(terms))])
    deg = max(len(t) for t in terms)
    msgs, chal, fin = prover.sumcheck_prove(dev, mles, coeffs, terms, nv, deg, prover.Transcript.stub(0xA7))
    omsgs, ochal, ofin = po.sumcheck_prove(tabs, coeffs, terms, nv, deg, po.StubTranscript(0xA7))
    assert np.array_equal(msgs, omsgs) and np.array_equal(chal, ochal) and np.array_equal(fin, ofin)
    for m in mles:
        m.free()


@pytest.mark.parametrize("seed", range(8))
def test_tower_random_specs_differential(dev, prover, seed):
    _tower_random_specs_differential(dev, prover, seed)


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("host_layers", [4, 8])
def test_tower_fused_eq_rounds_random_specs_differential(dev, prover, monkeypatch, seed, host_layers):
    """the same batches with the fused tower-layer kernel (sumcheck_tower.hip) taking every round of at least 2^6 pairs: the host completes
    each of its messages from q(1), the leading coefficient and the running claim, and proof and evaluations stay the oracle's bit for bit"""
    monkeypatch.setenv("CENO_HIP_TOWER_FAST_MIN_LOG", "6")
    monkeypatch.setenv("CENO_TOWER_HOST_LAYERS", str(host_layers))
    _tower_random_specs_differential(dev, prover, seed)


def test_tower_fused_eq_rounds_engage_only_on_the_tower_shape(dev, prover, monkeypatch):
    """ceno_hip_sumcheck_fused_eq_rounds: n - min_log leading rounds for a tower layer, none when switched off, none for other handles"""
    import ctypes as C

    lasts = [[po.rand_ext(1 << 10, 61), po.rand_ext(1 << 10, 62)], [po.rand_ext(1 << 8, 63), po.rand_ext(1 << 8, 64)]]
    lk = [po.rand_ext(1 << 10, 70 + j) for j in range(4)]
    pt = [prover.Tower.from_last_layer(dev, [dev.upload(x) for x in l]) for l in lasts]
    lt = prover.Tower.from_last_layer(dev, [dev.upload(x) for x in lk])
    rt = po.rand_ext(10, 77)
    alphas = po.rand_ext(4, 78)

    def fused_rounds(layer):
        pa = (C.c_void_p * 2)(pt[0].h, pt[1].h)
        la = (C.c_void_p * 1)(lt.h)
        h = C.c_void_p()
        dev.check(dev.L.ceno_hip_tower_layer_sumcheck_begin(dev.h, pa, 2, la, 1, layer, rt.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                            alphas.ctypes.data_as(C.POINTER(C.c_uint64)), None, C.byref(h)))
        n = dev.L.ceno_hip_sumcheck_fused_eq_rounds(h)
        dev.check(dev.L.ceno_hip_sumcheck_free(dev.h, h))
        return n

    monkeypatch.setenv("CENO_HIP_TOWER_FAST_MIN_LOG", "6")
    assert fused_rounds(10) == 4 and fused_rounds(8) == 2 and fused_rounds(6) == 0   # layer 10: only the first product tower and the LogUp tower have it
    monkeypatch.setenv("CENO_HIP_TOWER_FAST", "0")
    assert fused_rounds(10) == 0
    monkeypatch.delenv("CENO_HIP_TOWER_FAST")
    monkeypatch.delenv("CENO_HIP_TOWER_FAST_MIN_LOG")
    assert fused_rounds(10) == 0                                                        # default hand-over: rounds of 2^16 pairs and more
    from ceno_amd import Sumcheck

    tabs = [dev.upload(po.rand_ext(1 << 9, 80 + j)) for j in range(3)]
    sc = Sumcheck(dev, tabs, po.ext([(1, 0)]), [[0, 1, 2]], 9, 3)
    assert dev.L.ceno_hip_sumcheck_fused_eq_rounds(sc.h) == 0
    sc.free()
    for t in pt + [lt] + tabs:
        t.free()


def _tower_random_specs_differential(dev, prover, seed):
    """seeded random tower batches (0-3 product specs, 0-2 LogUp specs with or without numerators, heights 2-13 so that
    layers cross the tile / two-kernel thresholds): out-evals, every message, every per-round evaluation and the point
    equal the oracle's; the restated TowerVerify accepts"""
    import random

    rng = random.Random(400 + seed)

    def last(nv, s):
        return [po.rand_ext(1 << (nv - 1), s), po.rand_ext(1 << (nv - 1), s + 1)]

    n_p, n_l = rng.randint(0, 3), rng.randint(0, 2)
    if n_p + n_l == 0:
        n_p = 1
    hp = [rng.randint(2, 13) for _ in range(n_p)]
    hl = [rng.randint(2, 13) for _ in range(n_l)]
    pl = [last(h, 10 * i + seed) for i, h in enumerate(hp)]
    ql = [((last(h, 500 + 10 * i + seed) if rng.random() < 0.5 else None), last(h, 700 + 10 * i + seed)) for i, h in enumerate(hl)]
    specs_p = [po.infer_tower_product_witness(h, l) for h, l in zip(hp, pl)]
    specs_l = [po.infer_tower_logup_witness(p, q) for p, q in ql]
    towers_p = [prover.Tower.from_last_layer(dev, [dev.upload(x) for x in l]) for l in pl]
    towers_l = [prover.Tower.from_last_layer(dev, [dev.upload(p[0]) if p else None, dev.upload(p[1]) if p else None,
                                                   dev.upload(q[0]), dev.upload(q[1])]) for p, q in ql]
    out_evals, proof = prover.prove_tower_relation(dev, towers_p, towers_l, prover.Transcript.stub(seed))
    tr = po.StubTranscript(seed)
    po_ev = np.stack([np.stack([s[0][0][0], s[0][1][0]]) for s in specs_p]) if n_p else np.zeros((0, 2, 2), dtype=np.uint64)
    lo_ev = np.stack([np.stack([s[0][k][0] for k in range(4)]) for s in specs_l]) if n_l else np.zeros((0, 4, 2), dtype=np.uint64)
    flat = list(po_ev.reshape(-1, 2)) + list(lo_ev.reshape(-1, 2))
    for e in flat:
        tr.append_ext(tup(e))
    oproof = po.tower_prove(specs_p, specs_l, tr)
    max_h = max(hp + hl)
    assert np.array_equal(out_evals, np.array(flat, dtype=np.uint64).reshape(-1, 2))
    assert np.array_equal(proof.msgs, oproof.msgs)
    if n_p:
        assert np.array_equal(proof.prod_evals, oproof.prod_evals)
    if n_l:
        assert np.array_equal(proof.logup_evals, oproof.logup_evals)
    assert np.array_equal(proof.point[:max_h], oproof.point[:max_h])
    vt = po.StubTranscript(seed)
    for e in flat:
        vt.append_ext(tup(e))
    rc, *_ = po.tower_verify(po_ev, lo_ev, hp + hl, oproof, vt)
    assert rc == 0
    for t in towers_p + towers_l:
        t.free()


def test_no_device_memory_leaks_over_repeated_proofs(prover):
    """every handle returns its device memory to the pool: after warm-up, repeating sumcheck / tower / commit+open
    leaves pool_used where it was and does not grow the cached total"""
    from ceno_amd import Device

    d = Device(0)
    stream = d.stream_create()

    def one_pass(seed):
        tabs = [po.rand_ext(1 << 9, seed + j) for j in range(3)] + [po.rand_base(1 << 6, seed + 9)]
        mles = [d.upload(t) for t in tabs]
        prover.sumcheck_prove(d, mles, po.rand_ext(3, seed), [[0, 1, 2], [0, 1], [3, 3]], 9, 3, prover.Transcript.stub(seed))
        recs = [d.upload(po.rand_ext(1 << 5, seed + 20 + j)) for j in range(3)]
        t = prover.Tower.build_prod(d, recs, 1 << 5, (1, 0))
        lt = prover.Tower.build_logup(d, None, recs, 1 << 5, (5, 6))
        prover.prove_tower_relation(d, [t], [lt], prover.Transcript.stub(seed))
        t.free()
        lt.free()
        m = po.rand_base(40 * 5, seed + 40).reshape(40, 5)
        pcs = prover.PcsData(d, [m], 1, stream)
        pt = po.rand_ext(pcs.num_vars(0), seed + 41)
        ev = np.array([pcs.witness_mle(0, c).evaluate(pt) for c in range(5)], dtype=np.uint64)
        pcs.basefold_open([pt], [ev], 4, 3, prover.Transcript.stub(seed))
        pcs.free()
        for x in mles + recs:
            x.free()

    for s in range(3):
        one_pass(s)  # warm-up: pool buckets, twiddle tables, mailboxes
    d.sync()
    mi = d.mem_info()
    used0, cached0 = mi["pool_used"], mi["pool_cached"]
    for s in range(10):
        one_pass(100 + s)
    d.sync()
    mi = d.mem_info()
    used1, cached1 = mi["pool_used"], mi["pool_cached"]
    assert used1 == used0, (used0, used1)
    assert cached1 <= cached0 + (1 << 20), (cached0, cached1)
    d.close()


def test_filter_even_odd_and_memory_estimate(dev, prover):
    """filter_mle_even_odd_batch (scheme/gpu/util.rs:186-266) and estimate_sumcheck_memory (scheme/gpu/memory.rs:413-433):
    the estimate must cover what a sumcheck really takes from the pool"""
    import ctypes as C

    from ceno_amd.api import Mle

    for tab in (po.rand_base(1 << 9, 5), po.rand_ext(1 << 6, 6)):
        m = dev.upload(tab)
        for odd in (0, 1):
            h = C.c_void_p()
            dev.check(dev.L.ceno_hip_mle_filter_even_odd(dev.h, m.h, odd, None, C.byref(h)))
            out = Mle(dev, h)
            assert np.array_equal(out.download(), tab[odd::2])
            out.free()
    nv, k = 16, 5
    mles = [dev.synthetic(nv, j % 2 == 0, 40 + j) for j in range(k)]
    terms = [[0, 1, 2], [3, 4], [0, 4]]
    dev.sync()
    before = dev.mem_info()["pool_used"]
    from ceno_amd import Sumcheck

    sc = Sumcheck(dev, mles, po.rand_ext(3, 1), terms, nv, 3)
    sc.round(None)
    sc.round((3, 4))
    used = dev.mem_info()["pool_used"] - before
    nvs = (C.c_int * k)(*([nv] * k))
    est = dev.L.ceno_hip_sumcheck_estimate_memory(nv, 3, nvs, k, len(terms))
    assert 0 < used <= est <= 4 * used + (1 << 22), (used, est)
    sc.free()


def test_batched_main_constraints_grouping_variety(dev, prover):
    """the host groups terms by their extension-field factors: two selectors per chip (two groups), a term under both
    selectors, a term without any selector (ungrouped) and a term made of selectors only — the transcript must equal the
    oracle's single flat plan whatever the factoring"""
    gch = [(3, 4), (5, 6)]
    nv, n_w = 8, 4
    wit = [po.rand_base(1 << nv, 900 + j) for j in range(n_w)]
    p1, p2 = po.rand_ext(nv, 1), po.rand_ext(nv, 2)
    sels = [(po.SEL_PREFIX, 0, (1 << nv) - 9, 0, (), 0, p1), (po.SEL_PREFIX, 5, 100, 1, (), 0, p2)]
    s1, s2 = n_w, n_w + 1
    terms = [[s1, 0, 1], [s1, 2], [s2, 1, 3], [s2, 0, 0, 2], [s1, s2, 3], [0, 1, 2], [s1, s2], [s2, 3]]
    scalars = [[((2 + t, t), [2 + (t % 2)])] for t in range(len(terms))]
    mles = [dev.upload(t) for t in wit] + [None, None]
    job = dict(num_vars=nv, mles=mles, n_witin=n_w, n_fixed=0, n_structural=2, selectors=sels, n_exprs=2, max_degree=4, terms=terms,
               scalars=scalars)
    claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, [job], gch, prover.Transcript.stub(9))
    tr = po.StubTranscript(9)
    tr.append_label(b"combine subset evals")
    alpha = tr.sample_ext()
    chal = gch + [(1, 0), alpha]
    tables = wit + [po.selector_compute(s[0], s[6], s[1], s[2]) for s in sels]
    coeffs = []
    for monos in scalars:
        sc = (0, 0)
        for coeff, ids in monos:
            v = coeff
            for i in ids:
                v = po.e2_mul(v, chal[i])
            sc = po.e2_add(sc, v)
        coeffs.append(sc)
    omsgs, ochal, ofin = po.sumcheck_prove(tables, po.ext(coeffs), terms, nv, 4, tr)
    assert np.array_equal(msgs, omsgs) and np.array_equal(rt, ochal) and np.array_equal(evals, ofin)
    final_claim = po.sumcheck_expected_from_evals([nv] * len(tables), po.ext(coeffs), terms, nv, ochal, ofin)
    assert claimed == po.recover_claim_from_final(final_claim, omsgs, ochal)
