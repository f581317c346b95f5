"""GPU parity for the per-chip proof flow and the BASELINE.json configs at FULL size.

* create_chip_proof / build_tower_witness (ceno_zkvm/src/scheme/prover.rs:717-833, scheme/cpu/mod.rs:608-797) against the
  oracle, bit for bit, on small chips of every record shape (table circuits with numerators, a missing write set,
  non-power-of-two group sizes, a rotation argument).
* config #3 (benches/riscv_add.rs:86-141 shape): ADD-shaped chip, 2^20 rows x 22 columns, commit -> record inference ->
  three towers -> tower proof -> batched main constraints (one job) -> Basefold open, every proof object checked by the
  oracle's restated verifiers, the main sumcheck's last 12 rounds replayed on the oracle.
* config #4 shape on one GPU (scheme/cpu/mod.rs:1052-1390): 24 chips of 14..24 variables in ONE batched main sumcheck.
The commit / open path is PARITY UNPINNED against the reference (placeholder Poseidon2 constants).
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
    from ceno_amd import prover as p

    return p


def tup(a):
    return int(a[0]), int(a[1])


def e2_pow(a, k):
    r = (1, 0)
    for _ in range(k):
        r = po.e2_mul(r, a)
    return r


# ------------------------------------------------------------------------------------------------------------------
# synthetic ADD-shaped circuit (SURVEY.md section 8d "S-chip"): records are RLCs of witness columns with the two global
# challenges, the main constraints are selector x (degree-2 and degree-3 products of columns)
# ------------------------------------------------------------------------------------------------------------------
def record_plan(w, n_records, alpha, beta):
    b2 = po.e2_mul(beta, beta)
    terms, coeffs, out_terms = [], [], []
    for k in range(n_records):
        base = len(terms)
        terms += [[(2 * k) % w], [(2 * k + 1) % w], [(3 * k + 5) % w, (k + 7) % w]]
        coeffs += [beta, b2, alpha]
        out_terms.append([base, base + 1, base + 2])
    return po.ext(coeffs), terms, out_terms


def main_plan(w, s_id):
    terms = [[s_id, j, (j + 1) % w] for j in range(w)] + [[s_id, j, (j + 3) % w, (j + 5) % w] for j in range(0, w, 2)]
    # scalar_t = (3 + 5t, 11t + 1) * challenge[2 + t % 2]  (an alpha power), every third one also times the global beta
    scalars = [[(((3 + 5 * t) % P, (11 * t + 1) % P), [2 + (t % 2)] + ([1] if t % 3 == 0 else []))] for t in range(len(terms))]
    return terms, scalars


def oracle_scalars(scalars, chal):
    out = []
    for monos in scalars:
        sc = (0, 0)
        for coeff, ids in monos:
            v = coeff
            for i in ids:
                v = po.e2_mul(v, chal[i])
            sc = po.e2_add(sc, v)
        out.append(sc)
    return out


@pytest.mark.parametrize("shape", [(4, 4, 0, 8), (3, 0, 2, 0), (0, 1, 0, 1), (5, 2, 0, 3), (1, 1, 1, 0)])
def test_create_chip_proof_matches_oracle(dev, prover, shape):
    """every field of the chip proof equals the oracle's restatement of create_chip_proof"""
    num_reads, num_writes, num_lk_tables, num_lk = shape
    log2_n, w = 6, 9
    rows = 1 << log2_n
    n_lk_den = num_lk_tables if num_lk_tables else num_lk
    n_rec = num_reads + num_writes + num_lk_tables + n_lk_den
    alpha, beta = (0x1234567, 0x89ABCDE), (0x13579B, 0x2468AC)
    cols = [po.rand_base(rows, 700 + j) for j in range(w)]
    coeffs, terms, out_terms = record_plan(w, n_rec, alpha, beta)
    mles = [dev.upload(c) for c in cols]
    task = dict(mles=mles, n_witin=w, n_fixed=0, n_structural=0, num_instances=rows - 5, log2_num_instances=log2_n,
                num_reads=num_reads, num_writes=num_writes, num_lk_tables=num_lk_tables, num_lk=num_lk, record_coeffs=coeffs,
                record_terms=terms, record_out_terms=out_terms)
    proof = prover.create_chip_proof(dev, task, [alpha, beta], prover.Transcript.stub(21))
    # ---- oracle: wit_infer -> interleave -> towers -> out-evals into the transcript -> tower proof ----
    recs = [po.wit_infer(cols, coeffs[ts[0]: ts[-1] + 1], [terms[t] for t in ts], log2_n) for ts in out_terms]
    r_set, w_set = recs[:num_reads], recs[num_reads: num_reads + num_writes]
    lk_n = recs[num_reads + num_writes: num_reads + num_writes + num_lk_tables]
    lk_d = recs[num_reads + num_writes + num_lk_tables:]
    prod_specs, logup_specs, out_evals = [], [], []
    for group in (r_set, w_set):
        if group:
            limbs = po.interleaving_mles_to_mles(group, rows, 2, (1, 0))
            nv = int(limbs[0].shape[0]).bit_length()
            assert nv == log2_n + (max(1, len(group)) - 1).bit_length()  # group_num_vars (cpu/mod.rs:647-648)
            layers = po.infer_tower_product_witness(nv, limbs)
            prod_specs.append(layers)
            out_evals += [layers[0][0][0], layers[0][1][0]]
    if lk_d:
        ql = po.interleaving_mles_to_mles(lk_d, rows, 2, alpha)
        pl = po.interleaving_mles_to_mles(lk_n, rows, 2, alpha) if lk_n else None
        layers = po.infer_tower_logup_witness(pl, ql)
        logup_specs.append(layers)
        out_evals += [layers[0][k][0] for k in range(4)]
    tr = po.StubTranscript(21)
    for e in out_evals:
        tr.append_ext(tup(e))
    oproof = po.tower_prove(prod_specs, logup_specs, tr)
    assert np.array_equal(proof.tower_msgs, oproof.msgs)
    assert np.array_equal(proof.tower_point, oproof.point[: proof.tower_num_vars])
    if prod_specs:
        assert np.array_equal(proof.tower_prod_evals, oproof.prod_evals)
    if logup_specs:
        assert np.array_equal(proof.tower_logup_evals, oproof.logup_evals)
    got_out = [x for x in proof.r_out_evals] + [x for x in proof.w_out_evals] + [x for x in proof.lk_out_evals]
    assert len(got_out) == len(out_evals) and all(tup(a) == tup(b) for a, b in zip(got_out, out_evals))
    assert proof.rt_main.shape[0] == log2_n and np.array_equal(proof.rt_main, proof.tower_point[-log2_n:])
    # build_tower_witness on its own gives the same towers
    d_recs = [dev.upload(r) for r in recs]
    ev, pt, lt = prover.build_tower_witness(dev, d_recs, num_reads, num_writes, num_lk_tables, num_lk, log2_n, 0, [alpha, beta])
    assert [t.num_vars for t in pt] == [len(s) for s in prod_specs] and [t.num_vars for t in lt] == [len(s) for s in logup_specs]
    for t, spec in zip(pt + lt, prod_specs + logup_specs):
        last = t.num_vars - 1
        for limb in range(t.num_limbs):
            assert np.array_equal(t.layer(last, limb), spec[last][limb])


def test_create_chip_proof_with_rotation_matches_oracle(dev, prover):
    """a keccak-style chip: 2^2 instances x 2^5 rotation rows; the rotation argument runs at rt_main after the tower"""
    log2_n, rot_vars, w = 2, 5, 4
    nv = log2_n + rot_vars
    alpha, beta = (5, 6), (7, 8)
    src = po.rand_base(1 << nv, 31)
    cols = [src, po.rotation_next_base_mle(src, 5), po.rand_base(1 << nv, 32), po.rand_base(1 << nv, 33)]
    coeffs, terms, out_terms = record_plan(w, 3, alpha, beta)
    task = dict(mles=[dev.upload(c) for c in cols], n_witin=w, n_fixed=0, n_structural=0, num_instances=3, log2_num_instances=log2_n,
                rotation_vars=rot_vars, num_reads=1, num_writes=1, num_lk_tables=0, num_lk=1, record_coeffs=coeffs, record_terms=terms,
                record_out_terms=out_terms, rotation=dict(pairs=[(0, 1), (2, 3)], cyclic_subgroup_size=23, cyclic_group_log2=5))
    proof = prover.create_chip_proof(dev, task, [alpha, beta], prover.Transcript.stub(8))
    recs = [po.wit_infer(cols, coeffs[ts[0]: ts[-1] + 1], [terms[t] for t in ts], nv) for ts in out_terms]
    specs, out_evals = [], []
    for r in recs[:2]:
        limbs = po.interleaving_mles_to_mles([r], 1 << nv, 2, (1, 0))
        layers = po.infer_tower_product_witness(nv, limbs)
        specs.append(layers)
        out_evals += [layers[0][0][0], layers[0][1][0]]
    ll = po.infer_tower_logup_witness(None, po.interleaving_mles_to_mles(recs[2:], 1 << nv, 2, alpha))
    out_evals += [ll[0][k][0] for k in range(4)]
    tr = po.StubTranscript(8)
    for e in out_evals:
        tr.append_ext(tup(e))
    oproof = po.tower_prove(specs, [ll], tr)
    assert np.array_equal(proof.tower_msgs, oproof.msgs)
    rt_main = oproof.point[:nv][-nv:]
    exp = po.prove_rotation(cols, [(0, 1), (2, 3)], 23, 5, np.ascontiguousarray(rt_main), tr)
    assert np.array_equal(proof.rotation_msgs, exp[0]) and np.array_equal(proof.rotation_evals, exp[1])
    assert np.array_equal(proof.rotation_points, np.stack(exp[2:5]))


def test_main_constraints_public_instance_atoms(dev, prover):
    """scalar expressions may reference public-instance values (eval_by_expr_with_instance(.., &chip.pi, ..),
    scheme/cpu/mod.rs:1297-1304): atoms >= 2 + n_exprs select pi"""
    nv, w = 7, 3
    gch = [(3, 4), (5, 6)]
    pi = [(1000, 0), (77, 88)]
    wit = [po.rand_base(1 << nv, 40 + j) for j in range(w)]
    point = po.rand_ext(nv, 9)
    sel = (po.SEL_PREFIX, 0, (1 << nv) - 3, 0, (), 0, point)
    terms = [[w, 0, 1], [w, 2], [w, 1, 2, 0]]
    scalars = [[((1, 0), [2, 4])], [((2, 0), [3, 5]), ((7, 0), [4, 4, 0])], [((1, 1), [5])]]
    job = dict(num_vars=nv, mles=[dev.upload(t) for t in wit] + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2,
               max_degree=4, terms=terms, scalars=scalars, pi=pi)
    claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, [job], gch, prover.Transcript.stub(4))
    tr = po.StubTranscript(4)
    tr.append_label(b"combine subset evals")
    a = tr.sample_ext()
    chal = gch + [(1, 0), a] + pi
    coeffs = po.ext(oracle_scalars(scalars, chal))
    tables = wit + [po.selector_compute(sel[0], sel[6], sel[1], sel[2])]
    omsgs, ochal, ofin = po.sumcheck_prove(tables, coeffs, terms, nv, 4, tr)
    assert np.array_equal(msgs, omsgs) and np.array_equal(rt, ochal) and np.array_equal(evals, ofin)


# ------------------------------------------------------------------------------------------------------------------
# eq-factored main-constraint rounds (ceno_hip_sumcheck_begin_eq, csrc/sumcheck_gen.hip "EQ-FACTORED FORM"): Whole / Prefix selectors are
# declared as eq(., rt) on a row range; the rounds evaluate each chip's quotient at one point fewer and the host completes the messages
# from the chips' running claims.  The words must be those of the oracle's prover — for row ranges that start and end anywhere (boundary
# pairs in every round), several selectors per chip, chips whose selector is not of that form (generic rounds in the same sumcheck),
# and messages of 3, 4 and 5 points.
# ------------------------------------------------------------------------------------------------------------------
def _eq_case_jobs(dev, case, max_degree, degs=None):
    w = 4
    jobs, tabs, terms_all, scal_all, nvs = [], [], [], [], []
    batch_degree = max_degree
    for c, (nv, sels) in enumerate(case):
        max_degree = degs[c % len(degs)] if degs else batch_degree  # (chips of different degrees in one batch)
        cols = [po.rand_base(1 << nv, 7000 + 31 * c + j) for j in range(w)]
        point = po.rand_ext(nv, 500 + c)
        ns = len(sels)
        sel_t = [(k, off, n, sid, tuple(sp), snv, point) for sid, (k, off, n, sp, snv) in enumerate(sels)]
        terms = []
        for sid in range(ns):  # every selector gates a few products of 1 .. max_degree - 1 columns
            s = w + sid
            terms += [[s, (sid + j) % w, (sid + j + 1) % w][: min(3, max_degree)] for j in range(2)]
            terms += [[s, j % w] for j in range(2)]
            if max_degree >= 4:
                terms += [[s, 0, 2, 3], [s, 1, 2, 3]]
            if max_degree >= 5:
                terms += [[s, 0, 1, 2, 3]]
        scalars = [[((5 + 3 * t + c, 1 + t), [2 + (t % 2)])] for t in range(len(terms))]
        jobs.append(dict(num_vars=nv, mles=[dev.upload(t) for t in cols] + [None] * ns, n_witin=w, n_fixed=0, n_structural=ns, selectors=sel_t,
                         n_exprs=2, max_degree=max_degree, terms=terms, scalars=scalars))
        start = len(nvs)
        nvs += [nv] * (w + ns)
        tabs += cols + [po.selector_compute(k, point, off, n, tuple(sp), snv) for (k, off, n, sp, snv) in sels]
        terms_all += [[start + j for j in t] for t in terms]
        scal_all.append(scalars)
    return jobs, tabs, terms_all, scal_all, nvs


EQ_CASES = {
    # (num_vars, [(kind, offset, num_instances, sparse indices, sparse num_vars), ...]) per chip
    "prefix_ends_anywhere": [(15, [(po.SEL_PREFIX, 0, (1 << 15) - 5, (), 0)]), (14, [(po.SEL_PREFIX, 3, 100, (), 0)]),
                             (14, [(po.SEL_PREFIX, 1, (1 << 14) - 1, (), 0)]), (13, [(po.SEL_PREFIX, 4097, 2047, (), 0)])],
    "aligned_and_tiny": [(15, [(po.SEL_PREFIX, 1 << 14, 1 << 14, (), 0)]), (14, [(po.SEL_PREFIX, 0, 1, (), 0)]), (14, [(po.SEL_PREFIX, (1 << 14) - 1, 1, (), 0)]),
                         (13, [(po.SEL_PREFIX, 0, 64, (), 0)]), (13, [(po.SEL_PREFIX, 64, 64, (), 0)]), (13, [(po.SEL_PREFIX, 0, (1 << 12) + 1, (), 0)])],
    "whole_and_two_selectors": [(14, [(po.SEL_WHOLE, 0, 0, (), 0)]), (15, [(po.SEL_PREFIX, 0, 20000, (), 0), (po.SEL_PREFIX, 7, 12345, (), 0)]),
                                (13, [(po.SEL_PREFIX, 0, 2, (), 0), (po.SEL_WHOLE, 0, 0, (), 0)])],
    "generic_chip_in_the_batch": [(14, [(po.SEL_PREFIX, 0, 9999, (), 0)]), (13, [(po.SEL_ORDERED_SPARSE, 0, 200, (0, 2, 5), 3)]),
                                  (15, [(po.SEL_PREFIX, 5, 30001, (), 0), (po.SEL_ORDERED_SPARSE, 0, 1000, (1, 3), 2)])],
}


@pytest.mark.parametrize("max_degree", [4, 3, 5])
@pytest.mark.parametrize("name", sorted(EQ_CASES))
def test_eq_factored_main_constraints_match_the_oracle(dev, prover, monkeypatch, name, max_degree):
    gch = [(11, 22), (33, 44)]
    jobs, tabs, terms, scal_all, nvs = _eq_case_jobs(dev, EQ_CASES[name], max_degree)
    max_nv = max(nvs)
    L = dev.L
    before = L.ceno_hip_stat_eq_launches(dev.h)
    claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(5))
    launches = L.ceno_hip_stat_eq_launches(dev.h) - before
    assert launches >= max_nv - 2, "the eq-factored rounds did not run"
    t2 = po.StubTranscript(5)
    t2.append_label(b"combine subset evals")
    a = t2.sample_ext()
    pows = [e2_pow(a, i) for i in range(2 * len(jobs))]
    coeffs = []
    for c, sc in enumerate(scal_all):
        coeffs += oracle_scalars(sc, gch + pows[2 * c: 2 * c + 2])
    omsgs, ochal, ofin = po.sumcheck_prove(tabs, po.ext(coeffs), terms, max_nv, max_degree, t2)
    assert np.array_equal(omsgs, msgs) and np.array_equal(ochal, rt) and np.array_equal(ofin, evals)
    # the same proof with the declarations ignored
    monkeypatch.setenv("CENO_HIP_GEN_EQF", "0")
    before = L.ceno_hip_stat_eq_launches(dev.h)
    c2, m2, r2, e2 = prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(5))
    assert L.ceno_hip_stat_eq_launches(dev.h) == before
    assert c2 == claimed and np.array_equal(m2, msgs) and np.array_equal(e2, evals)
    if max_degree == 4:  # ... and through the staged first-round kernel instead of the direct one
        monkeypatch.delenv("CENO_HIP_GEN_EQF")
        monkeypatch.setenv("CENO_HIP_EQ_DIRECT0", "0")
        c3, m3, r3, e3 = prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(5))
        assert c3 == claimed and np.array_equal(m3, msgs) and np.array_equal(e3, evals)
    else:  # ... and with the small rounds on one workgroup per component (k_gen_eq) instead of one per component and slot (k_gen_eq_slots)
        monkeypatch.delenv("CENO_HIP_GEN_EQF")
        monkeypatch.setenv("CENO_HIP_EQ_SLOTS", "0")
        c3, m3, r3, e3 = prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(5))
        assert c3 == claimed and np.array_equal(m3, msgs) and np.array_equal(e3, evals)


@pytest.mark.parametrize("degs", [(5, 2, 4, 3), (2, 5), (3, 4, 2)])
@pytest.mark.parametrize("name", sorted(EQ_CASES))
def test_eq_factored_rounds_per_component_degree(dev, prover, monkeypatch, name, degs):
    """chips of DIFFERENT degrees in one batch (selector x column only: length 3 after the clamp; products of two, three, four columns): in
    the large rounds every component runs on the kernel of its own message length and the host extends its polynomial to the sumcheck's
    nodes (CENO_HIP_GEN_BY_DEGREE=2: in every round; also with the staged first round and with one workgroup per component in the small
    rounds).  Every variant produces the words of the oracle's prover."""
    gch = [(11, 22), (33, 44)]
    D = max(degs)
    jobs, tabs, terms, scal_all, nvs = _eq_case_jobs(dev, EQ_CASES[name], D, degs)
    max_nv = max(nvs)
    t2 = po.StubTranscript(5)
    t2.append_label(b"combine subset evals")
    a = t2.sample_ext()
    pows = [e2_pow(a, i) for i in range(2 * len(jobs))]
    coeffs = []
    for c, sc in enumerate(scal_all):
        coeffs += oracle_scalars(sc, gch + pows[2 * c: 2 * c + 2])
    omsgs, ochal, ofin = po.sumcheck_prove(tabs, po.ext(coeffs), terms, max_nv, D, t2)
    for switches in ({}, {"CENO_HIP_GEN_BY_DEGREE": "2"}, {"CENO_HIP_GEN_BY_DEGREE": "2", "CENO_HIP_EQ_DIRECT0": "0"},
                     {"CENO_HIP_GEN_BY_DEGREE": "2", "CENO_HIP_EQ_SLOTS": "0"}, {"CENO_HIP_GEN_BY_DEGREE": "0"}):
        for k in ("CENO_HIP_GEN_BY_DEGREE", "CENO_HIP_EQ_DIRECT0", "CENO_HIP_EQ_SLOTS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in switches.items():
            monkeypatch.setenv(k, v)
        before = dev.L.ceno_hip_stat_eq_launches(dev.h)
        claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(5))
        launches = dev.L.ceno_hip_stat_eq_launches(dev.h) - before
        assert np.array_equal(omsgs, msgs) and np.array_equal(ochal, rt) and np.array_equal(ofin, evals), switches
        if switches.get("CENO_HIP_GEN_BY_DEGREE") == "2" and name != "generic_chip_in_the_batch":
            assert launches > max_nv, "the per-degree launches did not run"


@pytest.mark.parametrize("max_degree", [4, 3])
def test_eq_factored_rounds_over_extension_columns(dev, prover, max_degree):
    """ceno_hip_sumcheck_begin_eq on a plan whose columns are EXTENSION-field tables from the start (a later GKR layer): the first round
    then runs on the staged eq kernel in its all-extension form (no base-field products), every slot wanted; two chips of different
    sizes, Prefix ranges with boundary pairs, one chip with two selectors at one point"""
    rng_seed = 900
    chips = [(14, [(5, 9000)]), (13, [(0, 8191), (100, 3000)])]
    tabs, mles, coeffs, terms, groups, decls, nvs = [], [], [], [], [], [], []
    for c, (nv, sels) in enumerate(chips):
        start = len(tabs)
        cols = [po.rand_ext(1 << nv, rng_seed + 17 * c + j) for j in range(3)]
        point = po.rand_ext(nv, 70 + c)
        tabs += cols
        for k, (off, n) in enumerate(sels):
            tabs.append(po.selector_compute(po.SEL_PREFIX, point, off, n))
            s_id = start + 3 + k
            t0 = len(terms)
            terms += [[start + (k + j) % 3, start + (k + j + 1) % 3][: max_degree - 1] for j in range(2)] + [[start + k % 3]]
            if max_degree >= 4:
                terms += [[start, start + 1, start + 2]]
            groups.append(([s_id], list(range(t0, len(terms)))))
            decls.append((s_id, point, off, off + n))
        nvs += [nv] * (3 + len(sels))
    coeffs = po.ext([(3 + 5 * t, 1 + t) for t in range(len(terms))])
    mles = [dev.upload(t) for t in tabs]
    max_nv = max(nvs)
    before = dev.L.ceno_hip_stat_eq_launches(dev.h)
    msgs, chal, fin = prover.sumcheck_prove(dev, mles, coeffs, terms, max_nv, max_degree, prover.Transcript.stub(21), groups=groups, eq_decls=decls)
    assert dev.L.ceno_hip_stat_eq_launches(dev.h) - before >= max_nv - 2
    full_terms = []
    for common, ts in groups:
        for t in ts:
            full_terms.append((t, list(common) + terms[t]))
    full_terms.sort()
    omsgs, ochal, ofin = po.sumcheck_prove(tabs, coeffs, [ft for _, ft in full_terms], max_nv, max_degree, po.StubTranscript(21))
    assert np.array_equal(omsgs, msgs) and np.array_equal(ochal, chal) and np.array_equal(ofin, fin)
    m2, c2, f2 = prover.sumcheck_prove(dev, mles, coeffs, terms, max_nv, max_degree, prover.Transcript.stub(21), groups=groups)  # no declarations
    assert np.array_equal(m2, msgs) and np.array_equal(f2, fin)
    if max_degree == 4:  # a declaration that does not describe its table is refused under CENO_HIP_EQ_VERIFY (set by tests/conftest.py)
        from ceno_amd.api import CenoHipError

        wrong = [(decls[0][0], decls[0][1], decls[0][2], decls[0][3] + 1)] + decls[1:]
        with pytest.raises(CenoHipError):
            prover.sumcheck_prove(dev, mles, coeffs, terms, max_nv, max_degree, prover.Transcript.stub(21), groups=groups, eq_decls=wrong)
    for m in mles:
        m.free()


# ------------------------------------------------------------------------------------------------------------------
# BASELINE config #3 at full size
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("log_rows", [12, 20])
def test_config3_add_chip_full_flow(dev, prover, log_rows):
    """commit -> 2 challenges -> create_chip_proof -> batched main constraints (1 job) -> open, at 2^20 rows x 22 columns
    (2^12: the same flow at a size where the tower proof is also compared with the oracle's prover bit for bit)"""
    w, n_rec, log_blowup, n_queries, pow_bits = 22, 16, 1, 100, 16
    rows = 1 << log_rows
    num_instances = rows - 3
    host = (np.random.default_rng(log_rows).integers(0, 1 << 62, size=(rows, w), dtype=np.uint64)) % np.uint64(P)
    host[num_instances:] = 0  # InstancePaddingStrategy::Default
    stream = dev.stream_create()
    pcs = prover.PcsData(dev, [host], log_blowup, stream)
    root = pcs.root()
    tr = prover.Transcript.stub(0xADD)
    tr.append_ext((int(root[0]), int(root[1])))
    tr.append_ext((int(root[2]), int(root[3])))
    alpha, beta = tr.sample_ext(), tr.sample_ext()      # prover.rs:528-531
    cols = [pcs.witness_mle(0, c) for c in range(w)]
    coeffs, terms, out_terms = record_plan(w, n_rec, alpha, beta)
    task = dict(mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, num_instances=num_instances, log2_num_instances=log_rows,
                num_reads=4, num_writes=4, num_lk_tables=0, num_lk=8, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
    proof = prover.create_chip_proof(dev, task, [alpha, beta], tr, stream)
    assert proof.tower_num_vars == log_rows + 3 and (proof.n_prod, proof.n_logup) == (2, 1)
    # ---- the restated TowerVerify accepts (scheme/verifier.rs:1372-1709) ----
    vt = po.StubTranscript(0xADD)
    vt.append_ext((int(root[0]), int(root[1])))
    vt.append_ext((int(root[2]), int(root[3])))
    assert vt.sample_ext() == alpha and vt.sample_ext() == beta
    for e in list(proof.r_out_evals) + list(proof.w_out_evals) + list(proof.lk_out_evals):
        vt.append_ext(tup(e))
    op = po.TowerProof(proof.tower_num_vars, 2, 1)
    op.msgs[:] = proof.tower_msgs
    op.prod_evals[:] = proof.tower_prod_evals
    op.logup_evals[:] = proof.tower_logup_evals
    rc, vpoint, pclaims, lp, lq = po.tower_verify(np.concatenate([proof.r_out_evals, proof.w_out_evals]), proof.lk_out_evals,
                                                  [log_rows + 2, log_rows + 2, log_rows + 3], op, vt)
    assert rc == 0 and np.array_equal(vpoint, proof.tower_point)
    # the verifier's leaf claim of the TALLEST tower (the LogUp one: its point is the final tower point) is an evaluation of the
    # interleaved lookup records: check it through an independent path — record inference, MLE evaluation of every record at
    # the row part of the point, the record index as eq-weights of the 3 low variables (interleaving, utils.rs:402-462)
    recs = dev.wit_infer(cols, coeffs, terms, out_terms, log_rows)
    eq_low = po.build_eq(np.ascontiguousarray(proof.tower_point[:3]))
    claim = (0, 0)
    for j in range(8):
        claim = po.e2_add(claim, po.e2_mul(tup(eq_low[j]), recs[8 + j].evaluate(proof.tower_point[3: 3 + log_rows])))
    assert claim == tup(lq[0]) and tup(lp[0]) == (1, 0)
    if log_rows <= 12:   # and the prover's messages equal the oracle prover's, bit for bit
        h_recs = [r.download() for r in recs]
        specs = [po.infer_tower_product_witness(log_rows + 2, po.interleaving_mles_to_mles(h_recs[4 * g: 4 * g + 4], rows, 2, (1, 0)))
                 for g in range(2)]
        ll = po.infer_tower_logup_witness(None, po.interleaving_mles_to_mles(h_recs[8:], rows, 2, alpha))
        t2 = po.StubTranscript(0xADD)
        t2.append_ext((int(root[0]), int(root[1])))
        t2.append_ext((int(root[2]), int(root[3])))
        t2.sample_ext(), t2.sample_ext()
        for e in list(proof.r_out_evals) + list(proof.w_out_evals) + list(proof.lk_out_evals):
            t2.append_ext(tup(e))
        assert np.array_equal(po.tower_prove(specs, [ll], t2).msgs, proof.tower_msgs)
    for r in recs:
        r.free()
    # ---- batched main constraints with this one job (prover.rs:577-586) ----
    mterms, mscalars = main_plan(w, w)
    sel = (po.SEL_PREFIX, 0, num_instances, 0, (), 0, proof.rt_main)
    job = dict(num_vars=log_rows, mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4,
               terms=mterms, scalars=mscalars)
    claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, [job], [alpha, beta], tr, stream)
    # verifier side: alpha powers, sumcheck_verify, final evaluations recomputed independently
    vt.append_label(b"combine subset evals")
    a = vt.sample_ext()
    mcoeffs = po.ext(oracle_scalars(mscalars, [alpha, beta, (1, 0), a]))
    vpoint, expected = po.sumcheck_verify(claimed, msgs, vt)
    assert np.array_equal(vpoint, rt)
    for c in range(w):
        assert cols[c].evaluate(rt) == tup(evals[c])                       # k_eval_dot: an independent kernel path
    assert po.selector_evaluate(po.SEL_PREFIX, proof.rt_main, rt, 0, num_instances) == tup(evals[w])   # succinct evaluator (selector.rs:247-363)
    final_claim = po.sumcheck_expected_from_evals([log_rows] * (w + 1), mcoeffs, mterms, log_rows, rt, evals)
    assert expected == final_claim and claimed == po.recover_claim_from_final(final_claim, msgs, rt)
    # the last rounds replay on the oracle from independently folded tables (fix_variables: k_fold)
    keep = min(12, log_rows)
    cut = log_rows - keep
    sel_tab = dev.selector_build(po.SEL_PREFIX, proof.rt_main, 0, num_instances)
    folded = [m.fix_variables(rt[:cut]).download() if cut else m.download() for m in cols + [sel_tab]]
    if cut == 0:
        folded = [np.stack([f, np.zeros_like(f)], axis=1) if f.ndim == 1 else f for f in folded]
    omsgs, _, ofin = po.sumcheck_prove(folded, mcoeffs, mterms, keep, 4, po.ReplayTranscript(rt[cut:]))
    assert np.array_equal(omsgs, msgs[cut:]) and np.array_equal(ofin, evals)
    for e in evals:
        vt.append_ext(tup(e))
    # ---- Basefold open at the main point (prover.rs:588-599); the oracle's verifier accepts ----
    oproof = pcs.basefold_open([rt], [evals[:w]], n_queries, pow_bits, tr)
    state = vt.state.s
    assert po.basefold_verify([(log_rows, w)], root.reshape(1, 4), [rt], [evals[:w]], log_blowup, n_queries, pow_bits, vt, oproof) == 0
    bad = oproof.copy()
    bad[len(bad) // 2] ^= 1   # somewhere inside the query answers
    vt.state.s = state
    assert po.basefold_verify([(log_rows, w)], root.reshape(1, 4), [rt], [evals[:w]], log_blowup, n_queries, pow_bits, vt, bad) != 0
    pcs.free()
    dev.stream_destroy(stream)


# ------------------------------------------------------------------------------------------------------------------
# BASELINE config #4 shape on one GPU: S-batched, 24 chips, max_nv = 24
# ------------------------------------------------------------------------------------------------------------------
def batched_jobs(dev, max_nv, w):
    sizes = [max_nv, max_nv - 2, max_nv - 2] + [max_nv - 4] * 5 + [max_nv - 6] * 8 + [max_nv - 10] * 8
    jobs, chips = [], []
    for c, nv in enumerate(sizes):
        cols = [dev.synthetic(nv, False, 1000 + 50 * c + j) for j in range(w)]
        point = np.array([[(i * 7919 + 13 + c) % P, (i * 104729 + 17) % P] for i in range(nv)], dtype=np.uint64)
        n_inst = max(1, (1 << nv) - 5 - c)
        sel = (po.SEL_PREFIX, 0, n_inst, 0, (), 0, point)
        terms = [[w, j, (j + 1) % w] for j in range(w)] + [[w, j, (j + 3) % w, (j + 5) % w] for j in range(0, w, 3)]
        scalars = [[((3 + t, 1), [2 + (t % 2)])] for t in range(len(terms))]
        jobs.append(dict(num_vars=nv, mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4,
                         terms=terms, scalars=scalars))
        chips.append(dict(nv=nv, cols=cols, point=point, n_inst=n_inst, terms=terms, scalars=scalars))
    return jobs, chips


@pytest.mark.parametrize("max_nv", [13, 24, 26])
def test_config4_batched_main_sumcheck_full_size(dev, prover, max_nv):
    """prove_batched_main_constraints over 24 chips of mixed sizes (front-load rule, scheme/verifier.rs:180-238) at max_nv = 26 —
    BASELINE config #4's stated size, 1.7 G table elements = 13.6 GB of base columns on one GPU — and 24 (424 M elements);
    13: the same plan small enough for the oracle's prover to produce every message"""
    w = 12
    gch = [(11, 22), (33, 44)]
    jobs, chips = batched_jobs(dev, max_nv, w)
    claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(5))
    vt = po.StubTranscript(5)
    vt.append_label(b"combine subset evals")
    a = vt.sample_ext()
    pows = [e2_pow(a, i) for i in range(2 * len(chips))]
    coeffs, terms, nvs = [], [], []
    for c, ch in enumerate(chips):
        start = len(nvs)
        nvs += [ch["nv"]] * (w + 1)
        coeffs += oracle_scalars(ch["scalars"], gch + pows[2 * c: 2 * c + 2])
        terms += [[start + j for j in t] for t in ch["terms"]]
    coeffs = po.ext(coeffs)
    vpoint, expected = po.sumcheck_verify(claimed, msgs, vt)
    assert np.array_equal(vpoint, rt)
    # final evaluations through independent paths: evaluate kernel for the witness columns, succinct evaluator for selectors
    off = 0
    for ch in chips:
        nv = ch["nv"]
        for j in (0, w // 2, w - 1):
            assert ch["cols"][j].evaluate(rt[:nv]) == tup(evals[off + j])
        assert po.selector_evaluate(po.SEL_PREFIX, ch["point"], rt[:nv], 0, ch["n_inst"]) == tup(evals[off + w])
        off += w + 1
    final_claim = po.sumcheck_expected_from_evals(nvs, coeffs, terms, max_nv, rt, evals)
    assert expected == final_claim and claimed == po.recover_claim_from_final(final_claim, msgs, rt)
    # replay on the oracle: every table folded (k_fold path) with the first `cut` challenges; the replay starts at round `cut`
    keep = min(12, max_nv)
    cut = max_nv - keep
    tables = []
    for ch in chips:
        nv = ch["nv"]
        sel_tab = dev.selector_build(po.SEL_PREFIX, ch["point"], 0, ch["n_inst"])
        for m in ch["cols"] + [sel_tab]:
            k = min(cut, nv)
            t = m.fix_variables(rt[:k]).download() if k else m.download()
            if t.ndim == 1:
                t = np.stack([t, np.zeros_like(t)], axis=1)
            tables.append(np.ascontiguousarray(t))
        sel_tab.free()
    if all(ch["nv"] > cut for ch in chips):
        omsgs, _, ofin = po.sumcheck_prove(tables, coeffs, terms, keep, 4, po.ReplayTranscript(rt[cut:]))
        assert np.array_equal(omsgs, msgs[cut:]) and np.array_equal(ofin, evals)
    if max_nv <= 13:  # the oracle's prover from round 0
        full = []
        for ch in chips:
            sel_tab = po.selector_compute(po.SEL_PREFIX, ch["point"], 0, ch["n_inst"])
            full += [m.download() for m in ch["cols"]] + [sel_tab]
        t2 = po.StubTranscript(5)
        t2.append_label(b"combine subset evals")
        t2.sample_ext()
        omsgs, ochal, ofin = po.sumcheck_prove(full, coeffs, terms, max_nv, 4, t2)
        assert np.array_equal(omsgs, msgs) and np.array_equal(ochal, rt) and np.array_equal(ofin, evals)


# ------------------------------------------------------------------------------------------------------------------
# config #4 at the reference's plan statistics: 48 chips, 22..96 columns, 60..250 monomials per chip (ceno_amd/synthetic.py
# wide_batched_jobs; shapes: gkr_iop/src/gkr/layer/zerocheck_layer.rs:86-207, ceno_zkvm/src/instructions.rs:48-83)
# ------------------------------------------------------------------------------------------------------------------
def _wide_oracle_plan(chips, gch, pows):
    coeffs, terms, nvs = [], [], []
    a0 = 0
    for ch in chips:
        start = len(nvs)
        nvs += [ch["nv"]] * (ch["w"] + ch["n_sel"])
        coeffs += oracle_scalars(ch["scalars"], gch + pows[a0: a0 + ch["n_exprs"]])
        terms += [[start + j for j in t] for t in ch["terms"]]
        a0 += ch["n_exprs"]
    return po.ext(coeffs), terms, nvs


@pytest.mark.parametrize("max_nv,switches", [(13, {}), (13, {"CENO_HIP_GEN_EQF": "0"}), (13, {"CENO_HIP_GEN_SPLIT": "0"}), (14, {"CENO_HIP_GEN_MIN_LOG": "4"}),
                                             (13, {"CENO_HIP_GEN_BY_DEGREE": "2"}), (14, {"CENO_HIP_GEN_BY_DEGREE": "2", "CENO_HIP_GEN_MIN_LOG": "4"}),
                                             (14, {"CENO_HIP_GEN_BY_DEGREE": "2", "CENO_HIP_EQ_DIRECT0": "0"}), (20, {}), (20, {"CENO_HIP_GEN_BY_DEGREE": "0"}),
                                             (13, {"CENO_PROVER_MAIN_LINCOMB": "0"}), (14, {"CENO_PROVER_MAIN_LINCOMB": "1", "CENO_HIP_GEN_MIN_LOG": "4"}),
                                             (20, {"CENO_PROVER_MAIN_LINCOMB": "0"})])
def test_wide_batched_main_constraints_match_the_oracle(dev, prover, monkeypatch, max_nv, switches):
    """the batched main sumcheck over 48 WIDE chips (22..96 base columns, 1..3 Prefix selectors, selector x column monomials for every
    column, selector x constant, a tail of degree 3..5 products): every message, challenge and final evaluation equals the oracle prover's
    (max_nv <= 14: from round 0; 20: verifier + independent evaluations + the last 12 rounds), with the eq-factored rounds, with the
    declarations ignored (generic rounds), without the column-block split of wide components, with the component tables on from 2^4
    rows so that the smallest chips take the same path as the largest, and with the per-degree launches of the large rounds (column blocks of
    selector x column terms on the kernel of length 3, the products on theirs) in EVERY round, with the direct and the staged first round;
    with the columns that are read only linearly combined into two tables per selector and evaluated afterwards (the default), with every
    such column kept in the sumcheck (CENO_PROVER_MAIN_LINCOMB=0), and with groups of a single column combined as well (=1)"""
    from ceno_amd import synthetic

    for k, v in switches.items():
        monkeypatch.setenv(k, v)
    gch = [(11, 22), (33, 44)]
    jobs, chips, _ = synthetic.wide_batched_jobs(dev, max_nv)
    D = max(j["max_degree"] for j in jobs)
    claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(5))
    vt = po.StubTranscript(5)
    vt.append_label(b"combine subset evals")
    a = vt.sample_ext()
    pows, acc = [], (1, 0)
    for _ in range(sum(ch["n_exprs"] for ch in chips)):
        pows.append(acc)
        acc = po.e2_mul(acc, a)
    coeffs, terms, nvs = _wide_oracle_plan(chips, gch, pows)
    vpoint, expected = po.sumcheck_verify(claimed, msgs, vt)
    assert np.array_equal(vpoint, rt)
    off = 0
    for ch in chips:
        nv, w = ch["nv"], ch["w"]
        for j in (0, w // 2, w - 1):
            assert ch["cols"][j].evaluate(rt[:nv]) == tup(evals[off + j])
        for si in range(ch["n_sel"]):
            assert po.selector_evaluate(po.SEL_PREFIX, ch["point"], rt[:nv], 0, ch["n_inst"][si]) == tup(evals[off + w + si])
        off += w + ch["n_sel"]
    final_claim = po.sumcheck_expected_from_evals(nvs, coeffs, terms, max_nv, rt, evals)
    assert expected == final_claim and claimed == po.recover_claim_from_final(final_claim, msgs, rt)
    if max_nv <= 14:  # the oracle's prover from round 0
        full = []
        for ch in chips:
            full += [m.download() for m in ch["cols"]]
            full += [po.selector_compute(po.SEL_PREFIX, ch["point"], 0, n) for n in ch["n_inst"]]
        t2 = po.StubTranscript(5)
        t2.append_label(b"combine subset evals")
        t2.sample_ext()
        omsgs, ochal, ofin = po.sumcheck_prove(full, coeffs, terms, max_nv, D, t2)
        assert np.array_equal(omsgs, msgs) and np.array_equal(ochal, rt) and np.array_equal(ofin, evals)
    else:       # the last 12 rounds on the oracle's prover, from tables folded by the independent fix_variables kernel
        keep = 12
        cut = max_nv - keep
        if all(ch["nv"] > cut for ch in chips):
            tables = []
            for ch in chips:
                sel_tabs = [dev.selector_build(po.SEL_PREFIX, ch["point"], 0, n) for n in ch["n_inst"]]
                for m in ch["cols"] + sel_tabs:
                    t = m.fix_variables(rt[:cut]).download()
                    if t.ndim == 1:
                        t = np.stack([t, np.zeros_like(t)], axis=1)
                    tables.append(np.ascontiguousarray(t))
                for m in sel_tabs:
                    m.free()
            omsgs, _, ofin = po.sumcheck_prove(tables, coeffs, terms, keep, D, po.ReplayTranscript(rt[cut:]))
            assert np.array_equal(omsgs, msgs[cut:]) and np.array_equal(ofin, evals)
    for ch in chips:
        for m in ch["cols"]:
            m.free()


def test_linear_only_columns_are_combined_and_evaluated_afterwards(dev, prover, monkeypatch):
    """prove_batched_main_constraints with columns that occur in `selector x column` monomials only (host/main_constraints.cpp): a column
    under ONE selector, under TWO selectors (it leaves the plan only if both combine it), one that also stands in a product (it stays), a
    selector with fewer linear columns than the threshold (its columns stay, and so does a column it shares with a larger group), several
    monomials of one column under one selector (coefficients add), chips of 2^3 .. 2^12 rows, a chip without any.  The proof equals the
    oracle prover's word for word with the combination on (default threshold 3, threshold 1) and off."""
    gch = [(11, 22), (33, 44)]
    w = 9
    cases = [  # (num_vars, selectors [(kind, offset, n)], terms over columns 0..w-1 and selectors w, w+1, ...)
        (12, [(po.SEL_PREFIX, 0, 4000), (po.SEL_PREFIX, 5, 3000)],
         [[w, 0], [w, 1], [w, 2], [w, 3], [w, 3], [w + 1, 2], [w + 1, 4], [w + 1, 5], [w + 1, 6], [w, 7, 8], [w, 7], [w], [w + 1]]),
        (10, [(po.SEL_PREFIX, 0, 1000), (po.SEL_WHOLE, 0, 0)],
         [[w, 0], [w, 1], [w, 2], [w, 3], [w + 1, 3], [w + 1, 4], [w, 5, 6], [w, 7, 8, 5]]),   # selector w+1: two linear columns, one shared
        (3, [(po.SEL_PREFIX, 1, 6)], [[w, j] for j in range(w)] + [[w]]),
        (11, [(po.SEL_PREFIX, 0, 2047)], [[w, j, (j + 1) % w] for j in range(w)]),              # no linear column at all
        (9, [(po.SEL_ORDERED_SPARSE, 0, 100, (0, 2), 2)], [[w, j] for j in range(5)] + [[w, 5, 6]]),  # a selector that is no eq table
    ]
    jobs, tabs, terms_all, scal_all, nvs = [], [], [], [], []
    for c, (nv, sels, terms) in enumerate(cases):
        cols = [po.rand_base(1 << nv, 8100 + 31 * c + j) for j in range(w)]
        point = po.rand_ext(nv, 600 + c)
        sel_t = []
        for sid, sl in enumerate(sels):
            kind, off, n = sl[0], sl[1], sl[2]
            sp, snv = (tuple(sl[3]), sl[4]) if len(sl) > 3 else ((), 0)
            sel_t.append((kind, off, n, sid, sp, snv, point))
        scalars = [[((7 + 5 * t + c, 2 + t), [2 + (t % 2)])] + ([((3 + t, 0), [t % 2, 2])] if t % 4 == 1 else []) for t in range(len(terms))]
        jobs.append(dict(num_vars=nv, mles=[dev.upload(t) for t in cols] + [None] * len(sels), n_witin=w, n_fixed=0, n_structural=len(sels),
                         selectors=sel_t, n_exprs=2, max_degree=max(len(t) for t in terms), terms=terms, scalars=scalars))
        start = len(nvs)
        nvs += [nv] * (w + len(sels))
        tabs += cols + [po.selector_compute(k, point, off, n, sp, snv) for (k, off, n, _sid, sp, snv, _pt) in sel_t]
        terms_all += [[start + j for j in t] for t in terms]
        scal_all.append(scalars)
    max_nv, D = max(nvs), max(j["max_degree"] for j in jobs)
    t2 = po.StubTranscript(5)
    t2.append_label(b"combine subset evals")
    a = t2.sample_ext()
    pows = [e2_pow(a, i) for i in range(2 * len(jobs))]
    coeffs = []
    for c, sc in enumerate(scal_all):
        coeffs += oracle_scalars(sc, gch + pows[2 * c: 2 * c + 2])
    omsgs, ochal, ofin = po.sumcheck_prove(tabs, po.ext(coeffs), terms_all, max_nv, D, t2)
    for mode in (None, "1", "0", "4"):
        if mode is None:
            monkeypatch.delenv("CENO_PROVER_MAIN_LINCOMB", raising=False)
        else:
            monkeypatch.setenv("CENO_PROVER_MAIN_LINCOMB", mode)
        claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(5))
        assert np.array_equal(omsgs, msgs) and np.array_equal(ochal, rt) and np.array_equal(ofin, evals), mode
        vpoint, expected = po.sumcheck_verify(claimed, msgs, po_stub_after_alpha())
        assert np.array_equal(vpoint, rt)


def po_stub_after_alpha():
    vt = po.StubTranscript(5)
    vt.append_label(b"combine subset evals")
    vt.sample_ext()
    return vt


@pytest.mark.parametrize("max_nv,n_chips,reps", [(18, 2, 60), (14, 48, 40)])
def test_wide_batched_main_is_the_same_in_every_run(dev, prover, max_nv, n_chips, reps):
    """Regression (round 5): the 16-byte stores that hand boundary values and component sums to the host are inline assembly, and a VMEM store
    of more than 64 bits reads its data registers late — without wait states inside the statement the compiler's next instruction could
    overwrite them (gfx940+: two wait states; the hazard recogniser does not look inside asm).  Under the load of an oversubscribed launch the
    first words (c0 of the first slots) of a boundary value went out corrupted in up to 87 % of the runs of this two-chip batch.  Every run
    must produce the words of the first."""
    from ceno_amd import synthetic

    jobs, chips, _ = synthetic.wide_batched_jobs(dev, max_nv, n_chips)
    mj = prover.MainJobs(jobs)
    gch = [(11, 22), (33, 44)]
    first = None
    for _ in range(reps):
        out = prover.prove_batched_main_constraints(dev, mj, gch, prover.Transcript.stub(5))
        if first is None:
            first = out
            continue
        assert out[0] == first[0] and np.array_equal(out[1], first[1]) and np.array_equal(out[3], first[3])
    for ch in chips:
        for m in ch["cols"]:
            m.free()


# ------------------------------------------------------------------------------------------------------------------
# multi-layer GKR circuit (a13): rotation argument first, then zerocheck -> linear -> sumcheck layers
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gkr_prove_multi_layer_with_rotation_first_matches_oracle(dev, prover, seed):
    """the order of the keccak-style harness (precompiles/lookup_keccakf.rs:1338-1397): prove_rotation at the out point, its
    left / right / target evaluations become out-evaluations of the first layer, then GKRCircuit::prove walks the layers
    (gkr.rs:72-115): a zerocheck layer with four selector groups (Prefix, Whole, OrderedSparse, a second Prefix at three
    different points), a linear layer, a plain sumcheck layer; claims flow through Single / Linear / Zero expressions"""
    import random

    rng = random.Random(77 + seed)
    log2_n, rot = 2, 5
    nv = log2_n + rot
    gch = [(rng.randrange(P), rng.randrange(P)) for _ in range(2)]
    pub_io = [(rng.randrange(P), 0), (rng.randrange(P), rng.randrange(P))]
    src = po.rand_base(1 << nv, 900 + seed)
    w = [src, po.rotation_next_base_mle(src, 5), po.rand_base(1 << nv, 910 + seed), po.rand_ext(1 << nv, 920 + seed)]
    v = [po.rand_base(1 << nv, 930 + seed), po.rand_ext(1 << nv, 931 + seed)]
    u = [po.rand_ext(1 << nv, 940 + seed + j) for j in range(3)]
    rt = po.rand_ext(nv, 950 + seed)
    d_w = [dev.upload(t) for t in w]
    # ---- rotation first ----
    pairs = [(0, 1), (2, 0)]
    tr_g, tr_o = prover.Transcript.stub(31 + seed), po.StubTranscript(31 + seed)
    got_rot = prover.prove_rotation(dev, d_w, pairs, 23, 5, rt, tr_g)
    exp_rot = po.prove_rotation(w, pairs, 23, 5, rt, tr_o)
    for a, b in zip(got_rot, exp_rot):
        assert np.array_equal(a, b)
    _, rev, origin, left, right = exp_rot
    # ---- claims: slot 0 = the tower-style claim at rt, 1..3 = rotation evaluations at their points ----
    n_ev = 12
    claims = [(rt, (rng.randrange(P), rng.randrange(P))), (left, tup(rev[0])), (right, tup(rev[1])), (origin, tup(rev[2]))] + [(None, (0, 0))] * (n_ev - 4)
    sel = lambda kind, sid, off, n, sparse=(), snv=0: (kind, sid, off, n, tuple(sparse), snv)
    s0, s1, s2, s3 = 4, 5, 6, 7  # structural slots of the four selectors inside layer 0's table list
    layer0 = dict(type=po.LAYER_ZEROCHECK, num_vars=nv, n_witin=4, n_fixed=0, n_structural=4,
                  groups=[(sel(po.SEL_PREFIX, 0, 0, (1 << nv) - 9), [("single", 0), ("linear", 0, (3, 1), (5, 0)), ("zero",)]),
                          (sel(po.SEL_WHOLE, 1, 0, 1 << nv), [("single", 1)]),
                          (sel(po.SEL_ORDERED_SPARSE, 2, 0, 3, (0, 2, 5, 17), 5), [("single", 2)]),
                          (sel(po.SEL_PREFIX, 3, 4, 40), [("single", 3)])],
                  n_exprs=4, max_degree=4,
                  terms=[[s0, 0, 1], [s0, 2], [s1, 1, 3], [s2, 0, 0, 2], [s3, 3], [s3, 1, 2]],
                  scalars=[[((1, 0), [2])], [((2, 0), [3, 0]), ((7, 1), [6])], [((1, 1), [4])], [((5, 0), [5, 7])], [((1, 0), [2, 1])], [((9, 9), [3])]],
                  in_eval_pos=[4, 5, 6, 7])
    layer1 = dict(type=po.LAYER_LINEAR, num_vars=nv, n_witin=2, n_fixed=0, n_structural=0, groups=[(None, [("single", 5)])], in_eval_pos=[8, 9])
    layer2 = dict(type=po.LAYER_SUMCHECK, num_vars=nv, n_witin=3, n_fixed=0, n_structural=0, groups=[(None, [("linear", 9, (2, 0), (1, 1))])], max_degree=3,
                  terms=[[0, 1, 2], [1], [2, 2]], scalars=[[((1, 0), [])], [((4, 0), [0, 2])], [((1, 2), [1])]], in_eval_pos=[10, 11])
    o_layers = [dict(layer0, mles=w + [None] * 4), dict(layer1, mles=v), dict(layer2, mles=u)]
    g_layers = [dict(layer0, mles=d_w + [None] * 4), dict(layer1, mles=[dev.upload(t) for t in v]), dict(layer2, mles=[dev.upload(t) for t in u])]
    got, got_claims = prover.gkr_prove(dev, g_layers, nv, claims, pub_io, gch, tr_g)
    exp, exp_claims = po.gkr_prove(o_layers, claims, pub_io, gch, tr_o)
    for li, ((gm, ge, gp), (em, ee, ep)) in enumerate(zip(got, exp)):
        if o_layers[li]["type"] != po.LAYER_LINEAR:
            assert np.array_equal(gm, em), f"layer {li} messages"
        assert np.array_equal(ge, np.asarray(ee, dtype=np.uint64).reshape(-1, 2)), f"layer {li} evaluations"
        assert np.array_equal(gp, np.asarray(ep, dtype=np.uint64).reshape(-1, 2)), f"layer {li} point"
    for (gpnt, gev), (epnt, eev) in zip(got_claims, exp_claims):
        assert gev == eev and ((gpnt is None and epnt is None) or np.array_equal(gpnt, epnt))
    assert tr_g.sample_ext() == tr_o.sample_ext()  # both transcripts absorbed exactly the same stream
    # a linear layer's evaluations are the tables at the layer point (independent path: the evaluate kernel)
    assert tup(got[1][1][1]) == po.mle_evaluate(v[1], got[0][2])


# ------------------------------------------------------------------------------------------------------------------
# whole create_proof flow on a small synthetic shard (the shape of BASELINE config #5 / metric M2), verified end to end
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("lanes", [1, 3])
def test_shard_flow_end_to_end_verifies(dev, prover, lanes):
    """(lanes > 1: the chip proofs run concurrently on their own host threads and HIP streams, the reference's chip scheduler;
    forked transcripts make the result independent of the interleaving)
    commit -> challenges -> per-chip proofs on forked transcripts -> merged fork samples -> batched main sumcheck -> one
    opening of all traces (ceno_zkvm/src/scheme/prover.rs:319-611), replayed by a verifier assembled from the oracle's restated
    TowerVerify / sumcheck verifier / Basefold verifier on ONE transcript: any deviation in what the prover binds, or in
    which order, makes a later challenge differ and the verification fail"""
    from ceno_amd import synthetic

    log_rows = (9, 8, 7, 6, 6)
    w = 22
    flow = synthetic.ShardFlow(dev, prover, w=w, n_queries=20, pow_bits=8, log_rows=log_rows)
    res = flow.run(lambda: prover.Transcript.stub(0x5A), lambda: prover.Transcript.stub(0xF0), lanes=lanes)
    a = flow.artifacts
    assert res["total_ms"] > 0
    vt = po.StubTranscript(0x5A)
    for root in a["roots"]:
        vt.append_ext((int(root[0]), int(root[1])))
        vt.append_ext((int(root[2]), int(root[3])))
    alpha, beta = vt.sample_ext(), vt.sample_ext()
    assert (alpha, beta) == (a["alpha"], a["beta"])
    # ---- chip proofs: each on its own fork ----
    for i, r in enumerate(log_rows):
        proof = a["chip_proofs"][i]
        ft = po.StubTranscript(0xF0)
        ft.append_ext(alpha)
        ft.append_ext(beta)
        for v in (i, i, (1 << r) - 3, 0):  # append_field_element: the stub absorbs a "BASE" marker, then the value
            _stub_absorb(ft, 0x4241534500000000)
            _stub_absorb(ft, v)
        for e in list(proof.r_out_evals) + list(proof.w_out_evals) + list(proof.lk_out_evals):
            ft.append_ext(tup(e))
        op = po.TowerProof(proof.tower_num_vars, 2, 1)
        op.msgs[:] = proof.tower_msgs
        op.prod_evals[:] = proof.tower_prod_evals
        op.logup_evals[:] = proof.tower_logup_evals
        rc, vpoint, *_ = po.tower_verify(np.concatenate([proof.r_out_evals, proof.w_out_evals]), proof.lk_out_evals, [r + 2, r + 2, r + 3], op, ft)
        assert rc == 0 and np.array_equal(vpoint, proof.tower_point) and np.array_equal(proof.rt_main, proof.tower_point[-r:])
        assert ft.sample_ext() == a["fork_samples"][i]
    for s_ in a["fork_samples"]:
        vt.append_ext(s_)
    # ---- batched main sumcheck ----
    vt.append_label(b"combine subset evals")
    al = vt.sample_ext()
    pows = [e2_pow(al, k) for k in range(2 * len(log_rows))]
    coeffs, terms, nvs = [], [], []
    for i, r in enumerate(log_rows):
        start = len(nvs)
        nvs += [r] * (w + 1)
        coeffs += oracle_scalars(a["mscalars"], [alpha, beta] + pows[2 * i: 2 * i + 2])
        terms += [[start + j for j in t] for t in a["mterms"]]
    vpoint, expected = po.sumcheck_verify(a["claimed"], a["msgs"], vt)
    assert np.array_equal(vpoint, a["rt"])
    final_claim = po.sumcheck_expected_from_evals(nvs, po.ext(coeffs), terms, max(log_rows), a["rt"], a["evals"])
    assert expected == final_claim
    for i, r in enumerate(log_rows):  # the selector evaluations the verifier computes itself (selector.rs:247-363)
        assert po.selector_evaluate(po.SEL_PREFIX, a["chip_proofs"][i].rt_main, a["rt"][:r], 0, (1 << r) - 3) == tup(a["evals"][i * (w + 1) + w])
    for e in a["evals"]:
        vt.append_ext(tup(e))
    # ---- opening of every trace at its own point ----
    shapes = [(r, w) for r in log_rows]
    assert po.basefold_verify(shapes, np.stack(a["roots"]), a["points"], a["open_evals"], 1, 20, 8, vt, a["open_proof"]) == 0
    flow.close()


@pytest.mark.gpu
def test_concurrent_lanes_produce_the_single_lane_proofs_every_time(dev, prover):
    """A soak of the chip scheduler: twelve chips of 2^8 .. 2^13 rows proved on one lane, then eight times over on four lanes (every lane
    beginning and releasing sumcheck handles, towers and record tables at the same moments: the pool's lock, the handles' arenas, the
    stream tags of recycled blocks).  Forked transcripts make a proof a function of its task alone: every word must come back
    identical, whatever the interleaving."""
    from ceno_amd import synthetic

    w = 22
    alpha, beta = (5, 6), (7, 8)
    coeffs, terms, out_terms = synthetic.record_plan(w, 16, alpha, beta)
    logs = (13, 8, 12, 9, 11, 10, 10, 11, 9, 12, 8, 13)
    cols = [[dev.synthetic(r, False, 0x700 + 37 * i + j) for j in range(w)] for i, r in enumerate(logs)]
    tasks = prover.ChipTasks([dict(circuit_idx=i, mles=cols[i], n_witin=w, n_fixed=0, n_structural=0, num_instances=(1 << r) - 3, log2_num_instances=r,
                                   num_reads=4, num_writes=4, num_lk_tables=0, num_lk=8, record_coeffs=coeffs, record_terms=terms,
                                   record_out_terms=out_terms) for i, r in enumerate(logs)])

    def run(lanes):
        forks = [prover.Transcript.stub(0xF0 + i) for i in range(len(logs))]
        proofs = prover.create_chip_proofs(dev, tasks, [alpha, beta], forks, lanes)
        return [(p.tower_msgs, p.tower_prod_evals, p.tower_logup_evals, p.tower_point, p.rt_main, p.r_out_evals, p.w_out_evals, p.lk_out_evals,
                 np.array(f.sample_ext(), dtype=np.uint64)) for p, f in zip(proofs, forks)]

    want = run(1)
    for rep in range(8):
        got = run(4)
        for i, (a, b) in enumerate(zip(want, got)):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (rep, i)


def _twelve_chips(d, prover, bad=None):
    from ceno_amd import synthetic

    w = 22
    alpha, beta = (5, 6), (7, 8)
    coeffs, terms, out_terms = synthetic.record_plan(w, 16, alpha, beta)
    logs = (13, 8, 12, 9, 11, 10, 10, 11, 9, 12, 8, 13)
    cols = [[d.synthetic(r, False, 0x700 + 37 * i + j) for j in range(w)] for i, r in enumerate(logs)]
    tasks = [dict(circuit_idx=i, mles=cols[i], n_witin=w, n_fixed=0, n_structural=0, num_instances=(1 << r) - 3, log2_num_instances=r,
                  num_reads=4, num_writes=4, num_lk_tables=0, num_lk=8, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
             for i, r in enumerate(logs)]
    if bad is not None:
        tasks[bad]["log2_num_instances"] += 1   # (the witness tables are shorter than the task says: utils.rs:713-723 refuses it)
    return prover.ChipTasks(tasks), cols, (alpha, beta), len(logs)


def _proofs_words(proofs, forks):
    return [None if p is None else (p.tower_msgs.tolist(), p.tower_prod_evals.tolist(), p.tower_logup_evals.tolist(), p.tower_point.tolist(),
                                    np.asarray(p.rt_main).tolist(), list(f.sample_ext())) for p, f in zip(proofs, forks)]


@pytest.mark.gpu
def test_one_bad_task_fails_alone_in_the_cohort_phase(dev, prover):
    """twelve chips through ceno_prover_create_chip_proofs (the cohort path: records, towers and tower layers of all chips in shared launches), one of
    them with a task its tables do not fit: that task's status says so, the other eleven proofs are the words of a run without it, nothing stays
    allocated, and the next run on the context is unharmed"""
    good, cols_g, (alpha, beta), n = _twelve_chips(dev, prover)
    forks = [prover.Transcript.stub(0xF0 + i) for i in range(n)]
    want = _proofs_words(prover.create_chip_proofs(dev, good, [alpha, beta], forks, 4), forks)
    dev.sync()
    base = dev.mem_info()["pool_used"]
    bad_tasks, cols_b, _, _ = _twelve_chips(dev, prover, bad=5)
    base_b = dev.mem_info()["pool_used"]
    forks = [prover.Transcript.stub(0xF0 + i) for i in range(n)]
    st = []
    got = _proofs_words(prover.create_chip_proofs(dev, bad_tasks, [alpha, beta], forks, 4, statuses=st), forks)
    assert st[5] != 0 and all(x == 0 for i, x in enumerate(st) if i != 5), st
    assert got[5] is None
    for i in range(n):
        if i != 5:
            assert got[i] == want[i], i
    dev.sync()
    assert dev.mem_info()["pool_used"] == base_b, "the failed run left device memory allocated"
    with pytest.raises(Exception):
        prover.create_chip_proofs(dev, bad_tasks, [alpha, beta], [prover.Transcript.stub(0xF0 + i) for i in range(n)], 4)   # (without statuses: raises)
    for c in cols_b:
        for m in c:
            m.free()
    forks = [prover.Transcript.stub(0xF0 + i) for i in range(n)]
    assert _proofs_words(prover.create_chip_proofs(dev, good, [alpha, beta], forks, 4), forks) == want
    for c in cols_g:
        for m in c:
            m.free()
    dev.sync()
    assert dev.mem_info()["pool_used"] <= base


@pytest.mark.gpu
def test_chip_proofs_fall_back_to_lanes_when_the_phase_cannot_be_booked(dev, prover):
    """the cohort path keeps every chip's towers resident and books the whole phase at once; on a context whose pool cannot promise that (pool_bytes)
    the phase runs chip by chip on the lanes, each task booked by itself (scheduler.rs:342-347,622-652) — same proofs"""
    from ceno_amd import Device

    tasks, cols, (alpha, beta), n = _twelve_chips(dev, prover)
    forks = [prover.Transcript.stub(0xF0 + i) for i in range(n)]
    want = _proofs_words(prover.create_chip_proofs(dev, tasks, [alpha, beta], forks, 4), forks)
    for c in cols:
        for m in c:
            m.free()
    d = Device(0, pool_bytes=1 << 30)   # the phase's booking is the sum of the estimates plus 1 GiB of cohort scratch: refused
    tasks, cols, _, _ = _twelve_chips(d, prover)
    d.L.ceno_hip_mem_booked_peak(d.h, 1)
    forks = [prover.Transcript.stub(0xF0 + i) for i in range(n)]
    got = _proofs_words(prover.create_chip_proofs(d, tasks, [alpha, beta], forks, 4), forks)
    assert got == want
    assert 0 < int(d.L.ceno_hip_mem_booked_peak(d.h, 0)) < (1 << 30)   # per-task bookings, never the whole phase
    for c in cols:
        for m in c:
            m.free()
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_chips", [1, 3])
def test_fewer_chips_than_the_threshold_forced_through_cohorts_write_the_same_proofs(dev, prover, monkeypatch, n_chips):
    """CENO_TOWER_COHORT_MIN_TASKS (default 8: a few chips have a lane each anyway, tools/dev/cohort_one_chip.py) = 1: one chip, and three, through
    the cohort launches — a chip per serving thread, a launch of one or a few jobs, groups of sub-cubes of a single chip — write the words the
    per-chip tower prover writes (which test_create_chip_proof_matches_oracle holds against the oracle)"""
    from ceno_amd import synthetic

    w = 22
    alpha, beta = (5, 6), (7, 8)
    coeffs, terms, out_terms = synthetic.record_plan(w, 16, alpha, beta)
    logs = (13, 9, 11)[:n_chips]
    cols = [[dev.synthetic(r, False, 0x900 + 41 * i + j) for j in range(w)] for i, r in enumerate(logs)]
    tasks = prover.ChipTasks([dict(circuit_idx=i, mles=cols[i], n_witin=w, n_fixed=0, n_structural=0, num_instances=(1 << r) - 3, log2_num_instances=r,
                                   num_reads=4, num_writes=4, num_lk_tables=0, num_lk=8, record_coeffs=coeffs, record_terms=terms,
                                   record_out_terms=out_terms) for i, r in enumerate(logs)])
    forks = [prover.Transcript.stub(0xF0 + i) for i in range(n_chips)]
    want = _proofs_words(prover.create_chip_proofs(dev, tasks, [alpha, beta], forks, 2), forks)
    monkeypatch.setenv("CENO_TOWER_COHORT_MIN_TASKS", "1")
    for last in ("19", "9"):
        monkeypatch.setenv("CENO_TOWER_COHORT_LAYERS", last)
        forks = [prover.Transcript.stub(0xF0 + i) for i in range(n_chips)]
        got = _proofs_words(prover.create_chip_proofs(dev, tasks, [alpha, beta], forks, 2), forks)
        assert got == want, f"cohort layers to {last}"
    for c in cols:
        for m in c:
            m.free()


def _stub_absorb(t, word):
    """one absorb step of the SplitMix stub transcript (oracle/oracle.c orc_stub_*; host/transcript.cpp Stub::absorb)"""
    M = (1 << 64) - 1
    z = ((t.state.s ^ word) + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    t.state.s = z ^ (z >> 31)


def test_pool_cache_does_not_grow_over_repeated_flows_with_lanes(prover):
    """The pool hands a block to another stream only once the stream that used it last has drained; blocks freed after a
    synchronisation must become free for everybody, or a flow that alternates streams (two-stream commit, lanes) allocates a
    share of its footprint afresh on every run (regression: +230 MB of cache per shard flow)"""
    from ceno_amd import Device, synthetic

    d = Device(0)
    flow = synthetic.ShardFlow(d, prover, w=22, n_queries=10, pow_bits=4, log_rows=(14, 13, 12, 11, 11))
    tr, fk = (lambda: prover.Transcript.stub(0x5A)), (lambda: prover.Transcript.stub(0xF0))
    for it in range(6):
        flow.run(tr, fk, lanes=(1, 3)[it % 2])
    base = d.mem_info()["pool_cached"]
    for it in range(16):
        flow.run(tr, fk, lanes=(1, 3)[it % 2])
    grown = d.mem_info()["pool_cached"] - base
    flow.close()
    d.close()
    assert grown <= base // 4 + (8 << 20), (base, grown)


@pytest.mark.parametrize("log_rows,n_reads,n_writes,n_lk", [(12, 4, 4, 8), (16, 1, 1, 1), (18, 4, 4, 8), (17, 2, 3, 5)])
def test_chip_proof_booking_estimate_covers_the_pool_high_water_mark(prover, log_rows, n_reads, n_writes, n_lk):
    """what the lane scheduler books for a chip proof (ceno_prover_chip_proof_estimate_bytes) against what the proof really takes from
    the pool (ceno_hip_mem_peak): the reference asserts its estimator against actual usage (scheme/gpu/memory.rs:54-145); here
    high-water <= estimate <= 2 x high-water + 8 MiB on chips of different heights and record counts"""
    from ceno_amd import Device, synthetic

    d = Device(0)
    w = 12
    cols = [d.synthetic(log_rows, False, 300 + j) for j in range(w)]
    alpha, beta = (5, 6), (7, 8)
    n_rec = n_reads + n_writes + n_lk
    coeffs, terms, out_terms = synthetic.record_plan(w, n_rec, alpha, beta)
    task = dict(mles=cols, n_witin=w, n_fixed=0, n_structural=0, num_instances=(1 << log_rows) - 3, log2_num_instances=log_rows, num_reads=n_reads,
                num_writes=n_writes, num_lk_tables=0, num_lk=n_lk, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
    est = prover.chip_proof_estimate_bytes(task)
    prover.create_chip_proof(d, task, [alpha, beta], prover.Transcript.stub(3))   # warm: code objects, pinned blocks
    d.sync()
    base = d.mem_info()["pool_used"]
    d.L.ceno_hip_mem_peak(d.h, 1)
    prover.create_chip_proof(d, task, [alpha, beta], prover.Transcript.stub(3))
    d.sync()
    peak = int(d.L.ceno_hip_mem_peak(d.h, 0)) - base
    print(f"chip 2^{log_rows} x ({n_reads},{n_writes},{n_lk}): high-water {peak / 2**20:.1f} MiB, estimate {est / 2**20:.1f} MiB, ratio {est / max(peak, 1):.2f}")
    assert peak <= est <= 2 * peak + (8 << 20), (peak, est)
    for m in cols:
        m.free()
    d.close()


def test_pool_limit_with_a_full_cache_does_not_fail_lanes(prover):
    """pool_bytes set, the cache full of parked blocks (what a large batch leaves behind) and a pipelined sumcheck alive (another
    lane proving): nothing can go back to the driver then (the trim gate), and `used + cached + request > limit` used to be
    reported as 'pool capacity exceeded' although the cache held everything that was needed (round-3 advisor finding).  A request
    now takes a larger idle block of the cache; failing that, a thread with no pipelined sumcheck of its own waits for the gate
    and trims, and a thread that has one gets a retryable error instead of a deadlock."""
    import threading

    from ceno_amd import Device, api
    from ceno_amd.api import CenoHipError

    limit = 1 << 30
    d = Device(0, pool_bytes=limit)
    parked = [d.synthetic(24, False, 700 + i) for i in range(6)]     # 6 x 128 MiB of the 1 GiB ...
    d.sync()
    for m in parked:
        m.free()                                                     # ... parked in the cache
    assert d.mem_info()["pool_cached"] >= 6 << 27
    nv = 16
    tabs = [d.synthetic(nv, True, 40 + j) for j in range(3)]
    sc = api.Sumcheck(d, tabs, po.ext([1]), [[0, 1, 2]], nv, 3)
    sc.set_pipelined(True)
    sc.round()                                                        # a pipelined sumcheck is alive from here on
    # (1) same thread, a larger idle block exists: a 100 MiB request is served by a parked 128 MiB block
    m1 = d.synthetic(23, True, 1)                                     # 128 MiB exactly: the ordinary reuse
    m2 = d.alloc(23, False)                                           # 64 MiB: no block of that size; 704 + 64 MiB fits under the limit
    big = d.mem_info()
    assert big["pool_used"] + big["pool_cached"] <= limit
    small = [d.alloc(23, False) for _ in range(3)]                    # 3 x 64 MiB more: over the limit with the cache counted in
    assert d.mem_info()["pool_used"] + d.mem_info()["pool_cached"] <= limit  # ... served from the parked 128 MiB blocks
    # (2) same thread, nothing large enough: retryable failure, not a hang
    with pytest.raises(CenoHipError) as ei:
        d.alloc(25, False)                                            # 256 MiB: larger than any parked block
    assert "retry" in str(ei.value)
    # (3) another thread (no pipelined sumcheck of its own) waits for the gate, trims and gets its block once this one is done
    got = {}

    def worker():
        try:
            got["m"] = d.alloc(25, False)
        except Exception as e:  # noqa: BLE001
            got["err"] = e

    t = threading.Thread(target=worker)
    t.start()
    t.join(0.3)
    assert t.is_alive(), "the other thread should be waiting for the trim gate"
    sc.free()                                                         # aborts the queued rounds: no pipelined sumcheck is alive any more
    t.join(20)
    assert not t.is_alive() and "m" in got, got.get("err")
    for m in [m1, m2, got["m"]] + small + tabs:
        m.free()
    d.close()


def test_trim_waiter_is_not_starved_by_back_to_back_sumchecks(prover):
    """(round-4 advisor finding) lanes that prove small pipelined sumchecks back to back rarely leave an instant with NO pipelined sumcheck
    alive, and nothing held new ones back while a thread waited for the trim gate: an over-the-limit request from a thread with nothing
    in flight starved until its time-out and reported a spurious OOM.  A waiting trimmer now holds back the threads that have nothing
    in flight; the request is served while the lanes keep proving."""
    import threading
    import time

    from ceno_amd import Device

    limit = 1 << 30
    d = Device(0, pool_bytes=limit)
    parked = [d.synthetic(24, False, 700 + i) for i in range(6)]     # 6 x 128 MiB parked in the cache of a 1 GiB pool
    d.sync()
    for m in parked:
        m.free()
    stop = threading.Event()
    count = [0, 0, 0, 0]
    errs = []

    def lane(t):
        try:
            st = d.stream_create()
            tabs = [d.synthetic(10, True, 40 + 7 * t + j) for j in range(3)]
            while not stop.is_set():
                prover.sumcheck_prove(d, tabs, po.ext([1]), [[0, 1, 2]], 10, 3, prover.Transcript.stub(t), stream=st)
                count[t] += 1
            for m in tabs:
                m.free()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=lane, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    time.sleep(0.5)
    before = sum(count)
    t0 = time.perf_counter()
    big = d.alloc(25, False)   # 256 MiB: larger than any parked block -> has to trim, i.e. needs the gate
    waited = time.perf_counter() - t0
    time.sleep(0.3)
    stop.set()
    for th in ths:
        th.join(60)
    assert not errs, errs
    assert waited < 1.0, f"the request waited {waited:.2f} s for the trim gate"
    assert before > 20 and sum(count) > before, "the lanes must keep proving after the trim"
    big.free()
    d.close()


def test_pipelined_handle_released_on_another_thread(prover):
    """(round-4 advisor finding) the count of pipelined sumchecks a thread holds belongs to the thread that BEGAN them: a handle released
    on another thread must neither leave the beginner unable to wait for the trim gate for ever nor decrement the releaser's count"""
    import ctypes as C
    import threading

    from ceno_amd import Device, api

    limit = 1 << 29
    d = Device(0, pool_bytes=limit)
    tabs = [d.synthetic(14, True, 40 + j) for j in range(3)]
    sc = api.Sumcheck(d, tabs, po.ext([1]), [[0, 1, 2]], 14, 3)
    sc.set_pipelined(True)
    sc.round()                                   # this thread holds one pipelined sumcheck
    th = threading.Thread(target=sc.free)        # ... released by another thread
    th.start()
    th.join(30)
    assert not th.is_alive()
    live, mid = C.c_int(-1), C.c_int(-1)
    d.check(d.L.ceno_hip_debug_state(d.h, C.byref(live), C.byref(mid)))
    assert (live.value, mid.value) == (0, 0)
    # the beginner's count is back at zero: with the cache full, an over-the-limit request of THIS thread waits for the gate (nothing is
    # alive: immediately) and trims instead of failing with 'retry'
    parked = [d.synthetic(24, False, 700 + i) for i in range(3)]     # 3 x 128 MiB parked
    d.sync()
    for m in parked:
        m.free()
    big = d.alloc(25, False)                     # 256 MiB: only possible after a trim
    big.free()
    for m in tabs:
        m.free()
    d.close()


def test_cache_over_the_soft_cap_does_not_stall_lanes(prover):
    """A phase that leaves many GB in the pool's cache (the 13 GB batch of config #4) followed by concurrent chip proofs: the
    pool's soft cap must not call hipFree — which waits for every stream of the device — while round kernels of the lanes are
    waiting for their host threads (regression: two lanes inside hipFree waited for each other's kernels until those gave up
    after their 60 s poll timeout: 'sumcheck round finished without publishing its message')"""
    import time

    from ceno_amd import Device, synthetic

    d = Device(0)
    big = [d.synthetic(27, False, 900 + i) for i in range(10)]   # 10 x 1 GiB ...
    d.sync()
    for m in big:
        m.free()                                                 # ... parked in the cache, untagged (the stream is idle)
    assert d.mem_info()["pool_cached"] >= 10 << 30
    flow = synthetic.ShardFlow(d, prover, w=22, n_queries=10, pow_bits=4, log_rows=(16, 15, 14, 13, 13, 12))
    t0 = time.perf_counter()
    for _ in range(3):
        flow.run(lambda: prover.Transcript.poseidon2(b"riscv"), lambda: prover.Transcript.poseidon2(b"fork"), lanes=4)
    assert time.perf_counter() - t0 < 30
    flow.close()
    # every handle is gone: the bookkeeping the trim and the k_mid budget depend on is back at zero (a counter that leaks upwards
    # would silently switch the trim off for good)
    import ctypes as C

    live, mid = C.c_int(-1), C.c_int(-1)
    d.check(d.L.ceno_hip_debug_state(d.h, C.byref(live), C.byref(mid)))
    assert (live.value, mid.value) == (0, 0)
    # ... and the parked 10 GB did go back to the driver in the moments when no pipelined sumcheck was alive
    assert d.mem_info()["pool_cached"] < (9 << 30)
    d.close()
