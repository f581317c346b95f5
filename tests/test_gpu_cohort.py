"""GPU parity of the cohort kernel (ceno_amd/csrc/tower_cohort.hip): many tower-layer sumchecks in ONE launch, one workgroup per job, each served
through its own pair of mailboxes — against the oracle's sumcheck prover of the same plan
  sum_x eq(x, rt) [ sum_i alpha_i a_i b_i + sum_k (an_k (p1 q2 + p2 q1) + ad_k q1 q2) ]     (CpuTowerProver::create_proof's layer sumcheck,
ceno_zkvm/src/scheme/cpu/mod.rs:417-494), message by message and evaluation by evaluation; and the sub-cube split of a layer of more than
2^13 entries (partial messages scaled by eq over the high variables add up to the whole layer's messages)."""
import ctypes as C

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


class JobC(C.Structure):
    _fields_ = [("tables", C.POINTER(C.c_void_p)), ("n_prod", C.c_int), ("n_logup", C.c_int), ("n", C.c_int), ("rt", po.u64p),
                ("alpha_prod", po.u64p), ("alpha_num", po.u64p), ("alpha_den", po.u64p), ("share_mailbox_of", C.c_int), ("scale", po.u64p), ("claim", po.u64p)]


def _oracle_layer(tabs, n, np_, nl, rt, a_prod, a_num, a_den, seed):
    """(messages (n, 3, 2), challenges (n, 2), final evaluations (K, 2)) of the layer sumcheck on the oracle, stub transcript"""
    eq = po.build_eq(rt)
    tables = [eq] + tabs
    coeffs, terms, m = [], [], 1
    for t in range(np_):
        coeffs.append(tuple(int(x) for x in a_prod[t]))
        terms.append([0, m, m + 1])
        m += 2
    for k in range(nl):
        an, ad = tuple(int(x) for x in a_num[k]), tuple(int(x) for x in a_den[k])
        coeffs += [an, an, ad]
        terms += [[0, m, m + 3], [0, m + 1, m + 2], [0, m + 2, m + 3]]
        m += 4
    msgs, chal, fin = po.sumcheck_prove(tables, po.ext(coeffs), terms, n, 3, po.StubTranscript(seed))
    # the sum the sumcheck proves (the cohort kernel takes it: it reports two values per round, the third comes from the running claim)
    final = po.sumcheck_expected_from_evals([n] * len(tables), po.ext(coeffs), terms, n, chal, fin)
    _oracle_layer.claim = po.recover_claim_from_final(final, msgs, chal)
    return msgs, chal, fin


def _run_cohort(dev, cases, group_scales=None):
    """cases: [(n, np, nl, host tables [2^n x 2], rt (n, 2), a_prod, a_num, a_den, challenges (n, 2))] -> [(msgs, fin)] served round-robin.
    group_scales (one ext per case): the cases are ONE group led by case 0 — one mailbox, one message per round (the device's sum of the
    cases' messages times their scales), returned as case 0's; the final evaluations stay per case"""
    share_first_mailbox = group_scales is not None
    L = dev.L
    keep, jobs = [], (JobC * len(cases))()
    for j, (n, np_, nl, tabs, rt, a_prod, a_num, a_den, _, claim) in enumerate(cases):
        mles = [dev.upload(t) for t in tabs]
        ptrs = (C.c_void_p * len(mles))(*[m.device_ptr for m in mles])
        rt_c, ap, an, ad = (np.ascontiguousarray(x, dtype=np.uint64) for x in (rt, a_prod, a_num, a_den))
        keep += [mles, ptrs, rt_c, ap, an, ad]
        J = jobs[j]
        J.tables, J.n_prod, J.n_logup, J.n = ptrs, np_, nl, n
        J.rt, J.alpha_prod, J.alpha_num, J.alpha_den = (x.ctypes.data_as(po.u64p) for x in (rt_c, ap, an, ad))
        cl = np.array([int(claim[0]), int(claim[1])], dtype=np.uint64)
        keep.append(cl)
        J.claim = cl.ctypes.data_as(po.u64p)
        if share_first_mailbox:
            sc = np.ascontiguousarray(group_scales[j], dtype=np.uint64)
            keep.append(sc)
            J.share_mailbox_of, J.scale = 1, sc.ctypes.data_as(po.u64p)
    st = dev.stream_create()
    h = C.c_void_p()
    dev.check(L.ceno_hip_tower_cohort_begin(dev.h, C.cast(jobs, C.c_void_p), len(cases), st, C.byref(h)))
    msgs = [np.zeros((c[0], 3, 2), dtype=np.uint64) for c in cases]
    fins = [np.zeros((1 + 2 * c[1] + 4 * c[2], 2), dtype=np.uint64) for c in cases]
    rnd, done = [0] * len(cases), [False] * len(cases)
    spins = 0
    while not all(done):
        spins += 1
        assert spins < 50_000_000, "a job never answered"
        for j, c in enumerate(cases):
            if done[j]:
                continue
            if share_first_mailbox and j > 0:
                rnd[j] = rnd[0]      # (a member follows its leader: only the leader has messages and a mailbox)
            if rnd[j] < c[0]:
                if share_first_mailbox and j > 0:
                    continue
                out = np.zeros(6, dtype=np.uint64)
                got = L.ceno_hip_tower_cohort_try_message(h, j, rnd[j], out.ctypes.data_as(po.u64p))
                assert got >= 0
                if got == 1:
                    msgs[j][rnd[j]] = out.reshape(3, 2)
                    ch = np.ascontiguousarray(c[8][rnd[j]], dtype=np.uint64)
                    assert L.ceno_hip_tower_cohort_send_challenge(h, j, rnd[j], ch.ctypes.data_as(po.u64p)) == 0
                    rnd[j] += 1
            else:
                got = L.ceno_hip_tower_cohort_try_final(h, j, fins[j].ctypes.data_as(po.u64p))
                assert got >= 0
                done[j] = got == 1
    dev.check(L.ceno_hip_tower_cohort_end(dev.h, h))
    dev.stream_destroy(st)
    for k in keep:
        if isinstance(k, list):
            for m in k:
                m.free()
    return list(zip(msgs, fins))


def _case(n, np_, nl, seed):
    tabs = [po.rand_ext(1 << n, 7000 + 31 * seed + j) for j in range(2 * np_ + 4 * nl)]
    rt = po.rand_ext(n, 7100 + seed)
    a_prod, a_num, a_den = po.rand_ext(max(np_, 1), 7200 + seed), po.rand_ext(max(nl, 1), 7300 + seed), po.rand_ext(max(nl, 1), 7400 + seed)
    omsgs, ochal, ofin = _oracle_layer(tabs, n, np_, nl, rt, a_prod, a_num, a_den, 0xC0 + seed)
    return (n, np_, nl, tabs, rt, a_prod, a_num, a_den, ochal, _oracle_layer.claim), (omsgs, ofin)


def test_cohort_jobs_of_every_shape_match_the_oracle(dev):
    """jobs of 1 .. 13 variables, every tower shape a chip proof produces (read + write + lookups, a table circuit's LogUp alone, a missing write
    set, two LogUp towers), all in ONE launch, answered in whatever order they report"""
    assert dev.L.ceno_hip_tower_cohort_max_vars() == 13
    shapes = [(1, 2, 1), (2, 1, 0), (3, 0, 1), (5, 2, 1), (8, 1, 1), (9, 2, 1), (10, 0, 1), (11, 2, 1), (12, 3, 2), (13, 2, 1), (13, 0, 1), (7, 3, 0)]
    cases, wants = zip(*[_case(n, np_, nl, s) for s, (n, np_, nl) in enumerate(shapes)])
    got = _run_cohort(dev, list(cases))
    for (n, np_, nl), (msgs, fin), (omsgs, ofin) in zip(shapes, got, wants):
        assert np.array_equal(msgs, omsgs), (n, np_, nl, "messages")
        assert np.array_equal(fin, ofin), (n, np_, nl, "final evaluations")


def test_many_jobs_at_once_are_all_resident(dev):
    """432 jobs (54 chains x 8 sub-cubes) of 2^13 entries in one launch: every workgroup waits for its own host without holding up another"""
    case, want = _case(13, 2, 1, 99)
    got = _run_cohort(dev, [case] * 432)
    for msgs, fin in got:
        assert np.array_equal(msgs, want[0]) and np.array_equal(fin, want[1])


@pytest.mark.parametrize("r,sub,grouped", [(14, 13, False), (16, 13, False), (16, 13, True), (15, 9, True), (12, 11, True)])
def test_sub_cubes_of_a_large_layer_add_up(dev, r, sub, grouped):
    """a layer of 2^r entries as 2^(r - sub) jobs over its top-bit sub-cubes: with every job fed the WHOLE layer's challenges, the sum of the
    jobs' messages scaled by eq(g; rt_high) equals the layer's message in each of the first `sub` rounds — added by this test, or by the device
    when the jobs form a group — and the jobs' final evaluations are the tables the host finishes the last r - sub rounds on"""
    np_, nl = 2, 1
    tabs = [po.rand_ext(1 << r, 9000 + j) for j in range(2 * np_ + 4 * nl)]
    rt = po.rand_ext(r, 9100)
    a_prod, a_num, a_den = po.rand_ext(np_, 9200), po.rand_ext(nl, 9300), po.rand_ext(nl, 9400)
    omsgs, ochal, ofin = _oracle_layer(tabs, r, np_, nl, rt, a_prod, a_num, a_den, 0xE0)
    G = 1 << (r - sub)
    claim = _oracle_layer.claim   # (the whole layer's: what a GROUP's leader is given)
    cases = []
    for g in range(G):
        sub_tabs = [np.ascontiguousarray(t[g << sub: (g + 1) << sub]) for t in tabs]
        if not grouped:   # a job that stands alone proves its own sub-cube's sum
            _oracle_layer(sub_tabs, sub, np_, nl, rt[:sub], a_prod, a_num, a_den, 0xE1)
        cases.append((sub, np_, nl, sub_tabs, rt[:sub], a_prod, a_num, a_den, ochal[:sub], claim if grouped else _oracle_layer.claim))
    eq_hi = po.build_eq(rt[sub:])
    # grouped: one mailbox and one message per round for all sub-cubes, added up on the device (what host/cohort.cpp launches)
    got = _run_cohort(dev, cases, group_scales=[eq_hi[g] for g in range(G)] if grouped else None)
    for i in range(sub):
        for e in range(3):
            if grouped:
                acc = tuple(int(x) for x in got[0][0][i][e])
            else:
                acc = (0, 0)
                for g in range(G):
                    acc = po.e2_add(acc, po.e2_mul(tuple(int(x) for x in eq_hi[g]), tuple(int(x) for x in got[g][0][i][e])))
            assert acc == tuple(int(x) for x in omsgs[i][e]), (i, e)
    # the gathered tables: table m over the high variables = the jobs' final evaluations (eq: times eq_hi); folding them with the remaining
    # challenges gives the layer's final evaluations
    K = 1 + 2 * np_ + 4 * nl
    for m in range(K):
        tab = np.array([got[g][1][m] for g in range(G)], dtype=np.uint64)
        if m == 0:
            tab = np.array([po.e2_mul(tuple(int(x) for x in tab[g]), tuple(int(x) for x in eq_hi[g])) for g in range(G)], dtype=np.uint64)
        assert po.mle_evaluate(tab, ochal[sub:]) == tuple(int(x) for x in ofin[m]), m


def test_towers_built_many_at_a_time_equal_towers_built_one_by_one(dev):
    """ceno_hip_tower_build_many (level-synchronous launches over all towers: csrc/tower.hip) against ceno_hip_tower_build_prod / _logup tower
    by tower, every layer and limb: product towers of 1 .. 9 records, LogUp towers with and without numerators, base and extension records,
    instance counts that are no power of two, towers too small for a layer kernel and towers of 2^17 entries, all in one call"""
    from ceno_amd import prover

    shapes = [("prod", 5, 12, 4096), ("prod", 1, 3, 5), ("logup", 3, 10, 1000, True), ("logup", 20, 12, 4096, False), ("prod", 9, 13, 8191),
              ("logup", 1, 2, 3, False), ("prod", 2, 16, 65536), ("logup", 4, 9, 512, True), ("prod", 3, 1, 2), ("logup", 7, 14, 16000, False)]
    specs, keep = [], []
    for j, sh in enumerate(shapes):
        kind, k, nv, n_inst = sh[:4]
        recs = [dev.synthetic(nv, (j + r) % 3 != 0, 0xB00 + 37 * j + r) for r in range(k)]
        keep += recs
        if kind == "prod":
            specs.append(("prod", recs, n_inst, (1, 0)))
        else:
            nums = None
            if sh[4]:
                nums = [dev.synthetic(nv, True, 0xD00 + 41 * j + r) for r in range(k)]
                keep += nums
            specs.append(("logup", nums, recs, n_inst, (77 + j, 5)))
    st = dev.stream_create()
    many = prover.Tower.build_many(dev, specs, stream=st)
    for sp, got in zip(specs, many):
        want = prover.Tower.build_prod(dev, sp[1], sp[2], sp[3], stream=st) if sp[0] == "prod" else prover.Tower.build_logup(dev, sp[1], sp[2], sp[3], sp[4], stream=st)
        assert (got.num_vars, got.num_limbs) == (want.num_vars, want.num_limbs)
        for layer in range(want.num_vars):
            for limb in range(want.num_limbs):
                assert np.array_equal(got.layer(layer, limb), want.layer(layer, limb)), (sp[0], layer, limb)
        want.free()
        got.free()
    dev.stream_destroy(st)
    for m in keep:
        m.free()
