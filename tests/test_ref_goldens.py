"""Parity against goldens dumped from the REAL reference (tools/goldens/dump_goldens.rs -> tests/golden/ref_goldens.json).

The reference's arithmetic lives in un-vendored Rust dependencies and this image has no Rust toolchain, so the file cannot
be produced here.  Until somebody runs the dump where cargo exists, every test below XFAILS with the reason
"PARITY UNPINNED" — visible in the report, not a silent skip.  With the file present the tests load the reference's
Poseidon2 table into the host challenger / the oracle / the device and compare, bit for bit: the extension product (W),
the permutation, label packing, a BasicTranscript script, a complete sumcheck proof, and report how the Basefold root
relates to ours.  `test_kit_plumbing_on_self_generated_file` runs the same checks on a file generated from this repository's
own (placeholder) implementation, so the plumbing itself is tested on every run.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.environ.get("CENO_REF_GOLDENS", os.path.join(ROOT, "tests", "golden", "ref_goldens.json"))
P = po.P
UNPINNED = ("PARITY UNPINNED: tests/golden/ref_goldens.json is absent - it has to be dumped from the reference with "
            "tools/goldens/dump_goldens.rs on a machine with cargo (see tools/goldens/README.md); W = 7, the Poseidon2-Goldilocks "
            "table, label packing, the BasicTranscript stream and the Basefold layout are checked against nothing until then")


def _load(path=None):
    path = path or GOLDEN
    if not os.path.exists(path):
        pytest.xfail(UNPINNED)
    return json.load(open(path))


@pytest.fixture(scope="module")
def plib():
    from ceno_amd import build, prover

    build.build_all()
    L = prover.plib()
    L.ceno_transcript_poseidon2_set_constants.restype = C.c_int
    L.ceno_transcript_poseidon2_set_constants.argtypes = [po.u64p, po.u64p, po.u64p]
    L.ceno_prover_test_label_to_field.restype = C.c_int
    L.ceno_prover_test_label_to_field.argtypes = [C.c_char_p, C.c_size_t, po.u64p, C.c_int]
    L.ceno_prover_test_poseidon2_permute_fast.restype = None
    L.ceno_prover_test_poseidon2_permute_fast.argtypes = [po.u64p]
    yield L
    L.ceno_transcript_poseidon2_set_constants(None, None, None)  # back to the placeholders for the rest of the session


def _constants(g):
    from ceno_amd import goldens

    return goldens.parse_constants(g)


def _oracle_params(ext, internal, diag):
    params = po.poseidon2_default_params().copy()
    params[:64] = ext.reshape(-1)
    params[64:86] = internal
    if diag is not None:
        params[86:94] = diag
    return params


def _install(L, g):
    ext, internal, diag = _constants(g)
    e, i = np.ascontiguousarray(ext.reshape(-1)), np.ascontiguousarray(internal)
    d = np.ascontiguousarray(diag) if diag is not None else None
    assert L.ceno_transcript_poseidon2_set_constants(po._p(e), po._p(i), po._p(d) if d is not None else None) == 0
    return _oracle_params(ext, internal, diag)


def check_ext_mul(L, g):
    L.ceno_prover_test_e2_mul.argtypes = [po.u64p] * 3
    L.ceno_prover_test_e2_inv.argtypes = [po.u64p] * 2
    a, b = np.array(g["ext_mul"]["a"], dtype=np.uint64), np.array(g["ext_mul"]["b"], dtype=np.uint64)
    o = np.zeros(2, dtype=np.uint64)
    L.ceno_prover_test_e2_mul(po._p(a), po._p(b), po._p(o))
    assert o.tolist() == g["ext_mul"]["ab"], "extension product differs: W of GoldilocksExt2 is not 7"
    assert list(po.e2_mul(tuple(int(x) for x in a), tuple(int(x) for x in b))) == g["ext_mul"]["ab"]
    L.ceno_prover_test_e2_inv(po._p(a), po._p(o))
    assert o.tolist() == g["ext_mul"]["a_inv"]


def check_permutation(L, g, params):
    for kat in g["poseidon2"]["kats"]:
        st = np.array(kat["in"], dtype=np.uint64)
        assert po.poseidon2_permute(st, params).tolist() == kat["out"], "oracle Poseidon2 != reference"
        got = st.copy()
        L.ceno_prover_test_poseidon2_permute_fast(po._p(got))
        assert got.tolist() == kat["out"], "host Poseidon2 != reference"


def check_labels(L, g):
    for item in g["bytes_to_field_elements"]:
        raw = item["label"].encode()
        out = np.zeros(16, dtype=np.uint64)
        n = L.ceno_prover_test_label_to_field(raw, len(raw), po._p(out), 16)
        assert out[:n].tolist() == item["elements"], f"label packing of {raw!r} differs"


def check_transcript(L, g):
    from ceno_amd import prover

    t = prover.Transcript.poseidon2(g["transcript"]["label"].encode())
    for step in g["transcript"]["script"]:
        op = step["op"]
        if op == "append_message":
            t.append_label(bytes(step["bytes"]))
        elif op == "append_ext":
            t.append_ext(tuple(step["value"]))
        elif op == "append_base":
            t.append_base(step["value"])
        elif op == "challenge":
            t.append_label(step["label"].encode())
            assert list(t.sample_ext()) == step["value"], "challenge differs at " + step["label"]
        elif op == "read_challenge":
            assert list(t.sample_ext()) == step["value"]
        elif op == "challenge_pows":
            t.append_label(b"combine subset evals")
            a = t.sample_ext()
            pows, acc = [], (1, 0)
            for _ in range(step["n"]):
                pows.append(list(acc))
                acc = po.e2_mul(acc, a)
            assert pows == step["value"]
        elif op == "sample_and_append_vec":
            t.append_label(step["label"].encode())
            assert [list(t.sample_ext()) for _ in range(step["n"])] == step["value"]
        elif op == "fork_sample":
            f = prover.Transcript.poseidon2(b"fork")
            f.append_ext(tuple(step["appended"]))
            assert list(f.sample_ext()) == step["value"]
        else:
            raise AssertionError("unknown script op " + op)


class _HostTranscriptForOracle:
    """the product's host Poseidon2 transcript behind the oracle's orc_transcript function table"""

    def __init__(self, label: bytes):
        from ceno_amd import prover

        self.t = prover.Transcript.poseidon2(label)
        LBL = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
        EXT = C.CFUNCTYPE(None, C.c_void_p, po.u64p)
        self._cb = (LBL(lambda _s, b, n: self.t.append_label(bytes(b[:n]))), EXT(lambda _s, e: self.t.append_ext((e[0], e[1]))), EXT(self._sample))
        self.tr = po.OrcTranscript(C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), C.cast(self._cb[2], C.c_void_p), None)

    def _sample(self, _s, out):
        out[0], out[1] = self.t.sample_ext()

    def ptr(self):
        return C.byref(self.tr)


def check_sumcheck_on_oracle(g):
    s = g["sumcheck"]
    tables = [np.array(t, dtype=np.uint64) for t in s["tables"]]
    msgs, chal, fin = po.sumcheck_prove(tables, po.ext([1]), [[0, 1, 2]], s["num_vars"], s["degree"], _HostTranscriptForOracle(s["label"].encode()))
    assert msgs.tolist() == s["messages"], "sumcheck messages differ from IOPProverState::prove"
    assert chal.tolist() == s["challenges"] and fin.tolist() == s["final_evals"]


def _flat_numbers(x):
    """every integer of a JSON value, in document order (the reference's serde layouts are not known here: proofs are compared as word sequences)"""
    if isinstance(x, bool):
        return []
    if isinstance(x, int):
        return [x]
    if isinstance(x, (list, tuple)):
        return [v for e in x for v in _flat_numbers(e)]
    if isinstance(x, dict):
        return [v for e in x.values() for v in _flat_numbers(e)]
    return []


def _contains_run(hay, needle):
    n = len(needle)
    if n == 0:
        return True
    first = needle[0]
    return any(hay[i] == first and hay[i: i + n] == needle for i in range(len(hay) - n + 1))


def _section(g, key):
    if key not in g:
        pytest.xfail(f"PARITY UNPINNED: the goldens file has no '{key}' section (dumped with an older tools/goldens/dump_goldens.rs)")
    return g[key]


def check_tower_on_oracle(g):
    """CpuTowerProver::create_proof (scheme/cpu/mod.rs:346-554) on one product and one LogUp spec: the point, every round message and every
    per-layer evaluation of the oracle's tower prover under the real transcript"""
    t = _section(g, "tower")
    prod_last = [np.array(l, dtype=np.uint64) for l in t["prod_last_layer"]]
    p_last = [np.array(l, dtype=np.uint64) for l in t["logup_p_last_layer"]]
    q_last = [np.array(l, dtype=np.uint64) for l in t["logup_q_last_layer"]]
    prod = po.infer_tower_product_witness(int(prod_last[0].shape[0]).bit_length(), prod_last)
    logup = po.infer_tower_logup_witness(p_last, q_last)
    proof = po.tower_prove([prod], [logup], _HostTranscriptForOracle(t["label"].encode()))
    nv = max(len(prod), len(logup))
    assert proof.point[:nv].tolist() == t["point"], "tower point differs from CpuTowerProver::create_proof"
    ref = _flat_numbers(t["proof"])
    if ref:
        off = 0
        for r in range(1, nv):  # the messages of every layer's sumcheck appear in the reference proof as one run of words
            words = proof.msgs[off: off + 6 * r].tolist()
            assert _contains_run(ref, words), f"tower layer {r}: round messages not found in the reference's TowerProofs"
            off += 6 * r


def check_rotation_on_oracle(g):
    r = _section(g, "rotation")
    table = np.array(r["table"], dtype=np.uint64)
    assert po.rotation_next_base_mle(table, r["cyclic_group_log2"]).tolist() == r["rotated"], "rotation_next_base_mle"
    eq = po.build_eq(np.array(r["point"], dtype=np.uint64))
    assert po.rotation_selector(eq, r["cyclic_subgroup_size"], r["cyclic_group_log2"]).tolist() == r["selector"], "rotation_selector"


def check_mixed_size_sumcheck_on_oracle(g):
    """IOPProverState::prove on a front-loaded plan: base columns of two chips of different sizes, each under its eq table, in monomial form"""
    m = _section(g, "mixed_size_sumcheck")
    a_cols = [np.array(c, dtype=np.uint64) for c in m["a_cols"]]
    b_cols = [np.array(c, dtype=np.uint64) for c in m["b_cols"]]
    tables = a_cols + [po.build_eq(np.array(m["a_point"], dtype=np.uint64))] + b_cols + [po.build_eq(np.array(m["b_point"], dtype=np.uint64))]
    msgs, chal, fin = po.sumcheck_prove(tables, po.ext([tuple(x) for x in m["scalars"]]), m["terms"], m["max_num_vars"], m["degree"],
                                        _HostTranscriptForOracle(m["label"].encode()))
    assert msgs.tolist() == m["messages"], "mixed-size sumcheck messages differ from IOPProverState::prove"
    assert chal.tolist() == m["challenges"] and fin.tolist() == m["final_evals"]


def test_extension_field_w(plib):
    check_ext_mul(plib, _load())


def test_poseidon2_permutation_and_constants(plib):
    g = _load()
    check_permutation(plib, g, _install(plib, g))


def test_label_packing(plib):
    check_labels(plib, _load())


def test_basic_transcript_script(plib):
    g = _load()
    _install(plib, g)
    check_transcript(plib, g)


def test_sumcheck_proof_on_the_oracle_under_the_real_transcript(plib):
    g = _load()
    _install(plib, g)
    check_sumcheck_on_oracle(g)


def test_tower_proof_on_the_oracle_under_the_real_transcript(plib):
    g = _load()
    _install(plib, g)
    check_tower_on_oracle(g)


def test_rotation_helpers_against_the_reference(plib):
    check_rotation_on_oracle(_load())


def test_mixed_size_sumcheck_on_the_oracle_under_the_real_transcript(plib):
    g = _load()
    _install(plib, g)
    check_mixed_size_sumcheck_on_oracle(g)


@pytest.mark.gpu
def test_opening_of_witness_and_fixed_commitments_against_the_reference(plib):
    """OpeningProver::open over two commitments: says whether our proof's words already coincide with the reference's batch_open, and xfails with
    the first difference when they do not (the layout of the commit path is unpinned like the root's)"""
    g = _load()
    b = _section(g, "basefold_two_commitments")
    ext, internal, diag = _constants(g)
    _install(plib, g)
    from ceno_amd import Device, api, prover

    dev = Device(0)
    api.poseidon2_set_constants(dev, ext.reshape(-1), internal, diag)
    stream = dev.stream_create()
    mats = lambda key: [np.array(m["values_row_major"], dtype=np.uint64).reshape(m["rows"], m["width"]) for m in b[key]]  # noqa: E731
    ref = _flat_numbers(b["proof"])
    hits = {}
    for log_blowup in (1, 2, 3):
        pw, pf = prover.PcsData(dev, mats("witness"), log_blowup, stream), prover.PcsData(dev, mats("fixed"), log_blowup, stream)
        tr = prover.Transcript.poseidon2(b["label"].encode())
        proof = pw.basefold_open([np.array(p_, dtype=np.uint64) for p_ in b["points"]], [np.array(e_, dtype=np.uint64) for e_ in b["evals"]], 100, 16, tr,
                                 more_commits=[pf])
        n = 4
        hits[log_blowup] = _contains_run(ref, proof[: 4 * n].tolist())   # the sumcheck messages of the opening: the first words of our flat proof
        pw.free()
        pf.free()
    api.poseidon2_set_constants(dev)
    dev.close()
    if not any(hits.values()):
        pytest.xfail(f"PARITY UNPINNED (Basefold opening): our opening's first round messages are not in the reference proof at any blow-up {hits}")


@pytest.mark.gpu
def test_sumcheck_proof_on_the_gpu_under_the_real_transcript(plib):
    g = _load()
    _install(plib, g)
    from ceno_amd import Device, prover

    s = g["sumcheck"]
    dev = Device(0)
    mles = [dev.upload(np.array(t, dtype=np.uint64)) for t in s["tables"]]
    msgs, chal, fin = prover.sumcheck_prove(dev, mles, po.ext([1]), [[0, 1, 2]], s["num_vars"], s["degree"], prover.Transcript.poseidon2(s["label"].encode()))
    assert msgs.tolist() == s["messages"] and chal.tolist() == s["challenges"] and fin.tolist() == s["final_evals"]
    dev.close()


@pytest.mark.gpu
def test_basefold_root_against_the_reference(plib):
    """the commit path's layout (rate, leaf hashing) is unpinned: this test says precisely whether the root already coincides,
    and xfails with the two roots when it does not"""
    g = _load()
    ext, internal, diag = _constants(g)
    from ceno_amd import Device, api, prover

    dev = Device(0)
    api.poseidon2_set_constants(dev, ext.reshape(-1), internal, diag)
    b = g["basefold"]
    host = np.array(b["values_row_major"], dtype=np.uint64).reshape(b["rows"], b["width"])
    stream = dev.stream_create()
    roots = {}
    for log_blowup in (1, 2, 3):
        pcs = prover.PcsData(dev, [host], log_blowup, stream)
        roots[log_blowup] = pcs.root().tolist()
        pcs.free()
    api.poseidon2_set_constants(dev)
    dev.close()
    flat = json.dumps(b["commitment"])
    hit = [k for k, r in roots.items() if all(str(x) in flat for x in r)]
    if not hit:
        pytest.xfail(f"PARITY UNPINNED (Basefold layout): reference commitment {flat[:200]} vs ours by blow-up {roots}")


# ------------------------------------------------------------------------------------------------------------------
# the kit's plumbing on a file generated from this repository's own implementation
# ------------------------------------------------------------------------------------------------------------------
def _self_goldens():
    from ceno_amd import prover

    params = po.poseidon2_default_params()
    a, b = (123456789, 987654321), (P - 5, 77)
    g = {"ext_mul": {"a": list(a), "b": list(b), "ab": list(po.e2_mul(a, b)), "a_inv": list(po.e2_inv(a))}}
    kats = []
    for seed in range(3):
        st = po.rand_base(8, 40 + seed)
        kats.append({"in": st.tolist(), "out": po.poseidon2_permute(st, params).tolist()})
    g["poseidon2"] = {"width": 8, "rate": 4, "kats": kats,
                      "round_constants": {"external": params[:64].reshape(8, 8).tolist(), "internal": params[64:86].tolist(), "diag": params[86:94].tolist()}}
    labels = [b"riscv", b"Internal round", b"combine subset evals"]
    g["bytes_to_field_elements"] = [{"label": l.decode(), "elements": [int.from_bytes(l[i: i + 8], "little") % P for i in range(0, len(l), 8)]} for l in labels]
    t = prover.Transcript.poseidon2(b"riscv")
    script = []
    for v in (26, 3):
        t.append_label(v.to_bytes(8, "little"))
        script.append({"op": "append_message", "bytes": list(v.to_bytes(8, "little"))})
    t.append_ext((5, 6))
    script.append({"op": "append_ext", "value": [5, 6]})
    t.append_label(b"Internal round")
    script.append({"op": "challenge", "label": "Internal round", "value": list(t.sample_ext())})
    t.append_base(99)
    script.append({"op": "append_base", "value": 99})
    script.append({"op": "read_challenge", "value": list(t.sample_ext())})
    g["transcript"] = {"label": "riscv", "script": script}
    tables = [po.rand_ext(16, 0xCE10 + j) for j in range(3)]
    msgs, chal, fin = po.sumcheck_prove(tables, po.ext([1]), [[0, 1, 2]], 4, 3, _HostTranscriptForOracle(b"sumcheck"))
    g["sumcheck"] = {"label": "sumcheck", "num_vars": 4, "degree": 3, "tables": [t.tolist() for t in tables], "messages": msgs.tolist(),
                     "challenges": chal.tolist(), "final_evals": fin.tolist()}
    # rounds 4 - 6: tower proof, rotation helpers, mixed-size sumcheck (the sections tools/goldens/dump_goldens.rs 7 - 9 write)
    prod_last = [po.rand_ext(16, 0x70 + l) for l in range(2)]
    p_last, q_last = [po.rand_ext(8, 0x7C + l) for l in range(2)], [po.rand_ext(8, 0x7A + l) for l in range(2)]
    proof = po.tower_prove([po.infer_tower_product_witness(5, prod_last)], [po.infer_tower_logup_witness(p_last, q_last)], _HostTranscriptForOracle(b"tower"))
    g["tower"] = {"label": "tower", "prod_last_layer": [x.tolist() for x in prod_last], "logup_p_last_layer": [x.tolist() for x in p_last],
                  "logup_q_last_layer": [x.tolist() for x in q_last], "point": proof.point[:5].tolist(),
                  "proof": {"proofs": proof.msgs.tolist(), "prod_specs_eval": proof.prod_evals.tolist(), "logup_specs_eval": proof.logup_evals.tolist()}}
    table, point = po.rand_base(128, 0x707), po.rand_ext(7, 0x708)
    g["rotation"] = {"cyclic_group_log2": 5, "cyclic_subgroup_size": 23, "table": table.tolist(), "rotated": po.rotation_next_base_mle(table, 5).tolist(),
                     "point": point.tolist(), "selector": po.rotation_selector(po.build_eq(point), 23, 5).tolist()}
    a_cols, b_cols = [po.rand_base(16, 0x910 + j) for j in range(3)], [po.rand_base(4, 0x920 + j) for j in range(2)]
    pa, pb, scal = po.rand_ext(4, 0x9A), po.rand_ext(2, 0x9B), po.rand_ext(4, 0x930)
    terms = [[3, 0, 1], [3, 2], [6, 4, 5], [6, 5]]
    tabs = a_cols + [po.build_eq(pa)] + b_cols + [po.build_eq(pb)]
    msgs, chal, fin = po.sumcheck_prove(tabs, scal, terms, 4, 3, _HostTranscriptForOracle(b"batched_main"))
    g["mixed_size_sumcheck"] = {"label": "batched_main", "max_num_vars": 4, "degree": 3, "a_cols": [c.tolist() for c in a_cols], "a_point": pa.tolist(),
                                "b_cols": [c.tolist() for c in b_cols], "b_point": pb.tolist(), "scalars": scal.tolist(), "terms": terms,
                                "messages": msgs.tolist(), "challenges": chal.tolist(), "final_evals": fin.tolist()}
    return g


def test_kit_plumbing_on_self_generated_file(plib, tmp_path):
    g = _self_goldens()
    path = tmp_path / "self_goldens.json"
    path.write_text(json.dumps(g))
    g = _load(str(path))
    params = _install(plib, g)
    check_ext_mul(plib, g)
    check_permutation(plib, g, params)
    check_labels(plib, g)
    check_transcript(plib, g)
    check_sumcheck_on_oracle(g)
    check_tower_on_oracle(g)
    check_rotation_on_oracle(g)
    check_mixed_size_sumcheck_on_oracle(g)
    # a wrong table must be noticed
    g["poseidon2"]["kats"][0]["out"][0] ^= 1
    with pytest.raises(AssertionError):
        check_permutation(plib, g, params)
