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
    # a wrong table must be noticed
    g["poseidon2"]["kats"][0]["out"][0] ^= 1
    with pytest.raises(AssertionError):
        check_permutation(plib, g, params)
