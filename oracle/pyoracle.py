"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/libceno_oracle.so (the plain-C CPU restatement) plus a tiny
pure-Python big-int model of Goldilocks / GoldilocksExt2 used to cross-check the C code on
small cases.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; nothing in ceno_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

P = 0xFFFFFFFF00000001
W = 7
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libceno_oracle.so")


def build(force: bool = False) -> str:
    import glob

    srcs = glob.glob(os.path.join(_HERE, "*.c")) + glob.glob(os.path.join(_HERE, "*.h")) + [os.path.join(_HERE, "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libceno_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


# ---------------------------------------------------------------------------------------------
# pure-python field model (independent of the C code)
# ---------------------------------------------------------------------------------------------
def e2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def e2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def e2_mul(a, b):
    return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def e2_inv(a):
    n = (a[0] * a[0] - W * a[1] * a[1]) % P
    ni = pow(n, P - 2, P)
    return (a[0] * ni % P, (-a[1]) * ni % P)


def splitmix64_at(seed: int, i: int) -> int:
    M = (1 << 64) - 1
    z = (seed + (i + 1) * 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def splitmix_gl(seed: int, i: int) -> int:
    z = splitmix64_at(seed, i)
    return z - P if z >= P else z


# ---------------------------------------------------------------------------------------------
# ctypes plumbing
# ---------------------------------------------------------------------------------------------
u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


class OrcMle(C.Structure):
    _fields_ = [("data", u64p), ("is_ext", C.c_int), ("num_vars", C.c_int)]


class OrcStubState(C.Structure):
    _fields_ = [("s", C.c_uint64)]


class OrcTranscript(C.Structure):
    _fields_ = [
        ("append_label", C.c_void_p),
        ("append_ext", C.c_void_p),
        ("sample_ext", C.c_void_p),
        ("self", C.c_void_p),
        ("reserved", C.c_void_p),
        ("append_base", C.c_void_p),
        ("sample_bits", C.c_void_p),
        ("fork", C.c_void_p),
        ("fork_free", C.c_void_p),
    ]


class OrcDuplexState(C.Structure):
    _fields_ = [("state", C.c_uint64 * 8), ("inp", C.c_uint64 * 4), ("n_in", C.c_int), ("n_out", C.c_int), ("params", C.c_uint64 * 138)]


class OrcTowerSpec(C.Structure):
    _fields_ = [("num_vars", C.c_int), ("layers", C.POINTER(u64p))]


class OrcTowerProof(C.Structure):
    _fields_ = [
        ("num_rounds", C.c_int),
        ("msgs", u64p),
        ("prod_evals", u64p),
        ("logup_evals", u64p),
        ("point", u64p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.environ.get("CENO_ORACLE_LIB") or _LIB_PATH)  # (override: the sanitizer build of the checker)
        _lib.orc_gl_mul.restype = C.c_uint64
        _lib.orc_gl_mul.argtypes = [C.c_uint64, C.c_uint64]
        _lib.orc_gl_mul_div.restype = C.c_uint64
        _lib.orc_gl_mul_div.argtypes = [C.c_uint64, C.c_uint64]
        _lib.orc_gl_inv.restype = C.c_uint64
        _lib.orc_gl_inv.argtypes = [C.c_uint64]
        _lib.orc_interleave_out_len.restype = C.c_size_t
        _lib.orc_interleave_out_len.argtypes = [C.c_int, C.c_size_t, C.c_int]
        _lib.orc_tower_msgs_words.restype = C.c_size_t
        _lib.orc_tower_msgs_words.argtypes = [C.c_int]
        _lib.orc_two_adic_generator.restype = C.c_uint64
        _lib.orc_two_adic_generator.argtypes = [C.c_int]
        _lib.orc_tr_sample_bits.restype = C.c_uint64
        _lib.orc_tr_sample_bits.argtypes = [C.c_void_p, C.c_int]
        _lib.orc_tr_check_witness.restype = C.c_int
        _lib.orc_tr_check_witness.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        _lib.orc_tr_grind.restype = C.c_uint64
        _lib.orc_tr_grind.argtypes = [C.c_void_p, C.c_int]
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _p32(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def ext(vals) -> np.ndarray:
    """list of (c0,c1) / ints -> uint64 array of shape (n,2)"""
    out = np.zeros((len(vals), 2), dtype=np.uint64)
    for i, v in enumerate(vals):
        if isinstance(v, (tuple, list, np.ndarray)):
            out[i, 0], out[i, 1] = int(v[0]) % P, int(v[1]) % P
        else:
            out[i, 0] = int(v) % P
    return out


def fill_splitmix(n_words: int, seed: int, word_offset: int = 0) -> np.ndarray:
    out = np.empty(n_words, dtype=np.uint64)
    lib().orc_fill_splitmix(_p(out), C.c_size_t(n_words), C.c_uint64(seed), C.c_uint64(word_offset))
    return out


def rand_ext(n: int, seed: int) -> np.ndarray:
    return fill_splitmix(2 * n, seed).reshape(n, 2)


def rand_base(n: int, seed: int) -> np.ndarray:
    return fill_splitmix(n, seed)


class StubTranscript:
    def __init__(self, seed: int = 0xF5):
        self.state = OrcStubState()
        lib().orc_stub_init(C.byref(self.state), C.c_uint64(seed))
        self.tr = OrcTranscript()
        lib().orc_stub_bind(C.byref(self.tr), C.byref(self.state))

    def ptr(self):
        return C.byref(self.tr)

    def append_label(self, b: bytes):
        buf = (C.c_uint8 * len(b)).from_buffer_copy(b) if b else (C.c_uint8 * 1)()
        lib().orc_stub_append_label(C.byref(self.state), buf, C.c_size_t(len(b)))

    def append_ext(self, e):
        a = ext([e]).reshape(2)
        lib().orc_stub_append_ext(C.byref(self.state), _p(a))

    def sample_ext(self) -> Tuple[int, int]:
        o = np.zeros(2, dtype=np.uint64)
        lib().orc_stub_sample_ext(C.byref(self.state), _p(o))
        return int(o[0]), int(o[1])

    def append_base(self, v: int):
        lib().orc_stub_append_base(C.byref(self.state), C.c_uint64(v))

    def sample_bits(self, bits: int) -> int:
        return int(lib().orc_tr_sample_bits(self.ptr(), bits))

    def sample_base(self) -> int:
        return self.sample_bits(64)

    def check_witness(self, bits: int, w: int) -> bool:
        return bool(lib().orc_tr_check_witness(self.ptr(), bits, C.c_uint64(w)))

    def grind(self, bits: int) -> int:
        return int(lib().orc_tr_grind(self.ptr(), bits))


class DuplexTranscript(StubTranscript):
    """the oracle's own Poseidon2 duplex challenger (oracle/transcript.c): p3 DuplexChallenger<_, _, 8, 4> rules"""

    def __init__(self, label: bytes = b"", params=None):
        self.params = poseidon2_default_params() if params is None else np.ascontiguousarray(params, dtype=np.uint64)
        self.state = OrcDuplexState()
        buf = (C.c_uint8 * max(1, len(label))).from_buffer_copy(label or b"\0")
        lib().orc_duplex_init(C.byref(self.state), _p(self.params), buf, C.c_size_t(len(label)))
        self.tr = OrcTranscript()
        lib().orc_duplex_bind(C.byref(self.tr), C.byref(self.state))

    def _call(self, name, *args):
        f = C.cast(getattr(self.tr, name), {
            "append_label": C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t),
            "append_ext": C.CFUNCTYPE(None, C.c_void_p, u64p),
            "sample_ext": C.CFUNCTYPE(None, C.c_void_p, u64p),
            "append_base": C.CFUNCTYPE(None, C.c_void_p, C.c_uint64),
        }[name])
        return f(self.tr.self, *args)

    def append_label(self, b: bytes):
        buf = (C.c_uint8 * max(1, len(b))).from_buffer_copy(b or b"\0")
        self._call("append_label", buf, len(b))

    def append_ext(self, e):
        a = ext([e]).reshape(2)
        self._call("append_ext", _p(a))

    def sample_ext(self) -> Tuple[int, int]:
        o = np.zeros(2, dtype=np.uint64)
        self._call("sample_ext", _p(o))
        return int(o[0]), int(o[1])

    def append_base(self, v: int):
        self._call("append_base", C.c_uint64(v))

    def export_state(self) -> np.ndarray:
        """[state 8][n_in][in 4][n_out][0, 0]: the layout of the host library's ceno_transcript export"""
        o = np.zeros(16, dtype=np.uint64)
        o[:8] = np.frombuffer(bytes(self.state.state), dtype=np.uint64)
        o[8] = self.state.n_in
        o[9:13] = np.frombuffer(bytes(self.state.inp), dtype=np.uint64)
        o[9 + self.state.n_in:13] = 0
        o[13] = self.state.n_out
        return o


class ReplayTranscript:
    """transcript that ignores what is appended and hands out a fixed list of challenges: replays the tail of a proof on the
    oracle from tables folded up to some round (tests at sizes the oracle cannot run from round 0)"""

    def __init__(self, challenges):
        self.q = [(int(c[0]), int(c[1])) for c in challenges]
        self.pos = 0
        LBL = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
        EXT = C.CFUNCTYPE(None, C.c_void_p, u64p)

        def sample(_self, out):
            c = self.q[self.pos]
            self.pos += 1
            out[0], out[1] = c

        self._cb = (LBL(lambda *_: None), EXT(lambda *_: None), EXT(sample))
        self.tr = OrcTranscript(C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), C.cast(self._cb[2], C.c_void_p), None)

    def ptr(self):
        return C.byref(self.tr)


def build_eq(point: np.ndarray) -> np.ndarray:
    n = point.shape[0]
    out = np.zeros((1 << n, 2), dtype=np.uint64)
    lib().orc_build_eq_x_r_vec(_p(np.ascontiguousarray(point)), n, _p(out))
    return out


def eq_eval(a: np.ndarray, b: np.ndarray) -> Tuple[int, int]:
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_eq_eval(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), a.shape[0], _p(o))
    return int(o[0]), int(o[1])


def mle_evaluate(evals: np.ndarray, point: np.ndarray) -> Tuple[int, int]:
    is_ext = int(evals.ndim == 2)
    nv = int(evals.shape[0]).bit_length() - 1
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_mle_evaluate(_p(np.ascontiguousarray(evals)), is_ext, nv, _p(np.ascontiguousarray(point)), _p(o))
    return int(o[0]), int(o[1])


def mle_fix_variable(evals: np.ndarray, r) -> np.ndarray:
    is_ext = int(evals.ndim == 2)
    nv = int(evals.shape[0]).bit_length() - 1
    out = np.zeros((1 << (nv - 1), 2), dtype=np.uint64)
    lib().orc_mle_fix_variable(_p(np.ascontiguousarray(evals)), is_ext, nv, _p(ext([r]).reshape(2)), _p(out))
    return out


def extrapolate_uni_poly(p0, evals: np.ndarray, x) -> Tuple[int, int]:
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_extrapolate_uni_poly(_p(ext([p0]).reshape(2)), _p(np.ascontiguousarray(evals)), evals.shape[0],
                                   _p(ext([x]).reshape(2)), _p(o))
    return int(o[0]), int(o[1])


def _mk_mles(mles: Sequence[np.ndarray]):
    arr = (OrcMle * len(mles))()
    keep = []
    for i, m in enumerate(mles):
        m = np.ascontiguousarray(m)
        keep.append(m)
        arr[i].data = _p(m)
        arr[i].is_ext = int(m.ndim == 2)
        arr[i].num_vars = int(m.shape[0]).bit_length() - 1
    return arr, keep


def _csr(terms: Sequence[Sequence[int]]):
    off = np.zeros(len(terms) + 1, dtype=np.uint32)
    idx: List[int] = []
    for t, s in enumerate(terms):
        idx.extend(s)
        off[t + 1] = len(idx)
    return off, np.array(idx if idx else [0], dtype=np.uint32)


def sumcheck_prove(mles: Sequence[np.ndarray], coeffs: np.ndarray, terms: Sequence[Sequence[int]], max_nv: int,
                   max_degree: int, tr: StubTranscript):
    arr, keep = _mk_mles(mles)
    off, idx = _csr(terms)
    msgs = np.zeros((max_nv, max_degree, 2), dtype=np.uint64)
    chal = np.zeros((max_nv, 2), dtype=np.uint64)
    fin = np.zeros((len(mles), 2), dtype=np.uint64)
    rc = lib().orc_sumcheck_prove(arr, len(mles), _p(np.ascontiguousarray(coeffs)), _p32(off), _p32(idx), len(terms),
                                  max_nv, max_degree, tr.ptr(), _p(msgs), _p(chal), _p(fin))
    if rc != 0:
        raise ValueError(f"orc_sumcheck_prove rc={rc}")
    return msgs, chal, fin


def sumcheck_verify(claimed_sum, msgs: np.ndarray, tr: StubTranscript):
    n, d = msgs.shape[0], msgs.shape[1]
    point = np.zeros((n, 2), dtype=np.uint64)
    exp = np.zeros(2, dtype=np.uint64)
    lib().orc_sumcheck_verify(_p(ext([claimed_sum]).reshape(2)), _p(np.ascontiguousarray(msgs)), n, d, tr.ptr(),
                              _p(point), _p(exp))
    return point, (int(exp[0]), int(exp[1]))


def sumcheck_expected_from_evals(mle_num_vars: Sequence[int], coeffs, terms, max_nv, point, final_evals):
    off, idx = _csr(terms)
    nv = (C.c_int * len(mle_num_vars))(*mle_num_vars)
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_sumcheck_expected_from_evals(nv, len(mle_num_vars), _p(np.ascontiguousarray(coeffs)), _p32(off),
                                           _p32(idx), len(terms), max_nv, _p(np.ascontiguousarray(point)),
                                           _p(np.ascontiguousarray(final_evals)), _p(o))
    return int(o[0]), int(o[1])


def recover_claim_from_final(final_claim, msgs, challenges):
    n, d = msgs.shape[0], msgs.shape[1]
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_recover_claim_from_final(_p(ext([final_claim]).reshape(2)), _p(np.ascontiguousarray(msgs)),
                                       _p(np.ascontiguousarray(challenges)), n, d, _p(o))
    return int(o[0]), int(o[1])


def dense_mt_workspace(k: int, n: int):
    """ping/pong buffers of sumcheck_dense_mt, touched (zero-filled) so that timing excludes page faults"""
    ping = [np.zeros((max(1, 1 << (n - 1)), 2), dtype=np.uint64) for _ in range(k)]
    pong = [np.zeros((max(1, 1 << max(n - 2, 0)), 2), dtype=np.uint64) for _ in range(k)]
    return ping, pong


def have_avx512() -> bool:
    return bool(lib().orc_have_avx512())


def sumcheck_dense_mt(tables: Sequence[np.ndarray], challenges: np.ndarray, threads: int = 0, workspace=None, avx512: bool = False):
    """tables: k ext arrays (2^n,2). Inputs are not modified.  avx512: the vector form (oracle/dense_avx512.c), same outputs"""
    k = len(tables)
    n = int(tables[0].shape[0]).bit_length() - 1
    bufs = [np.ascontiguousarray(t) for t in tables]
    ping, pong = workspace if workspace is not None else dense_mt_workspace(k, n)
    ptrs = (u64p * (3 * k))(*[_p(b) for b in bufs + ping + pong])
    msgs = np.zeros((n, k, 2), dtype=np.uint64)
    fin = np.zeros((k, 2), dtype=np.uint64)
    fn = lib().orc_sumcheck_dense_mt_avx512 if avx512 else lib().orc_sumcheck_dense_mt
    rc = fn(ptrs, k, n, _p(np.ascontiguousarray(challenges)), threads, _p(msgs), _p(fin))
    if rc != 0:
        raise ValueError(f"orc_sumcheck_dense_mt{'_avx512' if avx512 else ''} rc={rc}")
    return msgs, fin


def wit_infer(mles: Sequence[np.ndarray], coeffs: np.ndarray, terms, num_vars: int) -> np.ndarray:
    arr, keep = _mk_mles(mles)
    off, idx = _csr(terms)
    out = np.zeros((1 << num_vars, 2), dtype=np.uint64)
    rc = lib().orc_wit_infer(arr, len(mles), _p(np.ascontiguousarray(coeffs)), _p32(off), _p32(idx), len(terms),
                             num_vars, _p(out))
    if rc != 0:
        raise ValueError(f"orc_wit_infer rc={rc}")
    return out


SEL_WHOLE, SEL_PREFIX, SEL_ORDERED_SPARSE, SEL_QUARK_LT = 0, 1, 2, 3


def selector_compute(kind, out_point, offset=0, num_instances=0, sparse_indices=(), sparse_num_vars=0):
    n = out_point.shape[0]
    out = np.zeros((1 << n, 2), dtype=np.uint64)
    si = np.array(list(sparse_indices) or [0], dtype=np.uint32)
    rc = lib().orc_selector_compute(kind, _p(np.ascontiguousarray(out_point)), n, C.c_size_t(offset),
                                    C.c_size_t(num_instances), _p32(si), len(sparse_indices), sparse_num_vars, _p(out))
    if rc != 0:
        raise ValueError(f"orc_selector_compute rc={rc}")
    return out


def selector_evaluate(kind, out_point, in_point, offset=0, num_instances=0, sparse_indices=(), sparse_num_vars=0):
    n = out_point.shape[0]
    o = np.zeros(2, dtype=np.uint64)
    si = np.array(list(sparse_indices) or [0], dtype=np.uint32)
    rc = lib().orc_selector_evaluate(kind, _p(np.ascontiguousarray(out_point)), _p(np.ascontiguousarray(in_point)), n,
                                     C.c_size_t(offset), C.c_size_t(num_instances), _p32(si), len(sparse_indices),
                                     sparse_num_vars, _p(o))
    if rc != 0:
        raise ValueError(f"orc_selector_evaluate rc={rc}")
    return int(o[0]), int(o[1])


def eq_eval_less_or_equal_than(max_idx: int, a: np.ndarray, b: np.ndarray):
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_eq_eval_less_or_equal_than(C.c_uint64(max_idx), _p(np.ascontiguousarray(a)), a.shape[0],
                                         _p(np.ascontiguousarray(b)), b.shape[0], _p(o))
    return int(o[0]), int(o[1])


def eval_wellform_address_vec(offset, scaled, r, descending=False):
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_eval_wellform_address_vec(C.c_uint64(offset), C.c_uint64(scaled), _p(np.ascontiguousarray(r)),
                                        r.shape[0], int(descending), _p(o))
    return int(o[0]), int(o[1])


def eval_stacked_wellform_address_vec(r):
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_eval_stacked_wellform_address_vec(_p(np.ascontiguousarray(r)), r.shape[0], _p(o))
    return int(o[0]), int(o[1])


def eval_stacked_constant_vec(r):
    o = np.zeros(2, dtype=np.uint64)
    lib().orc_eval_stacked_constant_vec(_p(np.ascontiguousarray(r)), r.shape[0], _p(o))
    return int(o[0]), int(o[1])


# ---- tower witness -----------------------------------------------------------------------
def interleaving_mles_to_mles(mles: Sequence[np.ndarray], num_instances: int, num_limbs: int, default) -> List[np.ndarray]:
    arr, keep = _mk_mles(mles)
    out_len = lib().orc_interleave_out_len(len(mles), C.c_size_t(num_instances), num_limbs)
    outs = [np.zeros((out_len, 2), dtype=np.uint64) for _ in range(num_limbs)]
    ptrs = (u64p * num_limbs)(*[_p(o) for o in outs])
    rc = lib().orc_interleaving_mles_to_mles(arr, len(mles), C.c_size_t(num_instances), num_limbs,
                                             _p(ext([default]).reshape(2)), ptrs)
    if rc != 0:
        raise ValueError(f"orc_interleaving_mles_to_mles rc={rc}")
    return outs


def infer_tower_product_witness(num_vars: int, last_layer: Sequence[np.ndarray]) -> List[List[np.ndarray]]:
    layers = [[np.zeros((1 << l, 2), dtype=np.uint64) for _ in range(2)] for l in range(num_vars)]
    flat = [a for lay in layers for a in lay]
    ptrs = (u64p * len(flat))(*[_p(a) for a in flat])
    rc = lib().orc_infer_tower_product_witness(num_vars, _p(np.ascontiguousarray(last_layer[0])),
                                               _p(np.ascontiguousarray(last_layer[1])), ptrs)
    if rc != 0:
        raise ValueError(f"orc_infer_tower_product_witness rc={rc}")
    return layers


def infer_tower_logup_witness(p: Optional[Sequence[np.ndarray]], q: Sequence[np.ndarray]) -> List[List[np.ndarray]]:
    nv = int(q[0].shape[0]).bit_length() - 1
    layers = [[np.zeros((1 << l, 2), dtype=np.uint64) for _ in range(4)] for l in range(nv + 1)]
    flat = [a for lay in layers for a in lay]
    ptrs = (u64p * len(flat))(*[_p(a) for a in flat])
    p0 = _p(np.ascontiguousarray(p[0])) if p is not None else None
    p1 = _p(np.ascontiguousarray(p[1])) if p is not None else None
    rc = lib().orc_infer_tower_logup_witness(nv, p0, p1, _p(np.ascontiguousarray(q[0])),
                                             _p(np.ascontiguousarray(q[1])), ptrs)
    if rc != 0:
        raise ValueError(f"orc_infer_tower_logup_witness rc={rc}")
    return layers


class TowerProof:
    def __init__(self, max_nv: int, n_prod: int, n_logup: int):
        self.max_nv, self.n_prod, self.n_logup = max_nv, n_prod, n_logup
        R = max_nv - 1
        self.msgs = np.zeros(max(1, lib().orc_tower_msgs_words(max_nv)), dtype=np.uint64)
        self.prod_evals = np.zeros((max(1, n_prod), max(1, R), 2, 2), dtype=np.uint64)
        self.logup_evals = np.zeros((max(1, n_logup), max(1, R), 4, 2), dtype=np.uint64)
        self.point = np.zeros((max_nv + 1, 2), dtype=np.uint64)
        self.c = OrcTowerProof(R, _p(self.msgs), _p(self.prod_evals), _p(self.logup_evals), _p(self.point))

    def round_msgs(self, rnd: int) -> np.ndarray:
        """messages of tower round `rnd` (1-based): array (rnd, 3, 2)"""
        off = sum(r * 3 * 2 for r in range(1, rnd))
        return self.msgs[off: off + rnd * 6].reshape(rnd, 3, 2)


def _mk_specs(specs: Sequence[List[List[np.ndarray]]]):
    arr = (OrcTowerSpec * max(1, len(specs)))()
    keep = []
    for i, layers in enumerate(specs):
        flat = [np.ascontiguousarray(a) for lay in layers for a in lay]
        ptrs = (u64p * len(flat))(*[_p(a) for a in flat])
        keep.append((flat, ptrs))
        arr[i].num_vars = len(layers)
        arr[i].layers = ptrs
    return arr, keep


def tower_prove(prod_specs, logup_specs, tr: StubTranscript) -> TowerProof:
    max_nv = max([len(s) for s in prod_specs] + [len(s) for s in logup_specs])
    proof = TowerProof(max_nv, len(prod_specs), len(logup_specs))
    pa, k1 = _mk_specs(prod_specs)
    la, k2 = _mk_specs(logup_specs)
    rc = lib().orc_tower_prove(pa, len(prod_specs), la, len(logup_specs), tr.ptr(), C.byref(proof.c))
    if rc != 0:
        raise ValueError(f"orc_tower_prove rc={rc}")
    return proof


def tower_verify(prod_out_evals: np.ndarray, logup_out_evals: np.ndarray, num_variables: Sequence[int],
                 proof: TowerProof, tr: StubTranscript):
    n_prod, n_logup = proof.n_prod, proof.n_logup
    max_nv = max(num_variables)
    pt = np.zeros((max_nv + 1, 2), dtype=np.uint64)
    pc = np.zeros((max(1, n_prod), 2), dtype=np.uint64)
    lp = np.zeros((max(1, n_logup), 2), dtype=np.uint64)
    lq = np.zeros((max(1, n_logup), 2), dtype=np.uint64)
    nv = (C.c_int * len(num_variables))(*num_variables)
    po = np.ascontiguousarray(prod_out_evals) if n_prod else np.zeros((1, 2), dtype=np.uint64)
    lo = np.ascontiguousarray(logup_out_evals) if n_logup else np.zeros((1, 2), dtype=np.uint64)
    rc = lib().orc_tower_verify(_p(po), _p(lo), nv, n_prod, n_logup, C.byref(proof.c), tr.ptr(), _p(pt), _p(pc),
                                _p(lp), _p(lq))
    return rc, pt[:max_nv], pc, lp, lq


# ---- commit path (PARITY UNPINNED) ---------------------------------------------------------
def dft_bitrev(col: np.ndarray, inverse: bool = False) -> np.ndarray:
    col = np.ascontiguousarray(col, dtype=np.uint64)
    log_n = int(col.shape[0]).bit_length() - 1
    out = np.zeros_like(col)
    lib().orc_dft_bitrev(_p(col), log_n, int(inverse), _p(out))
    return out


def poseidon2_default_params() -> np.ndarray:
    p = np.zeros(138, dtype=np.uint64)
    lib().orc_poseidon2_default_params(_p(p))
    return p


def poseidon2_permute(state: np.ndarray, params: Optional[np.ndarray] = None) -> np.ndarray:
    params = poseidon2_default_params() if params is None else np.ascontiguousarray(params)
    s = np.ascontiguousarray(state, dtype=np.uint64).copy()
    lib().orc_poseidon2_permute(_p(s), _p(params))
    return s


def merkle_commit(col_major: np.ndarray, log_rows: int, width: int, params: Optional[np.ndarray] = None) -> List[np.ndarray]:
    """returns the tree levels, leaves first; each (n, 4)"""
    params = poseidon2_default_params() if params is None else np.ascontiguousarray(params)
    m = np.ascontiguousarray(col_major, dtype=np.uint64)
    out = np.zeros(4 * ((2 << log_rows) - 1), dtype=np.uint64)
    lib().orc_merkle_commit(_p(m), log_rows, width, _p(params), _p(out))
    levels, off = [], 0
    for l in range(log_rows + 1):
        n = 1 << (log_rows - l)
        levels.append(out[off: off + 4 * n].reshape(n, 4))
        off += 4 * n
    return levels


# ---- Basefold open / verify (a15) --------------------------------------------------------------
def fft_bitrev(col: np.ndarray) -> np.ndarray:
    a = np.array(col, dtype=np.uint64).copy()
    lib().orc_fft_bitrev(_p(a), int(a.shape[0]).bit_length() - 1)
    return a


def _bf_shapes(shapes, commit_sizes):
    n = len(shapes)
    sizes = [n] if commit_sizes is None else [int(x) for x in commit_sizes]
    assert sum(sizes) == n and all(x > 0 for x in sizes)
    return n, (C.c_int * len(sizes))(*sizes), (C.c_int * n)(*[int(s[0]) for s in shapes]), (C.c_int * n)(*[int(s[1]) for s in shapes]), len(sizes)


def _pp(xs):
    PP = C.POINTER(C.c_uint64)
    return (PP * len(xs))(*[x.ctypes.data_as(PP) for x in xs])


def _bf_args(traces, points, evals, commit_sizes=None):
    """traces: list of (rows, width) row-major base matrices -> column-major buffers + C pointer arrays.  commit_sizes: how many
    consecutive matrices each commitment holds (default: all of them in ONE commitment, as commit_traces does)"""
    shapes = [(int(t.shape[0]).bit_length() - 1, int(t.shape[1])) for t in traces]
    n, sizes, nv, width, nc = _bf_shapes(shapes, commit_sizes)
    cols = [np.ascontiguousarray(np.asarray(t, dtype=np.uint64).T) for t in traces]
    pts = [np.ascontiguousarray(p, dtype=np.uint64) for p in points]
    evs = [np.ascontiguousarray(e, dtype=np.uint64) for e in evals]
    return nc, sizes, nv, width, _pp(cols), _pp(pts), _pp(evs), (cols, pts, evs)


def basefold_proof_words(traces, rate_log: int, n_queries: int, commit_sizes=None) -> int:
    shapes = [(int(t.shape[0]).bit_length() - 1, int(t.shape[1])) for t in traces]
    n, sizes, nv, width, nc = _bf_shapes(shapes, commit_sizes)
    f = lib().orc_basefold_proof_words
    f.restype = C.c_size_t
    return int(f(nc, sizes, nv, width, rate_log, n_queries))


def basefold_open(traces, points, evals, rate_log: int, n_queries: int, pow_bits: int, transcript, params=None, commit_sizes=None) -> np.ndarray:
    params = poseidon2_default_params() if params is None else np.ascontiguousarray(params)
    nc, sizes, nv, width, tp, pp, ep, keep = _bf_args(traces, points, evals, commit_sizes)
    proof = np.zeros(basefold_proof_words(traces, rate_log, n_queries, commit_sizes), dtype=np.uint64)
    rc = lib().orc_basefold_open(nc, sizes, nv, width, tp, pp, ep, rate_log, n_queries, pow_bits, _p(params), transcript.ptr(), _p(proof))
    assert rc == 0, f"orc_basefold_open rc={rc}"
    return proof


def basefold_commit_roots(traces, rate_log: int, params=None, commit_sizes=None) -> np.ndarray:
    """(n_commits, 4): ONE root per commitment"""
    params = poseidon2_default_params() if params is None else np.ascontiguousarray(params)
    nc, sizes, nv, width, tp, _, _, keep = _bf_args(traces, [np.zeros((1, 2))] * len(traces), [np.zeros((1, 2))] * len(traces), commit_sizes)
    roots = np.zeros((nc, 4), dtype=np.uint64)
    lib().orc_basefold_commit_roots(nc, sizes, nv, width, tp, rate_log, _p(params), _p(roots))
    return roots


def basefold_verify(shapes, roots, points, evals, rate_log: int, n_queries: int, pow_bits: int, transcript, proof, params=None,
                    commit_sizes=None) -> int:
    """shapes: list of (num_vars, width). returns 0 when accepted"""
    params = poseidon2_default_params() if params is None else np.ascontiguousarray(params)
    n, sizes, nv, width, nc = _bf_shapes(shapes, commit_sizes)
    pts = [np.ascontiguousarray(p, dtype=np.uint64) for p in points]
    evs = [np.ascontiguousarray(e, dtype=np.uint64) for e in evals]
    r = np.ascontiguousarray(roots, dtype=np.uint64).reshape(nc, 4)
    pr = np.ascontiguousarray(proof, dtype=np.uint64)
    return int(lib().orc_basefold_verify(nc, sizes, nv, width, _p(r), _pp(pts), _pp(evs), rate_log, n_queries, pow_bits, _p(params),
                                         transcript.ptr(), _p(pr)))


# ---- mixed-height Merkle commitment (p3 MerkleTreeMmcs) ------------------------------------------------------
def mmcs_commit(mats_col_major: Sequence[np.ndarray], params=None) -> List[np.ndarray]:
    """mats_col_major: (width, rows) arrays, rows a power of two.  returns the digest layers, tallest first; each (n, 4)"""
    params = poseidon2_default_params() if params is None else np.ascontiguousarray(params)
    ms = [np.ascontiguousarray(m, dtype=np.uint64) for m in mats_col_major]
    n = len(ms)
    lr = (C.c_int * n)(*[int(m.shape[1]).bit_length() - 1 for m in ms])
    w = (C.c_int * n)(*[int(m.shape[0]) for m in ms])
    H = max(lr)
    out = np.zeros(4 * ((2 << H) - 1), dtype=np.uint64)
    lib().orc_mmcs_commit(n, lr, w, _pp(ms), _p(params), _p(out))
    levels, off = [], 0
    for l in range(H + 1):
        k = 1 << (H - l)
        levels.append(out[off: off + 4 * k].reshape(k, 4))
        off += 4 * k
    return levels


def mmcs_open(mats_col_major, levels, index: int):
    ms = [np.ascontiguousarray(m, dtype=np.uint64) for m in mats_col_major]
    n = len(ms)
    lr = (C.c_int * n)(*[int(m.shape[1]).bit_length() - 1 for m in ms])
    w = (C.c_int * n)(*[int(m.shape[0]) for m in ms])
    H = max(lr)
    flat = np.ascontiguousarray(np.concatenate([l.reshape(-1) for l in levels]))
    rows = np.zeros(sum(int(m.shape[0]) for m in ms), dtype=np.uint64)
    path = np.zeros((H, 4), dtype=np.uint64)
    lib().orc_mmcs_open(n, lr, w, _pp(ms), _p(flat), C.c_size_t(index), _p(rows), _p(path))
    return rows, path


def mmcs_verify(shapes, root, index: int, rows, path, params=None) -> int:
    """shapes: list of (log_rows, width); 0 when accepted"""
    params = poseidon2_default_params() if params is None else np.ascontiguousarray(params)
    n = len(shapes)
    lr = (C.c_int * n)(*[int(s[0]) for s in shapes])
    w = (C.c_int * n)(*[int(s[1]) for s in shapes])
    return int(lib().orc_mmcs_verify(n, lr, w, _p(np.ascontiguousarray(root, dtype=np.uint64)), C.c_size_t(index),
                                     _p(np.ascontiguousarray(rows, dtype=np.uint64)), _p(np.ascontiguousarray(path, dtype=np.uint64)), _p(params)))


# ---- rotation (a11) -------------------------------------------------------------------------
def cyclic_table(log2: int) -> np.ndarray:
    out = np.zeros(1 << log2, dtype=np.uint32)
    assert lib().orc_cyclic_table(log2, _p32(out)) == 0
    return out


def rotation_next_base_mle(table: np.ndarray, log2: int) -> np.ndarray:
    t = np.ascontiguousarray(table, dtype=np.uint64)
    out = np.zeros_like(t)
    assert lib().orc_rotation_next_base_mle(_p(t), int(t.shape[0]).bit_length() - 1, log2, _p(out)) == 0
    return out


def rotation_selector(eq: np.ndarray, subgroup_size: int, log2: int) -> np.ndarray:
    e = np.ascontiguousarray(eq)
    out = np.zeros_like(e)
    assert lib().orc_rotation_selector(_p(e), int(e.shape[0]).bit_length() - 1, subgroup_size, log2, _p(out)) == 0
    return out


def rotation_points(point: np.ndarray, log2: int):
    p = np.ascontiguousarray(point)
    l, r = np.zeros_like(p), np.zeros_like(p)
    assert lib().orc_rotation_points(_p(p), p.shape[0], log2, _p(l), _p(r)) == 0
    return l, r


def prove_rotation(wit: Sequence[np.ndarray], pairs: Sequence[Tuple[int, int]], subgroup_size: int, log2: int, rt: np.ndarray,
                   tr: StubTranscript):
    arr, keep = _mk_mles(wit)
    n = rt.shape[0]
    src = (C.c_int * len(pairs))(*[p[0] for p in pairs])
    tgt = (C.c_int * len(pairs))(*[p[1] for p in pairs])
    msgs = np.zeros((n, 2, 2), dtype=np.uint64)
    evals = np.zeros((3 * len(pairs), 2), dtype=np.uint64)
    origin, left, right = (np.zeros((n, 2), dtype=np.uint64) for _ in range(3))
    rc = lib().orc_prove_rotation(arr, len(wit), src, tgt, len(pairs), subgroup_size, log2, _p(np.ascontiguousarray(rt)), n,
                                  tr.ptr(), _p(msgs), _p(evals), _p(origin), _p(left), _p(right))
    if rc != 0:
        raise ValueError(f"orc_prove_rotation rc={rc}")
    return msgs, evals, origin, left, right


# ---- witness assignment of the ADD / SUB chips (witgen.c) ----
INSN_ADD, INSN_SUB = 1, 2  # ceno_emul InsnKind discriminants (rv32im.rs:166-172)
ARITH_COLMAP_FIELDS = 23   # 22 column ids in AddColumnMap / SubColumnMap order + num_cols


def step_records_r(cycles, pcs, kind, rs1, rs2, rd, rs1_vals, rs2_vals, rd_before, rd_after, prev_cycles) -> np.ndarray:
    """StepRecord::new_r_instruction for every entry -> (n, 136) uint8 array laid out as the emulator's #[repr(C)] struct"""
    n = len(cycles)
    nb = lib().orc_step_record_bytes()
    out = np.zeros((n, nb), dtype=np.uint8)
    L = lib()
    L.orc_step_record_r.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint8, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32, C.c_uint64]
    L.orc_step_record_r.restype = None
    for i in range(n):
        L.orc_step_record_r(out[i].ctypes.data, int(cycles[i]), int(pcs[i]), kind, rs1, rs2, rd, int(rs1_vals[i]), int(rs2_vals[i]),
                            int(rd_before[i]), int(rd_after[i]), int(prev_cycles[i]))
    return out


def witgen_arith(cols, is_sub: bool, records: np.ndarray, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of the chip's instances: (row-major n x num_cols matrix, dynamic-table counts (2^19), fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (ARITH_COLMAP_FIELDS,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[22])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_arith.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    L.orc_witgen_arith.restype = C.c_int
    rc = L.orc_witgen_arith(cols.ctypes.data, int(is_sub), recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc,
                            fetch_num_slots, out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_arith rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_ADDI = 11  # InsnKind::ADDI
ADDI_COLMAP_FIELDS = 19  # 18 column ids in AddiColumnMap order + num_cols


def step_records_i(cycles, pcs, kind, rs1, rd, imms, rs1_vals, rd_before, rd_after, prev_cycles) -> np.ndarray:
    """StepRecord::new_i_instruction for every entry -> (n, 136) uint8 array"""
    n = len(cycles)
    out = np.zeros((n, lib().orc_step_record_bytes()), dtype=np.uint8)
    L = lib()
    L.orc_step_record_i.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_int32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint64]
    L.orc_step_record_i.restype = None
    for i in range(n):
        L.orc_step_record_i(out[i].ctypes.data, int(cycles[i]), int(pcs[i]), kind, rs1, rd, int(imms[i]), int(rs1_vals[i]), int(rd_before[i]),
                            int(rd_after[i]), int(prev_cycles[i]))
    return out


def witgen_addi(cols, records: np.ndarray, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of the ADDI chip: (row-major n x num_cols matrix, dynamic-table counts, fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (ADDI_COLMAP_FIELDS,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[18])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_addi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_witgen_addi.restype = C.c_int
    rc = L.orc_witgen_addi(cols.ctypes.data, recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots,
                           out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_addi rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_JAL, INSN_AUIPC = 26, 42  # InsnKind::JAL; AUIPC follows LUI (u16limb_circuit feature)


def step_records_j(cycles, pcs, pcs_after, kind, rd, imms, rd_before, rd_after, prev_cycles) -> np.ndarray:
    """StepRecord::new_j_instruction for every entry -> (n, 136) uint8 array"""
    n = len(cycles)
    out = np.zeros((n, lib().orc_step_record_bytes()), dtype=np.uint8)
    L = lib()
    L.orc_step_record_j.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint64]
    L.orc_step_record_j.restype = None
    for i in range(n):
        L.orc_step_record_j(out[i].ctypes.data, int(cycles[i]), int(pcs[i]), int(pcs_after[i]), kind, rd, int(imms[i]), int(rd_before[i]), int(rd_after[i]),
                            int(prev_cycles[i]))
    return out


def _witgen_4tab(fn, n_cols, cols, records, indices, shard_offset, fetch_base_pc, fetch_num_slots):
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (n_cols + 1,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[n_cols])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    lk2 = np.zeros(1 << 16, dtype=np.uint32)
    lkx = np.zeros(1 << 16, dtype=np.uint32)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p]
    fn.restype = C.c_int
    rc = fn(cols.ctypes.data, recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots, out.ctypes.data,
            lkd.ctypes.data, lkf.ctypes.data, lk2.ctypes.data, lkx.ctypes.data)
    if rc != 0:
        raise ValueError(f"witgen rc={rc}")
    return out, lkd, lkf[:fetch_num_slots], lk2, lkx


def witgen_jal(cols, records, indices, shard_offset=0, fetch_base_pc=0, fetch_num_slots=0):
    """CPU assignment of the JAL chip: (matrix, dynamic counts, fetch counts, double-u8 counts, xor counts)"""
    return _witgen_4tab(lib().orc_witgen_jal, 13, cols, records, indices, shard_offset, fetch_base_pc, fetch_num_slots)


def witgen_auipc(cols, records, indices, shard_offset=0, fetch_base_pc=0, fetch_num_slots=0):
    """CPU assignment of the AUIPC chip: (matrix, dynamic counts, fetch counts, double-u8 counts, xor counts)"""
    return _witgen_4tab(lib().orc_witgen_auipc, 21, cols, records, indices, shard_offset, fetch_base_pc, fetch_num_slots)


INSN_SLL, INSN_SRL, INSN_SRA, INSN_SLLI, INSN_SRLI, INSN_SRAI = 6, 7, 8, 15, 16, 17  # InsnKind discriminants


def witgen_shift(cols, is_imm: bool, kind: int, records, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of a shift chip (kind 0 left, 1 logical right, 2 arithmetic right): (matrix, dynamic, fetch, double-u8, xor counts)"""
    nc = 40 if is_imm else 47
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (nc + 1,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[nc])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    lk2 = np.zeros(1 << 16, dtype=np.uint32)
    lkx = np.zeros(1 << 16, dtype=np.uint32)
    L = lib()
    L.orc_witgen_shift.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32] + [C.c_void_p] * 5
    L.orc_witgen_shift.restype = C.c_int
    rc = L.orc_witgen_shift(cols.ctypes.data, int(is_imm), int(kind), recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots,
                            out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data, lk2.ctypes.data, lkx.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_shift rc={rc}")
    return out, lkd, lkf[:fetch_num_slots], lk2, lkx


INSN_JALR = 27  # InsnKind::JALR


def step_records_jalr(cycles, pcs, pcs_after, rs1, rd, imms, rs1_vals, rd_before, rd_after, prev_cycles) -> np.ndarray:
    """an I-type record (StepRecord::new_i_instruction) whose pc.after is the jump target"""
    recs = step_records_i(cycles, pcs, INSN_JALR, rs1, rd, imms, rs1_vals, rd_before, rd_after, prev_cycles)
    recs[:, 12:16] = np.ascontiguousarray(np.asarray(pcs_after, dtype=np.uint64).astype("<u4")).view(np.uint8).reshape(-1, 4)
    return recs


def witgen_jalr(cols, records, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of the JALR chip: (row-major n x num_cols matrix, dynamic-table counts, fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (23,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[22])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_jalr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_witgen_jalr.restype = C.c_int
    rc = L.orc_witgen_jalr(cols.ctypes.data, recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots, out.ctypes.data,
                           lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_jalr rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_DIV, INSN_DIVU, INSN_REM, INSN_REMU = 32, 33, 34, 35


def witgen_div(cols, kind: int, records, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of DIV (kind 0) / DIVU (1) / REM (2) / REMU (3): (row-major n x num_cols matrix, dynamic-table counts, fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (40,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[39])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_div.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_witgen_div.restype = C.c_int
    rc = L.orc_witgen_div(cols.ctypes.data, int(kind), recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots, out.ctypes.data,
                          lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_div rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_MUL, INSN_MULH, INSN_MULHSU, INSN_MULHU = 28, 29, 30, 31


def witgen_mul(cols, kind: int, records, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of MUL (kind 0) / MULH (1) / MULHU (2) / MULHSU (3): (row-major n x num_cols matrix, dynamic-table counts, fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (27,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[26])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_mul.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_witgen_mul.restype = C.c_int
    rc = L.orc_witgen_mul(cols.ctypes.data, int(kind), recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots, out.ctypes.data,
                          lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_mul rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_LB, INSN_LH, INSN_LBU, INSN_LHU = 36, 37, 39, 40
NO_COLUMN = 0xFFFFFFFF


def load_sub_cols(ids, load_width: int, is_signed: bool, num_cols: int):
    """LoadSubColumnMap from a list of distinct ids: the fields the variant has take them in order, the others NO_COLUMN"""
    ids = list(ids)
    cols = [ids.pop(0) for _ in range(25)]
    cols += [ids.pop(0) for _ in range(3)] if load_width == 8 else [NO_COLUMN] * 3
    cols += [ids.pop(0)] if is_signed else [NO_COLUMN]
    return cols + [num_cols]


def witgen_load_sub(cols, load_width: int, is_signed: bool, records, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of LH / LHU / LB / LBU: (row-major n x num_cols matrix, dynamic-table counts, fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (30,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[29])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_load_sub.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
    L.orc_witgen_load_sub.restype = C.c_int
    rc = L.orc_witgen_load_sub(cols.ctypes.data, int(load_width), int(is_signed), recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc,
                               fetch_num_slots, out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_load_sub rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_SB, INSN_SH = 43, 44
INSN_LW, INSN_SW = 38, 45  # InsnKind::LW; SW = after LUI, AUIPC, SB, SH (u16limb_circuit feature)


def step_records_mem(is_store, cycles, pcs, kind, rs1, rs2_or_rd, imms, rs1_vals, rs2_vals, rd_before, rd_after, mem_addrs, mem_before, mem_after, prev_cycles,
                     mem_prev_cycles) -> np.ndarray:
    """load / store step records -> (n, 136) uint8 array"""
    n = len(cycles)
    out = np.zeros((n, lib().orc_step_record_bytes()), dtype=np.uint8)
    L = lib()
    L.orc_step_record_mem.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_int32] + [C.c_uint32] * 7 + \
        [C.c_uint64, C.c_uint64]
    L.orc_step_record_mem.restype = None
    for i in range(n):
        L.orc_step_record_mem(out[i].ctypes.data, int(is_store), int(cycles[i]), int(pcs[i]), kind, rs1, rs2_or_rd, int(imms[i]), int(rs1_vals[i]),
                              int(rs2_vals[i]), int(rd_before[i]), int(rd_after[i]), int(mem_addrs[i]), int(mem_before[i]), int(mem_after[i]),
                              int(prev_cycles[i]), int(mem_prev_cycles[i]))
    return out


def witgen_mem(cols, is_store, records: np.ndarray, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of the LW / SW chip: (row-major n x num_cols matrix, dynamic-table counts, fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    nc = {0: 23, 1: 23, 2: 24, 3: 29}[int(is_store)]  # 2 = SH, 3 = SB
    assert cols.shape == (nc + 1,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[nc])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_mem.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                 C.c_void_p]
    L.orc_witgen_mem.restype = C.c_int
    rc = L.orc_witgen_mem(cols.ctypes.data, int(is_store), recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots,
                          out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_mem rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_BEQ, INSN_BNE, INSN_BLT, INSN_BGE, INSN_BLTU, INSN_BGEU = 20, 21, 22, 23, 24, 25


def step_records_b(cycles, pcs, pcs_after, kind, rs1, rs2, imms, rs1_vals, rs2_vals, prev_cycles) -> np.ndarray:
    """B-type step records (rs1, rs2 read; no rd) -> (n, 136) uint8 array"""
    n = len(cycles)
    out = np.zeros((n, lib().orc_step_record_bytes()), dtype=np.uint8)
    L = lib()
    L.orc_step_record_b.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_int32, C.c_uint32, C.c_uint32,
                                    C.c_uint64]
    L.orc_step_record_b.restype = None
    for i in range(n):
        L.orc_step_record_b(out[i].ctypes.data, int(cycles[i]), int(pcs[i]), int(pcs_after[i]), kind, rs1, rs2, int(imms[i]), int(rs1_vals[i]),
                            int(rs2_vals[i]), int(prev_cycles[i]))
    return out


def witgen_branch(cols, is_eq: bool, flag: bool, records: np.ndarray, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of a branch chip (is_eq: BEQ / BNE with flag = is_beq, else BLT / BGE / BLTU / BGEU with flag = is_signed)"""
    nc = 19 if is_eq else 22
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (nc + 1,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[nc])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_branch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                    C.c_void_p, C.c_void_p]
    L.orc_witgen_branch.restype = C.c_int
    rc = L.orc_witgen_branch(cols.ctypes.data, int(is_eq), int(flag), recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc,
                             fetch_num_slots, out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_branch rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_SLTI, INSN_SLTIU = 18, 19


def witgen_slti(cols, is_signed: bool, records: np.ndarray, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of the SLTI / SLTIU chip: (row-major n x num_cols matrix, dynamic-table counts, fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (23,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[22])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_slti.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                  C.c_void_p]
    L.orc_witgen_slti.restype = C.c_int
    rc = L.orc_witgen_slti(cols.ctypes.data, int(is_signed), recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots,
                           out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_slti rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_SLT, INSN_SLTU = 9, 10


def witgen_slt(cols, is_signed: bool, records: np.ndarray, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of the SLT / SLTU chip: (row-major n x num_cols matrix, dynamic-table counts, fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (27,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[26])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_slt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                 C.c_void_p]
    L.orc_witgen_slt.restype = C.c_int
    rc = L.orc_witgen_slt(cols.ctypes.data, int(is_signed), recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots,
                          out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_slt rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_LUI = 41  # InsnKind::LUI (u16limb_circuit feature: after LHU = 40)
LUI_COLMAP_FIELDS = 17


def witgen_lui(cols, records: np.ndarray, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of the LUI chip: (row-major n x num_cols matrix, dynamic-table counts, fetch counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (LUI_COLMAP_FIELDS,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[16])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    L = lib()
    L.orc_witgen_lui.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_witgen_lui.restype = C.c_int
    rc = L.orc_witgen_lui(cols.ctypes.data, recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots,
                          out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_lui rc={rc}")
    return out, lkd, lkf[:fetch_num_slots]


INSN_XORI, INSN_ORI, INSN_ANDI = 12, 13, 14  # InsnKind discriminants
LOGIC_I_COLMAP_FIELDS = 25                   # 24 column ids in LogicIColumnMap order + num_cols


def witgen_logic_i(cols, records: np.ndarray, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of an ANDI / ORI / XORI chip: (row-major matrix, dynamic-table counts, fetch counts, the op's 2^16 table counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (LOGIC_I_COLMAP_FIELDS,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[24])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    lkl = np.zeros(1 << 16, dtype=np.uint32)
    L = lib()
    L.orc_witgen_logic_i.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    L.orc_witgen_logic_i.restype = C.c_int
    rc = L.orc_witgen_logic_i(cols.ctypes.data, recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots,
                              out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data, lkl.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_logic_i rc={rc}")
    return out, lkd, lkf[:fetch_num_slots], lkl


INSN_XOR, INSN_OR, INSN_AND = 3, 4, 5  # InsnKind discriminants (rv32im.rs:168-175)
LOGIC_COLMAP_FIELDS = 29                # 28 column ids in LogicRColumnMap order + num_cols


def witgen_logic_r(cols, records: np.ndarray, indices, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0):
    """CPU assignment of an AND / OR / XOR chip: (row-major n x num_cols matrix, dynamic-table counts, fetch counts, the op's 2^16 table counts)"""
    cols = np.ascontiguousarray(cols, dtype=np.uint32)
    assert cols.shape == (LOGIC_COLMAP_FIELDS,)
    idx = np.ascontiguousarray(indices, dtype=np.uint32)
    recs = np.ascontiguousarray(records)
    out = np.zeros((len(idx), int(cols[28])), dtype=np.uint64)
    lkd = np.zeros(1 << 19, dtype=np.uint32)
    lkf = np.zeros(max(fetch_num_slots, 1), dtype=np.uint32)
    lkl = np.zeros(1 << 16, dtype=np.uint32)
    L = lib()
    L.orc_witgen_logic_r.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    L.orc_witgen_logic_r.restype = C.c_int
    rc = L.orc_witgen_logic_r(cols.ctypes.data, recs.ctypes.data, idx.ctypes.data, len(idx), shard_offset, fetch_base_pc, fetch_num_slots,
                              out.ctypes.data, lkd.ctypes.data, lkf.ctypes.data, lkl.ctypes.data)
    if rc != 0:
        raise ValueError(f"orc_witgen_logic_r rc={rc}")
    return out, lkd, lkf[:fetch_num_slots], lkl


# ---- multi-layer GKR (control flow restated; arithmetic by the functions above) -----------------------------------
LAYER_ZEROCHECK, LAYER_LINEAR, LAYER_SUMCHECK = 0, 1, 2


def _scalars(scalars, chal):
    out = []
    for monos in scalars:
        sc = (0, 0)
        for coeff, ids in monos:
            v = (int(coeff[0]), int(coeff[1]))
            for i in ids:
                v = e2_mul(v, chal[i])
            sc = e2_add(sc, v)
        out.append(sc)
    return out


def gkr_prove(layers, claims, pub_io, challenges, tr):
    """GKRCircuit::prove (gkr_iop/src/gkr.rs:72-115) with Layer::prove / extract_claim_and_point / update_claims
    (gkr/layer.rs:198-243,289-322) and the three layer provers of gkr/layer/cpu/mod.rs (linear :47-66, sumcheck :72-96,
    zerocheck :102-238).  Same layer dictionaries as ceno_amd.prover.gkr_prove, with numpy tables instead of handles
    (None for a selector's structural slot).  Returns ([(msgs, evals, point)], final claims)."""
    claims = [(None if c[0] is None else np.ascontiguousarray(c[0], dtype=np.uint64).reshape(-1, 2), (int(c[1][0]), int(c[1][1]))) for c in claims]
    gch = [(int(c[0]), int(c[1])) for c in challenges]
    pio = [(int(c[0]), int(c[1])) for c in pub_io]

    def ev(e):
        if e[0] == "zero":
            return claims[0][0], (0, 0)
        if e[0] == "single":
            return claims[e[1]]
        pt, v = claims[e[1]]
        return pt, e2_add(e2_mul(v, (int(e[2][0]), int(e[2][1]))), (int(e[3][0]), int(e[3][1])))

    out = []
    for ly in layers:
        nv, tabs = ly["num_vars"], list(ly["mles"])
        gpts = [ev(g[1][0])[0] if g[1] else None for g in ly["groups"]]
        if ly["type"] == LAYER_LINEAR:
            point = gpts[0]
            evals = np.array([mle_evaluate(t, point) for t in tabs], dtype=np.uint64)
            for e in evals:
                tr.append_ext((int(e[0]), int(e[1])))
            msgs = np.zeros((0, 1, 2), dtype=np.uint64)
        else:
            if ly["type"] == LAYER_ZEROCHECK:
                tr.append_label(b"combine subset evals")
                a = tr.sample_ext()
                pows, acc = [], (1, 0)
                for _ in range(ly["n_exprs"]):
                    pows.append(acc)
                    acc = e2_mul(acc, a)
                chal = gch + pows + pio
                base = ly["n_witin"] + ly["n_fixed"]
                seen = set()
                for g, pt in zip(ly["groups"], gpts):
                    if g[0] is None or g[0][1] in seen:
                        continue
                    seen.add(g[0][1])
                    tabs[base + g[0][1]] = selector_compute(g[0][0], pt, g[0][2], g[0][3], g[0][4], g[0][5])
            else:
                chal = gch + pio
            coeffs = _scalars(ly["scalars"], chal)
            keep = [i for i, c in enumerate(coeffs) if c != (0, 0)]
            msgs, point, evals = sumcheck_prove(tabs, ext([coeffs[i] for i in keep]), [ly["terms"][i] for i in keep], nv, ly["max_degree"], tr)
            for e in evals:
                tr.append_ext((int(e[0]), int(e[1])))
        out.append((msgs, evals, point))
        for k, pos in enumerate(ly["in_eval_pos"][: len(tabs)]):
            claims[pos] = (np.array(point, dtype=np.uint64), (int(evals[k][0]), int(evals[k][1])))
    return out, claims
