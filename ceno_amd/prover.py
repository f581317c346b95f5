"""Python binding of the C++ host layer (libceno_prover.so, include/ceno_prover.h).

Names follow the reference: `sumcheck_prove` = IOPProverState::prove, `tower_create_proof` =
CpuTowerProver::create_proof, `prove_tower_relation` = TowerProver::prove_tower_relation.
"""
from __future__ import annotations

import ctypes as C
import time
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import SumcheckPlan, u32p, u64p
from .api import CenoHipError, Device, Mle, Sumcheck, _csr, _ext1, _p, _p32

_plib = None


class TowerProofC(C.Structure):
    _fields_ = [("num_rounds", C.c_int), ("msgs", u64p), ("prod_evals", u64p), ("logup_evals", u64p), ("point", u64p)]


def plib():
    global _plib
    if _plib is not None:
        return _plib
    _lib.lib()  # libceno_hip.so first (RTLD_GLOBAL)
    if not os.path.exists(_lib.PROVER_LIB_PATH):
        raise _lib.HipLibraryMissing(f"{_lib.PROVER_LIB_PATH} not found: run `python -m ceno_amd.build`")
    L = C.CDLL(_lib.PROVER_LIB_PATH)
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    vpp = C.POINTER(C.c_void_p)
    L.ceno_transcript_stub_new.restype = vp
    L.ceno_transcript_stub_new.argtypes = [C.c_uint64]
    L.ceno_transcript_free.restype = None
    L.ceno_transcript_free.argtypes = [vp]
    L.ceno_transcript_append_label.restype = None
    L.ceno_transcript_append_label.argtypes = [vp, C.c_char_p, sz]
    L.ceno_transcript_append_ext.restype = None
    L.ceno_transcript_append_ext.argtypes = [vp, u64p]
    L.ceno_transcript_sample_ext.restype = None
    L.ceno_transcript_sample_ext.argtypes = [vp, u64p]
    L.ceno_transcript_append_base.restype = None
    L.ceno_transcript_append_base.argtypes = [vp, C.c_uint64]
    L.ceno_transcript_sample_bits.restype = C.c_uint64
    L.ceno_transcript_sample_bits.argtypes = [vp, i]
    L.ceno_transcript_check_witness.restype = i
    L.ceno_transcript_check_witness.argtypes = [vp, i, C.c_uint64]
    L.ceno_transcript_clone.restype = vp
    L.ceno_transcript_clone.argtypes = [vp]
    L.ceno_transcript_export_state.restype = i
    L.ceno_transcript_export_state.argtypes = [vp, u64p]
    L.ceno_transcript_import_state.restype = i
    L.ceno_transcript_import_state.argtypes = [vp, u64p]
    L.ceno_prover_transcript_grind.restype = i
    L.ceno_prover_transcript_grind.argtypes = [vp, vp, i, vp, u64p]
    L.ceno_prover_sumcheck_prove.restype = i
    L.ceno_prover_sumcheck_prove.argtypes = [vp, vpp, C.POINTER(SumcheckPlan), vp, vp, u64p, u64p, u64p]
    L.ceno_prover_sumcheck_run.restype = i
    L.ceno_prover_sumcheck_run.argtypes = [vp, vp, i, i, i, vp, u64p, u64p, u64p]
    L.ceno_tower_msgs_words.restype = sz
    L.ceno_tower_msgs_words.argtypes = [i]
    L.ceno_prover_tower_create_proof.restype = i
    L.ceno_prover_tower_create_proof.argtypes = [vp, vpp, i, vpp, i, vp, vp, C.POINTER(TowerProofC)]
    L.ceno_prover_prove_tower_relation.restype = i
    L.ceno_prover_prove_tower_relation.argtypes = [vp, vpp, i, vpp, i, vp, vp, u64p, C.POINTER(TowerProofC)]
    L.ceno_prover_last_error.restype = C.c_char_p
    L.ceno_prover_last_error.argtypes = []
    L.ceno_prover_prove_rotation.restype = i
    L.ceno_prover_prove_rotation.argtypes = [vp, vpp, C.POINTER(i), C.POINTER(i), i, i, i, u64p, i, vp, vp, u64p, u64p, u64p, u64p, u64p]
    L.ceno_dist_unique_id.restype = i
    L.ceno_dist_unique_id.argtypes = [C.c_char_p]
    L.ceno_dist_comm_init.restype = i
    L.ceno_dist_comm_init.argtypes = [i, i, C.c_char_p, vpp]
    L.ceno_dist_comm_destroy.restype = None
    L.ceno_dist_comm_destroy.argtypes = [vp]
    L.ceno_dist_sumcheck_prove.restype = i
    L.ceno_dist_sumcheck_prove.argtypes = [vp, vp, vpp, C.POINTER(SumcheckPlan), i, vp, vp, u64p, u64p, u64p]
    L.ceno_dist_last_error.restype = C.c_char_p
    L.ceno_dist_last_error.argtypes = []
    for name in ("ceno_transcript_poseidon2_new",):
        if hasattr(L, name):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [C.c_char_p, sz]
    L.ceno_prover_test_gl_mul.restype = C.c_uint64
    L.ceno_prover_test_gl_mul.argtypes = [C.c_uint64, C.c_uint64]
    L.ceno_prover_test_gl_add.restype = C.c_uint64
    L.ceno_prover_test_gl_add.argtypes = [C.c_uint64, C.c_uint64]
    L.ceno_prover_test_gl_sub.restype = C.c_uint64
    L.ceno_prover_test_gl_sub.argtypes = [C.c_uint64, C.c_uint64]
    L.ceno_prover_test_gl_mul_small.restype = C.c_uint64
    L.ceno_prover_test_gl_mul_small.argtypes = [C.c_uint64, C.c_uint32]
    L.ceno_prover_test_e2_mul.restype = None
    L.ceno_prover_test_e2_mul.argtypes = [u64p, u64p, u64p]
    L.ceno_prover_test_gl_mul_ref.restype = C.c_uint64
    L.ceno_prover_test_gl_mul_ref.argtypes = [C.c_uint64, C.c_uint64]
    L.ceno_prover_test_gl_mul_nc.restype = C.c_uint64
    L.ceno_prover_test_gl_mul_nc.argtypes = [C.c_uint64, C.c_uint64]
    L.ceno_prover_test_gl_mul_add.restype = C.c_uint64
    L.ceno_prover_test_gl_mul_add.argtypes = [C.c_uint64] * 3
    L.ceno_prover_test_gl_mul_add2.restype = C.c_uint64
    L.ceno_prover_test_gl_mul_add2.argtypes = [C.c_uint64] * 4
    L.ceno_prover_test_e2_mul_ref.restype = None
    L.ceno_prover_test_e2_mul_ref.argtypes = [u64p, u64p, u64p]
    L.ceno_prover_test_e2_mul_pre.restype = None
    L.ceno_prover_test_e2_mul_pre.argtypes = [u64p, u64p, u64p]
    L.ceno_prover_test_e2_mul_nc.restype = None
    L.ceno_prover_test_e2_mul_nc.argtypes = [u64p, u64p, u64p]
    L.ceno_prover_test_e2_fma_pre.restype = None
    L.ceno_prover_test_e2_fma_pre.argtypes = [u64p, u64p, u64p, u64p]
    L.ceno_prover_test_e2_acc.restype = None
    L.ceno_prover_test_e2_acc.argtypes = [u64p, u64p, C.c_int, C.c_int, u64p]
    L.ceno_prover_test_e2_inv.restype = None
    L.ceno_prover_test_e2_inv.argtypes = [u64p, u64p]
    L.ceno_vp_builder_max_degree.restype = C.c_int
    L.ceno_vp_builder_max_degree.argtypes = [C.c_void_p]
    _plib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise CenoHipError(rc, (plib().ceno_prover_last_error() or b"").decode())


class Transcript:
    """handle on a ceno_transcript (reference: transcript::Transcript<E>)"""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)

    @classmethod
    def stub(cls, seed: int = 0xF5) -> "Transcript":
        return cls(plib().ceno_transcript_stub_new(C.c_uint64(seed)))

    @classmethod
    def poseidon2(cls, label: bytes = b"riscv") -> "Transcript":
        return cls(plib().ceno_transcript_poseidon2_new(label, len(label)))

    def append_label(self, b: bytes):
        plib().ceno_transcript_append_label(self.h, b, len(b))

    def append_ext(self, e):
        plib().ceno_transcript_append_ext(self.h, _p(_ext1(e)))

    def sample_ext(self) -> Tuple[int, int]:
        o = np.zeros(2, dtype=np.uint64)
        plib().ceno_transcript_sample_ext(self.h, _p(o))
        return int(o[0]), int(o[1])

    def append_base(self, v: int):
        plib().ceno_transcript_append_base(self.h, C.c_uint64(int(v)))

    def sample_bits(self, bits: int) -> int:
        return int(plib().ceno_transcript_sample_bits(self.h, bits))

    def sample_base(self) -> int:
        return self.sample_bits(64)

    def check_witness(self, bits: int, w: int) -> bool:
        return bool(plib().ceno_transcript_check_witness(self.h, bits, C.c_uint64(int(w))))

    def clone(self) -> "Transcript":
        h = plib().ceno_transcript_clone(self.h)
        if not h:
            raise RuntimeError("this transcript cannot fork")
        return Transcript(h)

    def export_state(self):
        """(kind, 16 words): kind 1 = Poseidon2 duplex [state 8][n_in][in 4][n_out][0][0], 0 = not exportable"""
        o = np.zeros(16, dtype=np.uint64)
        return int(plib().ceno_transcript_export_state(self.h, _p(o))), o

    def import_state(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint64)
        _check(plib().ceno_transcript_import_state(self.h, _p(w)))

    def grind(self, dev, bits: int, stream=None) -> int:
        """GrindingChallenger::grind: least witness accepted by check_witness on a clone; the transcript then observes it"""
        o = np.zeros(1, dtype=np.uint64)
        _check(plib().ceno_prover_transcript_grind(dev.h if dev is not None else None, self.h, bits, stream, _p(o)))
        return int(o[0])

    def __del__(self):
        try:
            if self.h:
                plib().ceno_transcript_free(self.h)
                self.h = None
        except Exception:
            pass


def make_plan(n_mles: int, coeffs: np.ndarray, terms, max_num_vars: int, max_degree: int, groups=None):
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 2)
    toff, tidx = _csr(terms)
    plan = SumcheckPlan()
    plan.num_mles, plan.num_terms = n_mles, len(terms)
    plan.term_coeffs, plan.term_offsets, plan.term_mle_idx = _p(coeffs), _p32(toff), _p32(tidx)
    keep = [coeffs, toff, tidx]
    if groups:
        goff, gidx = _csr([g[1] for g in groups])
        coff, cidx = _csr([g[0] for g in groups])
        plan.num_groups = len(groups)
        plan.group_term_offsets, plan.group_term_idx = _p32(goff), _p32(gidx)
        plan.common_offsets, plan.common_mle_idx = _p32(coff), _p32(cidx)
        keep += [goff, gidx, coff, cidx]
    plan.max_num_vars, plan.max_degree = max_num_vars, max_degree
    return plan, keep


def sumcheck_prove(dev: Device, mles: Sequence[Mle], coeffs: np.ndarray, terms, max_num_vars: int, max_degree: int,
                   tr: Transcript, groups=None, stream=None, eq_decls=None):
    """IOPProverState::prove — returns (msgs (n,d,2), challenges (n,2), final_evals (k,2)).
    eq_decls: [(mle index, point (nv,2), lo, hi)] — tables that ARE eq(., point) on the rows [lo, hi) (ceno_prover_sumcheck_prove_eq)"""
    plan, keep = make_plan(len(mles), coeffs, terms, max_num_vars, max_degree, groups)
    arr = (C.c_void_p * len(mles))(*[m.h for m in mles])
    msgs = np.zeros((max_num_vars, max_degree, 2), dtype=np.uint64)
    chal = np.zeros((max(max_num_vars, 1), 2), dtype=np.uint64)
    fin = np.zeros((len(mles), 2), dtype=np.uint64)
    if eq_decls:
        L = plib()
        L.ceno_prover_sumcheck_prove_eq.restype = C.c_int
        n = len(eq_decls)
        idx = (C.c_int * n)(*[int(d[0]) for d in eq_decls])
        pts = [np.ascontiguousarray(d[1], dtype=np.uint64) for d in eq_decls]
        ptp = (u64p * n)(*[_p(p_) for p_ in pts])
        lo = (C.c_size_t * n)(*[int(d[2]) for d in eq_decls])
        hi = (C.c_size_t * n)(*[int(d[3]) for d in eq_decls])
        _check(L.ceno_prover_sumcheck_prove_eq(dev.h, arr, C.byref(plan), n, idx, ptp, lo, hi, tr.h, stream,
                                               _p(msgs) if max_num_vars else None, _p(chal), _p(fin)))
        return msgs, chal[:max_num_vars], fin
    _check(plib().ceno_prover_sumcheck_prove(dev.h, arr, C.byref(plan), tr.h, stream,
                                             _p(msgs) if max_num_vars else None, _p(chal), _p(fin)))
    return msgs, chal[:max_num_vars], fin


class VirtualPolynomialsBuilder:
    """host-side plan builder (reference: multilinear_extensions::virtual_polys::VirtualPolynomialsBuilder)"""

    def __init__(self, dev: Device, max_num_vars: int):
        L = plib()
        vp, i = C.c_void_p, C.c_int
        L.ceno_vp_builder_new.restype = vp
        L.ceno_vp_builder_new.argtypes = [i]
        L.ceno_vp_builder_free.restype = None
        L.ceno_vp_builder_free.argtypes = [vp, vp]
        L.ceno_vp_builder_lift.restype = i
        L.ceno_vp_builder_lift.argtypes = [vp, vp, i]
        L.ceno_vp_builder_add_term.restype = i
        L.ceno_vp_builder_add_term.argtypes = [vp, u64p, C.POINTER(i), i]
        L.ceno_vp_builder_num_mles.restype = i
        L.ceno_vp_builder_num_mles.argtypes = [vp]
        L.ceno_vp_builder_prove.restype = i
        L.ceno_vp_builder_prove.argtypes = [vp, vp, i, vp, vp, u64p, u64p, u64p]
        self.dev, self.n = dev, max_num_vars
        self.h = C.c_void_p(L.ceno_vp_builder_new(max_num_vars))
        self._keep = []

    def lift(self, mle: Mle, owned: bool = False) -> int:
        r = plib().ceno_vp_builder_lift(self.h, mle.h, int(owned))
        if r < 0:
            _check(r)
        if owned:
            mle.h = None  # the builder frees it
        else:
            self._keep.append(mle)
        return r

    def add_term(self, scalar, product: Sequence[int]) -> int:
        arr = (C.c_int * len(product))(*product)
        r = plib().ceno_vp_builder_add_term(self.h, _p(_ext1(scalar)), arr, len(product))
        if r < 0:
            _check(r)
        return r

    def prove(self, transcript: "Transcript", max_degree: int = 0, stream=None):
        L = plib()
        k = L.ceno_vp_builder_num_mles(self.h)
        d = max_degree if max_degree > 0 else L.ceno_vp_builder_max_degree(self.h)
        msgs = np.zeros((self.n, d, 2), dtype=np.uint64)
        chal = np.zeros((self.n, 2), dtype=np.uint64)
        fin = np.zeros((k, 2), dtype=np.uint64)
        _check(L.ceno_vp_builder_prove(self.dev.h, self.h, max_degree, transcript.h, stream, _p(msgs), _p(chal), _p(fin)))
        return msgs, chal, fin

    def free(self):
        if getattr(self, "h", None) and self.dev.h:
            plib().ceno_vp_builder_free(self.dev.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class TowerSpecC(C.Structure):
    _fields_ = [("records", C.c_void_p), ("numerators", C.c_void_p), ("k", C.c_int), ("logup", C.c_int), ("num_instances", C.c_size_t),
                ("default2", C.c_uint64 * 2)]


class Tower:
    """built tower witness (reference: TowerProverSpec / GpuProverSpec)"""

    def __init__(self, dev: Device, h):
        self.dev, self.h = dev, h

    @classmethod
    def build_prod(cls, dev: Device, records: Sequence[Mle], num_instances: int, default=(1, 0), stream=None):
        arr = (C.c_void_p * len(records))(*[m.h for m in records])
        h = C.c_void_p()
        dev.check(dev.L.ceno_hip_tower_build_prod(dev.h, arr, len(records), num_instances, _p(_ext1(default)), stream, C.byref(h)))
        return cls(dev, h)

    @classmethod
    def build_logup(cls, dev: Device, p_records: Optional[Sequence[Mle]], q_records: Sequence[Mle], num_instances: int,
                    default, stream=None):
        q = (C.c_void_p * len(q_records))(*[m.h for m in q_records])
        p = (C.c_void_p * len(p_records))(*[m.h for m in p_records]) if p_records is not None else None
        h = C.c_void_p()
        dev.check(dev.L.ceno_hip_tower_build_logup(dev.h, p, q, len(q_records), num_instances, _p(_ext1(default)), stream, C.byref(h)))
        return cls(dev, h)

    @classmethod
    def build_many(cls, dev: Device, specs: Sequence[tuple], stream=None) -> list:
        """specs: ("prod", records, num_instances, default) or ("logup", p_records | None, q_records, num_instances, default): all towers in
        level-synchronous launches (ceno_hip_tower_build_many) — the same layers as build_prod / build_logup one by one"""
        arr = (TowerSpecC * len(specs))()
        keep = []
        for a, sp in zip(arr, specs):
            if sp[0] == "prod":
                _, recs, n_inst, dflt = sp
                nums = None
            else:
                _, nums, recs, n_inst, dflt = sp
            r = (C.c_void_p * len(recs))(*[m.h for m in recs])
            nm = (C.c_void_p * len(nums))(*[m.h for m in nums]) if nums is not None else None
            keep += [r, nm]
            a.records, a.numerators = C.cast(r, C.c_void_p), (C.cast(nm, C.c_void_p) if nm is not None else None)
            a.k, a.logup, a.num_instances = len(recs), int(sp[0] == "logup"), n_inst
            d = _ext1(dflt)
            a.default2[0], a.default2[1] = int(d[0]), int(d[1])
        hs = (C.c_void_p * len(specs))()
        dev.check(dev.L.ceno_hip_tower_build_many(dev.h, C.cast(arr, C.c_void_p), len(specs), stream, hs))
        return [cls(dev, C.c_void_p(h)) for h in hs]

    @classmethod
    def from_last_layer(cls, dev: Device, limbs: Sequence[Optional[Mle]], stream=None):
        arr = (C.c_void_p * len(limbs))(*[(m.h if m is not None else None) for m in limbs])
        h = C.c_void_p()
        dev.check(dev.L.ceno_hip_tower_from_last_layer(dev.h, arr, len(limbs), stream, C.byref(h)))
        return cls(dev, h)

    @property
    def num_vars(self) -> int:
        return self.dev.L.ceno_hip_tower_num_vars(self.h)

    @property
    def num_limbs(self) -> int:
        return self.dev.L.ceno_hip_tower_num_limbs(self.h)

    def layer(self, layer: int, limb: int) -> np.ndarray:
        h = C.c_void_p()
        self.dev.check(self.dev.L.ceno_hip_tower_layer(self.dev.h, self.h, layer, limb, C.byref(h)))
        m = Mle(self.dev, h)
        out = m.download()
        m.free()
        return out

    def out_evals(self, stream=None) -> np.ndarray:
        out = np.zeros((self.num_limbs, 2), dtype=np.uint64)
        self.dev.check(self.dev.L.ceno_hip_tower_out_evals(self.dev.h, self.h, _p(out), stream))
        return out

    def free(self):
        if getattr(self, "h", None) and self.dev.h:
            self.dev.L.ceno_hip_tower_free(self.dev.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class TowerProof:
    def __init__(self, max_nv: int, n_prod: int, n_logup: int):
        self.max_nv, self.n_prod, self.n_logup = max_nv, n_prod, n_logup
        R = max_nv - 1
        self.msgs = np.zeros(max(1, plib().ceno_tower_msgs_words(max_nv)), dtype=np.uint64)
        self.prod_evals = np.zeros((max(1, n_prod), max(1, R), 2, 2), dtype=np.uint64)
        self.logup_evals = np.zeros((max(1, n_logup), max(1, R), 4, 2), dtype=np.uint64)
        self.point = np.zeros((max_nv + 1, 2), dtype=np.uint64)
        self.c = TowerProofC(R, _p(self.msgs), _p(self.prod_evals), _p(self.logup_evals), _p(self.point))


def tower_create_proof(dev: Device, prod: Sequence[Tower], logup: Sequence[Tower], tr: Transcript, stream=None) -> TowerProof:
    max_nv = max([t.num_vars for t in prod] + [t.num_vars for t in logup])
    proof = TowerProof(max_nv, len(prod), len(logup))
    pa = (C.c_void_p * max(1, len(prod)))(*[t.h for t in prod])
    la = (C.c_void_p * max(1, len(logup)))(*[t.h for t in logup])
    _check(plib().ceno_prover_tower_create_proof(dev.h, pa, len(prod), la, len(logup), tr.h, stream, C.byref(proof.c)))
    return proof


def prove_tower_relation(dev: Device, prod: Sequence[Tower], logup: Sequence[Tower], tr: Transcript, stream=None):
    max_nv = max([t.num_vars for t in prod] + [t.num_vars for t in logup])
    proof = TowerProof(max_nv, len(prod), len(logup))
    pa = (C.c_void_p * max(1, len(prod)))(*[t.h for t in prod])
    la = (C.c_void_p * max(1, len(logup)))(*[t.h for t in logup])
    out_evals = np.zeros((2 * len(prod) + 4 * len(logup), 2), dtype=np.uint64)
    _check(plib().ceno_prover_prove_tower_relation(dev.h, pa, len(prod), la, len(logup), tr.h, stream, _p(out_evals), C.byref(proof.c)))
    return out_evals, proof


class RcclComm:
    """RCCL communicator for the C++ sharded prover (ceno_amd/host/dist.cpp).  `dist` is an initialised
    torch.distributed module (or None for world 1): it only carries the 128-byte unique id."""

    def __init__(self, world: int, rank: int, dist=None):
        if "CENO_RCCL_PATH" not in os.environ:  # share the RCCL build torch.distributed already loaded
            try:
                import torch

                cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
                if os.path.exists(cand):
                    os.environ["CENO_RCCL_PATH"] = cand
            except Exception:
                pass
        L = plib()
        idbuf = C.create_string_buffer(128)
        if rank == 0:
            rc = L.ceno_dist_unique_id(idbuf)
            if rc != 0:
                raise CenoHipError(rc, (L.ceno_dist_last_error() or b"").decode())
        if world > 1:
            obj = [bytes(idbuf.raw)]
            dist.broadcast_object_list(obj, src=0)
            idbuf = C.create_string_buffer(obj[0], 128)
        h = C.c_void_p()
        rc = L.ceno_dist_comm_init(world, rank, idbuf, C.byref(h))
        if rc != 0:
            raise CenoHipError(rc, (L.ceno_dist_last_error() or b"").decode())
        self.h, self.world, self.rank = h, world, rank

    def nranks(self) -> int:
        """the rank count RCCL itself reports for this communicator (ncclCommCount)"""
        L = plib()
        L.ceno_dist_comm_rccl_ranks.restype = C.c_int
        L.ceno_dist_comm_rccl_ranks.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        r = C.c_int(-1)
        n = L.ceno_dist_comm_rccl_ranks(self.h, C.byref(r))
        if n < 0:
            raise CenoHipError(n, (L.ceno_dist_last_error() or b"").decode())
        return int(n)

    def close(self):
        if getattr(self, "h", None):
            plib().ceno_dist_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShmComm:
    """Communicator whose per-round exchange goes through a host shared-memory segment (one node; see
    include/ceno_prover.h ceno_dist_comm_attach_shm).  `dist` only carries the segment name and two barriers."""

    def __init__(self, world: int, rank: int, dist=None, name: Optional[str] = None):
        L = plib()
        L.ceno_dist_comm_attach_shm.restype = C.c_int
        L.ceno_dist_comm_attach_shm.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.ceno_dist_shm_unlink.restype = C.c_int
        L.ceno_dist_shm_unlink.argtypes = [C.c_char_p]
        L.ceno_dist_shm_selftest.restype = C.c_int
        L.ceno_dist_shm_selftest.argtypes = [C.c_void_p, C.c_int]
        if name is None:
            name = f"/ceno_dist_{os.getpid()}_{int.from_bytes(os.urandom(4), 'little'):08x}" if rank == 0 else None
        h = C.c_void_p()

        def attach(create):
            rc = L.ceno_dist_comm_attach_shm(C.byref(h), world, rank, name.encode(), int(create))
            if rc != 0:
                raise CenoHipError(rc, (L.ceno_dist_last_error() or b"").decode())

        # every rank must leave this constructor the same way (all succeed or all raise): a rank that failed alone would
        # leave the others waiting in the next collective
        err = None
        if rank == 0:
            try:
                attach(True)  # the segment exists and is initialised before anybody learns its name
            except Exception as e:  # noqa: BLE001
                err, name = e, None
        if world > 1 and dist is not None:
            obj = [name]
            dist.broadcast_object_list(obj, src=0)
            name = obj[0]
        if name is None:
            raise err if err is not None else RuntimeError("rank 0 could not create the shared-memory segment")
        if rank != 0:
            try:
                attach(False)
            except Exception as e:  # noqa: BLE001
                err = e
        if world > 1 and dist is not None:
            oks = [None] * world
            dist.all_gather_object(oks, err is None)
        else:
            oks = [err is None]
        if rank == 0:
            L.ceno_dist_shm_unlink(name.encode())  # mappings stay valid; nothing is left behind in /dev/shm
        if not all(oks):
            if h:
                L.ceno_dist_comm_destroy(h)
            raise err if err is not None else RuntimeError("another rank could not attach the shared-memory segment")
        self.h, self.world, self.rank, self.name = h, world, rank, name

    def selftest(self, iters: int = 1000) -> int:
        return plib().ceno_dist_shm_selftest(self.h, iters)

    def close(self):
        if getattr(self, "h", None):
            plib().ceno_dist_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DistClassC(C.Structure):
    _fields_ = [("num_vars", C.c_int), ("sharded", C.c_int), ("num_mles", C.c_int), ("mles", C.POINTER(C.c_void_p)),
                ("num_terms", C.c_int), ("term_coeffs", u64p), ("term_offsets", u32p), ("term_mle_idx", u32p)]


def dist_batched_sumcheck_prove(dev: Device, comm, classes: Sequence[dict], n_total: int, max_degree: int, tr: Transcript, stream):
    """mixed-size batched sumcheck across ranks (C++ driver, shared-memory exchange).  classes: dicts with num_vars
    (global), sharded, mles (Mle handles of this rank), coeffs (T,2), terms [[class-local ids]].
    Returns (msgs, challenges, [final evals per class])."""
    L = plib()
    L.ceno_dist_batched_sumcheck_prove.restype = C.c_int
    arr = (DistClassC * len(classes))()
    keep = []
    for i, c in enumerate(classes):
        mh = (C.c_void_p * len(c["mles"]))(*[m.h for m in c["mles"]])
        coeffs = np.ascontiguousarray(c["coeffs"], dtype=np.uint64).reshape(-1, 2)
        toff, tidx = _csr(c["terms"])
        K = arr[i]
        K.num_vars, K.sharded, K.num_mles, K.mles = int(c["num_vars"]), int(bool(c["sharded"])), len(c["mles"]), mh
        K.num_terms, K.term_coeffs, K.term_offsets, K.term_mle_idx = len(c["terms"]), _p(coeffs), _p32(toff), _p32(tidx)
        keep += [mh, coeffs, toff, tidx]
    total = sum(len(c["mles"]) for c in classes)
    msgs = np.zeros((n_total, max_degree, 2), dtype=np.uint64)
    chal = np.zeros((n_total, 2), dtype=np.uint64)
    fin = np.zeros((total, 2), dtype=np.uint64)
    rc = L.ceno_dist_batched_sumcheck_prove(dev.h, comm.h, arr, len(classes), n_total, max_degree, tr.h, stream, _p(msgs), _p(chal), _p(fin))
    if rc != 0:
        raise CenoHipError(rc, (L.ceno_dist_last_error() or b"").decode())
    out, off = [], 0
    for c in classes:
        out.append(fin[off: off + len(c["mles"])])
        off += len(c["mles"])
    return msgs, chal, out


def dist_sumcheck_prove(dev: Device, comm, mles: Sequence[Mle], coeffs: np.ndarray, terms, n_total: int,
                        max_degree: int, tr: Transcript, stream):
    """sharded IOPProverState::prove — `mles` are this rank's shards; returns the global proof"""
    n_local = n_total - (comm.world.bit_length() - 1)
    plan, keep = make_plan(len(mles), coeffs, terms, n_local, max_degree)
    arr = (C.c_void_p * len(mles))(*[m.h for m in mles])
    msgs = np.zeros((n_total, max_degree, 2), dtype=np.uint64)
    chal = np.zeros((max(n_total, 1), 2), dtype=np.uint64)
    fin = np.zeros((len(mles), 2), dtype=np.uint64)
    rc = plib().ceno_dist_sumcheck_prove(dev.h, comm.h, arr, C.byref(plan), n_total, tr.h, stream, _p(msgs), _p(chal), _p(fin))
    if rc != 0:
        raise CenoHipError(rc, (plib().ceno_dist_last_error() or b"").decode())
    return msgs, chal[:n_total], fin


def prove_rotation(dev: Device, wit: Sequence[Mle], pairs: Sequence[Tuple[int, int]], cyclic_subgroup_size: int,
                   cyclic_group_log2: int, rt: np.ndarray, tr: Transcript, stream=None):
    """prove_rotation (gkr_iop/src/gkr/layer/cpu/mod.rs:249-389) -> (msgs (n,2,2), evals (3*pairs,2), origin, left, right)"""
    rt = np.ascontiguousarray(rt, dtype=np.uint64).reshape(-1, 2)
    n = rt.shape[0]
    arr = (C.c_void_p * len(wit))(*[m.h for m in wit])
    src = (C.c_int * len(pairs))(*[p[0] for p in pairs])
    tgt = (C.c_int * len(pairs))(*[p[1] for p in pairs])
    msgs = np.zeros((n, 2, 2), dtype=np.uint64)
    evals = np.zeros((3 * len(pairs), 2), dtype=np.uint64)
    origin, left, right = (np.zeros((n, 2), dtype=np.uint64) for _ in range(3))
    _check(plib().ceno_prover_prove_rotation(dev.h, arr, src, tgt, len(pairs), cyclic_subgroup_size, cyclic_group_log2, _p(rt), n,
                                             tr.h, stream, _p(msgs), _p(evals), _p(origin), _p(left), _p(right)))
    return msgs, evals, origin, left, right


class MainJobC(C.Structure):
    _fields_ = [
        ("circuit_idx", C.c_int), ("num_vars", C.c_int), ("n_witin", C.c_int), ("n_fixed", C.c_int), ("n_structural", C.c_int),
        ("mles", C.POINTER(C.c_void_p)), ("n_selectors", C.c_int), ("sel_kind", C.POINTER(C.c_int)),
        ("sel_offset", C.POINTER(C.c_size_t)), ("sel_num_instances", C.POINTER(C.c_size_t)), ("sel_structural_id", C.POINTER(C.c_int)),
        ("sel_sparse_indices", C.POINTER(u32p)), ("sel_n_sparse", C.POINTER(C.c_int)), ("sel_sparse_num_vars", C.POINTER(C.c_int)),
        ("sel_points", C.POINTER(u64p)), ("n_exprs", C.c_int), ("max_degree", C.c_int), ("n_terms", C.c_int),
        ("term_offsets", u32p), ("term_mle_idx", u32p), ("scalar_offsets", u32p), ("mono_coeffs", u64p),
        ("mono_chal_offsets", u32p), ("mono_chal_idx", u32p), ("n_pi", C.c_int), ("pi", u64p),
    ]


class MainJobs:
    """the C view (ceno_main_job array) of a list of job dicts, marshalled once"""

    def __init__(self, jobs: Sequence[dict]):
        self.arr = (MainJobC * len(jobs))()
        self.keep = [jobs]  # the job dicts own the device tables the handles point to
        self.n = len(jobs)
        self.total_mles = 0
        self.max_nv = max(j["num_vars"] for j in jobs)
        self.max_deg = max(j["max_degree"] for j in jobs)
        keep = self.keep
        for c, j in enumerate(jobs):
            J = self.arr[c]
            J.circuit_idx, J.num_vars = j.get("circuit_idx", c), j["num_vars"]
            J.n_witin, J.n_fixed, J.n_structural = j["n_witin"], j["n_fixed"], j["n_structural"]
            mh = (C.c_void_p * len(j["mles"]))(*[(m.h if m is not None else None) for m in j["mles"]])
            self.total_mles += len(j["mles"])
            sels = j["selectors"]
            ns = len(sels)
            kinds = (C.c_int * max(ns, 1))(*[s[0] for s in sels])
            offs = (C.c_size_t * max(ns, 1))(*[s[1] for s in sels])
            nins = (C.c_size_t * max(ns, 1))(*[s[2] for s in sels])
            sids = (C.c_int * max(ns, 1))(*[s[3] for s in sels])
            sp_arrays = [np.array(list(s[4]) or [0], dtype=np.uint32) for s in sels]
            spp = (u32p * max(ns, 1))(*[_p32(a) for a in sp_arrays])
            nsp = (C.c_int * max(ns, 1))(*[len(s[4]) for s in sels])
            snv = (C.c_int * max(ns, 1))(*[s[5] for s in sels])
            pts = [np.ascontiguousarray(s[6], dtype=np.uint64) for s in sels]
            ptp = (u64p * max(ns, 1))(*[_p(p) for p in pts])
            toff, tidx = _csr(j["terms"])
            soff = np.zeros(len(j["terms"]) + 1, dtype=np.uint32)
            mono_c, mono_off, mono_idx = [], [0], []
            for t, monos in enumerate(j["scalars"]):
                for coeff, ids in monos:
                    mono_c.append([int(coeff[0]), int(coeff[1])])
                    mono_idx.extend(ids)
                    mono_off.append(len(mono_idx))
                soff[t + 1] = len(mono_c)
            mono_c = np.array(mono_c if mono_c else [[0, 0]], dtype=np.uint64)
            mono_off = np.array(mono_off, dtype=np.uint32)
            mono_idx = np.array(mono_idx if mono_idx else [0], dtype=np.uint32)
            J.mles, J.n_selectors = mh, ns
            J.sel_kind, J.sel_offset, J.sel_num_instances, J.sel_structural_id = kinds, offs, nins, sids
            J.sel_sparse_indices, J.sel_n_sparse, J.sel_sparse_num_vars, J.sel_points = spp, nsp, snv, ptp
            J.n_exprs, J.max_degree, J.n_terms = j["n_exprs"], j["max_degree"], len(j["terms"])
            J.term_offsets, J.term_mle_idx, J.scalar_offsets = _p32(toff), _p32(tidx), _p32(soff)
            J.mono_coeffs, J.mono_chal_offsets, J.mono_chal_idx = _p(mono_c), _p32(mono_off), _p32(mono_idx)
            pi = np.ascontiguousarray(j.get("pi", []), dtype=np.uint64).reshape(-1, 2)
            J.n_pi, J.pi = pi.shape[0], (_p(pi) if pi.shape[0] else None)
            keep += [pi, mh, kinds, offs, nins, sids, sp_arrays, spp, nsp, snv, pts, ptp, toff, tidx, soff, mono_c, mono_off, mono_idx]


def prove_batched_main_constraints(dev: Device, jobs, global_challenges, tr: Transcript, stream=None):
    """jobs: dicts with keys num_vars, mles (witness++fixed++structural, None allowed for replaced structural slots),
    n_witin, n_fixed, n_structural, selectors [(kind, offset, num_instances, structural_id, sparse_indices, sparse_num_vars, point)],
    n_exprs, max_degree, terms [[mle ids]], scalars [[(coeff_ext, [challenge ids])...] per term], optional pi [(c0, c1)...]
    (public-instance values: challenge ids >= 2 + n_exprs select them) — or a MainJobs built from such a list once.
    Returns (claimed_sum, msgs (n,d,2), global_rt (n,2), evals (total_mles,2))."""
    L = plib()
    L.ceno_prover_prove_batched_main_constraints.restype = C.c_int
    mj = jobs if isinstance(jobs, MainJobs) else MainJobs(jobs)
    gc = np.array([[int(global_challenges[0][0]), int(global_challenges[0][1])],
                   [int(global_challenges[1][0]), int(global_challenges[1][1])]], dtype=np.uint64)
    claimed = np.zeros(2, dtype=np.uint64)
    msgs = np.zeros((mj.max_nv, mj.max_deg, 2), dtype=np.uint64)
    rt = np.zeros((mj.max_nv, 2), dtype=np.uint64)
    evals = np.zeros((mj.total_mles, 2), dtype=np.uint64)
    nv_o, d_o = C.c_int(), C.c_int()
    t0 = time.perf_counter()
    rc = L.ceno_prover_prove_batched_main_constraints(dev.h, mj.arr, mj.n, _p(gc), tr.h, stream, _p(claimed), _p(msgs), _p(rt),
                                                      _p(evals), C.byref(nv_o), C.byref(d_o))
    prove_batched_main_constraints.last_native_ms = (time.perf_counter() - t0) * 1e3
    _check(rc)
    return (int(claimed[0]), int(claimed[1])), msgs, rt, evals


def dist_prove_batched_main_constraints(dev: Device, comm, jobs_local, global_challenges, tr: Transcript, q: int = 0, stream=None):
    """ceno_dist_prove_batched_main_constraints: prove_batched_main_constraints over row-sharded tables (the block layout of
    dist_create_chip_proof: shard_rows(column, world, rank, q)).  jobs_local: as for prove_batched_main_constraints with THIS rank's tables and the
    GLOBAL num_vars / selector ranges / points.  Returns what prove_batched_main_constraints returns for the whole tables."""
    L = plib()
    L.ceno_dist_prove_batched_main_constraints.restype = C.c_int
    mj = jobs_local if isinstance(jobs_local, MainJobs) else MainJobs(jobs_local)
    gc = np.array([[int(global_challenges[0][0]), int(global_challenges[0][1])],
                   [int(global_challenges[1][0]), int(global_challenges[1][1])]], dtype=np.uint64)
    claimed = np.zeros(2, dtype=np.uint64)
    msgs = np.zeros((mj.max_nv, mj.max_deg, 2), dtype=np.uint64)
    rt = np.zeros((mj.max_nv, 2), dtype=np.uint64)
    evals = np.zeros((mj.total_mles, 2), dtype=np.uint64)
    nv_o, d_o = C.c_int(), C.c_int()
    L.ceno_dist_prove_batched_main_constraints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, u64p, C.c_void_p, C.c_void_p, u64p, u64p, u64p, u64p,
                                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]
    _check(L.ceno_dist_prove_batched_main_constraints(dev.h, comm, C.cast(mj.arr, C.c_void_p), mj.n, q, _p(gc), tr.h, stream, _p(claimed), _p(msgs), _p(rt),
                                                      _p(evals), C.byref(nv_o), C.byref(d_o)))
    return (int(claimed[0]), int(claimed[1])), msgs, rt, evals


class EvalExprC(C.Structure):
    _fields_ = [("kind", C.c_int), ("idx", C.c_int), ("c0", C.c_uint64 * 2), ("c1", C.c_uint64 * 2)]


class GkrLayerC(C.Structure):
    _fields_ = [("type", C.c_int), ("num_vars", C.c_int), ("n_witin", C.c_int), ("n_fixed", C.c_int), ("n_structural", C.c_int),
                ("mles", C.POINTER(C.c_void_p)), ("n_groups", C.c_int), ("group_sel_kind", C.POINTER(C.c_int)),
                ("group_sel_structural_id", C.POINTER(C.c_int)), ("group_sel_offset", C.POINTER(C.c_size_t)),
                ("group_sel_num_instances", C.POINTER(C.c_size_t)), ("group_sel_sparse_indices", C.POINTER(u32p)),
                ("group_sel_n_sparse", C.POINTER(C.c_int)), ("group_sel_sparse_num_vars", C.POINTER(C.c_int)), ("group_expr_offsets", u32p),
                ("out_exprs", C.POINTER(EvalExprC)), ("n_exprs", C.c_int), ("max_degree", C.c_int), ("n_terms", C.c_int),
                ("term_offsets", u32p), ("term_mle_idx", u32p), ("scalar_offsets", u32p), ("mono_coeffs", u64p),
                ("mono_chal_offsets", u32p), ("mono_chal_idx", u32p), ("n_in_evals", C.c_int), ("in_eval_pos", C.POINTER(C.c_int))]


LAYER_ZEROCHECK, LAYER_LINEAR, LAYER_SUMCHECK = 0, 1, 2


def gkr_prove(dev: Device, layers: Sequence[dict], max_num_vars: int, claims: Sequence, pub_io, challenges, tr: Transcript, stream=None):
    """GKRCircuit::prove (gkr_iop/src/gkr.rs:72-115).  layers (output side first): dicts with type, num_vars, mles (witin ++ fixed ++
    structural, None for selector slots), n_witin, n_fixed, n_structural, groups [(selector | None, [exprs])] with selector =
    (kind, structural_id, offset, num_instances, sparse_indices, sparse_num_vars) and expr = ("zero",) | ("single", idx) |
    ("linear", idx, c0, c1), n_exprs, max_degree, terms, scalars (as in prove_batched_main_constraints), in_eval_pos.
    claims: [(point (k,2) or None, eval)] per evaluation slot.
    Returns ([(msgs, evals, point)] per layer, final claims [(point, eval)])."""
    L = plib()
    L.ceno_prover_gkr_prove.restype = C.c_int
    arr = (GkrLayerC * len(layers))()
    keep, outs = [], []
    for li, ly in enumerate(layers):
        G = arr[li]
        n_m = len(ly["mles"])
        mh = (C.c_void_p * n_m)(*[(m.h if m is not None else None) for m in ly["mles"]])
        groups = ly["groups"]
        ng = len(groups)
        kinds = (C.c_int * ng)(*[(g[0][0] if g[0] is not None else -1) for g in groups])
        sids = (C.c_int * ng)(*[(g[0][1] if g[0] is not None else 0) for g in groups])
        offs = (C.c_size_t * ng)(*[(g[0][2] if g[0] is not None else 0) for g in groups])
        nins = (C.c_size_t * ng)(*[(g[0][3] if g[0] is not None else 0) for g in groups])
        sp_arrays = [np.array(list(g[0][4]) or [0], dtype=np.uint32) if g[0] is not None else np.zeros(1, dtype=np.uint32) for g in groups]
        spp = (u32p * ng)(*[_p32(a) for a in sp_arrays])
        nsp = (C.c_int * ng)(*[(len(g[0][4]) if g[0] is not None else 0) for g in groups])
        snv = (C.c_int * ng)(*[(g[0][5] if g[0] is not None else 0) for g in groups])
        eoff = np.zeros(ng + 1, dtype=np.uint32)
        flat = []
        for gi, g in enumerate(groups):
            flat += list(g[1])
            eoff[gi + 1] = len(flat)
        exprs = (EvalExprC * max(1, len(flat)))()
        for k, e in enumerate(flat):
            exprs[k].kind = {"zero": 0, "single": 1, "linear": 2}[e[0]]
            exprs[k].idx = e[1] if len(e) > 1 else 0
            if e[0] == "linear":
                exprs[k].c0[0], exprs[k].c0[1] = int(e[2][0]), int(e[2][1])
                exprs[k].c1[0], exprs[k].c1[1] = int(e[3][0]), int(e[3][1])
        terms = ly.get("terms", [])
        toff, tidx = _csr(terms)
        soff = np.zeros(len(terms) + 1, dtype=np.uint32)
        mono_c, mono_off, mono_idx = [], [0], []
        for t, monos in enumerate(ly.get("scalars", [])):
            for coeff, ids in monos:
                mono_c.append([int(coeff[0]), int(coeff[1])])
                mono_idx.extend(ids)
                mono_off.append(len(mono_idx))
            soff[t + 1] = len(mono_c)
        mono_c = np.array(mono_c if mono_c else [[0, 0]], dtype=np.uint64)
        mono_off = np.array(mono_off, dtype=np.uint32)
        mono_idx = np.array(mono_idx if mono_idx else [0], dtype=np.uint32)
        inpos = (C.c_int * max(1, len(ly["in_eval_pos"])))(*ly["in_eval_pos"])
        G.type, G.num_vars = ly["type"], ly["num_vars"]
        G.n_witin, G.n_fixed, G.n_structural, G.mles = ly["n_witin"], ly["n_fixed"], ly["n_structural"], mh
        G.n_groups, G.group_sel_kind, G.group_sel_structural_id, G.group_sel_offset, G.group_sel_num_instances = ng, kinds, sids, offs, nins
        G.group_sel_sparse_indices, G.group_sel_n_sparse, G.group_sel_sparse_num_vars = spp, nsp, snv
        G.group_expr_offsets, G.out_exprs = _p32(eoff), exprs
        G.n_exprs, G.max_degree, G.n_terms = ly.get("n_exprs", 0), ly.get("max_degree", 1), len(terms)
        G.term_offsets, G.term_mle_idx, G.scalar_offsets = _p32(toff), _p32(tidx), _p32(soff)
        G.mono_coeffs, G.mono_chal_offsets, G.mono_chal_idx = _p(mono_c), _p32(mono_off), _p32(mono_idx)
        G.n_in_evals, G.in_eval_pos = len(ly["in_eval_pos"]), inpos
        keep += [mh, kinds, sids, offs, nins, sp_arrays, spp, nsp, snv, eoff, exprs, toff, tidx, soff, mono_c, mono_off, mono_idx, inpos]
        nv, d = ly["num_vars"], ly.get("max_degree", 1)
        outs.append((np.zeros((max(nv, 1), d, 2), dtype=np.uint64), np.zeros((n_m, 2), dtype=np.uint64), np.zeros((max(nv, 1), 2), dtype=np.uint64)))
    n_ev = len(claims)
    pts = [np.ascontiguousarray(c[0], dtype=np.uint64).reshape(-1, 2) if c[0] is not None else np.zeros((0, 2), dtype=np.uint64) for c in claims]
    cp = (u64p * n_ev)(*[(_p(p) if p.shape[0] else None) for p in pts])
    cl = (C.c_int * n_ev)(*[p.shape[0] for p in pts])
    ce = np.array([[int(c[1][0]), int(c[1][1])] for c in claims], dtype=np.uint64)
    pio = np.ascontiguousarray(pub_io, dtype=np.uint64).reshape(-1, 2)
    gc = np.array([[int(c[0]), int(c[1])] for c in challenges], dtype=np.uint64)
    om = (u64p * len(layers))(*[_p(o[0]) for o in outs])
    oe = (u64p * len(layers))(*[_p(o[1]) for o in outs])
    op = (u64p * len(layers))(*[_p(o[2]) for o in outs])
    ocp = np.zeros((n_ev, max(max_num_vars, 1), 2), dtype=np.uint64)
    ocl = (C.c_int * n_ev)()
    oce = np.zeros((n_ev, 2), dtype=np.uint64)
    _check(L.ceno_prover_gkr_prove(dev.h, arr, len(layers), max_num_vars, n_ev, cp, cl, _p(ce), _p(pio) if pio.shape[0] else None, pio.shape[0], _p(gc),
                                   tr.h, stream, om, oe, op, _p(ocp), ocl, _p(oce)))
    res = [(o[0][: ly["num_vars"]], o[1], o[2][: ly["num_vars"]]) for o, ly in zip(outs, layers)]
    final = [(ocp[i, : ocl[i]].copy() if ocl[i] else None, (int(oce[i, 0]), int(oce[i, 1]))) for i in range(n_ev)]
    return res, final


class TowerWitnessC(C.Structure):
    _fields_ = [("prod", C.c_void_p * 2), ("n_prod", C.c_int), ("logup", C.c_void_p * 1), ("n_logup", C.c_int),
                ("has_r", C.c_int), ("has_w", C.c_int), ("has_lk", C.c_int),
                ("r_out_evals", C.c_uint64 * 4), ("w_out_evals", C.c_uint64 * 4), ("lk_out_evals", C.c_uint64 * 8)]


def build_tower_witness(dev: Device, records: Sequence[Mle], num_reads: int, num_writes: int, num_lk_tables: int, num_lk: int,
                        log2_num_instances: int, rotation_vars: int, challenges, stream=None):
    """CpuProver::build_tower_witness (scheme/cpu/mod.rs:608-757) -> (out_evals dict, prod towers, logup towers)"""
    L = plib()
    L.ceno_prover_build_tower_witness.restype = C.c_int
    L.ceno_prover_build_tower_witness.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  u64p, C.c_void_p, C.POINTER(TowerWitnessC)]
    arr = (C.c_void_p * max(1, len(records)))(*[m.h for m in records])
    ch = np.array([[int(c[0]), int(c[1])] for c in challenges], dtype=np.uint64)
    tw = TowerWitnessC()
    _check(L.ceno_prover_build_tower_witness(dev.h, arr, num_reads, num_writes, num_lk_tables, num_lk, log2_num_instances, rotation_vars,
                                             _p(ch), stream, C.byref(tw)))
    evals = {"r": np.array(tw.r_out_evals, dtype=np.uint64).reshape(2, 2) if tw.has_r else None,
             "w": np.array(tw.w_out_evals, dtype=np.uint64).reshape(2, 2) if tw.has_w else None,
             "lk": np.array(tw.lk_out_evals, dtype=np.uint64).reshape(4, 2) if tw.has_lk else None}
    prod = [Tower(dev, C.c_void_p(tw.prod[i])) for i in range(tw.n_prod)]
    logup = [Tower(dev, C.c_void_p(tw.logup[i])) for i in range(tw.n_logup)]
    return evals, prod, logup


class ChipTaskC(C.Structure):
    _fields_ = [("circuit_idx", C.c_int), ("num_instances", C.c_size_t), ("log2_num_instances", C.c_int), ("rotation_vars", C.c_int),
                ("n_witin", C.c_int), ("n_fixed", C.c_int), ("n_structural", C.c_int), ("mles", C.POINTER(C.c_void_p)),
                ("num_reads", C.c_int), ("num_writes", C.c_int), ("num_lk_tables", C.c_int), ("num_lk", C.c_int),
                ("n_record_terms", C.c_int), ("record_coeffs", u64p), ("record_term_offsets", u32p), ("record_term_mle_idx", u32p),
                ("record_out_term_offsets", u32p), ("n_rotation_pairs", C.c_int), ("rotation_source_idx", C.POINTER(C.c_int)),
                ("rotation_target_idx", C.POINTER(C.c_int)), ("cyclic_subgroup_size", C.c_int), ("cyclic_group_log2", C.c_int)]


class ChipProofC(C.Structure):
    _fields_ = [("num_instances", C.c_size_t), ("n_r_out", C.c_int), ("n_w_out", C.c_int), ("n_lk_out", C.c_int),
                ("r_out_evals", C.c_uint64 * 4), ("w_out_evals", C.c_uint64 * 4), ("lk_out_evals", C.c_uint64 * 8),
                ("tower_num_vars", C.c_int), ("n_prod", C.c_int), ("n_logup", C.c_int), ("tower", TowerProofC),
                ("num_var_with_rotation", C.c_int), ("rt_main", u64p), ("n_rotation_pairs", C.c_int), ("rotation_msgs", u64p),
                ("rotation_evals", u64p), ("rotation_points", u64p)]


class _ProofBlock:
    """the C proofs of one phase: released together when the last Python proof that reads them is gone (a Rust caller owns the C buffers the same
    way; copying 54 proofs out eagerly cost the harness ~0.9 ms per shard)"""

    def __init__(self, outs, n):
        self.outs, self.n = outs, n

    def __del__(self):
        try:
            L = plib()
            for i in range(self.n):
                L.ceno_chip_proof_free(C.byref(self.outs[i]))
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass


class ChipProof:
    """ZKVMChipProof (ceno_zkvm/src/scheme.rs:59-76): copied out of the C structure at once, or — given the block that owns the C buffers — array by
    array when first read"""

    _LAZY = ("tower_msgs", "tower_prod_evals", "tower_logup_evals", "tower_point", "rt_main", "rotation_msgs", "rotation_evals", "rotation_points")

    def __init__(self, c: ChipProofC, owner: Optional[_ProofBlock] = None):
        nv = c.tower_num_vars
        self._c, self._owner = c, owner
        self.num_instances = c.num_instances
        self.r_out_evals = np.array(c.r_out_evals, dtype=np.uint64).reshape(2, 2)[: c.n_r_out]
        self.w_out_evals = np.array(c.w_out_evals, dtype=np.uint64).reshape(2, 2)[: c.n_w_out]
        self.lk_out_evals = np.array(c.lk_out_evals, dtype=np.uint64).reshape(4, 2)[: c.n_lk_out]
        self.tower_num_vars, self.n_prod, self.n_logup = nv, c.n_prod, c.n_logup
        self.n_rotation_pairs = c.n_rotation_pairs
        if owner is None:  # (the C buffers go away with the caller: everything now)
            for k in self._LAZY:
                if k.startswith("rotation") and not c.n_rotation_pairs:
                    continue
                getattr(self, k)
            self._c = None

    def __getattr__(self, name):  # (only reached for attributes not set yet: the lazy arrays)
        if name not in ChipProof._LAZY or self.__dict__.get("_c") is None:
            raise AttributeError(name)
        c = self._c

        def arr(ptr, n):   # (np.ctypeslib.as_array builds an array interface per call: ~30 us; a buffer view of the same words: ~2 us)
            if not n:
                return np.zeros(0, dtype=np.uint64)
            return np.frombuffer((C.c_uint64 * n).from_address(C.addressof(ptr.contents)), dtype=np.uint64).copy()

        nv, R, n = c.tower_num_vars, c.tower_num_vars - 1, c.num_var_with_rotation
        if name == "tower_msgs":
            v = arr(c.tower.msgs, int(plib().ceno_tower_msgs_words(nv)))
        elif name == "tower_prod_evals":
            v = arr(c.tower.prod_evals, c.n_prod * R * 4).reshape(c.n_prod, R, 2, 2)
        elif name == "tower_logup_evals":
            v = arr(c.tower.logup_evals, c.n_logup * R * 8).reshape(c.n_logup, R, 4, 2)
        elif name == "tower_point":
            v = arr(c.tower.point, 2 * nv).reshape(nv, 2)
        elif name == "rt_main":
            v = arr(c.rt_main, 2 * n).reshape(n, 2)
        elif not c.n_rotation_pairs:
            raise AttributeError(name)
        elif name == "rotation_msgs":
            v = arr(c.rotation_msgs, n * 4).reshape(n, 2, 2)
        elif name == "rotation_evals":
            v = arr(c.rotation_evals, 6 * c.n_rotation_pairs).reshape(-1, 2)
        else:
            v = arr(c.rotation_points, 6 * n).reshape(3, n, 2)
        self.__dict__[name] = v
        return v

    def tower_round_msgs(self, rnd: int) -> np.ndarray:
        off = sum(r * 6 for r in range(1, rnd))
        return self.tower_msgs[off: off + rnd * 6].reshape(rnd, 3, 2)


def _marshal_chip_task(task: dict):
    """dict -> (ChipTaskC, objects that must stay alive while it is used)"""
    T = ChipTaskC()
    mh = (C.c_void_p * len(task["mles"]))(*[(m.h if m is not None else None) for m in task["mles"]])
    coeffs = np.ascontiguousarray(task["record_coeffs"], dtype=np.uint64).reshape(-1, 2)
    toff, tidx = _csr(task["record_terms"])
    ooff = np.zeros(len(task["record_out_terms"]) + 1, dtype=np.uint32)
    for o, ts in enumerate(task["record_out_terms"]):
        assert list(ts) == list(range(int(ooff[o]), int(ooff[o]) + len(ts))), "record terms must be listed output by output"
        ooff[o + 1] = ooff[o] + len(ts)
    T.circuit_idx, T.num_instances = task.get("circuit_idx", 0), task["num_instances"]
    T.log2_num_instances, T.rotation_vars = task["log2_num_instances"], task.get("rotation_vars", 0)
    T.n_witin, T.n_fixed, T.n_structural, T.mles = task["n_witin"], task["n_fixed"], task["n_structural"], mh
    T.num_reads, T.num_writes, T.num_lk_tables, T.num_lk = task["num_reads"], task["num_writes"], task["num_lk_tables"], task["num_lk"]
    T.n_record_terms, T.record_coeffs, T.record_term_offsets, T.record_term_mle_idx = len(task["record_terms"]), _p(coeffs), _p32(toff), _p32(tidx)
    T.record_out_term_offsets = _p32(ooff)
    keep = [mh, coeffs, toff, tidx, ooff]
    rot = task.get("rotation")
    if rot:
        src = (C.c_int * len(rot["pairs"]))(*[p[0] for p in rot["pairs"]])
        tgt = (C.c_int * len(rot["pairs"]))(*[p[1] for p in rot["pairs"]])
        T.n_rotation_pairs, T.rotation_source_idx, T.rotation_target_idx = len(rot["pairs"]), src, tgt
        T.cyclic_subgroup_size, T.cyclic_group_log2 = rot["cyclic_subgroup_size"], rot["cyclic_group_log2"]
        keep += [src, tgt]
    return T, keep


def create_chip_proof(dev: Device, task: dict, challenges, tr: Transcript, stream=None) -> ChipProof:
    """ZKVMProver::create_chip_proof (scheme/prover.rs:717-833).  task: mles (witness ++ fixed ++ structural), n_witin, n_fixed,
    n_structural, num_instances, log2_num_instances, rotation_vars (0), num_reads, num_writes, num_lk_tables, num_lk,
    record_coeffs (T,2), record_terms [[mle ids]], record_out_terms [[term ids]] (consecutive), optional rotation
    dict(pairs, cyclic_subgroup_size, cyclic_group_log2)."""
    L = plib()
    L.ceno_prover_create_chip_proof.restype = C.c_int
    L.ceno_prover_create_chip_proof.argtypes = [C.c_void_p, C.POINTER(ChipTaskC), u64p, C.c_void_p, C.c_void_p, C.POINTER(ChipProofC)]
    L.ceno_chip_proof_free.restype = None
    L.ceno_chip_proof_free.argtypes = [C.POINTER(ChipProofC)]
    if isinstance(task, ChipTasks):  # marshalled once by the caller (a Rust caller hands the struct over directly): the first task of the list
        T, keep = task.arr[0], None
    else:
        T, keep = _marshal_chip_task(task)
    ch = np.array([[int(c[0]), int(c[1])] for c in challenges], dtype=np.uint64)
    out = ChipProofC()
    _check(L.ceno_prover_create_chip_proof(dev.h, C.byref(T), _p(ch), tr.h, stream, C.byref(out)))
    try:
        return ChipProof(out)
    finally:
        L.ceno_chip_proof_free(C.byref(out))


class LocalGroup:
    """in-process group of `world` virtual ranks (threads of one process sharing a device): ceno_dist_local_group"""

    def __init__(self, world: int):
        L = plib()
        L.ceno_dist_local_group_create.restype = C.c_void_p
        L.ceno_dist_local_group_create.argtypes = [C.c_int]
        L.ceno_dist_local_group_destroy.restype = None
        L.ceno_dist_local_group_destroy.argtypes = [C.c_void_p]
        L.ceno_dist_comm_init_local.restype = C.c_int
        L.ceno_dist_comm_init_local.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ceno_dist_comm_destroy.restype = None
        L.ceno_dist_comm_destroy.argtypes = [C.c_void_p]
        self.world = world
        self.h = L.ceno_dist_local_group_create(world)
        if not self.h:
            raise CenoHipError(-1, "local group: world must be a power of two")
        self.comms = []
        for r in range(world):
            c = C.c_void_p()
            _check(L.ceno_dist_comm_init_local(self.h, r, C.byref(c)))
            self.comms.append(c)

    def close(self):
        L = plib()
        for c in self.comms:
            L.ceno_dist_comm_destroy(c)
        self.comms = []
        if self.h:
            L.ceno_dist_local_group_destroy(self.h)
            self.h = None


def shard_rows(full: np.ndarray, world: int, rank: int, q: int) -> np.ndarray:
    """the rows of a column that rank `rank` holds under ceno_dist_create_chip_proof's layout: index bits [q, q + log2 world) == rank, in order"""
    k = world.bit_length() - 1
    idx = np.arange(full.shape[0])
    return np.ascontiguousarray(full[((idx >> q) & (world - 1)) == rank]) if k else full


def dist_create_chip_proof(dev: Device, comm, task_local: dict, log2_num_instances_global: int, q: int, challenges, tr: Transcript, stream=None) -> ChipProof:
    """ceno_dist_create_chip_proof: the chip proof over row-sharded columns (task_local: this rank's tables, log2_num_instances = the local height).
    `comm`: a communicator handle (LocalGroup.comms[rank], ShmComm(...).h, RcclComm(...).h)"""
    L = plib()
    L.ceno_dist_create_chip_proof.restype = C.c_int
    L.ceno_dist_create_chip_proof.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ChipTaskC), C.c_int, C.c_int, u64p, C.c_void_p, C.c_void_p, C.POINTER(ChipProofC)]
    L.ceno_chip_proof_free.restype = None
    L.ceno_chip_proof_free.argtypes = [C.POINTER(ChipProofC)]
    T, keep = _marshal_chip_task(task_local)
    ch = np.array([[int(c[0]), int(c[1])] for c in challenges], dtype=np.uint64)
    out = ChipProofC()
    _check(L.ceno_dist_create_chip_proof(dev.h, comm, C.byref(T), log2_num_instances_global, q, _p(ch), tr.h, stream, C.byref(out)))
    try:
        return ChipProof(out)
    finally:
        L.ceno_chip_proof_free(C.byref(out))


def dist_basefold_open(dev: Device, comm, log_rows, widths, log_blowup: int, trace_ptrs, cw_row_ptrs, subtree, top, points, evals,
                       n_queries: int, pow_bits: int, tr: Transcript, stream) -> np.ndarray:
    """ceno_dist_basefold_open[_mmcs]: the opening of a commitment made across ranks by ceno_dist_commit_traces_mmcs (log_rows: one height for
    all matrices, or a list with one height per matrix).
    widths[m][g]; trace_ptrs[m] = this rank's columns of matrix m (device pointer), cw_row_ptrs[m] = its codeword rows of ALL columns (the
    commit's output); subtree / top = the commit's trees.  Returns the flat proof = ceno_prover_basefold_open's of the single-device commitment."""
    L = plib()
    n, world = len(widths), len(widths[0])
    L.ceno_dist_basefold_open.restype = C.c_int
    L.ceno_dist_basefold_open.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.c_void_p, C.c_void_p, C.POINTER(u64p), C.POINTER(u64p), C.c_int, C.c_int, C.c_void_p, C.c_void_p, u64p]
    L.ceno_dist_basefold_open_mmcs.restype = C.c_int
    L.ceno_dist_basefold_open_mmcs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)] + L.ceno_dist_basefold_open.argtypes[4:]
    L.ceno_prover_basefold_proof_words_meta.restype = C.c_size_t
    L.ceno_prover_basefold_proof_words_meta.argtypes = [C.c_int] * 5
    total_w = int(sum(sum(row) for row in widths))
    mixed = not isinstance(log_rows, (int, np.integer))
    max_log = max(int(x) for x in log_rows) if mixed else int(log_rows)
    proof = np.zeros(int(L.ceno_prover_basefold_proof_words_meta(n, total_w, max_log, log_blowup, n_queries)), dtype=np.uint64)
    wa = (C.c_int * (n * world))(*[int(w) for row in widths for w in row])
    tp = (C.c_void_p * n)(*[C.c_void_p(int(x)) for x in trace_ptrs])
    cp = (C.c_void_p * n)(*[C.c_void_p(int(x)) for x in cw_row_ptrs])
    pts = [np.ascontiguousarray(x, dtype=np.uint64) for x in points]
    evs = [np.ascontiguousarray(x, dtype=np.uint64) for x in evals]
    pp = (u64p * n)(*[_p(x) for x in pts])
    ep = (u64p * n)(*[_p(x) for x in evs])
    if mixed:
        lr = (C.c_int * n)(*[int(x) for x in log_rows])
        rc = L.ceno_dist_basefold_open_mmcs(dev.h, comm, n, lr, wa, log_blowup, tp, cp, subtree, top, pp, ep, n_queries, pow_bits, tr.h, stream, _p(proof))
    else:
        rc = L.ceno_dist_basefold_open(dev.h, comm, n, log_rows, wa, log_blowup, tp, cp, subtree, top, pp, ep, n_queries, pow_bits, tr.h, stream, _p(proof))
    _check(rc)
    return proof


class DistCommitViewC(C.Structure):
    _fields_ = [("n_mats", C.c_int), ("log_rows", C.POINTER(C.c_int)), ("widths", C.POINTER(C.c_int)), ("local_trace_cols", C.POINTER(C.c_void_p)),
                ("local_cw_rows", C.POINTER(C.c_void_p)), ("subtree", C.c_void_p), ("top", C.c_void_p)]


def dist_basefold_open_commits(dev: Device, comm, commits, log_blowup: int, points, evals, n_queries: int, pow_bits: int, tr: Transcript, stream) -> np.ndarray:
    """ceno_dist_basefold_open_commits: ONE opening of several commitments made across ranks (witness + fixed).  commits: dicts with log_rows
    [per matrix], widths [m][g], trace_ptrs [m], cw_row_ptrs [m], subtree, top; points / evals over the matrices of all commitments in order."""
    L = plib()
    world = len(commits[0]["widths"][0])
    nc = len(commits)
    views = (DistCommitViewC * nc)()
    keep = []
    for c, cm in enumerate(commits):
        n = len(cm["log_rows"])
        lr = (C.c_int * n)(*[int(x) for x in cm["log_rows"]])
        wa = (C.c_int * (n * world))(*[int(w) for row in cm["widths"] for w in row])
        tp = (C.c_void_p * n)(*[C.c_void_p(int(x)) for x in cm["trace_ptrs"]])
        cp = (C.c_void_p * n)(*[C.c_void_p(int(x)) for x in cm["cw_row_ptrs"]])
        keep += [lr, wa, tp, cp]
        views[c] = DistCommitViewC(n, lr, wa, tp, cp, cm["subtree"], cm["top"])
    L.ceno_prover_basefold_proof_words_commits.restype = C.c_size_t
    L.ceno_prover_basefold_proof_words_commits.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int]
    nm = (C.c_int * nc)(*[len(cm["log_rows"]) for cm in commits])
    tw = (C.c_int * nc)(*[int(sum(sum(row) for row in cm["widths"])) for cm in commits])
    ml = (C.c_int * nc)(*[max(int(x) for x in cm["log_rows"]) for cm in commits])
    proof = np.zeros(int(L.ceno_prover_basefold_proof_words_commits(nc, nm, tw, ml, log_blowup, n_queries)), dtype=np.uint64)
    n_all = sum(len(cm["log_rows"]) for cm in commits)
    pts = [np.ascontiguousarray(x, dtype=np.uint64) for x in points]
    evs = [np.ascontiguousarray(x, dtype=np.uint64) for x in evals]
    assert len(pts) == n_all and len(evs) == n_all
    pp = (u64p * n_all)(*[_p(x) for x in pts])
    ep = (u64p * n_all)(*[_p(x) for x in evs])
    L.ceno_dist_basefold_open_commits.restype = C.c_int
    L.ceno_dist_basefold_open_commits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(DistCommitViewC), C.c_int, C.POINTER(u64p), C.POINTER(u64p), C.c_int,
                                                  C.c_int, C.c_void_p, C.c_void_p, u64p]
    _check(L.ceno_dist_basefold_open_commits(dev.h, comm, nc, views, log_blowup, pp, ep, n_queries, pow_bits, tr.h, stream, _p(proof)))
    return proof


def chip_proof_estimate_bytes(task: dict) -> int:
    """ceno_prover_chip_proof_estimate_bytes: what the lane scheduler books for this chip proof"""
    L = plib()
    L.ceno_prover_chip_proof_estimate_bytes.restype = C.c_size_t
    L.ceno_prover_chip_proof_estimate_bytes.argtypes = [C.POINTER(ChipTaskC)]
    T, keep = _marshal_chip_task(task)
    return int(L.ceno_prover_chip_proof_estimate_bytes(C.byref(T)))


class ChipTasks:
    """the C view of a list of chip tasks, marshalled once (a Rust caller hands the structs over directly)"""

    def __init__(self, tasks: Sequence[dict]):
        self.n = len(tasks)
        self.arr = (ChipTaskC * self.n)()
        self.keep = []
        for i, t in enumerate(tasks):
            T, keep = _marshal_chip_task(t)
            self.arr[i] = T
            self.keep.append(keep)


def create_chip_proofs(dev: Device, tasks, challenges, transcripts: Sequence[Transcript], lanes: int, statuses: Optional[list] = None) -> List[ChipProof]:
    """the chip-proof phase of create_proof on the C++ scheduler (ceno_prover_create_chip_proofs; prover.rs:556-570,
    scheduler.rs:231-336): one forked transcript per task, `lanes` concurrent lanes on the context's own lane streams, results in
    task order.  `statuses` (a list): receives every task's return code; a failed task then yields None instead of raising"""
    L = plib()
    L.ceno_prover_create_chip_proofs.restype = C.c_int
    L.ceno_prover_create_chip_proofs.argtypes = [C.c_void_p, C.POINTER(ChipTaskC), C.c_int, u64p, C.POINTER(C.c_void_p), C.c_int,
                                                 C.POINTER(ChipProofC), C.POINTER(C.c_int)]
    L.ceno_chip_proof_free.restype = None
    L.ceno_chip_proof_free.argtypes = [C.POINTER(ChipProofC)]
    ct = tasks if isinstance(tasks, ChipTasks) else ChipTasks(tasks)
    assert len(transcripts) == ct.n
    ch = np.array([[int(c[0]), int(c[1])] for c in challenges], dtype=np.uint64)
    trs = (C.c_void_p * ct.n)(*[t.h for t in transcripts])
    outs = (ChipProofC * ct.n)()
    status = (C.c_int * ct.n)()
    t0 = time.perf_counter()
    rc = L.ceno_prover_create_chip_proofs(dev.h, ct.arr, ct.n, _p(ch), trs, lanes, outs, status)
    create_chip_proofs.last_native_ms = (time.perf_counter() - t0) * 1e3   # (the C call alone: what remains is this wrapper's marshalling)
    try:
        if statuses is not None:
            statuses[:] = [int(status[i]) for i in range(ct.n)]
            if rc != 0 and all(x == 0 for x in statuses):
                _check(rc)   # (a failure that is nobody's in particular)
            return [ChipProof(outs[i]) if status[i] == 0 else None for i in range(ct.n)]
        _check(rc)
        return [ChipProof(outs[i]) for i in range(ct.n)]
    finally:
        for i in range(ct.n):
            L.ceno_chip_proof_free(C.byref(outs[i]))


def run_chip_proofs(dev: Device, tasks, challenges, fork_parent: Transcript, bind_words: Sequence[Sequence[int]], lanes: int):
    """ZKVMProver::run_chip_proofs (ceno_prover_run_chip_proofs; prover.rs:618-710): every task's transcript is forked from `fork_parent` inside the
    library and bound to the challenges and to bind_words[i] (task id, circuit index, instance counts); returns (proofs, one sample per fork)"""
    L = plib()
    L.ceno_prover_run_chip_proofs.restype = C.c_int
    L.ceno_prover_run_chip_proofs.argtypes = [C.c_void_p, C.POINTER(ChipTaskC), C.c_int, u64p, C.c_void_p, u64p, u32p, C.c_int, C.POINTER(ChipProofC), u64p,
                                              C.POINTER(C.c_int)]
    L.ceno_chip_proof_free.restype = None
    L.ceno_chip_proof_free.argtypes = [C.POINTER(ChipProofC)]
    ct = tasks if isinstance(tasks, ChipTasks) else ChipTasks(tasks)
    assert len(bind_words) == ct.n
    ch = np.array([[int(c[0]), int(c[1])] for c in challenges], dtype=np.uint64)
    offs = np.zeros(ct.n + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(w) for w in bind_words])
    words = np.array([int(v) for w in bind_words for v in w], dtype=np.uint64) if offs[-1] else np.zeros(1, dtype=np.uint64)
    outs = (ChipProofC * ct.n)()
    status = (C.c_int * ct.n)()
    samples = np.zeros((ct.n, 2), dtype=np.uint64)
    t0 = time.perf_counter()
    rc = L.ceno_prover_run_chip_proofs(dev.h, ct.arr, ct.n, _p(ch), fork_parent.h, _p(words), _p32(offs), lanes, outs, _p(samples), status)
    create_chip_proofs.last_native_ms = (time.perf_counter() - t0) * 1e3
    block = _ProofBlock(outs, ct.n)   # (frees the C proofs — also when the check below raises)
    _check(rc)
    return [ChipProof(outs[i], block) for i in range(ct.n)], [(int(a), int(b)) for a, b in samples]


class PcsData:
    """committed traces (reference: PCS::CommitmentWithWitness returned by commit_traces)"""

    def __init__(self, dev: Device, matrices: Sequence[np.ndarray], log_blowup: int, stream, device_ptrs=None):
        """matrices: host row-major (rows, width) arrays; or device_ptrs = [(device pointer, rows, width)] for device-resident
        row-major matrices (ceno_prover_commit_traces_dev)"""
        L = plib()
        vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
        self._declare(L)
        self.dev, self.stream, self.log_blowup = dev, stream, log_blowup
        if device_ptrs is not None:
            n = len(device_ptrs)
            self.shapes = [(int(r), int(w)) for _, r, w in device_ptrs]
            ptrs = (u64p * n)(*[C.cast(C.c_void_p(int(p_)), u64p) for p_, _, _ in device_ptrs])
            rows = (sz * n)(*[r for _, r, _ in device_ptrs])
            widths = (sz * n)(*[w for _, _, w in device_ptrs])
            h = vp()
            _check(L.ceno_prover_commit_traces_dev(dev.h, ptrs, rows, widths, n, log_blowup, stream, C.byref(h)))
            self.h = h
            return
        mats = [np.ascontiguousarray(m, dtype=np.uint64) for m in matrices]
        self.shapes = [m.shape for m in mats]
        ptrs = (u64p * len(mats))(*[_p(m) for m in mats])
        rows = (sz * len(mats))(*[m.shape[0] for m in mats])
        widths = (sz * len(mats))(*[m.shape[1] for m in mats])
        h = vp()
        _check(L.ceno_prover_commit_traces(dev.h, ptrs, rows, widths, len(mats), log_blowup, stream, C.byref(h)))
        self.h = h

    @staticmethod
    def _declare(L):
        vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
        L.ceno_prover_commit_traces.restype = i
        L.ceno_prover_commit_traces.argtypes = [vp, C.POINTER(u64p), C.POINTER(sz), C.POINTER(sz), i, i, vp, C.POINTER(vp)]
        L.ceno_pcs_data_num_vars.restype = i
        L.ceno_pcs_data_num_vars.argtypes = [vp, i]
        L.ceno_pcs_data_root.restype = i
        L.ceno_pcs_data_root.argtypes = [vp, vp, u64p, vp]
        L.ceno_pcs_data_witness_mle.restype = i
        L.ceno_pcs_data_witness_mle.argtypes = [vp, vp, i, sz, C.POINTER(vp)]
        L.ceno_pcs_data_opening_words.restype = sz
        L.ceno_pcs_data_opening_words.argtypes = [vp]
        L.ceno_pcs_data_open.restype = i
        L.ceno_pcs_data_open.argtypes = [vp, vp, sz, u64p, vp]
        L.ceno_pcs_data_free.restype = None
        L.ceno_pcs_data_free.argtypes = [vp, vp]
        L.ceno_prover_basefold_proof_words.restype = sz
        L.ceno_prover_basefold_proof_words.argtypes = [C.POINTER(vp), i, i]
        L.ceno_prover_basefold_open.restype = i
        L.ceno_prover_basefold_open.argtypes = [vp, C.POINTER(vp), i, C.POINTER(u64p), C.POINTER(u64p), i, i, vp, vp, u64p]
        L.ceno_prover_commit_traces_dev.restype = i
        L.ceno_prover_commit_traces_dev.argtypes = [vp, C.POINTER(u64p), C.POINTER(sz), C.POINTER(sz), i, i, vp, C.POINTER(vp)]

    @classmethod
    def reserve(cls, dev: Device, shapes: Sequence[Tuple[int, int]], log_blowup: int, stream) -> "PcsData":
        """ceno_prover_commit_reserve: the commitment's storage for matrices of (num_instances, width) that are PRODUCED on the device;
        write matrix m COLUMN-major at trace_ptr(m) (rows(m) words per column), then finish()"""
        L = plib()
        vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
        L.ceno_prover_commit_reserve.restype = i
        L.ceno_prover_commit_reserve.argtypes = [vp, C.POINTER(sz), C.POINTER(sz), i, i, vp, C.POINTER(vp)]
        L.ceno_prover_commit_finish.restype = i
        L.ceno_prover_commit_finish.argtypes = [vp, vp, vp]
        L.ceno_pcs_data_trace_ptr.restype = vp
        L.ceno_pcs_data_trace_ptr.argtypes = [vp, i]
        L.ceno_pcs_data_rows.restype = sz
        L.ceno_pcs_data_rows.argtypes = [vp, i]
        self = cls.__new__(cls)
        cls._declare(L)
        self.dev, self.stream, self.log_blowup = dev, stream, log_blowup
        n = len(shapes)
        rows = (sz * n)(*[int(r) for r, _ in shapes])
        widths = (sz * n)(*[int(w) for _, w in shapes])
        h = vp()
        _check(L.ceno_prover_commit_reserve(dev.h, rows, widths, n, log_blowup, stream, C.byref(h)))
        self.h = h
        self.shapes = [(int(L.ceno_pcs_data_rows(h, m)), int(w)) for m, (_, w) in enumerate(shapes)]
        return self

    def trace_ptr(self, matrix: int) -> int:
        return int(plib().ceno_pcs_data_trace_ptr(self.h, matrix) or 0)

    def rows(self, matrix: int) -> int:
        return int(plib().ceno_pcs_data_rows(self.h, matrix))

    def finish(self):
        """ceno_prover_commit_finish: encode + hash what the producers wrote (queued on the commitment's stream behind them)"""
        _check(plib().ceno_prover_commit_finish(self.dev.h, self.h, self.stream))

    def num_vars(self, matrix: int) -> int:
        return plib().ceno_pcs_data_num_vars(self.h, matrix)

    def root(self) -> np.ndarray:
        """PCS::get_pure_commitment: ONE root for all the matrices of the commitment"""
        out = np.zeros(4, dtype=np.uint64)
        _check(plib().ceno_pcs_data_root(self.dev.h, self.h, _p(out), self.stream))
        return out

    def witness_mle(self, matrix: int, col: int) -> Mle:
        h = C.c_void_p()
        _check(plib().ceno_pcs_data_witness_mle(self.dev.h, self.h, matrix, col, C.byref(h)))
        m = Mle(self.dev, h)
        m._parent = self
        return m

    def open(self, index: int):
        """MerkleTreeMmcs::open_batch at row `index` of the tallest codeword: (rows of every matrix, path (H, 4))"""
        words = int(plib().ceno_pcs_data_opening_words(self.h))
        out = np.zeros(words, dtype=np.uint64)
        _check(plib().ceno_pcs_data_open(self.dev.h, self.h, index, _p(out), self.stream))
        wsum = sum(int(w) for _, w in self.shapes)
        rows, off = [], 0
        for _, w in self.shapes:
            rows.append(out[off: off + w].copy())
            off += w
        return rows, out[wsum:].reshape(-1, 4).copy()

    def open_row(self, matrix: int, index: int):
        """row `index` of matrix `matrix`'s codeword + the commitment's path of the corresponding tallest-height row"""
        nmax = max(self.num_vars(m) for m in range(len(self.shapes)))
        rows, path = self.open(index << (nmax - self.num_vars(matrix)))
        return rows[matrix], path

    def basefold_open(self, points: Sequence[np.ndarray], evals: Sequence[np.ndarray], n_queries: int, pow_bits: int,
                      transcript: "Transcript", more_commits: Sequence["PcsData"] = ()) -> np.ndarray:
        """OpeningProver::open (ceno_zkvm/src/scheme/hal.rs:284-294): every matrix at its own point; `more_commits` are further
        commitments opened in the same proof (the fixed commitment), their points / evals follow this one's.
        Returns the flat proof (layout: include/ceno_prover.h)."""
        L = plib()
        commits = [self] + list(more_commits)
        n = sum(len(c.shapes) for c in commits)
        pts = [np.ascontiguousarray(p, dtype=np.uint64) for p in points]
        evs = [np.ascontiguousarray(e, dtype=np.uint64) for e in evals]
        assert len(pts) == n and len(evs) == n
        ch = (C.c_void_p * len(commits))(*[c.h for c in commits])
        proof = np.zeros(int(L.ceno_prover_basefold_proof_words(ch, len(commits), n_queries)), dtype=np.uint64)
        pp = (u64p * n)(*[_p(x) for x in pts])
        ep = (u64p * n)(*[_p(x) for x in evs])
        t0 = time.perf_counter()
        rc = L.ceno_prover_basefold_open(self.dev.h, ch, len(commits), pp, ep, n_queries, pow_bits, transcript.h, self.stream, _p(proof))
        PcsData.last_open_native_ms = (time.perf_counter() - t0) * 1e3
        _check(rc)
        return proof

    def free(self):
        if getattr(self, "h", None) and self.dev.h:
            plib().ceno_pcs_data_free(self.dev.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
