"""Thin Python handles over the C ABI (include/ceno_hip.h).

These classes are plumbing for tests, bench.py and the torch.distributed driver: every compute call
goes through libceno_hip.so.  Field elements are numpy uint64; an extension element is a pair
[c0, c1]; an ext table has shape (2^nv, 2), a base table shape (2^nv,).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import SumcheckPlan, u32p, u64p

P = 0xFFFFFFFF00000001


class CenoHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ceno_hip error {code}: {msg}")
        self.code = code


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _p32(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def _ext1(e) -> np.ndarray:
    a = np.zeros(2, dtype=np.uint64)
    a[0], a[1] = int(e[0]), int(e[1])
    return a


_COLMAP_TYPES: dict = {}
_COLMAPS: dict = {}


def _colmap(n_cols: int, cols, cls=None):
    """the column map of a witness kernel ({cols[n_cols], num_cols}: the reference's `extract_*_column_map`, made once per circuit) as a ctypes
    structure — cached: building a Structure class and filling it cost ~40 us per call, 45 calls per shard"""
    key = (n_cols, cls, tuple(int(c) for c in cols[: n_cols + 1]))
    m = _COLMAPS.get(key)
    if m is None:
        if cls is None:
            cls = _COLMAP_TYPES.get(n_cols)
            if cls is None:
                cls = _COLMAP_TYPES[n_cols] = type(f"ColumnMap{n_cols}", (C.Structure,), {"_fields_": [("cols", C.c_uint32 * n_cols), ("num_cols", C.c_uint32)]})
        m = cls()
        for k in range(n_cols):
            m.cols[k] = key[2][k]
        m.num_cols = key[2][n_cols]
        if len(_COLMAPS) < 4096:
            _COLMAPS[key] = m
    return m


class Device:
    """ceno_hip_ctx: one per process/GPU (reference: process-global CUDA_HAL, gkr_iop/src/gpu/mod.rs:53-66)."""

    def __init__(self, device: int = 0, pool_bytes: int = 0):
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = self.L.ceno_hip_init(device, pool_bytes, C.byref(h))
        if rc != 0:
            raise CenoHipError(rc, (self.L.ceno_hip_last_error(None) or b"").decode())
        self.h = h
        self.device = device

    def check(self, rc: int):
        if rc != 0:
            raise CenoHipError(rc, (self.L.ceno_hip_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.ceno_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- streams / memory ----
    def stream_create(self) -> C.c_void_p:
        s = C.c_void_p()
        self.check(self.L.ceno_hip_stream_create(self.h, C.byref(s)))
        return s

    def stream_create_lane(self, lane: int) -> C.c_void_p:
        s = C.c_void_p()
        self.check(self.L.ceno_hip_stream_create_lane(self.h, lane, C.byref(s)))
        return s

    def stream_destroy(self, s):
        self.check(self.L.ceno_hip_stream_destroy(self.h, s))

    def sync(self, stream=None):
        self.check(self.L.ceno_hip_stream_sync(self.h, stream))

    def mem_info(self):
        f, t, u, c = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        self.check(self.L.ceno_hip_mem_info(self.h, C.byref(f), C.byref(t), C.byref(u), C.byref(c)))
        return {"free": f.value, "total": t.value, "pool_used": u.value, "pool_cached": c.value}

    def mem_trim(self):
        self.check(self.L.ceno_hip_mem_trim(self.h))

    # ---- MLEs ----
    def alloc(self, num_vars: int, is_ext: bool) -> "Mle":
        h = C.c_void_p()
        self.check(self.L.ceno_hip_mle_alloc(self.h, num_vars, int(is_ext), C.byref(h)))
        return Mle(self, h)

    def upload(self, table: np.ndarray, stream=None) -> "Mle":
        table = np.ascontiguousarray(table, dtype=np.uint64)
        is_ext = table.ndim == 2
        n = table.shape[0]
        assert n & (n - 1) == 0 and n >= 1
        h = C.c_void_p()
        self.check(self.L.ceno_hip_mle_upload(self.h, _p(table), n.bit_length() - 1, int(is_ext), stream, C.byref(h)))
        return Mle(self, h)

    def wrap(self, device_ptr: int, num_vars: int, is_ext: bool) -> "Mle":
        h = C.c_void_p()
        self.check(self.L.ceno_hip_mle_wrap(self.h, C.c_void_p(device_ptr), num_vars, int(is_ext), C.byref(h)))
        return Mle(self, h)

    def evaluate_prefix_batch(self, cols: Sequence["Mle"], point: np.ndarray, stream=None) -> np.ndarray:
        """cols[j](point[:num_vars_j]) for base-field tables, one pass (ceno_hip_mle_evaluate_prefix_batch); returns (n, 2) words"""
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 2)
        arr = (C.c_void_p * max(len(cols), 1))(*[m.h for m in cols])
        out = np.zeros((len(cols), 2), dtype=np.uint64)
        self.check(self.L.ceno_hip_mle_evaluate_prefix_batch(self.h, len(cols), arr, _p(point), point.shape[0], stream, _p(out)))
        return out

    def lincomb_base_batch(self, groups: Sequence[Sequence["Mle"]], coeffs: Sequence[np.ndarray], stream=None):
        """per group g: (sum_j c_j.c0 col_j, sum_j c_j.c1 col_j) as two base-field tables (ceno_hip_lincomb_base_batch)"""
        offs, flat = [0], []
        for g in groups:
            flat += list(g)
            offs.append(len(flat))
        co = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.uint64).reshape(-1, 2) for c in coeffs]), dtype=np.uint64)
        offs = np.asarray(offs, dtype=np.uint32)
        arr = (C.c_void_p * max(len(flat), 1))(*[m.h for m in flat])
        o0 = (C.c_void_p * len(groups))()
        o1 = (C.c_void_p * len(groups))()
        self.check(self.L.ceno_hip_lincomb_base_batch(self.h, len(groups), offs.ctypes.data_as(C.POINTER(C.c_uint32)), arr, _p(co), stream, o0, o1))
        return [(Mle(self, C.c_void_p(o0[g])), Mle(self, C.c_void_p(o1[g]))) for g in range(len(groups))]

    def synthetic(self, num_vars: int, is_ext: bool, seed: int, word_offset: int = 0, stream=None) -> "Mle":
        self.check(self.L.ceno_hip_stream_bind(self.h, stream))  # the block is for work on `stream` (pool tags)
        m = self.alloc(num_vars, is_ext)
        self.check(self.L.ceno_hip_mle_fill_splitmix(self.h, m.h, C.c_uint64(seed), C.c_uint64(word_offset), stream))
        return m

    def zeros(self, num_vars: int, is_ext: bool = False, stream=None) -> "Mle":
        """a table of zeros (ceno_hip_mle_fill_zero); its words double as raw device storage for lookup counters (two u32 counters per word)"""
        self.check(self.L.ceno_hip_stream_bind(self.h, stream))
        m = self.alloc(num_vars, is_ext)
        self.check(self.L.ceno_hip_mle_fill_zero(self.h, m.h, stream))
        return m

    # ---- a shard's witness generation as one session (ceno_hip_witgen_session_begin / _end) ----
    def witgen_session_begin(self, tables: Sequence[Tuple[int, int]], stream=None):
        """tables: [(device pointer of the u32 counters, slots)] — every lookup table the shard's chips count into"""
        n = len(tables)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(int(p_)) for p_, _ in tables])
        slots = (C.c_size_t * n)(*[int(s_) for _, s_ in tables])
        self.check(self.L.ceno_hip_witgen_session_begin(self.h, ptrs, slots, n, stream))

    def witgen_session_end(self, stream=None):
        self.check(self.L.ceno_hip_witgen_session_end(self.h, stream))

    def lk_to_mlt_column(self, counters_ptr: int, n: int, column_ptr: int, rows_padded: int, stream=None):
        """a table circuit's `mlt` witness column from the device counters (ceno_hip_lk_to_mlt_column)"""
        self.check(self.L.ceno_hip_lk_to_mlt_column(self.h, C.c_void_p(int(counters_ptr)), n, C.c_void_p(int(column_ptr)), rows_padded, stream))

    def eq_build(self, point: np.ndarray, scalar=None, stream=None) -> "Mle":
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 2)
        h = C.c_void_p()
        sc = _p(_ext1(scalar)) if scalar is not None else None
        pp = _p(point) if point.shape[0] else None
        self.check(self.L.ceno_hip_eq_build(self.h, pp, point.shape[0], sc, stream, C.byref(h)))
        return Mle(self, h)

    def selector_build(self, kind: int, point: np.ndarray, offset=0, num_instances=0, sparse_indices=(),
                       sparse_num_vars=0, stream=None) -> "Mle":
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 2)
        si = np.array(list(sparse_indices) or [0], dtype=np.uint32)
        h = C.c_void_p()
        self.check(self.L.ceno_hip_selector_build(self.h, kind, _p(point) if point.shape[0] else None, point.shape[0],
                                                  offset, num_instances, _p32(si), len(sparse_indices), sparse_num_vars,
                                                  stream, C.byref(h)))
        return Mle(self, h)

    def rotation_next_base_mle(self, m: "Mle", cyclic_group_log2: int, stream=None) -> "Mle":
        h = C.c_void_p()
        self.check(self.L.ceno_hip_rotation_next_base_mle(self.h, m.h, cyclic_group_log2, stream, C.byref(h)))
        return Mle(self, h)

    def rotation_selector_build(self, point: np.ndarray, cyclic_subgroup_size: int, cyclic_group_log2: int, stream=None) -> "Mle":
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 2)
        h = C.c_void_p()
        self.check(self.L.ceno_hip_rotation_selector_build(self.h, _p(point), point.shape[0], cyclic_subgroup_size,
                                                           cyclic_group_log2, stream, C.byref(h)))
        return Mle(self, h)

    def wit_infer(self, mles: Sequence["Mle"], coeffs: np.ndarray, terms: Sequence[Sequence[int]],
                  out_terms: Sequence[Sequence[int]], num_vars: int, stream=None) -> List["Mle"]:
        """out o = sum over term ids in out_terms[o]; terms must be listed so that each output owns a
        contiguous range (the CSR the C ABI takes)."""
        order = [t for o in out_terms for t in o]
        assert order == list(range(len(terms))), "terms must be grouped contiguously per output"
        toff, tidx = _csr(terms)
        ooff = np.zeros(len(out_terms) + 1, dtype=np.uint32)
        for o, ts in enumerate(out_terms):
            ooff[o + 1] = ooff[o] + len(ts)
        arr = (C.c_void_p * len(mles))(*[m.h for m in mles])
        outs = (C.c_void_p * len(out_terms))()
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
        self.check(self.L.ceno_hip_wit_infer(self.h, arr, len(mles), _p(coeffs), _p32(toff), _p32(tidx), len(terms),
                                             _p32(ooff), len(out_terms), num_vars, stream, outs))
        return [Mle(self, C.c_void_p(o)) for o in outs]

    # ---- profiling hooks ----
    def prof_enable(self, on=True):
        """True / 1: unpipelined, events around the bare kernels; 2: pipelined sumchecks stay pipelined (events around the queued rounds)"""
        self.check(self.L.ceno_hip_prof_enable(self.h, int(on)))

    def prof_reset(self):
        self.check(self.L.ceno_hip_prof_reset(self.h))

    def prof_get(self):
        ms, n, b = C.c_double(), C.c_uint64(), C.c_double()
        self.check(self.L.ceno_hip_prof_get(self.h, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value


def _csr(terms: Sequence[Sequence[int]]):
    off = np.zeros(len(terms) + 1, dtype=np.uint32)
    idx: List[int] = []
    for t, s in enumerate(terms):
        idx.extend(s)
        off[t + 1] = len(idx)
    return off, np.array(idx if idx else [0], dtype=np.uint32)


class Mle:
    """device multilinear polynomial (reference: MultilinearExtensionGpu, gkr_iop/src/gpu/mod.rs:157-370)"""

    def __init__(self, dev: Device, h, borrowed: bool = False):
        self.dev, self.h, self.borrowed = dev, h, borrowed

    @property
    def num_vars(self) -> int:
        return self.dev.L.ceno_hip_mle_num_vars(self.h)

    @property
    def is_ext(self) -> bool:
        return bool(self.dev.L.ceno_hip_mle_is_ext(self.h))

    @property
    def device_ptr(self) -> int:
        return self.dev.L.ceno_hip_mle_device_ptr(self.h) or 0

    def download(self, stream=None) -> np.ndarray:
        n = 1 << self.num_vars
        out = np.empty((n, 2) if self.is_ext else (n,), dtype=np.uint64)
        self.dev.check(self.dev.L.ceno_hip_mle_download(self.dev.h, self.h, _p(out), stream))
        return out

    def evaluate(self, point: np.ndarray, stream=None) -> Tuple[int, int]:
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 2)
        assert point.shape[0] == self.num_vars
        o = np.zeros(2, dtype=np.uint64)
        self.dev.check(self.dev.L.ceno_hip_mle_evaluate(self.dev.h, self.h, _p(point) if point.shape[0] else None, _p(o), stream))
        return int(o[0]), int(o[1])

    def fix_variables(self, point: np.ndarray, stream=None) -> "Mle":
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 2)
        h = C.c_void_p()
        self.dev.check(self.dev.L.ceno_hip_mle_fix_variables(self.dev.h, self.h, _p(point) if point.shape[0] else None,
                                                             point.shape[0], stream, C.byref(h)))
        return Mle(self.dev, h)

    def view_chunk(self, sub_vars: int, chunk: int) -> "Mle":
        h = C.c_void_p()
        self.dev.check(self.dev.L.ceno_hip_mle_view_chunk(self.dev.h, self.h, sub_vars, chunk, C.byref(h)))
        m = Mle(self.dev, h)
        m._parent = self  # keep the parent alive (gkr_iop/src/gpu/mod.rs:244-253)
        return m

    def fill_zero(self, stream=None):
        self.dev.check(self.dev.L.ceno_hip_mle_fill_zero(self.dev.h, self.h, stream))

    def free(self):
        if self.h and not self.borrowed and self.dev.h:
            self.dev.L.ceno_hip_mle_free(self.dev.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Sumcheck:
    """round-granular generic sumcheck (reference: prove_generic_sumcheck_gpu, layer/gpu/mod.rs:259-271)"""

    def __init__(self, dev: Device, mles: Sequence[Mle], coeffs: np.ndarray, terms: Sequence[Sequence[int]],
                 max_num_vars: int, max_degree: int, groups: Optional[Sequence[Tuple[Sequence[int], Sequence[int]]]] = None,
                 stream=None, _handle=None):
        self.dev = dev
        self.mles = list(mles)  # keep inputs alive
        self.n, self.d = max_num_vars, max_degree
        if _handle is not None:
            self.h = _handle
            return
        coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 2)
        assert coeffs.shape[0] == len(terms)
        toff, tidx = _csr(terms)
        plan = SumcheckPlan()
        plan.num_mles, plan.num_terms = len(mles), len(terms)
        plan.term_coeffs, plan.term_offsets, plan.term_mle_idx = _p(coeffs), _p32(toff), _p32(tidx)
        keep = [coeffs, toff, tidx]
        if groups:
            goff, gidx = _csr([g[1] for g in groups])
            coff, cidx = _csr([g[0] for g in groups])
            plan.num_groups = len(groups)
            plan.group_term_offsets, plan.group_term_idx = _p32(goff), _p32(gidx)
            plan.common_offsets, plan.common_mle_idx = _p32(coff), _p32(cidx)
            keep += [goff, gidx, coff, cidx]
        else:
            plan.num_groups = 0
        plan.max_num_vars, plan.max_degree = max_num_vars, max_degree
        arr = (C.c_void_p * len(mles))(*[m.h for m in mles])
        h = C.c_void_p()
        dev.check(dev.L.ceno_hip_sumcheck_begin(dev.h, arr, C.byref(plan), stream, C.byref(h)))
        self.h = h

    def set_pipelined(self, on: bool = True):
        self.dev.check(self.dev.L.ceno_hip_sumcheck_set_pipelined(self.dev.h, self.h, int(on)))

    def round(self, challenge=None) -> np.ndarray:
        out = np.zeros((self.d, 2), dtype=np.uint64)
        ch = _p(_ext1(challenge)) if challenge is not None else None
        self.dev.check(self.dev.L.ceno_hip_sumcheck_round(self.dev.h, self.h, ch, _p(out)))
        return out

    def round_dev(self, challenge, dev_out_ptr: int):
        ch = _p(_ext1(challenge)) if challenge is not None else None
        self.dev.check(self.dev.L.ceno_hip_sumcheck_round_dev(self.dev.h, self.h, ch, C.c_void_p(dev_out_ptr)))

    def finish(self, last_challenge) -> np.ndarray:
        out = np.zeros((len(self.mles), 2), dtype=np.uint64)
        ch = _p(_ext1(last_challenge)) if last_challenge is not None else None
        self.dev.check(self.dev.L.ceno_hip_sumcheck_finish(self.dev.h, self.h, ch, _p(out)))
        return out

    def free(self):
        if getattr(self, "h", None) and self.dev.h:
            self.dev.L.ceno_hip_sumcheck_free(self.dev.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class OpenRounds:
    """the commit phase's sumcheck of a batch opening over matrices of mixed heights, one launch per round for all of them
    (ceno_hip_open_rounds_*, include/ceno_hip.h; protocol: ceno_recursion_v2/src/pcs/mod.rs:1111-1316).  eq / f: extension tables (Mle) per matrix."""

    def __init__(self, dev: Device, eq: Sequence[Mle], f: Sequence[Mle], num_vars: Sequence[int], stream=None):
        assert len(eq) == len(f) == len(num_vars)
        self.dev, self.keep, self.n_mats, self.n = dev, (list(eq), list(f)), len(eq), max(num_vars)
        pe = (C.c_void_p * len(eq))(*[m.device_ptr for m in eq])
        pf = (C.c_void_p * len(f))(*[m.device_ptr for m in f])
        nv = (C.c_int * len(num_vars))(*num_vars)
        h = C.c_void_p()
        dev.check(dev.L.ceno_hip_open_rounds_begin(dev.h, len(eq), pe, pf, nv, stream, C.byref(h)))
        self.h = h

    def round(self, challenge_prev=None) -> np.ndarray:
        out = np.zeros((2, 2), dtype=np.uint64)
        ch = _p(_ext1(challenge_prev)) if challenge_prev is not None else None
        self.dev.check(self.dev.L.ceno_hip_open_rounds_round(self.dev.h, self.h, ch, _p(out)))
        return out

    def finish(self, challenge_last=None) -> np.ndarray:
        out = np.zeros((self.n_mats, 2), dtype=np.uint64)
        ch = _p(_ext1(challenge_last)) if challenge_last is not None else None
        self.dev.check(self.dev.L.ceno_hip_open_rounds_finish(self.dev.h, self.h, ch, _p(out)))
        return out

    def free(self):
        if getattr(self, "h", None) and self.dev.h:
            self.dev.L.ceno_hip_open_rounds_free(self.dev.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- Basefold commit path over raw device memory (torch tensors or Mle.device_ptr) -----------------
def ntt_batch(dev: Device, dev_ptr: int, log_n: int, n_cols: int, inverse: bool = False, stream=None):
    dev.check(dev.L.ceno_hip_ntt_batch(dev.h, C.c_void_p(dev_ptr), log_n, n_cols, int(inverse), stream))


def rs_encode(dev: Device, src_ptr: int, log_n: int, n_cols: int, log_blowup: int, dst_ptr: int, stream=None):
    dev.check(dev.L.ceno_hip_rs_encode(dev.h, C.c_void_p(src_ptr), log_n, n_cols, log_blowup, C.c_void_p(dst_ptr), stream))


def transpose(dev: Device, src_ptr: int, rows: int, width: int, dst_ptr: int, stream=None):
    dev.check(dev.L.ceno_hip_transpose(dev.h, C.c_void_p(src_ptr), rows, width, C.c_void_p(dst_ptr), stream))


class ArithColumnMap(C.Structure):
    """ceno_hip_add_column_map / ceno_hip_sub_column_map (identical layout: 22 column ids + num_cols)"""
    _fields_ = [("cols", C.c_uint32 * 22), ("num_cols", C.c_uint32)]


def witgen_arith(dev: Device, cols, is_sub: bool, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int,
                 rows_padded: int, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0,
                 lk_fetch_ptr: int = 0, stream=None):
    """hal.witgen.witgen_add / witgen_sub (ceno_zkvm/src/instructions/gpu/dispatch.rs:509-571): `cols` = the 22 column ids in
    AddColumnMap / SubColumnMap field order followed by num_cols; all pointers are device pointers"""
    m = _colmap(22, cols, ArithColumnMap)
    f = dev.L.ceno_hip_witgen_sub if is_sub else dev.L.ceno_hip_witgen_add
    dev.check(f(dev.h, C.byref(m), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset, fetch_base_pc,
                fetch_num_slots, C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None), C.c_void_p(lk_fetch_ptr or None),
                stream))


class AddiColumnMap(C.Structure):
    """ceno_hip_addi_column_map: 18 column ids + num_cols"""
    _fields_ = [("cols", C.c_uint32 * 18), ("num_cols", C.c_uint32)]


def witgen_addi(dev: Device, cols, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
                shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0, stream=None):
    """hal.witgen.witgen_addi (GpuWitgenKind::Addi): `cols` = the 18 column ids in AddiColumnMap field order followed by num_cols"""
    m = _colmap(18, cols, AddiColumnMap)
    dev.check(dev.L.ceno_hip_witgen_addi(dev.h, C.byref(m), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset,
                                         fetch_base_pc, fetch_num_slots, C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None),
                                         C.c_void_p(lk_fetch_ptr or None), stream))


def _witgen_4tab(dev: Device, fn, n_cols: int, cols, records_ptr, num_records, indices_ptr, n, witness_ptr, rows_padded, shard_offset, fetch_base_pc,
                 fetch_num_slots, lk_dynamic_ptr, lk_fetch_ptr, lk_double_u8_ptr, lk_xor_ptr, stream):
    m = _colmap(n_cols, cols)
    dev.check(fn(dev.h, C.byref(m), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset, fetch_base_pc, fetch_num_slots,
                 C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None), C.c_void_p(lk_fetch_ptr or None),
                 C.c_void_p(lk_double_u8_ptr or None), C.c_void_p(lk_xor_ptr or None), stream))


def witgen_jal(dev: Device, cols, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
               shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0,
               lk_double_u8_ptr: int = 0, lk_xor_ptr: int = 0, stream=None):
    """hal.witgen.witgen_jal (GpuWitgenKind::Jal): `cols` = the 13 column ids in JalColumnMap field order followed by num_cols"""
    _witgen_4tab(dev, dev.L.ceno_hip_witgen_jal, 13, cols, records_ptr, num_records, indices_ptr, n, witness_ptr, rows_padded, shard_offset, fetch_base_pc,
                 fetch_num_slots, lk_dynamic_ptr, lk_fetch_ptr, lk_double_u8_ptr, lk_xor_ptr, stream)


def witgen_auipc(dev: Device, cols, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
                 shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0,
                 lk_double_u8_ptr: int = 0, lk_xor_ptr: int = 0, stream=None):
    """hal.witgen.witgen_auipc (GpuWitgenKind::Auipc): `cols` = the 21 column ids in AuipcColumnMap field order followed by num_cols"""
    _witgen_4tab(dev, dev.L.ceno_hip_witgen_auipc, 21, cols, records_ptr, num_records, indices_ptr, n, witness_ptr, rows_padded, shard_offset,
                 fetch_base_pc, fetch_num_slots, lk_dynamic_ptr, lk_fetch_ptr, lk_double_u8_ptr, lk_xor_ptr, stream)


def witgen_slt(dev: Device, cols, is_signed: bool, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
               shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0, stream=None):
    """hal.witgen.witgen_slt (GpuWitgenKind::Slt): `cols` = the 26 column ids in SltColumnMap field order followed by num_cols"""
    m = _colmap(26, cols)
    dev.check(dev.L.ceno_hip_witgen_slt(dev.h, C.byref(m), int(is_signed), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset,
                                        fetch_base_pc, fetch_num_slots, C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None),
                                        C.c_void_p(lk_fetch_ptr or None), stream))


def witgen_slti(dev: Device, cols, is_signed: bool, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
                shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0, stream=None):
    """hal.witgen.witgen_slti (GpuWitgenKind::Slti): `cols` = the 22 column ids in SltiColumnMap field order followed by num_cols"""
    m = _colmap(22, cols)
    dev.check(dev.L.ceno_hip_witgen_slti(dev.h, C.byref(m), int(is_signed), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset,
                                         fetch_base_pc, fetch_num_slots, C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None),
                                         C.c_void_p(lk_fetch_ptr or None), stream))


def witgen_branch(dev: Device, cols, is_eq: bool, flag: bool, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int,
                  rows_padded: int, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0,
                  stream=None):
    """hal.witgen.witgen_branch_cmp (is_eq False; flag = is_signed, 22 column ids) / witgen_branch_eq (is_eq True; flag = is_beq, 19 column ids)"""
    nc = 19 if is_eq else 22

    m = _colmap(nc, cols)
    fn = dev.L.ceno_hip_witgen_branch_eq if is_eq else dev.L.ceno_hip_witgen_branch_cmp
    dev.check(fn(dev.h, C.byref(m), int(flag), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset, fetch_base_pc,
                 fetch_num_slots, C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None), C.c_void_p(lk_fetch_ptr or None), stream))


def witgen_shift(dev: Device, cols, is_imm: bool, kind: int, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
                 shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0,
                 lk_double_u8_ptr: int = 0, lk_xor_ptr: int = 0, stream=None):
    """hal.witgen.witgen_shift_r / witgen_shift_i (kind 0 = left, 1 = logical right, 2 = arithmetic right): `cols` = the 47 / 40 column ids in
    ShiftRColumnMap / ShiftIColumnMap field order followed by num_cols"""
    nc = 40 if is_imm else 47

    m = _colmap(nc, cols)
    fn = dev.L.ceno_hip_witgen_shift_i if is_imm else dev.L.ceno_hip_witgen_shift_r
    dev.check(fn(dev.h, C.byref(m), int(kind), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset, fetch_base_pc, fetch_num_slots,
                 C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None), C.c_void_p(lk_fetch_ptr or None),
                 C.c_void_p(lk_double_u8_ptr or None), C.c_void_p(lk_xor_ptr or None), stream))


def witgen_jalr(dev: Device, cols, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
                shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0, stream=None):
    """hal.witgen.witgen_jalr (GpuWitgenKind::Jalr): `cols` = the 22 column ids in JalrColumnMap field order followed by num_cols"""
    m = _colmap(22, cols)
    dev.check(dev.L.ceno_hip_witgen_jalr(dev.h, C.byref(m), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset, fetch_base_pc,
                                         fetch_num_slots, C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None),
                                         C.c_void_p(lk_fetch_ptr or None), stream))


NO_COLUMN = 0xFFFFFFFF


def witgen_div(dev: Device, cols, div_kind: int, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
               shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0, stream=None):
    """hal.witgen.witgen_div (div_kind 0 DIV, 1 DIVU, 2 REM, 3 REMU): `cols` = the 39 column ids in DivColumnMap field order followed by num_cols"""
    m = _colmap(39, cols)
    dev.check(dev.L.ceno_hip_witgen_div(dev.h, C.byref(m), int(div_kind), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset,
                                        fetch_base_pc, fetch_num_slots, C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None),
                                        C.c_void_p(lk_fetch_ptr or None), stream))


def witgen_mul(dev: Device, cols, mul_kind: int, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
               shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0, stream=None):
    """hal.witgen.witgen_mul (mul_kind 0 MUL, 1 MULH, 2 MULHU, 3 MULHSU): `cols` = the 26 column ids in MulColumnMap field order (NO_COLUMN in
    rd_high / rs1_ext / rs2_ext for MUL) followed by num_cols"""
    m = _colmap(26, cols)
    dev.check(dev.L.ceno_hip_witgen_mul(dev.h, C.byref(m), int(mul_kind), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset,
                                        fetch_base_pc, fetch_num_slots, C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None),
                                        C.c_void_p(lk_fetch_ptr or None), stream))


def witgen_load_sub(dev: Device, cols, load_width: int, is_signed: bool, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int,
                    rows_padded: int, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0,
                    stream=None):
    """hal.witgen.witgen_load_sub (LH / LHU / LB / LBU): `cols` = the 29 column ids in LoadSubColumnMap field order (NO_COLUMN for the Option fields
    the variant does not have) followed by num_cols"""
    m = _colmap(29, cols)
    dev.check(dev.L.ceno_hip_witgen_load_sub(dev.h, C.byref(m), int(load_width), int(is_signed), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n,
                                             shard_offset, fetch_base_pc, fetch_num_slots, C.c_void_p(witness_ptr), rows_padded,
                                             C.c_void_p(lk_dynamic_ptr or None), C.c_void_p(lk_fetch_ptr or None), stream))


def witgen_mem(dev: Device, cols, is_store, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
               shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0, stream=None):
    """hal.witgen.witgen_lw / witgen_sw / witgen_sh / witgen_sb: `cols` = the 23 (LW, SW), 24 (SH) or 29 (SB) column ids in the chip's
    ColumnMap field order followed by num_cols; `is_store` = 0 / False (LW), 1 / True (SW), 2 (SH) or 3 (SB)"""
    nc = {0: 23, 1: 23, 2: 24, 3: 29}[int(is_store)]

    m = _colmap(nc, cols)
    fn = {0: dev.L.ceno_hip_witgen_lw, 1: dev.L.ceno_hip_witgen_sw, 2: dev.L.ceno_hip_witgen_sh, 3: dev.L.ceno_hip_witgen_sb}[int(is_store)]
    dev.check(fn(dev.h, C.byref(m), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset, fetch_base_pc, fetch_num_slots,
                 C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None), C.c_void_p(lk_fetch_ptr or None), stream))


class LuiColumnMap(C.Structure):
    """ceno_hip_lui_column_map: 16 column ids + num_cols"""
    _fields_ = [("cols", C.c_uint32 * 16), ("num_cols", C.c_uint32)]


def witgen_lui(dev: Device, cols, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int, rows_padded: int,
               shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0, lk_fetch_ptr: int = 0, stream=None):
    """hal.witgen.witgen_lui (GpuWitgenKind::Lui): `cols` = the 16 column ids in LuiColumnMap field order followed by num_cols"""
    m = _colmap(16, cols, LuiColumnMap)
    dev.check(dev.L.ceno_hip_witgen_lui(dev.h, C.byref(m), C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n, shard_offset,
                                        fetch_base_pc, fetch_num_slots, C.c_void_p(witness_ptr), rows_padded, C.c_void_p(lk_dynamic_ptr or None),
                                        C.c_void_p(lk_fetch_ptr or None), stream))


class LogicIColumnMap(C.Structure):
    """ceno_hip_logic_i_column_map: 24 column ids + num_cols"""
    _fields_ = [("cols", C.c_uint32 * 24), ("num_cols", C.c_uint32)]


def witgen_logic_i(dev: Device, cols, logic_kind: int, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int,
                   rows_padded: int, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0,
                   lk_fetch_ptr: int = 0, lk_logic_ptr: int = 0, stream=None):
    """hal.witgen.witgen_logic_i (GpuWitgenKind::LogicI): `cols` = the 24 column ids in LogicIColumnMap field order followed by num_cols"""
    m = _colmap(24, cols, LogicIColumnMap)
    dev.check(dev.L.ceno_hip_witgen_logic_i(dev.h, C.byref(m), logic_kind, C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n,
                                            shard_offset, fetch_base_pc, fetch_num_slots, C.c_void_p(witness_ptr), rows_padded,
                                            C.c_void_p(lk_dynamic_ptr or None), C.c_void_p(lk_fetch_ptr or None), C.c_void_p(lk_logic_ptr or None),
                                            stream))


class LogicRColumnMap(C.Structure):
    """ceno_hip_logic_r_column_map: 28 column ids + num_cols"""
    _fields_ = [("cols", C.c_uint32 * 28), ("num_cols", C.c_uint32)]


def witgen_logic_r(dev: Device, cols, logic_kind: int, records_ptr: int, num_records: int, indices_ptr: int, n: int, witness_ptr: int,
                   rows_padded: int, shard_offset: int = 0, fetch_base_pc: int = 0, fetch_num_slots: int = 0, lk_dynamic_ptr: int = 0,
                   lk_fetch_ptr: int = 0, lk_logic_ptr: int = 0, stream=None):
    """hal.witgen.witgen_logic_r (ceno_zkvm/src/instructions/gpu/dispatch.rs:574-611): `cols` = the 28 column ids in LogicRColumnMap
    field order followed by num_cols; logic_kind 0 AND / 1 OR / 2 XOR; all pointers are device pointers"""
    m = _colmap(28, cols, LogicRColumnMap)
    dev.check(dev.L.ceno_hip_witgen_logic_r(dev.h, C.byref(m), logic_kind, C.c_void_p(records_ptr), num_records, C.c_void_p(indices_ptr), n,
                                            shard_offset, fetch_base_pc, fetch_num_slots, C.c_void_p(witness_ptr), rows_padded,
                                            C.c_void_p(lk_dynamic_ptr or None), C.c_void_p(lk_fetch_ptr or None), C.c_void_p(lk_logic_ptr or None),
                                            stream))


def poseidon2_permute(dev: Device, states_ptr: int, n: int, stream=None):
    dev.check(dev.L.ceno_hip_poseidon2_permute(dev.h, C.c_void_p(states_ptr), n, stream))


def poseidon2_set_constants(dev: Device, ext_rc=None, int_rc=None, int_diag=None):
    def p(a):
        return _p(np.ascontiguousarray(a, dtype=np.uint64)) if a is not None else None

    dev.check(dev.L.ceno_hip_poseidon2_set_constants(dev.h, p(ext_rc), p(int_rc), p(int_diag)))


class Merkle:
    """Poseidon2 Merkle tree over the rows of a column-major matrix (reference: basefold PcsData)"""

    def __init__(self, dev: Device, col_major_ptr: int, log_rows: int, width: int, stream=None):
        self.dev, self.log_rows = dev, log_rows
        h = C.c_void_p()
        dev.check(dev.L.ceno_hip_merkle_commit(dev.h, C.c_void_p(col_major_ptr), log_rows, width, stream, C.byref(h)))
        self.h = h

    def root(self, stream=None) -> np.ndarray:
        out = np.zeros(4, dtype=np.uint64)
        self.dev.check(self.dev.L.ceno_hip_merkle_root(self.dev.h, self.h, _p(out), stream))
        return out

    def open(self, index: int, stream=None) -> np.ndarray:
        out = np.zeros((max(self.log_rows, 1), 4), dtype=np.uint64)
        self.dev.check(self.dev.L.ceno_hip_merkle_open(self.dev.h, self.h, index, _p(out), stream))
        return out[: self.log_rows]

    def free(self):
        if getattr(self, "h", None) and self.dev.h:
            self.dev.L.ceno_hip_merkle_free(self.dev.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
