"""Loads the reference's Poseidon2-Goldilocks constants from a `ref_goldens.json` produced by tools/goldens/ (see README
"Closing parity") into the device library and the host transcript.  Without that file both run on placeholder constants
(PARITY UNPINNED) and `is_pinned()` is False."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "tests", "golden", "ref_goldens.json")


def path() -> str:
    return os.environ.get("CENO_REF_GOLDENS", DEFAULT)


def parse_constants(g):
    """(external 8x8, internal 22, diag 8 or None) as uint64 arrays from whatever shape the dump produced"""
    rc = g["poseidon2"]["round_constants"]
    if isinstance(rc, dict):
        ext = np.array(rc["external"], dtype=np.uint64).reshape(8, 8)
        internal = np.array(rc["internal"], dtype=np.uint64).reshape(22)
        diag = np.array(rc["diag"], dtype=np.uint64).reshape(8) if rc.get("diag") is not None else None
        return ext, internal, diag
    flat = np.array(rc, dtype=np.uint64).reshape(-1)
    if flat.size == 8 * 8 + 22:          # external initial (4 x 8), internal (22), external terminal (4 x 8)
        ext = np.concatenate([flat[:32], flat[32 + 22:]]).reshape(8, 8)
        return ext, flat[32: 32 + 22].copy(), None
    if flat.size == 30 * 8:              # one row of 8 per round, internal rounds use word 0
        rows = flat.reshape(30, 8)
        return np.concatenate([rows[:4], rows[26:]]), rows[4:26, 0].copy(), None
    raise ValueError(f"unrecognised round-constant table of {flat.size} words")


def install(dev=None, file: str | None = None) -> bool:
    """Installs the constants of `file` (default: CENO_REF_GOLDENS or tests/golden/ref_goldens.json) into the host
    transcript and, when `dev` is given, the device library.  Returns False (and changes nothing) if the file is absent."""
    from . import api, prover

    file = file or path()
    if not os.path.exists(file):
        return False
    ext, internal, diag = parse_constants(json.load(open(file)))
    u64p = C.POINTER(C.c_uint64)

    def p(a):
        return np.ascontiguousarray(a, dtype=np.uint64).ctypes.data_as(u64p) if a is not None else None

    L = prover.plib()
    L.ceno_transcript_poseidon2_set_constants.restype = C.c_int
    L.ceno_transcript_poseidon2_set_constants.argtypes = [u64p, u64p, u64p]
    e, i = np.ascontiguousarray(ext.reshape(-1)), np.ascontiguousarray(internal)
    if L.ceno_transcript_poseidon2_set_constants(p(e), p(i), p(diag)) != 0:
        raise RuntimeError("ceno_transcript_poseidon2_set_constants failed")
    if dev is not None:
        api.poseidon2_set_constants(dev, e, i, diag)
    return True
