#!/usr/bin/env python3
"""Per-kernel roofline table: times each streaming primitive at chip-flow sizes on one MI355X and
prints achieved GB/s against the ALGORITHMIC bytes of SURVEY.md §8(d) (wall time over several back-to-back calls,
so launch overhead is included; peak 8000 GB/s)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ceno_amd import Device, api, prover

dev = Device(0)
P = api.P
rows_log, w = 20, 22
rows = 1 << rows_log


def timed(f, reps=5):
    f()
    dev.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    dev.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = []


def row(name, secs, alg_bytes, note=""):
    out.append({"kernel": name, "ms": round(secs * 1e3, 4), "algorithmic_MB": round(alg_bytes / 1e6, 1),
                "GBps": round(alg_bytes / secs / 1e9, 1), "frac_of_8TBps": round(alg_bytes / secs / 8e12, 3), "note": note})


pt24 = np.array([[(i * 7919 + 13) % P, (i * 104729 + 17) % P] for i in range(24)], dtype=np.uint64)
# eq build: 16 * 2^n written
t = timed(lambda: dev.eq_build(pt24).free())
row("eq_build nv=24", t, 16 * (1 << 24))
# evaluate: s * 2^n read
f_ext = dev.synthetic(24, True, 1)
f_base = dev.synthetic(24, False, 2)
row("mle_evaluate ext nv=24", timed(lambda: f_ext.evaluate(pt24)), 16 * (1 << 24))
row("mle_evaluate base nv=24", timed(lambda: f_base.evaluate(pt24)), 8 * (1 << 24))
# fix one variable: 2s read + 16/2 written per input pair
row("fix_variable ext nv=24", timed(lambda: f_ext.fix_variables(pt24[:1]).free()), (16 + 8) * (1 << 24))
# wit_infer: (8 * in_cols + 16 * out_cols) * rows
cols = [dev.synthetic(rows_log, False, 10 + j) for j in range(w)]
alpha, beta = (0x1234567, 0x89abcde), (0x13579b, 0x2468ac)
b2 = ((beta[0] * beta[0] + 7 * beta[1] * beta[1]) % P, (2 * beta[0] * beta[1]) % P)
terms, coeffs, out_terms = [], [], []
for k in range(16):
    base = len(terms)
    terms += [[(2 * k) % w], [(2 * k + 1) % w], [(3 * k + 5) % w, (k + 7) % w]]
    coeffs += [beta, b2, alpha]
    out_terms.append([base, base + 1, base + 2])
coeffs = np.array(coeffs, dtype=np.uint64)
recs = dev.wit_infer(cols, coeffs, terms, out_terms, rows_log)


def wi():
    for m in dev.wit_infer(cols, coeffs, terms, out_terms, rows_log):
        m.free()


row("wit_infer 22 base -> 16 ext, 2^20 rows", timed(wi), (8 * w + 16 * 16) * rows)
# tower build (product): 48 * 2^n per tower of n variables
row("tower_build_prod 4 records x 2^20 (nv=22)", timed(lambda: prover.Tower.build_prod(dev, recs[:4], rows, (1, 0)).free()), 48 * (1 << 22),
    "plus the interleave read of 4 x 16 MB")
row("tower_build_logup 8 records x 2^20 (nv=23)", timed(lambda: prover.Tower.build_logup(dev, None, recs[8:16], rows, alpha).free()),
    2 * 48 * (1 << 23), "4 limbs per layer")
# transpose
rm = dev.synthetic((rows * w - 1).bit_length(), False, 0xADD)
cm = torch.empty(rows * w, dtype=torch.int64, device="cuda:0")
row("transpose 2^20 x 22", timed(lambda: api.transpose(dev, rm.device_ptr, rows, w, cm.data_ptr())), 16 * rows * w)
# RS encode (blow-up 2): SURVEY target <= 48 N per column, N = codeword length
cw = torch.empty((rows << 1) * w, dtype=torch.int64, device="cuda:0")
row("rs_encode 22 x 2^20 -> 2^21", timed(lambda: api.rs_encode(dev, cm.data_ptr(), rows_log, w, 1, cw.data_ptr())), 48 * (rows << 1) * w,
    "3 passes move 44 N per column")
# Merkle: 8 w N leaf read + 64 N tree
row("merkle_commit 2^21 x 22", timed(lambda: api.Merkle(dev, cw.data_ptr(), rows_log + 1, w).free(), reps=3), (8 * w + 64) * (rows << 1),
    "integer-ALU bound: 6 + 1 Poseidon2 permutations per row")
# on-device witness generation of the ADD chip: 136 B step record read + 22 x 8 B written per instance (+ 9 lookup counts)
n_w = rows
rng = np.random.default_rng(3)
rec = np.zeros((n_w, 17), dtype=np.uint64)
v1, v2 = rng.integers(0, 1 << 32, n_w, dtype=np.uint64), rng.integers(0, 1 << 32, n_w, dtype=np.uint64)
pcs = np.uint64(0x2000) + np.uint64(4) * (np.arange(n_w, dtype=np.uint64) % np.uint64(4096))
rec[:, 0] = 4 + 4 * np.arange(n_w, dtype=np.uint64)
rec[:, 1] = pcs | ((pcs + np.uint64(4)) << np.uint64(32))
rec[:, 4] = np.uint64(1 | (2 << 8) | (3 << 16) | (4 << 24))
rec[:, 5] = np.uint64(0x00010101) << np.uint64(32)
rec[:, 6] = np.uint64((2 << 8) // 4) | (v1 << np.uint64(32))
rec[:, 8] = np.uint64((3 << 8) // 4) | (v2 << np.uint64(32))
rec[:, 10] = np.uint64((4 << 8) // 4)
rec[:, 11] = (v1 + v2) & np.uint64(0xFFFFFFFF)
d_rec = torch.from_numpy(rec.view(np.int64)).to("cuda:0")
d_idx = torch.arange(n_w, dtype=torch.int32, device="cuda:0")
d_w = torch.empty(22 * n_w, dtype=torch.int64, device="cuda:0")
d_lkd = torch.zeros(1 << 19, dtype=torch.int32, device="cuda:0")
d_lkf = torch.zeros(4096, dtype=torch.int32, device="cuda:0")
wcols = list(range(22)) + [22]
row("witgen_add 2^20 instances (22 columns + lookup counts)",
    timed(lambda: api.witgen_arith(dev, wcols, False, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w.data_ptr(), n_w, 0, 0x2000, 4096,
                                   d_lkd.data_ptr(), d_lkf.data_ptr())), (136 + 4 + 8 * 22) * n_w)
row("witgen_add 2^20 instances, witness only",
    timed(lambda: api.witgen_arith(dev, wcols, False, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w.data_ptr(), n_w, 0, 0x2000, 4096)),
    (136 + 4 + 8 * 22) * n_w)
d_w28 = torch.empty(28 * n_w, dtype=torch.int64, device="cuda:0")
d_lkl = torch.zeros(1 << 16, dtype=torch.int32, device="cuda:0")
lcols = list(range(28)) + [28]
row("witgen_logic_r (XOR) 2^20 instances (28 columns + lookup counts: 6 range, fetch, 4 byte-table)",
    timed(lambda: api.witgen_logic_r(dev, lcols, 2, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w28.data_ptr(), n_w, 0, 0x2000, 4096,
                                     d_lkd.data_ptr(), d_lkf.data_ptr(), d_lkl.data_ptr())), (136 + 4 + 8 * 28) * n_w)
row("witgen_logic_r 2^20 instances, witness only",
    timed(lambda: api.witgen_logic_r(dev, lcols, 2, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w28.data_ptr(), n_w, 0, 0x2000, 4096)),
    (136 + 4 + 8 * 28) * n_w)
acols = list(range(18)) + [18]
row("witgen_addi 2^20 instances (18 columns + lookup counts)",
    timed(lambda: api.witgen_addi(dev, acols, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w.data_ptr(), n_w, 0, 0x2000, 4096,
                                  d_lkd.data_ptr(), d_lkf.data_ptr())), (136 + 4 + 8 * 18) * n_w)
# the widest and the most arithmetic of the later chips on the same records (R-type records serve them: rs1, rs2, rd are all present)
d_w47 = torch.empty(47 * n_w, dtype=torch.int64, device="cuda:0")
d_lkx = torch.zeros(1 << 16, dtype=torch.int32, device="cuda:0")
scols = list(range(47)) + [47]
row("witgen_shift_r (SRA) 2^20 instances (47 columns + lookup counts: 11 range, fetch, 2 double-byte, 1 xor)",
    timed(lambda: api.witgen_shift(dev, scols, False, 2, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w47.data_ptr(), n_w, 0, 0x2000, 4096,
                                   d_lkd.data_ptr(), d_lkf.data_ptr(), d_lkl.data_ptr(), d_lkx.data_ptr())), (136 + 4 + 8 * 47) * n_w)
row("witgen_shift_r 2^20 instances, witness only",
    timed(lambda: api.witgen_shift(dev, scols, False, 2, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w47.data_ptr(), n_w, 0, 0x2000, 4096)),
    (136 + 4 + 8 * 47) * n_w)
mcols = list(range(26)) + [26]
row("witgen_mul (MULH) 2^20 instances (26 columns + lookup counts: 16 range of which 4 are 18-bit, fetch)",
    timed(lambda: api.witgen_mul(dev, mcols, 1, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w47.data_ptr(), n_w, 0, 0x2000, 4096,
                                 d_lkd.data_ptr(), d_lkf.data_ptr())), (136 + 4 + 8 * 26) * n_w)
dcols = list(range(39)) + [39]
row("witgen_div (DIV) 2^20 instances (39 columns, four field inversions per row + lookup counts: 17 range, fetch)",
    timed(lambda: api.witgen_div(dev, dcols, 0, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w47.data_ptr(), n_w, 0, 0x2000, 4096,
                                 d_lkd.data_ptr(), d_lkf.data_ptr())), (136 + 4 + 8 * 39) * n_w)
row("witgen_div 2^20 instances, witness only",
    timed(lambda: api.witgen_div(dev, dcols, 0, d_rec.data_ptr(), n_w, d_idx.data_ptr(), n_w, d_w47.data_ptr(), n_w, 0, 0x2000, 4096)),
    (136 + 4 + 8 * 39) * n_w)
print(json.dumps(out, indent=1))
