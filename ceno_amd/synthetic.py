"""Synthetic workloads of SURVEY.md section 8d, shared by bench.py and tools/: the ADD-shaped chip ("S-chip", BASELINE.json
config #3, harness shape ceno_zkvm/benches/riscv_add.rs:86-141) and the mixed-size batch of chips ("S-batched", config #4,
ceno_zkvm/src/scheme/cpu/mod.rs:1052-1390).  The real ComposedConstrainSystem of the ADD chip needs the Rust front end, so the
plans below only have its SHAPE: 22 base witness columns, 4 read + 4 write + 8 lookup records (RLCs of columns with the two
global challenges), main constraints = selector x (degree-2 and degree-3 products of columns)."""
from __future__ import annotations

import time

import numpy as np

from .api import P


def _e2_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def record_plan(w: int, n_records: int, alpha, beta):
    """record_k = beta * col[2k] + beta^2 * col[2k+1] + alpha * col[3k+5] * col[k+7]  (ids mod w)"""
    b2 = _e2_mul(beta, beta)
    terms, coeffs, out_terms = [], [], []
    for k in range(n_records):
        base = len(terms)
        terms += [[(2 * k) % w], [(2 * k + 1) % w], [(3 * k + 5) % w, (k + 7) % w]]
        coeffs += [beta, b2, alpha]
        out_terms.append([base, base + 1, base + 2])
    return np.array(coeffs, dtype=np.uint64), terms, out_terms


def main_plan(w: int, s_id: int, step3: int = 2):
    """selector x [ w degree-2 products + w/step3 degree-3 products ]; scalar_t = (3 + 5t, 11t + 1) * alpha_pow[t % 2]"""
    terms = [[s_id, j, (j + 1) % w] for j in range(w)] + [[s_id, j, (j + 3) % w, (j + 5) % w] for j in range(0, w, step3)]
    scalars = [[(((3 + 5 * t) % P, (11 * t + 1) % P), [2 + (t % 2)])] for t in range(len(terms))]
    return terms, scalars


def batched_sizes(max_nv: int):
    return [max_nv, max_nv - 2, max_nv - 2] + [max_nv - 4] * 5 + [max_nv - 6] * 8 + [max_nv - 10] * 8


def batched_jobs(dev, max_nv: int = 24, w: int = 12):
    """24 chips of max_nv .. max_nv - 10 variables, 12 base columns + one Prefix selector each, 16 constraint terms"""
    jobs, elems = [], 0
    for c, nv in enumerate(batched_sizes(max_nv)):
        cols = [dev.synthetic(nv, False, 1000 + 50 * c + j) for j in range(w)]
        point = np.array([[(i * 7919 + 13 + c) % P, (i * 104729 + 17) % P] for i in range(nv)], dtype=np.uint64)
        sel = (1, 0, max(1, (1 << nv) - 5 - c), 0, (), 0, point)  # Prefix selector
        terms = [[w, j, (j + 1) % w] for j in range(w)] + [[w, j, (j + 3) % w, (j + 5) % w] for j in range(0, w, 3)]
        scalars = [[((3 + t, 1), [2 + (t % 2)])] for t in range(len(terms))]
        jobs.append(dict(num_vars=nv, mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4,
                         terms=terms, scalars=scalars))
        elems += (w + 1) << nv
    return jobs, elems


def batched_algorithmic_bytes(max_nv: int = 24, w: int = 12) -> float:
    """a base-field table costs 40 B per element over the whole sumcheck (read 8 B twice before the first
    fold can happen, then the ext schedule on the half-size table), an ext table 48 B (3 x 16)"""
    return float(sum((40 * w + 48) << nv for nv in batched_sizes(max_nv)))


class ChipFlow:
    """config #3: device-resident 2^log_rows x w base trace -> commit -> 2 challenges -> create_chip_proof ->
    batched main constraints (one job) -> Basefold open.  `run()` returns per-phase wall times in ms."""

    def __init__(self, dev, prover, log_rows: int = 20, w: int = 22, log_blowup: int = 1, n_queries: int = 100, pow_bits: int = 16):
        self.dev, self.prover = dev, prover
        self.log_rows, self.w, self.log_blowup, self.n_queries, self.pow_bits = log_rows, w, log_blowup, n_queries, pow_bits
        self.rows = 1 << log_rows
        # RowMajorMatrix::rand (benches/riscv_add.rs:88): uniform base-field words, row-major, resident in HBM
        self.trace = dev.synthetic((self.rows * w - 1).bit_length(), False, 0xADD)
        self.stream = dev.stream_create()

    def algorithmic_bytes(self) -> dict:
        rows, w, n = self.rows, self.w, self.log_rows
        N = rows << self.log_blowup
        n_rec = 16
        tower = 2 * 48 * (1 << (n + 2)) + 96 * (1 << (n + 3))                    # build: two product towers, one LogUp tower
        tower_proof = sum(3 * 16 * (1 + 2 * 2 + 4) * (1 << r) for r in range(1, n + 3))   # eq + 2x(a,b) + (p1,p2,q1,q2) per layer
        return {
            "commit": 16 * rows * w + 48 * N * w + 8 * w * N + 64 * N,             # transpose, RS encode (<= 3 passes), leaves + tree
            "chip_proof": (8 * w + 16 * n_rec) * rows + tower + tower_proof,       # record inference + tower build + tower proof
            "main": (40 * w + 48) * rows,
            "open": 8 * w * (N + rows) + 16 * 3 * N,                               # batch codeword + trace columns, fold chain
        }

    def run(self, transcript_factory) -> dict:
        dev, prover = self.dev, self.prover
        w, n, rows = self.w, self.log_rows, self.rows

        def timed(f):
            dev.sync()
            t0 = time.perf_counter()
            r = f()
            dev.sync()
            return r, (time.perf_counter() - t0) * 1e3

        res = {}
        pcs, res["commit_ms"] = timed(lambda: prover.PcsData(dev, None, self.log_blowup, self.stream, device_ptrs=[(self.trace.device_ptr, rows, w)]))
        tr = transcript_factory()
        root = pcs.root(0)
        tr.append_ext((int(root[0]), int(root[1])))
        tr.append_ext((int(root[2]), int(root[3])))
        alpha, beta = tr.sample_ext(), tr.sample_ext()
        cols = [pcs.witness_mle(0, c) for c in range(w)]
        coeffs, terms, out_terms = record_plan(w, 16, alpha, beta)
        task = dict(mles=cols, n_witin=w, n_fixed=0, n_structural=0, num_instances=rows - 3, log2_num_instances=n, num_reads=4, num_writes=4,
                    num_lk_tables=0, num_lk=8, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
        proof, res["chip_proof_ms"] = timed(lambda: prover.create_chip_proof(dev, task, [alpha, beta], tr, self.stream))
        mterms, mscalars = main_plan(w, w)
        sel = (1, 0, rows - 3, 0, (), 0, proof.rt_main)
        job = dict(num_vars=n, mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4,
                   terms=mterms, scalars=mscalars)
        (claimed, msgs, rt, evals), res["main_ms"] = timed(lambda: prover.prove_batched_main_constraints(dev, [job], [alpha, beta], tr, self.stream))
        oproof, res["open_ms"] = timed(lambda: pcs.basefold_open([rt], [evals[:w]], self.n_queries, self.pow_bits, tr))
        res["total_ms"] = res["commit_ms"] + res["chip_proof_ms"] + res["main_ms"] + res["open_ms"]
        res["tower_num_vars"] = proof.tower_num_vars
        res["open_proof_bytes"] = int(oproof.size * 8)
        for m in cols:
            m.free()
        pcs.free()
        return res

    def close(self):
        self.trace.free()
        self.dev.stream_destroy(self.stream)
