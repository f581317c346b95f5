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
        root = pcs.root()
        tr.append_ext((int(root[0]), int(root[1])))
        tr.append_ext((int(root[2]), int(root[3])))
        alpha, beta = tr.sample_ext(), tr.sample_ext()
        cols = [pcs.witness_mle(0, c) for c in range(w)]
        coeffs, terms, out_terms = record_plan(w, 16, alpha, beta)
        task = dict(mles=cols, n_witin=w, n_fixed=0, n_structural=0, num_instances=rows - 3, log2_num_instances=n, num_reads=4, num_writes=4,
                    num_lk_tables=0, num_lk=8, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
        ct = prover.ChipTasks([task])  # the C view of the task, marshalled outside the timed call
        proof, res["chip_proof_ms"] = timed(lambda: prover.create_chip_proof(dev, ct, [alpha, beta], tr, self.stream))
        mterms, mscalars = main_plan(w, w)
        sel = (1, 0, rows - 3, 0, (), 0, proof.rt_main)
        job = dict(num_vars=n, mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4,
                   terms=mterms, scalars=mscalars)
        mj = prover.MainJobs([job])  # the C view of the job, marshalled outside the timed call (a Rust caller hands the structs over directly)
        (claimed, msgs, rt, evals), res["main_ms"] = timed(lambda: prover.prove_batched_main_constraints(dev, mj, [alpha, beta], tr, self.stream))
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


class ShardFlow:
    """BASELINE.json metric M2 shape ("e2e prover sec for 2^20 cycles") on synthetic data: one shard whose 2^20 cycles are spread
    over eight ADD-shaped opcode chips of 2^19 .. 2^13 rows (22 base columns, 4 + 4 + 8 records, 33 main-constraint terms each),
    proved the way ZKVMProver::create_proof does it (ceno_zkvm/src/scheme/prover.rs:319-611): commit every trace, bind the
    commitment and draw the two record challenges, one chip proof per circuit on a forked transcript (tower relation), merge one
    sample per fork, ONE batched main-constraint sumcheck over all chips, ONE Basefold opening of all traces.  Witness
    generation and the emulator are upstream of the path and not part of the time."""

    LOG_ROWS = (19, 18, 17, 16, 15, 14, 13, 13)

    def __init__(self, dev, prover, w: int = 22, log_blowup: int = 1, n_queries: int = 100, pow_bits: int = 16, log_rows=None):
        self.dev, self.prover, self.w = dev, prover, w
        self.log_blowup, self.n_queries, self.pow_bits = log_blowup, n_queries, pow_bits
        if log_rows is not None:
            self.LOG_ROWS = tuple(log_rows)   # (tests run the same flow on a small shard)
        else:
            assert sum(1 << r for r in self.LOG_ROWS) == 1 << 20
        self.traces = [dev.synthetic(((1 << r) * w - 1).bit_length(), False, 0x5A0 + i) for i, r in enumerate(self.LOG_ROWS)]
        self.stream = dev.stream_create()

    def run(self, transcript_factory, fork_factory, lanes: int = 1) -> dict:
        """the chip proofs run on `lanes` concurrent lanes of the product's C++ scheduler (ceno_prover_create_chip_proofs: one worker
        thread and one context-owned HIP stream per lane, largest chip first; the reference's chip scheduler,
        ceno_zkvm/src/scheme/scheduler.rs:231-336: forks make the chip transcripts independent, results are merged in task order)"""
        dev, prover, w = self.dev, self.prover, self.w

        def timed(f):
            dev.sync()
            t0 = time.perf_counter()
            r = f()
            dev.sync()
            return r, (time.perf_counter() - t0) * 1e3

        res = {}
        ptrs = [(t.device_ptr, 1 << r, w) for t, r in zip(self.traces, self.LOG_ROWS)]
        pcs, res["commit_ms"] = timed(lambda: prover.PcsData(dev, None, self.log_blowup, self.stream, device_ptrs=ptrs))
        tr = transcript_factory()
        root = pcs.root()                                           # ONE commitment for all traces (PCS::write_commitment, prover.rs)
        tr.append_ext((int(root[0]), int(root[1])))
        tr.append_ext((int(root[2]), int(root[3])))
        alpha, beta = tr.sample_ext(), tr.sample_ext()              # prover.rs:528-531
        coeffs, terms, out_terms = record_plan(w, 16, alpha, beta)
        mterms, mscalars = main_plan(w, w)
        chips, jobs = [], []

        n_chips = len(self.LOG_ROWS)
        cols_all = [[pcs.witness_mle(i, c) for c in range(w)] for i in range(n_chips)]
        tasks = prover.ChipTasks([dict(circuit_idx=i, mles=cols_all[i], n_witin=w, n_fixed=0, n_structural=0, num_instances=(1 << r) - 3,
                                       log2_num_instances=r, num_reads=4, num_writes=4, num_lk_tables=0, num_lk=8, record_coeffs=coeffs,
                                       record_terms=terms, record_out_terms=out_terms) for i, r in enumerate(self.LOG_ROWS)])

        def chip_proofs():
            # forks make the chip transcripts independent (prover.rs:556-558, 682-689); the proofs run on the product's own
            # scheduler — ceno_prover_create_chip_proofs: C++ worker threads on the context's lane streams, largest chip first,
            # booked against the pool (scheduler.rs:231-336) — and come back in task order (scheduler.rs:303-304)
            forks = []
            for i, r in enumerate(self.LOG_ROWS):
                fork = fork_factory()
                fork.append_ext(alpha)
                fork.append_ext(beta)
                for v in (i, i, (1 << r) - 3, 0):
                    fork.append_base(v)
                forks.append(fork)
            proofs = prover.create_chip_proofs(dev, tasks, [alpha, beta], forks, max(1, lanes))
            for i in range(n_chips):
                chips.append((cols_all[i], proofs[i], forks[i].sample_ext()))
            for _, _, s in chips:                                     # one sample per fork back into the main transcript (prover.rs:567-570)
                tr.append_ext(s)

        _, res["chip_proofs_ms"] = timed(chip_proofs)
        for i, r in enumerate(self.LOG_ROWS):
            cols, proof, _ = chips[i]
            sel = (1, 0, (1 << r) - 3, 0, (), 0, proof.rt_main)
            jobs.append(dict(circuit_idx=i, num_vars=r, mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2,
                             max_degree=4, terms=mterms, scalars=mscalars))
        mj = prover.MainJobs(jobs)
        (claimed, msgs, rt, evals), res["batched_main_ms"] = timed(lambda: prover.prove_batched_main_constraints(dev, mj, [alpha, beta], tr, self.stream))
        # (kept for tools/dev/dbg_shard_digest.py: what the opening is asked to prove, available even when the opening fails)
        self.pre_open = dict(roots=[pcs.root()], alpha=alpha, chip_proofs=[c[1] for c in chips], msgs=msgs, rt=rt, evals=evals)
        points = [rt[:r] for r in self.LOG_ROWS]
        ev = [evals[i * (w + 1): i * (w + 1) + w] for i in range(len(self.LOG_ROWS))]
        oproof, res["open_ms"] = timed(lambda: pcs.basefold_open(points, ev, self.n_queries, self.pow_bits, tr))
        res["total_ms"] = res["commit_ms"] + res["chip_proofs_ms"] + res["batched_main_ms"] + res["open_ms"]
        res["e2e_prover_sec_for_2p20_cycles"] = res["total_ms"] / 1e3
        res["open_proof_bytes"] = int(oproof.size * 8)
        # everything a verifier needs (tests/test_gpu_flows.py replays the whole transcript with the oracle's verifiers)
        self.artifacts = dict(roots=[pcs.root()], alpha=alpha, beta=beta, chip_proofs=[c[1] for c in chips],
                              fork_samples=[c[2] for c in chips], claimed=claimed, msgs=msgs, rt=rt, evals=evals, points=points, open_evals=ev,
                              open_proof=oproof, mterms=mterms, mscalars=mscalars)
        for cols, _, _ in chips:
            for m in cols:
                m.free()
        pcs.free()
        return res

    def close(self):
        for t in self.traces:
            t.free()
        self.dev.stream_destroy(self.stream)
