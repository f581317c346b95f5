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


# ------------------------------------------------------------------------------------------------------------------
# config #4 at the REFERENCE's plan statistics (round-4 verdict, item 1).  What the monomial form of a real chip looks like
# (gkr_iop/src/gkr/layer/zerocheck_layer.rs:86-207: every out-eval expression is multiplied by its group's selector, the groups are
# summed with the alpha powers and the sum is monomialised; ceno_zkvm/src/instructions.rs:48-83: the r / w / lk / zero groups of an
# opcode chip SHARE one Prefix selector, tables/shard_ram.rs:575-578: a few chips have two or three): per chip 22..96 base columns, one
# (sometimes 2-3) selectors, and 60..250 monomials = selector x column for (nearly) every column — the records are RLCs of columns —
# plus a `selector x constant` monomial per selector, plus a tail of selector x (2..4 columns) from the zero constraints, whose columns
# come from a small cluster (the limbs and carries the arithmetic constraints tie together).
# ------------------------------------------------------------------------------------------------------------------
WIDE_SPREAD = [0, 1, 2, 2, 3, 3] + [4] * 4 + [5] * 4 + [6] * 6 + [7] * 6 + [8] * 6 + [9] * 4 + [10] * 4 + [11] * 4 + [12] * 4   # 48 chips
WIDE_WIDTHS = [22, 30, 26, 40, 34, 64, 22, 48, 80, 96, 26, 30, 56, 22, 34, 40, 96, 22, 64, 26, 48, 30, 80, 22,
               34, 56, 26, 40, 22, 96, 30, 48, 64, 22, 26, 80, 34, 40, 22, 56, 30, 96, 26, 48, 22, 64, 34, 80]


def wide_plan(c: int, w: int, n_exprs: int):
    """the monomial plan of wide chip `c`: (n_sel, terms [[mle ids]], scalars [[(coeff, [challenge ids])]], max_degree).  MLE ids: columns
    0 .. w - 1, selectors w .. w + n_sel - 1.  Deterministic in (c, w)."""
    rng = np.random.RandomState(0x51DE + 7919 * c + w)
    n_sel = 3 if c % 12 == 11 else (2 if c % 6 == 5 else 1)
    terms, scalars = [], []

    def scalar(t):
        monos = [((int(rng.randint(1, 1 << 30)), int(rng.randint(0, 1 << 30))), [2 + int(rng.randint(0, n_exprs))])]
        if t % 3 == 0:  # a column that several records share: its coefficient is a sum of challenge monomials
            monos.append(((int(rng.randint(1, 1 << 30)), 0), [int(rng.randint(0, 2)), 2 + int(rng.randint(0, n_exprs))]))
        return monos

    for si in range(n_sel):
        sel = w + si
        cols = range(w) if si == 0 else [j for j in range(w) if (j + si) % 3 == 0]
        for j in cols:                      # selector x column: the record RLCs
            terms.append([sel, j])
            scalars.append(scalar(len(terms)))
        terms.append([sel])                 # selector x constant (the constants of the records)
        scalars.append(scalar(len(terms)))
    # the zero constraints: products of 2 .. 4 columns of a cluster of ~ w / 4 columns under the first selector
    cluster = [int(x) for x in rng.choice(w, size=max(6, w // 4), replace=False)]
    n_nl = max(8, min(250 - len(terms), int(w * 0.9) + int(rng.randint(0, 12))))
    max_deg = 3
    for t in range(n_nl):
        u = rng.randint(0, 100)
        k = 2 if u < 70 else (3 if u < 92 else 4)
        if c % 4 != 0 and k == 4:           # degree 5 (selector x 4 columns) only on every fourth chip
            k = 3
        fs = [cluster[int(x)] for x in rng.randint(0, len(cluster), size=k)]  # repeated factors happen (x^2 terms)
        terms.append([w] + fs)
        scalars.append(scalar(len(terms)))
        max_deg = max(max_deg, k + 1)
    return n_sel, terms, scalars, max_deg


def wide_sizes(max_nv: int):
    return [max(1, max_nv - d) for d in WIDE_SPREAD]


def wide_batched_jobs(dev, max_nv: int = 24, n_chips: int = 48, point_of=None):
    """48 chips of 2^(max_nv - 12) .. 2^max_nv rows and 22 .. 96 columns (>= 2000 monomials over >= 1500 MLEs in ONE sumcheck).
    Returns (jobs, chips, table_elements): `chips` carries what a checker needs (columns, points, ranges, terms, scalars)."""
    jobs, chips, elems = [], [], 0
    n_exprs = 4
    for c, nv in enumerate(wide_sizes(max_nv)[:n_chips]):
        w = WIDE_WIDTHS[c % len(WIDE_WIDTHS)]
        n_sel, terms, scalars, deg = wide_plan(c, w, n_exprs)
        cols = [dev.synthetic(nv, False, 0x77000 + 131 * c + j) for j in range(w)]
        point = np.array([[(i * 7919 + 13 + c) % P, (i * 104729 + 17) % P] for i in range(nv)], dtype=np.uint64)
        sels, ranges = [], []
        for si in range(n_sel):
            n_inst = max(1, (1 << nv) - 3 - 5 * c - ((1 << nv) // 3) * si)  # the further selectors cover a shorter prefix
            sels.append((1, 0, n_inst, si, (), 0, point))
            ranges.append(n_inst)
        jobs.append(dict(num_vars=nv, mles=cols + [None] * n_sel, n_witin=w, n_fixed=0, n_structural=n_sel, selectors=sels, n_exprs=n_exprs,
                         max_degree=deg, terms=terms, scalars=scalars))
        chips.append(dict(nv=nv, w=w, n_sel=n_sel, cols=cols, point=point, n_inst=ranges, terms=terms, scalars=scalars, n_exprs=n_exprs))
        elems += (w + n_sel) << nv
    return jobs, chips, elems


def eq_form_mult_equivalents(jobs, degree: int) -> float:
    """extension-field multiplication equivalents of one batched main sumcheck in the eq-factored form (k_gen_eq row), the
    yardstick the VALU instruction counts of two plans are compared on: per chip and pair a fold is 2 per table, a monomial of nf column
    factors under its selector costs D - 2 multiply-accumulates for nf <= 1 and 2 + (nf - 1)(D - 2) (+ nf - 1 when it reaches the leading
    coefficient, nf = D - 1) otherwise, a selector group D - 1; a chip of 2^nv rows has 2^nv - 1 pairs over its rounds"""
    tot = 0.0
    for j in jobs:
        n_sel = j["n_structural"]
        first_sel = j["n_witin"] + j["n_fixed"]
        per_pair = 2.0 * len(j["mles"]) + (degree - 1) * n_sel
        for t in j["terms"]:
            nf = sum(1 for m in t if m < first_sel)
            per_pair += (degree - 2) if nf <= 1 else 2 + (nf - 1) * (degree - 2) + (nf - 1 if nf == degree - 1 else 0)
        tot += per_pair * ((1 << j["num_vars"]) - 1)
    return tot


def wide_algorithmic_bytes(max_nv: int = 24) -> float:
    """as batched_algorithmic_bytes: 40 B per base-column element, 48 B per selector element"""
    tot = 0
    for c, nv in enumerate(wide_sizes(max_nv)):
        w = WIDE_WIDTHS[c % len(WIDE_WIDTHS)]
        n_sel = 3 if c % 12 == 11 else (2 if c % 6 == 5 else 1)
        tot += (40 * w + 48 * n_sel) << nv
    return float(tot)


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
