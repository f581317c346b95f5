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


# ------------------------------------------------------------------------------------------------------------------
# metric M2 on a shard with the REFERENCE's population (round-5 verdict, item 1): the 45 RV32IM opcode circuits that have on-device
# witness generation (every opcode circuit but ECALL, ceno_zkvm/src/instructions/riscv/rv32im.rs:124-204), their row counts following an
# instruction mix (a few hot opcodes, a long tail: ~2^18 .. ~2^10 instances, NOT powers of two), the table circuits of rv32im.rs:223-230,
# 580-587 — dynamic range (2^19 rows, two structural columns: tables/range/range_impl.rs:15-52), double-u8 (2^16), AND / OR / XOR / LTU
# (2^16 rows, three FIXED columns a, b, c: tables/ops/ops_impl.rs:17-50), the program table (fixed pc + instruction fields,
# tables/program.rs) — each with ONE witness column `mlt` and one LogUp table record, and two wide circuits standing for the ECALL /
# precompile class whose witness comes from the host side.  Record and constraint EXPRESSIONS are synthetic of the right shape (the real
# ComposedConstrainSystem needs the Rust front end): r / w / lk record counts per opcode class, main constraints from `wide_plan`.
# ------------------------------------------------------------------------------------------------------------------
NO_COLUMN = 0xFFFFFFFF
# (name, call kind, mapped columns, call arguments, (reads, writes, lookups), weight in the instruction mix)
OPCODE_KINDS = [
    ("ADDI", "addi", 18, (), (3, 3, 7), 170), ("LW", "mem", 23, (0,), (4, 4, 9), 140), ("ADD", "arith", 22, (False,), (4, 4, 9), 90),
    ("SW", "mem", 23, (1,), (4, 4, 9), 80), ("BNE", "branch", 19, (True, False), (3, 3, 7), 60), ("BEQ", "branch", 19, (True, True), (3, 3, 7), 50),
    ("SLLI", "shift", 40, (True, 0), (3, 3, 16), 40), ("JAL", "jal", 13, (), (2, 2, 5), 30), ("LUI", "lui", 16, (), (2, 2, 6), 30),
    ("ANDI", "logic_i", 24, (0,), (3, 3, 9), 30), ("LBU", "load_sub", 28, (8, False), (4, 4, 12), 25), ("JALR", "jalr", 22, (), (3, 3, 9), 20),
    ("AND", "logic_r", 28, (0,), (4, 4, 11), 20), ("OR", "logic_r", 28, (1,), (4, 4, 11), 20), ("SUB", "arith", 22, (True,), (4, 4, 11), 20),
    ("BLT", "branch", 22, (False, True), (3, 3, 9), 20), ("SRLI", "shift", 40, (True, 1), (3, 3, 16), 20), ("XOR", "logic_r", 28, (2,), (4, 4, 11), 15),
    ("SLTU", "slt", 26, (False,), (4, 4, 11), 15), ("BGE", "branch", 22, (False, True), (3, 3, 9), 15), ("BLTU", "branch", 22, (False, False), (3, 3, 9), 15),
    ("BGEU", "branch", 22, (False, False), (3, 3, 9), 15), ("SB", "mem", 29, (3,), (4, 4, 12), 15), ("MUL", "mul", 22, (0,), (4, 4, 11), 15),
    ("AUIPC", "auipc", 21, (), (2, 2, 8), 10), ("LHU", "load_sub", 25, (16, False), (4, 4, 11), 10), ("SH", "mem", 24, (2,), (4, 4, 10), 10),
    ("SRAI", "shift", 40, (True, 2), (3, 3, 16), 10), ("ORI", "logic_i", 24, (1,), (3, 3, 9), 8), ("XORI", "logic_i", 24, (2,), (3, 3, 9), 6),
    ("SLTIU", "slti", 22, (False,), (3, 3, 9), 6), ("LB", "load_sub", 29, (8, True), (4, 4, 12), 5), ("LH", "load_sub", 26, (16, True), (4, 4, 11), 5),
    ("SLL", "shift", 47, (False, 0), (4, 4, 18), 5), ("SRL", "shift", 47, (False, 1), (4, 4, 18), 5), ("SLT", "slt", 26, (True,), (4, 4, 11), 4),
    ("SLTI", "slti", 22, (True,), (3, 3, 9), 4), ("MULHU", "mul", 26, (2,), (4, 4, 13), 4), ("SRA", "shift", 47, (False, 2), (4, 4, 18), 3),
    ("DIVU", "div", 39, (1,), (4, 4, 20), 3), ("REMU", "div", 39, (3,), (4, 4, 20), 3), ("MULH", "mul", 26, (1,), (4, 4, 13), 2),
    ("DIV", "div", 39, (0,), (4, 4, 20), 2), ("REM", "div", 39, (2,), (4, 4, 20), 2), ("MULHSU", "mul", 26, (3,), (4, 4, 13), 1),
]
assert len(OPCODE_KINDS) == 45
# wide circuits without device witness generation (ECALL / precompile class): (name, width, log2 rows at 2^20 cycles, (r, w, lk))
WIDE_KINDS = [("ECALL_A", 64, 12, (4, 4, 12)), ("ECALL_B", 96, 10, (6, 6, 16))]
# table circuits: (name, counters it is fed from, log2 rows, fixed columns, structural columns)
TABLE_KINDS = [("DynamicRange", "dyn", 19, 0, 2), ("DoubleU8", "du8", 16, 0, 2), ("AndTable", "and", 16, 3, 0), ("OrTable", "or", 16, 3, 0),
               ("XorTable", "xor", 16, 3, 0), ("LtuTable", "ltu", 16, 3, 0), ("Program", "fetch", None, 7, 0)]


def opcode_counts(log_cycles: int):
    """instances per opcode circuit: the mix's weights scaled to 2^log_cycles cycles in all, at least one each"""
    total = 1 << log_cycles
    wsum = sum(k[5] for k in OPCODE_KINDS)
    counts = [max(1, (total * k[5]) // wsum) for k in OPCODE_KINDS]
    counts[0] += total - sum(counts)   # the remainder goes to the hottest opcode
    assert sum(counts) == total and min(counts) >= 1
    return counts


def synthetic_step_records(n: int, fetch_base_pc: int, fetch_slots: int, offset: int = 0, seed: int = 0x57E9):
    """n StepRecords in the emulator's #[repr(C)] layout (136 bytes = 17 words each; ceno_emul/src/tracer.rs:33-60; the offsets the
    witgen kernels read: ceno_amd/csrc/witgen.hip): plausible values in every field a chip reads — cycles 4 apart, earlier previous
    cycles, pcs inside the program, register word addresses, a memory operation."""
    rng = np.random.RandomState(seed)
    i = np.arange(n, dtype=np.uint64)
    u = np.uint64
    rec = np.zeros((n, 17), dtype=np.uint64)
    cyc = u(offset) + u(4) + u(4) * i
    pc = u(fetch_base_pc) + u(4) * (rng.randint(0, fetch_slots, n).astype(np.uint64))
    nxt = pc + u(4)
    jump = rng.randint(0, 8, n) == 0
    nxt[jump] = u(fetch_base_pc) + u(4) * rng.randint(0, fetch_slots, int(jump.sum())).astype(np.uint64)

    def r32(k):
        # a mix of small and full-width operands (edge cases included)
        v = rng.randint(0, 1 << 32, k, dtype=np.uint64)
        small = rng.randint(0, 4, k) == 0
        v[small] = rng.randint(0, 1 << 12, int(small.sum())).astype(np.uint64)
        return v

    def prev(k):
        p = (rng.randint(0, 1 << 30, k).astype(np.uint64) % (cyc - u(offset))) + u(offset)
        p[rng.randint(0, 5, k) == 0] = 0  # first access of the shard
        return p

    regs = rng.randint(1, 32, (n, 3)).astype(np.uint64)
    imm = (rng.randint(-2048, 2048, n).astype(np.int64) & 0xFFFFFFFF).astype(np.uint64)
    rec[:, 0] = cyc
    rec[:, 1] = pc | (nxt << u(32))
    rec[:, 4] = u(1) | (regs[:, 0] << u(8)) | (regs[:, 1] << u(16)) | (regs[:, 2] << u(24)) | (imm << u(32))
    rec[:, 5] = u(0x01010101) << u(32)   # has_rs1, has_rs2, has_rd, has_memory_op (the Option discriminants: every chip finds its operands)
    rec[:, 16] = u(0xFFFFFFFF)           # syscall_index: none
    rec[:, 6] = ((regs[:, 0] << u(8)) >> u(2)) | (r32(n) << u(32))
    rec[:, 7] = prev(n)
    rec[:, 8] = ((regs[:, 1] << u(8)) >> u(2)) | (r32(n) << u(32))
    rec[:, 9] = prev(n)
    rec[:, 10] = ((regs[:, 2] << u(8)) >> u(2)) | (r32(n) << u(32))
    rec[:, 11] = r32(n)
    rec[:, 12] = prev(n)
    rec[:, 13] = (u(0x20000000) >> u(2)) + rng.randint(0, 1 << 20, n).astype(np.uint64) | (r32(n) << u(32))
    rec[:, 14] = r32(n)
    rec[:, 15] = prev(n)
    return rec


def _table_plan(n_cols: int, n_exprs: int, c: int):
    """main constraints of a table circuit: selector x every column (the LogUp record is an RLC of them) + selector x constant; degree 2"""
    rng = np.random.RandomState(0x7AB1E + c)
    sel = n_cols
    terms = [[sel, j] for j in range(n_cols)] + [[sel]]
    scalars = [[((int(rng.randint(1, 1 << 30)), int(rng.randint(0, 1 << 30))), [2 + int(rng.randint(0, n_exprs))])] for _ in terms]
    return terms, scalars


class ShardFlowWide:
    """BASELINE.json metric M2 ("e2e prover sec for 2^20 cycles") on a synthetic shard with the reference's population, proved the way
    ZKVMProver::create_proof does it (ceno_zkvm/src/scheme/prover.rs:319-611) with the witness PRODUCED on the device:

      step records resident in HBM (the emulator's output, upstream) ->
      witness generation, every opcode circuit's kernel writing column-major INTO the commitment's storage (ceno_prover_commit_reserve),
        all chips counting into one set of per-XCD lookup counters (ceno_hip_witgen_session_*), the table circuits' `mlt` columns built from the
        counters on the device (ceno_hip_lk_to_mlt_column) — no PCIe transfer anywhere ->
      commit_traces (witness; the FIXED commitment of the op tables and the program table is set-up, outside the time: keygen) ->
      bind the roots, two challenges -> one chip proof per circuit (~54) on forked transcripts over the lane scheduler with VRAM booking
        (scheme/scheduler.rs:231-470) -> ONE batched main-constraint sumcheck on the wide plans ->
      ONE opening of witness + fixed commitment (OpeningProver::open, scheme/hal.rs:284-294)."""

    FETCH_BASE_PC = 0x20000000

    def __init__(self, dev, prover, log_cycles: int = 20, log_blowup: int = 1, n_queries: int = 100, pow_bits: int = 16, log_program: int = None,
                 include_wide: bool = True):
        from . import api

        self.dev, self.prover, self.api = dev, prover, api
        self.log_cycles, self.log_blowup, self.n_queries, self.pow_bits = log_cycles, log_blowup, n_queries, pow_bits
        self.log_program = log_program if log_program is not None else max(4, min(17, log_cycles - 3))
        self.fetch_slots = 1 << self.log_program
        self.stream = dev.stream_create()
        self.n_exprs = 4
        # ---- the emulator's output: step records + per-circuit step indices, resident on the device ----
        n = 1 << log_cycles
        recs = synthetic_step_records(n, self.FETCH_BASE_PC, self.fetch_slots)
        words = recs.reshape(-1)
        pad = 1 << (words.size - 1).bit_length()
        self.records = dev.upload(np.concatenate([words, np.zeros(pad - words.size, dtype=np.uint64)]))
        self.n_records = n
        counts = opcode_counts(log_cycles)
        perm = np.random.RandomState(0x1D5).permutation(n).astype(np.uint32)
        self.chips = []     # dicts: name, kind, w, n_inst, nv, rows, (r, w, lk), witgen call, index buffer
        off = 0
        for k, (name, call, ncols, args, rec_shape, _wt) in enumerate(OPCODE_KINDS):
            cnt = counts[k]
            idx = np.sort(perm[off: off + cnt])   # a circuit sees its steps in execution order
            off += cnt
            iw = np.zeros(1 << max(1, ((cnt + 1) // 2 - 1).bit_length()), dtype=np.uint64)
            iw.view(np.uint32)[:cnt] = idx
            self.chips.append(dict(name=name, cls="opcode", call=call, args=args, w=ncols, n_inst=cnt, rec_shape=rec_shape, idx=dev.upload(iw)))
        if include_wide:
            for name, w, lr, rec_shape in WIDE_KINDS:
                lr_ = max(2, lr - (20 - log_cycles))
                self.chips.append(dict(name=name, cls="wide", w=w, n_inst=(1 << lr_) - (5 if lr_ >= 4 else 1), rec_shape=rec_shape))
        for name, src, lr, n_fixed, n_struct in TABLE_KINDS:
            lr_ = self.log_program if lr is None else lr
            self.chips.append(dict(name=name, cls="table", src=src, w=1, n_inst=1 << lr_, n_fixed=n_fixed, n_struct=n_struct))
        for c, ch in enumerate(self.chips):
            ch["circuit_idx"] = c
            ch["rows"] = max(2, 1 << (ch["n_inst"] - 1).bit_length())
            ch["nv"] = ch["rows"].bit_length() - 1
        # ---- lookup counters (u32, two per word of a zeroed table) ----
        self.counter_slots = {"dyn": 1 << 19, "fetch": self.fetch_slots, "du8": 1 << 16, "and": 1 << 16, "or": 1 << 16, "xor": 1 << 16, "ltu": 1 << 16}
        self.counters = {k: dev.zeros(max(1, (v // 2 - 1).bit_length()), False) for k, v in self.counter_slots.items()}
        # ---- set-up (keygen): the fixed commitment and the structural columns of the table circuits ----
        fixed_ptrs, self.fixed_src = [], []
        self.structural = {}
        for ch in self.chips:
            if ch["cls"] != "table":
                continue
            if ch["n_fixed"]:
                t = dev.synthetic((ch["rows"] * ch["n_fixed"] - 1).bit_length(), False, 0xF1D0 + ch["circuit_idx"])
                self.fixed_src.append(t)
                ch["fixed_matrix"] = len(fixed_ptrs)
                fixed_ptrs.append((t.device_ptr, ch["rows"], ch["n_fixed"]))
            if ch["n_struct"]:
                self.structural[ch["circuit_idx"]] = [dev.synthetic(ch["nv"], False, 0x57C0 + 16 * ch["circuit_idx"] + j) for j in range(ch["n_struct"])]
        self.fixed_pcs = prover.PcsData(dev, None, log_blowup, self.stream, device_ptrs=fixed_ptrs)
        dev.sync()

    # -- one opcode circuit's witness generation, writing at `wptr` (column-major, `rows` words per column) --
    def _witgen(self, ch, wptr: int, rows: int, st=None):
        api, dev = self.api, self.dev
        st = self.stream if st is None else st
        T = {k: m.device_ptr for k, m in self.counters.items()}
        w, call, a = ch["w"], ch["call"], ch["args"]
        common = (self.records.device_ptr, self.n_records, ch["idx"].device_ptr, ch["n_inst"], wptr, rows, 0, self.FETCH_BASE_PC, self.fetch_slots)
        nat = list(range(w))
        if call == "arith":
            api.witgen_arith(dev, nat + [w], a[0], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "addi":
            api.witgen_addi(dev, nat + [w], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "logic_r":
            api.witgen_logic_r(dev, nat + [w], a[0], *common, T["dyn"], T["fetch"], T[("and", "or", "xor")[a[0]]], stream=st)
        elif call == "logic_i":
            api.witgen_logic_i(dev, nat + [w], a[0], *common, T["dyn"], T["fetch"], T[("and", "or", "xor")[a[0]]], stream=st)
        elif call == "lui":
            api.witgen_lui(dev, nat + [w], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "jal":
            api.witgen_jal(dev, nat + [w], *common, T["dyn"], T["fetch"], T["du8"], T["xor"], stream=st)
        elif call == "auipc":
            api.witgen_auipc(dev, nat + [w], *common, T["dyn"], T["fetch"], T["du8"], T["xor"], stream=st)
        elif call == "jalr":
            api.witgen_jalr(dev, nat + [w], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "slt":
            api.witgen_slt(dev, nat + [w], a[0], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "slti":
            api.witgen_slti(dev, nat + [w], a[0], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "branch":
            api.witgen_branch(dev, nat + [w], a[0], a[1], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "shift":
            api.witgen_shift(dev, nat + [w], a[0], a[1], *common, T["dyn"], T["fetch"], T["du8"], T["xor"], stream=st)
        elif call == "mul":
            cols = nat[:22] + (nat[22:26] if a[0] else [NO_COLUMN] * 4) + [w]
            api.witgen_mul(dev, cols, a[0], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "div":
            api.witgen_div(dev, nat + [w], a[0], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "mem":
            api.witgen_mem(dev, nat + [w], a[0], *common, T["dyn"], T["fetch"], stream=st)
        elif call == "load_sub":
            ids = list(nat)
            cols = [ids.pop(0) for _ in range(25)]
            cols += [ids.pop(0) for _ in range(3)] if a[0] == 8 else [NO_COLUMN] * 3
            cols += [ids.pop(0)] if a[1] else [NO_COLUMN]
            api.witgen_load_sub(dev, cols + [w], a[0], a[1], *common, T["dyn"], T["fetch"], stream=st)
        else:
            raise ValueError(call)

    def generate_witness(self):
        """the witness of the whole shard, on the device, inside the commitment's storage; returns the reserved PcsData (finish() commits)"""
        dev, st = self.dev, self.stream
        pcs = self.prover.PcsData.reserve(dev, [(ch["n_inst"], ch["w"]) for ch in self.chips], self.log_blowup, st)
        for m in self.counters.values():
            m.fill_zero(st)
        dev.witgen_session_begin([(self.counters[k].device_ptr, v) for k, v in self.counter_slots.items()], st)
        # the 45 witness kernels are small (2^10 .. 2^18 rows, ~1.3 ms of device time one after the other): they go round-robin over four streams
        # (the session's per-XCD counter copies take their atomics from any stream); the tables' mlt columns wait for all of them
        if not hasattr(self, "_wit_streams"):
            self._wit_streams = [st] + [dev.stream_create() for _ in range(3)]
        try:
            for c, ch in enumerate(self.chips):
                assert pcs.rows(c) == ch["rows"]
                if ch["cls"] == "opcode":
                    self._witgen(ch, pcs.trace_ptr(c), ch["rows"], self._wit_streams[c % len(self._wit_streams)])
                elif ch["cls"] == "wide":
                    # the host side's witness arriving in place: whole power-of-two blocks of columns of the column-major matrix
                    ptr, left, blk = pcs.trace_ptr(c), ch["w"], 0
                    while left:
                        p2 = 1 << (left.bit_length() - 1)
                        m = dev.wrap(ptr, ch["nv"] + p2.bit_length() - 1, False)
                        dev.check(dev.L.ceno_hip_mle_fill_splitmix(dev.h, m.h, 0xECA11 + 97 * c + blk, 0, st))
                        m.free()
                        ptr += 8 * p2 * ch["rows"]
                        left -= p2
                        blk += 1
        finally:
            dev.witgen_session_end(st)   # (waits for the other streams by itself)
        for c, ch in enumerate(self.chips):
            if ch["cls"] == "table":
                dev.lk_to_mlt_column(self.counters[ch["src"]].device_ptr, min(ch["n_inst"], self.counter_slots[ch["src"]]), pcs.trace_ptr(c), ch["rows"], st)
        return pcs

    def _chip_mles(self, pcs, c):
        """(witness ++ fixed ++ structural tables of circuit c, n_witin, n_fixed, n_structural without selectors)"""
        ch = self.chips[c]
        cols = [pcs.witness_mle(c, j) for j in range(ch["w"])]
        if ch["cls"] != "table":
            return cols, ch["w"], 0, 0
        fixed = [self.fixed_pcs.witness_mle(ch["fixed_matrix"], j) for j in range(ch["n_fixed"])] if ch["n_fixed"] else []
        struct = self.structural.get(c, [])
        return cols + fixed + list(struct), 1, len(fixed), len(struct)

    def run(self, transcript_factory, fork_factory, lanes: int = 4) -> dict:
        dev, prover = self.dev, self.prover

        def timed(f):
            dev.sync()
            t0 = time.perf_counter()
            r = f()
            dev.sync()
            return r, (time.perf_counter() - t0) * 1e3

        self.free_last()
        res = {}
        pcs, res["witgen_ms"] = timed(self.generate_witness)
        _, res["commit_ms"] = timed(pcs.finish)
        tr = transcript_factory()
        for root in (self.fixed_pcs.root(), pcs.root()):            # fixed commitment (vk), then the witness commitment (prover.rs:343-368)
            tr.append_ext((int(root[0]), int(root[1])))
            tr.append_ext((int(root[2]), int(root[3])))
        alpha, beta = tr.sample_ext(), tr.sample_ext()
        n_chips = len(self.chips)
        mles_all, tasks = [], []
        for c, ch in enumerate(self.chips):
            mles, n_wit, n_fix, n_str = self._chip_mles(pcs, c)
            mles_all.append((mles, n_wit, n_fix, n_str))
            n_all = len(mles)
            if ch["cls"] == "table":
                # one LogUp TABLE record: numerator = mlt, denominator = an RLC of the table's fixed / structural columns
                b2 = _e2_mul(beta, beta)
                terms = [[0], [1], [2 % n_all if n_all > 2 else 1], [1, (n_all - 1)]]
                coeffs = np.array([(1, 0), beta, b2, alpha], dtype=np.uint64)
                out_terms = [[0], [1, 2, 3]]
                shape = dict(num_reads=0, num_writes=0, num_lk_tables=1, num_lk=0)
            else:
                nr, nw, nl = ch["rec_shape"]
                coeffs, terms, out_terms = record_plan(ch["w"], nr + nw + nl, alpha, beta)
                shape = dict(num_reads=nr, num_writes=nw, num_lk_tables=0, num_lk=nl)
            tasks.append(dict(circuit_idx=c, mles=mles, n_witin=n_wit, n_fixed=n_fix, n_structural=n_str, num_instances=ch["n_inst"],
                              log2_num_instances=ch["nv"], record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms, **shape))
        ct = prover.ChipTasks(tasks)
        res["chip_proof_booking_estimates_bytes_sum"] = int(sum(prover.chip_proof_estimate_bytes(t) for t in tasks))
        proofs, samples = [], []

        # every task's transcript is forked INSIDE the library from one parent and bound to the challenges and the task's words (task id, circuit
        # index, instance counts: ZKVMProver::run_chip_proofs, prover.rs:618-710); one sample per fork comes back for the main transcript
        bind_words = [(c, c, ch["n_inst"], 0) for c, ch in enumerate(self.chips)]

        def chip_proofs():
            pr, sm = prover.run_chip_proofs(dev, ct, [alpha, beta], fork_factory(), bind_words, max(1, lanes))
            proofs.extend(pr)
            samples.extend(sm)
            for s_ in samples:
                tr.append_ext(s_)

        base_used = dev.mem_info()["pool_used"]
        dev.L.ceno_hip_mem_peak(dev.h, 1)
        dev.L.ceno_hip_mem_booked_peak(dev.h, 1)
        _, res["chip_proofs_ms"] = timed(chip_proofs)
        res["chip_proofs_native_ms"] = float(getattr(prover.create_chip_proofs, "last_native_ms", 0.0))   # the C++ call inside it
        res["chip_proofs_pool_high_water_bytes"] = int(dev.L.ceno_hip_mem_peak(dev.h, 0)) - int(base_used)
        res["chip_proofs_booked_high_water_bytes"] = int(dev.L.ceno_hip_mem_booked_peak(dev.h, 0))
        jobs, plans = [], []
        for c, ch in enumerate(self.chips):
            mles, n_wit, n_fix, n_str = mles_all[c]
            n_cols = len(mles)
            if ch["cls"] == "table":
                n_sel, deg = 1, 2
                terms, scalars = _table_plan(n_cols, self.n_exprs, c)
            else:
                n_sel, terms, scalars, deg = wide_plan(c, n_cols, self.n_exprs)
            sels = []
            for si in range(n_sel):
                n_inst = max(1, ch["n_inst"] - (ch["n_inst"] // 3) * si)   # the further selectors cover a shorter prefix
                sels.append((1, 0, n_inst, n_str + si, (), 0, proofs[c].rt_main))
            jobs.append(dict(circuit_idx=c, num_vars=ch["nv"], mles=mles + [None] * n_sel, n_witin=n_wit, n_fixed=n_fix, n_structural=n_str + n_sel,
                             selectors=sels, n_exprs=self.n_exprs, max_degree=deg, terms=terms, scalars=scalars))
            plans.append(dict(terms=terms, scalars=scalars, n_sel=n_sel, sel_n_inst=[s[2] for s in sels], n_cols=n_cols))
        mj = prover.MainJobs(jobs)
        (claimed, msgs, rt, evals), res["batched_main_ms"] = timed(lambda: prover.prove_batched_main_constraints(dev, mj, [alpha, beta], tr, self.stream))
        # ---- the opening: every witness matrix, then every fixed matrix, at its circuit's prefix of the sumcheck point ----
        points, ev, off, fixed_pts, fixed_ev = [], [], 0, [], []
        for c, ch in enumerate(self.chips):
            mles, n_wit, n_fix, n_str = mles_all[c]
            points.append(rt[: ch["nv"]])
            ev.append(evals[off: off + n_wit])
            if n_fix:
                fixed_pts.append(rt[: ch["nv"]])
                fixed_ev.append(evals[off + n_wit: off + n_wit + n_fix])
            off += len(jobs[c]["mles"])
        oproof, res["open_ms"] = timed(lambda: pcs.basefold_open(points + fixed_pts, ev + fixed_ev, self.n_queries, self.pow_bits, tr, more_commits=[self.fixed_pcs]))
        res["total_ms"] = res["witgen_ms"] + res["commit_ms"] + res["chip_proofs_ms"] + res["batched_main_ms"] + res["open_ms"]
        res["batched_main_native_ms"] = float(getattr(prover.prove_batched_main_constraints, "last_native_ms", 0.0))   # (the C++ calls inside the phases)
        res["open_native_ms"] = float(getattr(prover.PcsData, "last_open_native_ms", 0.0))
        res["e2e_prover_sec_for_2p20_cycles"] = res["total_ms"] / 1e3
        res["open_proof_bytes"] = int(oproof.size * 8)
        res["n_chips"], res["lanes"] = n_chips, lanes
        self.artifacts = dict(roots=[self.fixed_pcs.root(), pcs.root()], alpha=alpha, beta=beta, chip_proofs=proofs, fork_samples=samples, claimed=claimed,
                              msgs=msgs, rt=rt, evals=evals, points=points + fixed_pts, open_evals=ev + fixed_ev, open_proof=oproof, plans=plans,
                              tasks=tasks, jobs=jobs)
        self.last_pcs = pcs   # (tests read the generated witness before free_last())
        return res

    def free_last(self):
        a = getattr(self, "artifacts", None)
        if a:
            for j in a["jobs"]:
                for m in j["mles"]:
                    if m is not None and getattr(m, "_parent", None) is not None:
                        m.free()
        if getattr(self, "last_pcs", None) is not None:
            self.last_pcs.free()
            self.last_pcs = None

    def population(self) -> dict:
        """what the shard holds (for the bench line)"""
        by = {}
        for ch in self.chips:
            d = by.setdefault(ch["cls"], dict(chips=0, instances=0, padded_rows=0, cells=0, widths=set()))
            d["chips"] += 1
            d["instances"] += ch["n_inst"]
            d["padded_rows"] += ch["rows"]
            d["cells"] += ch["rows"] * ch["w"]
            d["widths"].add(ch["w"])
        for d in by.values():
            d["widths"] = [min(d["widths"]), max(d["widths"])]
        return by

    def close(self):
        self.free_last()
        self.fixed_pcs.free()
        for t in self.fixed_src:
            t.free()
        for ms in self.structural.values():
            for m in ms:
                m.free()
        for m in self.counters.values():
            m.free()
        for ch in self.chips:
            if "idx" in ch:
                ch["idx"].free()
        self.records.free()
        for s_ in getattr(self, "_wit_streams", [])[1:]:
            self.dev.stream_destroy(s_)
        self.dev.stream_destroy(self.stream)
