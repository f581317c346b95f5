#!/usr/bin/env python3
"""bench.py — sumcheck + MLE-fold throughput (Goldilocks-ext mults/sec) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched through
torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

A "step" = one complete sumcheck of prod_{k<3} f_k over synthetic GoldilocksExt2 tables already
resident in HBM: begin, n rounds (each: fused fold+accumulate pass, message to the host transcript,
challenge back), finish.  N=1 workload: BASELINE.json config "sumcheck, 3 MLEs, Goldilocks-ext2" at the
size the metric is quoted on, nv=26 (3 x 2^26 x 16 B = 3.2 GB).  N>1: STRONG scaling is the headline — the nv=26
hypercube split over the N ranks by its top log2 N bits (BASELINE's metric is quoted at nv=26; config #4 is "nv=26 sharded
over 8") — and the weak-scaling run (a 2^26 shard per rank of a 2^(26+log2 N) hypercube) is timed beside it and reported under
`weak_scaling`; one exchange of the d partial evaluations per round.

Algorithmic work: ext mults = d^2 (2^n - 1), d = 3; bytes = 3*d*16*2^n.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 3          # tables in the product == degree
SEED0 = 0xCE10
TR_SEED = 0xF5


def csrc_sha16() -> str:
    """what a committed counters file is valid for: the kernel sources (ceno_amd/csrc/*, file names and contents)"""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ceno_amd", "csrc", "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_valu_counters():
    """(counters, source path, stale): the committed --pmc pass (tools/r06_valu_counters.sh -> profiles/r*_valu_counters.json).  The file is stamped
    with the hash of the kernel sources it was measured on; when the tree's kernels differ, the instruction counts are NOT used (valu_frac null,
    "stale_counters": true) — a count from another build times this run's milliseconds would be a wrong fraction beside live numbers."""
    import glob

    vc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_counters.json")))
    if not vc:
        return {}, None, False
    d = json.load(open(vc[-1]))
    stale = d.get("csrc_sha16") != csrc_sha16()
    return d, os.path.relpath(vc[-1], ROOT), stale


def valu_headline(kernel_ms_per_sumcheck: float, n_local: int) -> dict:
    """VALU-issue fraction of the dense kernel on THIS run's kernel time: wave-level SQ_INSTS_VALU of one nv = 26 sumcheck from the committed
    --pmc pass x 4.3 cycles per instruction / (1024 SIMDs x 2.4 GHz x time); null when the counters are stale or for another size"""
    d, src, stale = load_valu_counters()
    vc = d.get("sumcheck_nv26")
    if not vc or n_local != 26 or kernel_ms_per_sumcheck <= 0:
        return {"valu_frac": None}
    if stale:
        return {"valu_frac": None, "stale_counters": True, "valu": {"source": src}}
    cpi, simds, ghz = d.get("cycles_per_valu_inst", 4.3), d.get("simds", 1024), d.get("clock_ghz", 2.4)
    frac = vc["SQ_INSTS_VALU"] * cpi / (simds * ghz * 1e9 * kernel_ms_per_sumcheck * 1e-3)
    out = {"valu_frac": frac, "valu": {"SQ_INSTS_VALU_per_sumcheck": vc["SQ_INSTS_VALU"], "cycles_per_inst": cpi, "simds": simds, "clock_ghz": ghz, "source": src}}
    if "fold_rounds_valu_issue_frac" in vc:
        out["valu"]["fold_rounds_valu_issue_frac"] = vc["fold_rounds_valu_issue_frac"]
    return out


def cpu_baseline(nv: int = 26):
    """the oracle's OpenMP fused sumcheck (a port, not the Rust/rayon reference binary) on the host cores,
    in a child process with a clean OpenMP environment"""
    import subprocess

    env = {k: v for k, v in os.environ.items() if not k.startswith(("OMP_", "GOMP_", "KMP_", "MKL_"))}
    env["OMP_PROC_BIND"] = "false"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), str(nv)], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError(out.stderr[-300:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def extra_measurements(dev, prover, new_transcript, transcript_name, reps: int = 3):
    """BASELINE.json configs #2, #3 and #4 at their own sizes (best of `reps`, wall clock around synchronous host entries,
    inputs resident in HBM), each with its algorithmic bytes against the 8 TB/s HBM peak"""
    from ceno_amd import synthetic

    # VALU-bound extras: wave-level VALU instruction counts of ONE run of the workload from the committed --pmc pass (tools/r05_valu_counters.sh ->
    # profiles/r*_valu_counters.json); valu_frac = SQ_INSTS_VALU x measured cycles per instruction / (SIMDs x clock x time of THIS run)
    valu_counters, valu_source, valu_stale = load_valu_counters()

    def roof(alg_bytes, ms, note=None, valu_key=None):
        gbps = alg_bytes / (ms * 1e-3) / 1e9
        r = {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0, "algorithmic_bytes": alg_bytes}
        if note:
            r["note"] = note
        vc = valu_counters.get(valu_key) if valu_key else None
        if valu_key:
            r["bound"] = "valu"  # integer-ALU issue, not HBM: `frac` stays the HBM fraction of the algorithmic bytes (the contract's field)
            r["hbm_frac"] = r["frac"]
        if vc and valu_stale:
            r["valu_frac"] = None
            r["stale_counters"] = True
            r["valu"] = {"source": valu_source, "note": "measured on other kernel sources than this tree's (csrc_sha16 differs): re-run tools/r06_valu_counters.sh"}
        elif vc:
            cpi, simds, ghz = valu_counters.get("cycles_per_valu_inst", 4.3), valu_counters.get("simds", 1024), valu_counters.get("clock_ghz", 2.4)
            insts = vc["SQ_INSTS_VALU"]
            frac = insts * cpi / (simds * ghz * 1e9 * ms * 1e-3)
            r["valu"] = {"SQ_INSTS_VALU": insts, "cycles_per_inst": cpi, "simds": simds, "clock_ghz": ghz, "source": valu_source}
            if valu_key == "chip_flow.commit":
                # Poseidon2's mix holds full-rate 32-bit adds and moves (2.3-2.7 cycles, profiles/r01_valu_issue_rates.txt): the 4.3-cycle
                # figure of the multiply / carry instructions overstates it — at 4.0 cycles per instruction the count already fills the SIMDs
                r["valu"]["note"] = ("instruction mix includes full-rate adds / moves: insts x 4.3 cycles exceeds the SIMD time of the phase "
                                     f"({frac:.2f}); the VALU port is saturated (profiles/r04_merkle_sq_pmc.json)")
                r["valu_frac"] = None
            else:
                r["valu_frac"] = frac
        return r

    def best_of(f):
        best = None
        for _ in range(reps):
            dev.sync()
            t0 = time.perf_counter()
            f()
            dev.sync()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        return best

    out = {"transcript": transcript_name}
    # config #2: single sumcheck instance, 3 ext MLEs x nv=22
    m22 = [dev.synthetic(22, True, SEED0 + j) for j in range(K)]
    one = np.array([[1, 0]], dtype=np.uint64)
    ms = best_of(lambda: prover.sumcheck_prove(dev, m22, one, [list(range(K))], 22, K, new_transcript()))
    out["nv22"] = {"workload": "config #2: single sumcheck instance, 3 ext MLEs x nv=22", "ms": ms,
                   "ext_mults_per_s": K * K * ((1 << 22) - 1) / (ms * 1e-3), "roofline": roof(3 * K * 16 * (1 << 22), ms)}
    for m in m22:
        m.free()
    # independent nv = 26 sumchecks IN FLIGHT (one host thread + stream each, as the chip scheduler runs chip proofs): the small rounds of
    # one instance overlap the streaming rounds of another.  Reported beside the headline, which stays one instance at a time.
    import threading
    flight = {}
    for n_inst in (1, 2, 3):
        insts = [[dev.synthetic(26, True, SEED0 + 100 * (t + 1) + j) for j in range(K)] for t in range(n_inst)]
        streams = [dev.stream_create() for _ in range(n_inst)]
        per = 4

        calls = [[] for _ in range(n_inst)]

        def work(t, reps=per):
            for _ in range(reps):
                c0 = time.perf_counter()
                prover.sumcheck_prove(dev, insts[t], one, [list(range(K))], 26, K, new_transcript(), stream=streams[t])
                calls[t].append((time.perf_counter() - c0) * 1e3)

        # warm-up of EVERY instance on its own stream, concurrently: the first sumcheck on a new stream takes its 2.4 GB of working buffers
        # from the driver inside what used to be the timed region.  Two boxes (the round-4 driver run and one round-5 run) reported two
        # instances SLOWER than one (3.3 / 5.5 vs 2.9 ms) — a constant ~20 ms on top of the expected total, whatever the instance count;
        # the dev boxes never showed it (tools/dev/flight_ab.py: 2.92 / 2.75 / 2.68 ms, hipMalloc 250 us each).  Steady state is what the
        # extra reports; the slowest and fastest single call are reported so that a one-off stall is visible as such.
        ths = [threading.Thread(target=work, args=(t, 1)) for t in range(n_inst)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dev.sync()
        for c_ in calls:
            c_.clear()
        # Three timed repetitions, the best one reported and all of them listed.  What the slow ones are (profiles/r05_in_flight_stall_trace.txt):
        # a pause of ~10 ms that hits EVERY concurrent call at the same moment (15.38 | 15.36 ms in the same position of both threads, 18.2 | 18.0 |
        # 17.8 of all three) with no allocation or other driver call of this process anywhere near it — a device-wide event outside the
        # library, seen once per few hundred milliseconds of GPU time on some boxes and never on others.
        runs = []
        for _rep in range(3):
            for c_ in calls:
                c_.clear()
            ths = [threading.Thread(target=work, args=(t,)) for t in range(n_inst)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dev.sync()
            dt_ = time.perf_counter() - t0
            flat = [x for c_ in calls for x in c_]
            runs.append((dt_, min(flat), max(flat)))
        dt, cmin, cmax = min(runs)
        flight[str(n_inst)] = {"ms_per_sumcheck": dt / (per * n_inst) * 1e3, "ext_mults_per_s": K * K * ((1 << 26) - 1) * per * n_inst / dt,
                               "call_ms_min": cmin, "call_ms_max": cmax,
                               "repetitions_ms_per_sumcheck": [round(r_[0] / (per * n_inst) * 1e3, 4) for r_ in runs],
                               "repetitions_call_ms_max": [round(r_[2], 3) for r_ in runs]}
        for row in insts:
            for m in row:
                m.free()
        for st_ in streams:
            dev.stream_destroy(st_)
    out["nv26_instances_in_flight"] = dict(flight, workload="independent nv=26 sumchecks (3 ext MLEs each) proved concurrently, one host thread and stream "
                                                               "per instance; aggregate throughput")
    # config #4 on one GPU: batched main-constraint sumcheck over 24 chips, at the config's stated size (max_nv = 26) and at 24
    for max_nv, key in ((26, "batched_main_nv26"), (24, "batched_main")):
        jobs, elems = synthetic.batched_jobs(dev, max_nv, 12)
        mj = prover.MainJobs(jobs)  # the C view of the job list, marshalled once (a Rust caller hands the structs over directly)
        ms = best_of(lambda: prover.prove_batched_main_constraints(dev, mj, [(11, 22), (33, 44)], new_transcript()))
        out[key] = {"workload": f"config #4{' shape' if max_nv != 26 else ''}: prove_batched_main_constraints, 24 chips of {max_nv - 10}..{max_nv} "
                                "variables, 12 base columns + selector, 16 terms of degree <= 4 each", "ms": ms, "table_elements": elems,
                    "roofline": roof(synthetic.batched_algorithmic_bytes(max_nv, 12), ms, "integer-ALU bound: DESIGN.md section 3", valu_key=key)}
        for j in jobs:
            for m in j["mles"]:
                if m is not None:
                    m.free()
        del mj, jobs
    # config #4 at the REFERENCE's plan statistics (zerocheck_layer.rs:86-207, instructions.rs:48-83): 48 chips of 2^12..2^24 rows, 22..96 base
    # columns, 1..3 selectors, 42..250 monomials each (selector x column for every column, selector x constant, products of 2..4 columns)
    os.environ["CENO_HIP_PLAN_REPORT"] = "1"
    try:
        jobs, _chips, elems = synthetic.wide_batched_jobs(dev, 24)
        mj = prover.MainJobs(jobs)
        before = dev.L.ceno_hip_stat_eq_launches(dev.h)
        ms = best_of(lambda: prover.prove_batched_main_constraints(dev, mj, [(11, 22), (33, 44)], new_transcript()))
        launches = int(dev.L.ceno_hip_stat_eq_launches(dev.h) - before) // reps
        classes = json.loads(dev.L.ceno_hip_plan_report(dev.h).decode() or "[]")
        deg = max(j["max_degree"] for j in jobs)
        mult_eq = synthetic.eq_form_mult_equivalents(jobs, deg)
        out["batched_main_wide"] = {
            "workload": "config #4 at the reference's plan statistics: prove_batched_main_constraints, 48 chips of 2^12..2^24 rows x 22..96 base columns, "
                        "1..3 Prefix selectors, 42..250 monomials per chip (selector x column for every column, selector x constant, products of 2..4 columns)",
            "ms": ms, "chips": len(jobs), "tables": sum(len(j["mles"]) for j in jobs), "monomials": sum(len(j["terms"]) for j in jobs),
            "max_degree": deg, "table_elements": elems, "eq_launches_per_sumcheck": launches, "ext_mult_equivalents": mult_eq,
            "mult_equivalents_per_s": mult_eq / (ms * 1e-3),
            "classes_of_2p16_rows_and_more_on_eq_factored_kernel": all(c["path"] == "eq-factored" for c in classes if c["num_vars"] >= 16),
            "classes": [{k: c[k] for k in ("num_vars", "tables", "terms", "path", "components", "tables_staged", "pairs_per_tile_log2")} for c in classes],
            "roofline": roof(synthetic.wide_algorithmic_bytes(24), ms, "integer-ALU bound: DESIGN.md section 3", valu_key="batched_main_wide")}
        for j in jobs:
            for m in j["mles"]:
                if m is not None:
                    m.free()
        del mj, jobs, _chips
    finally:
        os.environ.pop("CENO_HIP_PLAN_REPORT", None)
    # config #3: ADD-shaped chip, 2^20 rows x 22 columns, commit -> chip proof -> main constraints -> open
    flow = synthetic.ChipFlow(dev, prover, 20, 22)
    best = None
    for _ in range(reps + 2):  # latency-bound flows on a shared host: best of five (a neighbour's burst costs a whole repetition)
        r = flow.run(new_transcript)
        if best is None or r["total_ms"] < best["total_ms"]:
            best = r
    ab = flow.algorithmic_bytes()
    flow.close()
    best["workload"] = "config #3: ADD-shaped chip, 2^20 rows x 22 base columns, blow-up 2, 100 queries, 16-bit proof of work"
    best["roofline"] = roof(sum(ab.values()), best["total_ms"], "commit is integer-ALU bound (Poseidon2): DESIGN.md section 3")
    best["roofline_by_phase"] = {k: roof(v, best[f"{k}_ms"], valu_key=("chip_flow.commit" if k == "commit" else None)) for k, v in ab.items()}
    out["chip_flow"] = best
    # metric M2 shape: one synthetic shard of 2^20 cycles through the whole create_proof flow
    shard = synthetic.ShardFlow(dev, prover)
    fork = (lambda: prover.Transcript.poseidon2(b"fork")) if transcript_name == "poseidon2" else (lambda: prover.Transcript.stub(0xF0))
    bs, by_lanes = None, {}
    for lanes in (1, 4, 8):  # chip proofs serially, then on 4 and 8 lanes of the C++ scheduler (ceno_prover_create_chip_proofs: context-owned lane streams)
        for _ in range(reps + 2):
            r = shard.run(new_transcript, fork, lanes=lanes)
            by_lanes[lanes] = min(by_lanes.get(lanes, 1e30), r["total_ms"])
            if bs is None or r["total_ms"] < bs["total_ms"]:
                bs = dict(r, chip_proof_lanes=lanes)
    shard.close()
    bs["total_ms_by_chip_proof_lanes"] = by_lanes
    bs["workload"] = ("metric M2 shape: synthetic shard, 2^20 cycles over 8 ADD-shaped chips of 2^19..2^13 rows x 22 columns: commit all traces, "
                      "8 chip proofs (tower relation) on forked transcripts, one batched main sumcheck, one Basefold opening; witness generation "
                      "and the emulator are upstream and excluded")
    out["shard_e2e"] = bs
    # metric M2 on a shard with the REFERENCE's population (rv32im.rs:124-230,580-587): 45 opcode circuits with on-device witness generation
    # writing into the commitment's storage, 2 wide circuits, 7 table circuits (mlt from the device lookup counters), a fixed commitment
    # opened beside the witness commitment; chip proofs on 1 / 4 / 8 / 16 requested lanes (the scheduler runs at most 8 at once for such a batch)
    bw = None
    try:  # (one extra must not take the others down)
        wide = synthetic.ShardFlowWide(dev, prover)
        bw, w_lanes = None, {}
        for lanes in (1, 4, 8, 16):
            bl = None
            for _ in range(reps if lanes > 1 else 2):
                r = wide.run(new_transcript, fork, lanes=lanes)
                if bl is None or r["total_ms"] < bl["total_ms"]:
                    bl = r
            w_lanes[str(lanes)] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in bl.items()
                                   if k.endswith("_ms") or k.endswith("_bytes") or k.endswith("_bytes_sum")}
            if bw is None or bl["total_ms"] < bw["total_ms"]:
                bw = dict(bl, chip_proof_lanes=lanes)
        # the same shard with every chip's tower proof on its own lane (round 5's path: CENO_TOWER_COHORT_LAYERS=0) — what the cohort layers
        # (host/cohort.cpp: records, towers and tower layers 6..19 of all 54 chips in shared launches) are measured against
        prev = os.environ.get("CENO_TOWER_COHORT_LAYERS")
        os.environ["CENO_TOWER_COHORT_LAYERS"] = "0"
        try:
            bn = None
            for _ in range(reps):
                r = wide.run(new_transcript, fork, lanes=8)
                if bn is None or r["total_ms"] < bn["total_ms"]:
                    bn = r
        finally:
            if prev is None:
                os.environ.pop("CENO_TOWER_COHORT_LAYERS", None)
            else:
                os.environ["CENO_TOWER_COHORT_LAYERS"] = prev
        wide.free_last()
        pop = wide.population()
        wide.close()
        bw["per_chip_tower_proofs_8_lanes"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in bn.items() if k.endswith("_ms")}
        bw["by_chip_proof_lanes"] = w_lanes
        bw["population"] = pop
        bw["workload"] = ("metric M2 on a shard with the reference's population: 2^20 cycles over 45 opcode circuits (13..47 columns, ~2^10..~2^18 instances "
                          "following an instruction mix, witness generated ON THE DEVICE from resident step records straight into the commitment's storage), "
                          "2 wide circuits (64 / 96 columns), 7 table circuits of 2^16..2^19 rows (mlt from the device lookup counters; 3-7 fixed columns in a "
                          "FIXED commitment, or 2 structural columns): witgen -> commit -> 54 chip proofs (records, towers and the middle tower layers of all chips in "
                          "shared launches — cohorts: layers 6..19 —, the larger layers on the lane scheduler; VRAM booked for the phase) -> one "
                          "batched main sumcheck on the wide plans (degree <= 5) -> one opening of witness + fixed commitment; the emulator is upstream "
                          "and excluded, the fixed commitment is set-up (keygen)")
        out["shard_e2e_wide"] = bw
    except Exception as e:  # noqa: BLE001
        out["shard_e2e_wide"] = {"error": f"{type(e).__name__}: {e}"}
        bw = None
    # config #1 SHAPE (the reference's own CPU-runnable case is the fibonacci program at 2^10 steps; its emulator and opcode circuits are upstream
    # of the path and not here): the same create_proof flow on 2^10 synthetic cycles — what a proof costs when nothing but latency is left
    small = synthetic.ShardFlow(dev, prover, log_rows=(9, 8, 7, 7))
    bsm = None
    for _ in range(reps + 2):
        r = small.run(new_transcript, fork, lanes=4)
        if bsm is None or r["total_ms"] < bsm["total_ms"]:
            bsm = r
    small.close()
    bsm = {k: v for k, v in bsm.items() if k.endswith("_ms") or k == "open_proof_bytes"}
    bsm["workload"] = ("config #1 shape only: 2^10 synthetic cycles over 4 ADD-shaped chips of 2^9..2^7 rows x 22 columns through commit, chip proofs "
                       "(4 lanes), batched main sumcheck and opening; NOT the fibonacci guest (emulator / opcode circuits are outside the path)")
    out["small_shard_e2e"] = bsm
    out["chip_flow_ms"], out["batched_main_ms"], out["nv22_ms"] = best["total_ms"], out["batched_main"]["ms"], out["nv22"]["ms"]
    out["batched_main_nv26_ms"] = out["batched_main_nv26"]["ms"]
    out["batched_main_wide_ms"] = out["batched_main_wide"]["ms"]
    # metric M2: from the shard with the reference's population; the eight-chip shard of rounds 3-5 beside it
    out["shard_e2e_sec"] = bw["e2e_prover_sec_for_2p20_cycles"] if bw else None
    out["shard_e2e_8chips_sec"] = bs["e2e_prover_sec_for_2p20_cycles"]
    return out


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process (what `python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N` would do: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, the same
    argv), let them write to the inherited stdout / stderr (rank 0 prints the ONE JSON line), and return the first non-zero exit code.  The
    parent never initialises the GPU.  A rank that dies takes the others along after a grace period (exact PIDs, no patterns)."""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port,
                CENO_BENCH_SELF_LAUNCHED="1")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.setdefault("OMP_NUM_THREADS", "1")  # what torch.distributed.run sets for its workers
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0"))
             for r in range(n)]
    rc, deadline = 0, None
    live = list(procs)
    while live:
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 128 - c
                deadline = time.time() + float(os.environ.get("CENO_BENCH_PEER_GRACE_S", "30"))
                print(f"bench.py: rank {procs.index(p)} exited with {c}; the other ranks get "
                      f"{os.environ.get('CENO_BENCH_PEER_GRACE_S', '30')} s to leave", file=sys.stderr)
        if deadline is not None and time.time() > deadline:
            for p in live:
                p.kill()
            deadline = time.time() + 1e9
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nv", type=int, default=26, help="variables of the hypercube the metric is quoted on (strong scaling: of the WHOLE "
                                                       "hypercube, split over the ranks; weak scaling: of every rank's shard)")
    ap.add_argument("--scaling", choices=["strong", "weak", "both"], default="both",
                    help="N > 1: strong = the nv-variable hypercube split over the ranks (BASELINE's metric is quoted at nv=26: the headline), "
                         "weak = an nv-variable shard per rank; both (default) times both and reports the strong one as `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the config #2 / #3 / #4 shaped measurements reported under `extra`")
    ap.add_argument("--transcript", choices=["poseidon2", "stub"], default="poseidon2",
                    help="Fiat-Shamir transcript of the timed steps (the other one is timed too and reported beside it)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  Nothing here has touched the GPU (no torch, no HIP, no
        # library of ours imported yet), and the ranks are CHILD processes — never an exec, never a restart of a process that holds a device.
        sys.exit(self_launch(args.gpus))
    if os.environ.get("CENO_BENCH_FAIL_RANK") and os.environ.get("CENO_BENCH_FAIL_RANK") == os.environ.get("RANK"):  # tests: a rank that dies
        sys.exit(7)
    # torch is the multi-rank plumbing (torch.distributed over RCCL, its device tensors for the flags): one rank needs none of it, and on a
    # box whose image is cold the import alone has cost minutes (tests/conftest.py).  CENO_BENCH_IMPORT_TORCH=1 imports it anyway.
    torch = None
    if world > 1 or os.environ.get("CENO_BENCH_IMPORT_TORCH") == "1":
        import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("CENO_BENCH_SINGLE_DEVICE"):  # dry run of the multi-rank flow on a 1-GPU box: all ranks on cuda:0, gloo
            local_rank = 0
            torch.cuda.set_device(0)
            dist_mod.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        dist = dist_mod
    elif torch is not None:
        torch.cuda.set_device(local_rank)
    # everything imported so far (torch: ~10^6 objects) moves to the collector's permanent generation: a collection that falls into an
    # extra's timing loop or a worker thread later only walks what the benchmark itself allocated
    gc.collect()
    gc.freeze()

    from ceno_amd import Device
    from ceno_amd import dist as cdist
    from ceno_amd import prover

    dev = Device(local_rank)
    from ceno_amd import goldens
    poseidon2_pinned = goldens.install(dev)  # reference constants when tests/golden/ref_goldens.json exists (README "Closing parity")
    log_w = world.bit_length() - 1
    tdev = (f"cuda:{local_rank}" if dist is not None and dist.get_backend() == "nccl" else "cpu")

    if torch is not None:
        device_synchronize = torch.cuda.synchronize
    else:  # the same call underneath: hipDeviceSynchronize on the current device (Device(local_rank) selected it)
        import ctypes

        _hip = None
        for _name in ("/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"):
            try:
                _hip = ctypes.CDLL(_name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("bench.py: libamdhip64.so not found")

        def device_synchronize():
            rc = _hip.hipDeviceSynchronize()
            if rc != 0:
                raise RuntimeError(f"hipDeviceSynchronize failed: {rc}")

    def barrier():
        if dist is not None:
            dist.barrier()
        device_synchronize()  # torch.cuda.synchronize() with ranks; hipDeviceSynchronize directly with one
        dev.sync()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ONE = np.array([[1, 0]], dtype=np.uint64)
    TERMS = [list(range(K))]
    # Fiat-Shamir on the host between every two rounds: the Poseidon2 duplex challenger (what the reference's BasicTranscript
    # is; 2 permutations per round on the critical path) by default, the SplitMix stub for comparison
    factories = {"poseidon2": lambda: prover.Transcript.poseidon2(b"riscv"), "stub": lambda: prover.Transcript.stub(TR_SEED)}
    other = "stub" if args.transcript == "poseidon2" else "poseidon2"
    names = {"shm": "host shared-memory exchange of the d partial evaluations per round from the C++ host loop",
             "rccl": "ncclAllGather of the d partial evaluations per round on the kernels' stream from the C++ host loop (RCCL over xGMI)",
             "torch": "torch.distributed all_gather (python loop)"}
    stream = dev.stream_create() if world > 1 else None
    comms = {}        # exchange kind -> communicator, created once and shared by the strong and the weak run
    comm_info = {}    # e.g. the rank count RCCL itself reports

    def measure(n_local: int, kinds=None) -> dict:
        """all the timings of one hypercube size: every rank holds shard `rank` (2^n_local elements per table) of a
        2^(n_local + log2 world) hypercube.  kinds: the exchanges to validate and time (N > 1)"""
        n_total = n_local + log_w
        # shard `rank` of table j: words [rank * 2 * 2^n_local, ...) of the SplitMix stream seeded SEED0 + j
        mles = [dev.synthetic(n_local, True, SEED0 + j, word_offset=rank * 2 * (1 << n_local)) for j in range(K)]
        dev.sync()
        tr = {"new": factories[args.transcript]}
        cur = {"comm": None}

        def step_torch():  # per-round all-gather issued from Python through torch.distributed (RCCL underneath)
            eng = cdist.HipShardEngine(dev, mles)
            return cdist.sharded_sumcheck_prove(eng, n_total, K, tr["new"](), dist=dist, world=world, rank=rank)

        def step_native():  # the same protocol driven from C++ (host/dist.cpp) over the communicator in `cur`
            return prover.dist_sumcheck_prove(dev, cur["comm"], mles, ONE, TERMS, n_total, K, tr["new"](), stream)

        def step_single():
            return prover.sumcheck_prove(dev, mles, ONE, TERMS, n_total, K, tr["new"]())

        out = {"n_local": n_local, "n_total": n_total, "collective": "none", "collective_ms": {}, "validated": []}
        step = step_single
        if world > 1:
            # Three drivers of the same protocol; each candidate must reproduce the torch.distributed path's proof on every rank
            # before it is timed:
            #   shm  : C++ loop, per-round partials exchanged through host shared memory (the messages are in host memory
            #          anyway for the transcript) — no device collective on the round path
            #   rccl : C++ loop, ncclAllGather on the kernels' HIP stream + early gather of small shards
            #   torch: Python loop over torch.distributed all_gather (RCCL underneath)
            try:
                reference = step_torch()
            except Exception as e:  # without it a candidate is accepted when it runs on every rank (its proof is replicated by construction)
                print(f"bench.py: torch.distributed reference path failed on rank {rank}: {e}", file=sys.stderr)
                reference = None
            order = list(kinds) if kinds is not None else [x for x in os.environ.get("CENO_BENCH_EXCHANGE", "shm,rccl").split(",") if x]
            ok_kinds = []
            for kind in order:
                ok = 1
                try:
                    if kind not in comms:
                        comms[kind] = prover.ShmComm(world, rank, dist) if kind == "shm" else prover.RcclComm(world, rank, dist)
                        if kind == "rccl":
                            comm_info["rccl_ranks"] = comms[kind].nranks()
                    cur["comm"] = comms[kind]
                    got = step_native()
                    ok = 1 if reference is None or all(np.array_equal(x, y) for x, y in zip(got, reference)) else 0
                except Exception as e:  # keep looking
                    print(f"bench.py: {kind} exchange unavailable on rank {rank}: {e}", file=sys.stderr)
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=tdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    ok_kinds.append(kind)

            def time_steps(fn, n):
                for _ in range(min(2, args.warmup + 1)):
                    fn()
                barrier()
                t_ = time.perf_counter()
                for _ in range(n):
                    fn()
                barrier()
                return max_over_ranks(time.perf_counter() - t_) / n * 1e3

            for kind in ok_kinds:
                cur["comm"] = comms[kind]
                out["collective_ms"][kind] = time_steps(step_native, max(2, args.steps // 2))
            out["validated"] = sorted(ok_kinds)
            # north_star names the RCCL collective: when it validates it carries the headline and the shared-memory exchange (an
            # optimisation of the same protocol: the partials are in host memory anyway) is reported beside it; CENO_BENCH_HEADLINE
            # = fastest picks whichever exchange measured fastest instead
            pick = None
            if ok_kinds:
                fastest = min(ok_kinds, key=lambda k_: out["collective_ms"][k_])
                pick = "rccl" if ("rccl" in ok_kinds and os.environ.get("CENO_BENCH_HEADLINE", "rccl") != "fastest") else fastest
                out["fastest_exchange"] = fastest
            if pick is None:
                out["collective_ms"]["torch"] = time_steps(step_torch, max(2, args.steps // 2))
                step, out["collective"] = step_torch, names["torch"]
            else:
                cur["comm"] = comms[pick]
                step = step_native
                out["collective"] = names[pick] + (" (checked against the torch.distributed path)" if reference is not None
                                                   else " (unchecked: reference path failed)")
            out["headline_exchange"] = pick or "torch"

        for _ in range(args.warmup):
            step()
        # an UNINSTRUMENTED pass of the same steps first (reported as ms_per_step_uninstrumented: what the event recording costs) ...
        gc.collect()
        gc.disable()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        out["dt_plain"] = max_over_ranks(time.perf_counter() - t0)
        # ... then the TIMED REGION: exactly `steps` steps, with HIP events on the launch stream around every launch of the dominant kernel
        # (prof mode 2: the sumchecks stay pipelined, exactly as in the pass above; a queued round ends when its finishing workgroup has the
        # next challenge, so the events' sum is part of — never more than — the wall time of this same region: the roofline numbers and
        # `value` come from ONE pass).  The interpreter's cyclic garbage collector stays out of it (as in `timeit`): with torch imported a
        # full collection walks ~10^6 objects and takes ~35 ms — ten sumchecks — whenever the allocation count of the process happens to
        # cross its threshold (measured: it fell into the timed region from 24 steps on, 2.8 -> 4.0 ms per step).
        dev.prof_enable(2)
        dev.prof_reset()
        barrier()
        t0 = time.perf_counter()
        marks = [t0]
        for _ in range(args.steps):
            step()
            marks.append(time.perf_counter())  # (a step returns with its final evaluations on the host: the stamps cost ~0.1 us each)
        barrier()
        out["dt"] = max_over_ranks(time.perf_counter() - t0)
        # the single steps of THIS rank, so that a one-off pause of the device inside the region (profiles/r05_in_flight_stall_trace.txt: ~10 ms, some
        # boxes) is visible as such next to the mean that `value` is made of
        per = sorted((b_ - a_) * 1e3 for a_, b_ in zip(marks, marks[1:]))
        out["step_ms"] = {"min": per[0], "median": per[len(per) // 2], "max": per[-1]} if per else {}
        gc.enable()
        out["kernel_ms"], out["launches"], out["prof_bytes"] = dev.prof_get()
        dev.prof_enable(False)
        # the same steps with the other transcript (untimed for `value`; reported as ms_per_step_<name>)
        tr["new"] = factories[other]
        step()
        gc.collect()
        gc.disable()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        out["dt_other"] = time.perf_counter() - t1
        gc.enable()
        for m in mles:
            m.free()
        return out

    def dist_commit_extra() -> dict:
        """N > 1: the multi-rank commitment (ceno_dist_commit_traces_mmcs: column-sharded RS encoding, ONE grouped ncclSend / ncclRecv
        re-shard by rows, local sub-tree, replicated top levels — DESIGN.md section 6) on two small synthetic traces of different
        heights, validated against the single-device root computed by rank 0, then timed.  Guarded exactly like the RCCL sumcheck arm:
        without a validated RCCL communicator it reports why and the bench goes on; a rank that fails reports it through the
        all-reduce; a collective that never returns is caught by the caller's watchdog (non-zero exit of the process)."""
        # RCCL where this launch has a validated communicator; otherwise the shared segment's host-staged bulk exchange (ranks sharing one GPU, a
        # node whose RCCL is unusable): the same entry points, validated the same way — its times say nothing about xGMI and are labelled
        bulk = "rccl" if "rccl" in comms else ("shm" if "shm" in comms else None)
        if bulk is None:
            return {"status": "skipped: no communicator with a bulk exchange on this launch"}
        import ctypes as C

        from ceno_amd import dist as cdist2

        log_rows, cols_per_rank, blow = [14, 10], [2, 1], 1
        ok, root, info = 1, None, {}
        try:
            mine = [dev.synthetic(lr + (c.bit_length() - 1), False, 0xC0117 + 97 * m + rank) for m, (lr, c) in enumerate(zip(log_rows, cols_per_rank))]
            dev.sync()
            widths = [[c] * world for c in cols_per_rank]
            cst = dev.stream_create()

            def run_once():
                r_ = cdist2.sharded_commit_mmcs_native(dev, comms[bulk].h, [m_.device_ptr for m_ in mine], widths, log_rows, blow, rank, cst)
                dev.sync(cst)
                for key in ("subtree", "top"):
                    if r_.get(key):
                        dev.L.ceno_hip_merkle_free(dev.h, r_[key])
                return r_["root"]

            root = run_once()
            if rank == 0:  # the same traces on ONE device: all ranks' columns side by side, row-major on the host -> commit_traces
                mats = []
                for m, (lr, c) in enumerate(zip(log_rows, cols_per_rank)):
                    cols = []
                    for g in range(world):
                        t = dev.synthetic(lr + (c.bit_length() - 1), False, 0xC0117 + 97 * m + g)
                        cols.append(t.download().reshape(c, 1 << lr))
                        t.free()
                    mats.append(np.ascontiguousarray(np.concatenate(cols, axis=0).T))
                pcs = prover.PcsData(dev, mats, blow, stream)
                want = pcs.root()
                pcs.free()
                ok = 1 if np.array_equal(np.asarray(want, dtype=np.uint64).reshape(-1), np.asarray(root, dtype=np.uint64).reshape(-1)) else 0
                info["root_matches_single_device"] = bool(ok)
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: multi-rank commit failed on rank {rank}: {e}", file=sys.stderr)
            ok, info["error"] = 0, f"{type(e).__name__}: {e}"
        flag = torch.tensor([ok], dtype=torch.int32, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            return {"status": "failed validation (reported, bench continues)", **info}
        reps = 5
        barrier()
        t_ = time.perf_counter()
        for _ in range(reps):
            run_once()
        barrier()
        ms = max_over_ranks(time.perf_counter() - t_) / reps * 1e3
        for m_ in mine:
            m_.free()
        # ... and the OPENING of a commitment made across the ranks (ceno_dist_basefold_open: one matrix of 2^14 rows, 2 columns per rank):
        # validated word for word against the single-device opening rank 0 computes, then timed
        try:
            info["dist_open"] = dist_open_part(cdist2, cst, bulk)
        except Exception as e:  # noqa: BLE001
            info["dist_open"] = {"status": f"failed: {type(e).__name__}: {e}"}
        return {"status": "ok", "ms": ms, "bulk_exchange": "RCCL (grouped ncclSend / ncclRecv)" if bulk == "rccl" else "shared segment, host-staged (no RCCL communicator)",
                "workload": f"ceno_dist_commit_traces_mmcs: traces of 2^{log_rows[0]} x {cols_per_rank[0] * world} and 2^{log_rows[1]} x "
                f"{cols_per_rank[1] * world} base elements column-sharded over {world} ranks, blow-up 2, ONE root", **info}

    def dist_open_part(cdist2, cst, bulk) -> dict:
        lr, c, blow, nq, pow_bits = 14, 2, 1, 20, 8
        mine = dev.synthetic(lr + 1, False, 0x0BE11 + rank)
        dev.sync()
        widths = [[c] * world]
        com = cdist2.sharded_commit_mmcs_native(dev, comms[bulk].h, [mine.device_ptr], widths, [lr], blow, rank, cst)
        dev.sync(cst)
        point = np.array([[i * 7919 + 13, i * 104729 + 17] for i in range(lr)], dtype=np.uint64)
        evals = [np.zeros((c * world, 2), dtype=np.uint64)]  # (one height class: the claimed evaluations do not enter the proof's words)

        def run_once():
            return prover.dist_basefold_open(dev, comms[bulk].h, lr, widths, blow, [mine.device_ptr], [t.data_ptr() for t in com["codeword_rows"]],
                                             com["subtree"], com["top"], [point], evals, nq, pow_bits, factories[args.transcript](), cst)

        got = run_once()
        ok = 1
        if rank == 0:
            cols = []
            for g in range(world):
                t = dev.synthetic(lr + 1, False, 0x0BE11 + g)
                cols.append(t.download().reshape(c, 1 << lr))
                t.free()
            pcs = prover.PcsData(dev, [np.ascontiguousarray(np.concatenate(cols, axis=0).T)], blow, stream)
            want = pcs.basefold_open([point], evals, nq, pow_bits, factories[args.transcript]())
            pcs.free()
            ok = 1 if np.array_equal(want, got) else 0
        flag = torch.tensor([ok], dtype=torch.int32, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            return {"status": "failed validation: the multi-rank opening differs from the single-device opening"}
        barrier()
        t_ = time.perf_counter()
        for _ in range(3):
            run_once()
        barrier()
        ms = max_over_ranks(time.perf_counter() - t_) / 3 * 1e3
        for key in ("subtree", "top"):
            if com.get(key):
                dev.L.ceno_hip_merkle_free(dev.h, com[key])
        mine.free()
        return {"status": "ok", "ms": ms, "proof_equals_single_device": True,
                "workload": f"ceno_dist_basefold_open: one trace of 2^{lr} x {c * world} base elements committed across {world} ranks, {nq} queries, {pow_bits}-bit proof of work"}

    def dist_chip_proof_extra() -> dict:
        """N > 1: the GKR half of config #3's chip across the ranks (ceno_dist_create_chip_proof): an ADD-shaped chip of 2^20
        rows x 22 base columns, rows dealt to the ranks in blocks of 2^10 (block-cyclic), record inference + three towers + tower proof, the
        per-round partial sums exchanged through the shared segment.  Validated against the single-device proof rank 0 computes from the whole
        columns, then timed (max over ranks)."""
        from ceno_amd import synthetic

        if "shm" not in comms:
            return {"status": "skipped: no shared-memory communicator on this launch"}
        log_rows, w, q = int(os.environ.get("CENO_BENCH_DIST_CHIP_LOG_ROWS", "20")), 22, 10
        if log_rows < q + log_w + 1:
            return {"status": f"skipped: 2^{log_rows} rows are too few for {world} ranks at blocks of 2^{q}"}
        alpha, beta = (0x1234567, 0x89ABCDE), (0x13579B, 0x2468AC)
        coeffs, terms, out_terms = synthetic.record_plan(w, 16, alpha, beta)
        idx = np.arange(1 << log_rows)
        mine = ((idx >> q) & (world - 1)) == rank
        local, full = [], []
        for j in range(w):
            c = dev.synthetic(log_rows, False, 0xADD0 + j)
            local.append(dev.upload(np.ascontiguousarray(c.download()[mine])))
            if rank == 0:
                full.append(c)
            else:
                c.free()
        task = dict(mles=local, n_witin=w, n_fixed=0, n_structural=0, num_instances=(1 << log_rows) - 3, log2_num_instances=log_rows - log_w, num_reads=4,
                    num_writes=4, num_lk_tables=0, num_lk=8, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
        cst = dev.stream_create()

        def run_once():
            return prover.dist_create_chip_proof(dev, comms["shm"].h, task, log_rows, q, [alpha, beta], factories[args.transcript](), cst)

        got = run_once()
        # ... and the chip's main-constraint sumcheck on the SAME row layout (ceno_dist_prove_batched_main_constraints): the chip flow's plan,
        # a Prefix selector at the rt_main of the proof
        mterms, mscalars = synthetic.main_plan(w, w)
        sel = (1, 0, (1 << log_rows) - 3, 0, (), 0, np.ascontiguousarray(got.rt_main))

        def main_job(tables):
            return [dict(num_vars=log_rows, mles=list(tables) + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4,
                         terms=mterms, scalars=mscalars)]

        def run_main():
            return prover.dist_prove_batched_main_constraints(dev, comms["shm"].h, main_job(local), [alpha, beta], factories[args.transcript](), q, cst)

        got_main = run_main()
        ok, single_ms, single_main_ms = 1, None, None
        if rank == 0:
            ftask = dict(task, mles=full, log2_num_instances=log_rows)
            want = prover.create_chip_proof(dev, ftask, [alpha, beta], factories[args.transcript](), cst)
            ok = 1 if (np.array_equal(want.tower_msgs, got.tower_msgs) and np.array_equal(want.tower_point, got.tower_point) and
                       np.array_equal(want.tower_prod_evals, got.tower_prod_evals) and np.array_equal(want.tower_logup_evals, got.tower_logup_evals)) else 0
            want_main = prover.prove_batched_main_constraints(dev, main_job(full), [alpha, beta], factories[args.transcript](), cst)
            if not (want_main[0] == got_main[0] and all(np.array_equal(a_, b_) for a_, b_ in zip(want_main[1:], got_main[1:]))):
                ok = 0
            bm = 1e9
            for _ in range(3):
                dev.sync()
                t_ = time.perf_counter()
                prover.prove_batched_main_constraints(dev, main_job(full), [alpha, beta], factories[args.transcript](), cst)
                dev.sync()
                bm = min(bm, (time.perf_counter() - t_) * 1e3)
            single_main_ms = bm
            best = 1e9
            for _ in range(3):
                dev.sync()
                t_ = time.perf_counter()
                prover.create_chip_proof(dev, ftask, [alpha, beta], factories[args.transcript](), cst)
                dev.sync()
                best = min(best, (time.perf_counter() - t_) * 1e3)
            single_ms = best
            for m_ in full:
                m_.free()
        flag = torch.tensor([ok], dtype=torch.int32, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            return {"status": "failed validation: the sharded chip proof or main-constraint sumcheck differs from the single-device one"}
        best = 1e9
        for _ in range(3):
            barrier()
            t_ = time.perf_counter()
            run_once()
            dev.sync()
            best = min(best, max_over_ranks(time.perf_counter() - t_) * 1e3)
        best_main = 1e9
        for _ in range(3):
            barrier()
            t_ = time.perf_counter()
            run_main()
            dev.sync()
            best_main = min(best_main, max_over_ranks(time.perf_counter() - t_) * 1e3)
        for m_ in local:
            m_.free()
        dev.stream_destroy(cst)
        return {"status": "ok", "ms": best, "single_device_ms_on_rank0": single_ms, "proof_equals_single_device": True,
                "main_constraints": {"ms": best_main, "single_device_ms_on_rank0": single_main_ms, "equals_single_device": True,
                                     "workload": "ceno_dist_prove_batched_main_constraints: the same chip's main-constraint sumcheck (33 terms of degree <= 4, one "
                                                 "Prefix selector at rt_main) on the same row layout"},
                "workload": f"ceno_dist_create_chip_proof: ADD-shaped chip, 2^{log_rows} rows x {w} base columns, 4 + 4 + 8 records, rows dealt to {world} ranks in "
                            f"blocks of 2^{q}; record inference, tower witness, tower proof; shared-memory exchange"}

    def line(m: dict, scaling: str) -> dict:
        n_local, n_total = m["n_local"], m["n_total"]
        dt, kernel_ms, launches = m["dt"], m["kernel_ms"], m["launches"]
        value = K * K * ((1 << n_total) - 1) * args.steps / dt
        alg_bytes_per_step = 3 * K * 16 * (1 << n_local)          # 3*d*s*2^n per sumcheck (per GPU)
        achieved = alg_bytes_per_step * args.steps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        peak = 8000.0
        return {
            "metric": f"Goldilocks-ext mults/sec in sumcheck nv={n_total if scaling == 'strong' else n_local}",  # BASELINE.json metric at the default --nv 26
            "value": value,
            "unit": "ext-mults/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_uninstrumented": m["dt_plain"] / args.steps * 1e3,
            "step_ms": m.get("step_ms", {}),
            f"ms_per_step_{args.transcript}": dt / args.steps * 1e3,
            f"ms_per_step_{other}": m["dt_other"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "poseidon2_constants": "reference" if poseidon2_pinned else "placeholder (PARITY UNPINNED)",
            "config": {
                "workload": (f"single sumcheck instance, {K} MLEs x nv={n_total}, Goldilocks-ext2 (16 B/elem), degree {K}, " +
                             (f"hypercube split over {world} GPUs (nv={n_local} per GPU), " if world > 1 else "") +
                             f"{args.transcript} Fiat-Shamir transcript on host, inputs resident in HBM"),
                "global_num_vars": n_total,
                "num_vars_per_gpu": n_local,
                "sharding": "none" if world == 1 else f"top-{log_w}-bits over {world} GPUs, all-gather of partials per round",
                "collective": m["collective"],
            },
            **({"collective_ms": m["collective_ms"], "exchanges_validated": m["validated"], "headline_exchange": m.get("headline_exchange"),
                "fastest_exchange": m.get("fastest_exchange"), **comm_info} if world > 1 else {}),
            "roofline": {
                # the contract's field: HBM is the roof `frac` is taken against.  The kernel is CO-LIMITED (profiles/r05_dense_overlap.json,
                # r04_dense_kernel_floor.json): its fold rounds issue VALU instructions at ~0.86 of capacity while its schedule moves its bytes at
                # ~0.84 of what that access pattern can reach — `valu_frac` / `schedule_ceiling_frac` below say so on this run's time
                "bound": "hbm",
                "co_limited_by": "valu",
                **valu_headline(kernel_ms / args.steps if args.steps else 0.0, n_local),
                "schedule_min_bytes": 4 * K * 16 * (1 << n_local),   # 192 * 2^n: one round per pass (Fiat-Shamir) reads T0 twice; 144 * 2^n is the yardstick
                "kernel": "k_dense<3,*> (fused fold + round-polynomial accumulate)",
                "achieved": achieved,
                "peak": peak,
                "unit": "GB/s",
                "frac": achieved / peak,
                "traffic": None,
                "launches": int(launches),
                "avg_launch_ms": kernel_ms / launches if launches else None,
                # HIP events around every k_dense launch of the TIMED steps (pipelined: a launch ends when its finishing workgroup has the next
                # challenge, so kernel_ms_per_sumcheck <= ms_per_step by construction; rounds of <= 2^15 pairs run on k_mid / k_tail and are not
                # in it).  `rocprofv3 --kernel-trace --stats` of this command (profiles/r*_sumcheck_nv26_kernel_stats.csv): sum the
                # TotalDurationNs of the k_dense rows and divide by the calls of the read-only round-0 instantiation k_dense<3, 0, *>.
                "kernel_ms_per_sumcheck": kernel_ms / args.steps if args.steps else None,
                "algorithmic_bytes_per_launch": alg_bytes_per_step * args.steps / launches if launches else None,
                "schedule_bytes_per_step": m["prof_bytes"] / args.steps if args.steps else None,
                "schedule_gbps": (m["prof_bytes"] / (kernel_ms * 1e-3) / 1e9) if kernel_ms > 0 else None,
                # against what THIS schedule can reach on this chip (tools/ubench_bw.hip, profiles/r03_dense_kernel_ab.json): the read-only
                # round streams at 6.45 TB/s, the fold rounds (read four elements, write two per table) at 5.79 TB/s; the fraction is
                # (time the schedule's bytes need at those ceilings) / (measured kernel time)
                "schedule_ceiling_frac": ((K * 16 * (1 << n_local) / 6.45e12 + (m["prof_bytes"] / args.steps - K * 16 * (1 << n_local)) / 5.79e12)
                                          / (kernel_ms * 1e-3 / args.steps)) if kernel_ms > 0 and args.steps else None,
                "schedule_ceilings_gbps": {"read_only_round": 6450.0, "fold_rounds": 5790.0},
            },
        }

    def run_sizes(kinds):
        strong = weak = None
        if args.scaling in ("strong", "both") and args.nv - log_w >= 1:
            strong = line(measure(args.nv - log_w, kinds), "strong")
        if args.scaling in ("weak", "both") or strong is None:
            weak = line(measure(args.nv, kinds), "weak")
        r_ = strong if strong is not None else weak
        if strong is not None and weak is not None:
            r_["weak_scaling"] = {k: weak[k] for k in ("metric", "value", "unit", "ms_per_step", "scaling", "config", "collective_ms", "roofline")}
        return r_

    if world == 1:
        res = line(measure(args.nv), "weak")  # one GPU: strong and weak coincide; `weak` is what the single-GPU line has always said
    else:
        # N > 1.  FIRST a complete measurement that needs no RCCL call of this library (the shared-memory exchange, validated against the
        # torch.distributed path): whatever happens to the RCCL arm afterwards, this line exists.  THEN the RCCL communicator is created and
        # one sharded sumcheck validated under a watchdog; only if every rank reports success is the measurement repeated over RCCL — which
        # then carries the headline (north_star names the RCCL collective; CENO_BENCH_HEADLINE=fastest picks the faster one).  A collective
        # that never returns: every rank notices after CENO_BENCH_RCCL_TIMEOUT_S, rank 0 prints the shared-memory line with the reason,
        # and the processes leave through os._exit (their stream cannot be drained any more) — a reported fallback, not a hang.
        wanted = [x for x in os.environ.get("CENO_BENCH_EXCHANGE", "shm,rccl").split(",") if x]
        base = [k for k in wanted if k != "rccl"] or ["shm"]
        res = run_sizes(base)
        if "rccl" in wanted:
            import threading

            probe = {"ok": 0, "why": "timeout: the RCCL communicator or its first all-gather did not return"}

            def rccl_probe():
                try:
                    if os.environ.get("CENO_BENCH_FAKE_RCCL_HANG") == "1":  # tests: what a collective that never returns looks like
                        time.sleep(1e6)
                    n_probe = 10
                    pm = [dev.synthetic(n_probe, True, SEED0 + j, word_offset=rank * 2 * (1 << n_probe)) for j in range(K)]
                    ref = cdist.sharded_sumcheck_prove(cdist.HipShardEngine(dev, pm), n_probe + log_w, K, prover.Transcript.stub(TR_SEED), dist=dist,
                                                       world=world, rank=rank)
                    comms["rccl"] = prover.RcclComm(world, rank, dist)
                    comm_info["rccl_ranks"] = comms["rccl"].nranks()
                    got = prover.dist_sumcheck_prove(dev, comms["rccl"], pm, ONE, TERMS, n_probe + log_w, K, prover.Transcript.stub(TR_SEED), stream)
                    dev.sync()
                    same = all(np.array_equal(x, y) for x, y in zip(got, ref))
                    probe["ok"], probe["why"] = (1, "ok") if same else (0, "the RCCL path's proof differs from the torch.distributed path's")
                    for m_ in pm:
                        m_.free()
                except Exception as e:  # noqa: BLE001
                    probe["ok"], probe["why"] = 0, f"{type(e).__name__}: {e}"

            th = threading.Thread(target=rccl_probe, daemon=True)
            th.start()
            th.join(float(os.environ.get("CENO_BENCH_RCCL_TIMEOUT_S", "90")))
            hung = th.is_alive()
            flag = torch.tensor([0 if hung else probe["ok"]], dtype=torch.int32, device=tdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                over_rccl = run_sizes(["rccl"])
                if "rccl" in over_rccl.get("exchanges_validated", []):
                    for key in ("collective_ms",):
                        over_rccl[key] = dict(res.get(key, {}), **over_rccl.get(key, {}))
                    over_rccl["exchanges_validated"] = sorted(set(res.get("exchanges_validated", [])) | {"rccl"})
                    if "weak_scaling" in over_rccl and "weak_scaling" in res:
                        over_rccl["weak_scaling"]["collective_ms"] = dict(res["weak_scaling"].get("collective_ms", {}), **over_rccl["weak_scaling"].get("collective_ms", {}))
                    fastest = min(over_rccl["collective_ms"], key=lambda k_: over_rccl["collective_ms"][k_])
                    over_rccl["fastest_exchange"] = fastest
                    if os.environ.get("CENO_BENCH_HEADLINE", "rccl") == "fastest" and fastest != "rccl":
                        res["collective_ms"], res["exchanges_validated"], res["fastest_exchange"] = over_rccl["collective_ms"], over_rccl["exchanges_validated"], fastest
                        res.update(comm_info)
                    else:
                        res = over_rccl
                else:
                    res["rccl"] = "unavailable: validated in the probe but not at the benchmark size"
            else:
                res["rccl"] = "unavailable: " + (probe["why"] if not hung else "timeout after CENO_BENCH_RCCL_TIMEOUT_S: the RCCL communicator or its first all-gather did not return")
                print(f"bench.py: RCCL arm unavailable on rank {rank}: {res['rccl']}", file=sys.stderr)
                comms.pop("rccl", None)
                if hung:  # this process cannot synchronise its device any more: report and leave
                    # exit status 0 (the shared-memory measurement is complete and valid), but the line says so at its top level: anything
                    # that only looks at the exit status still finds `degraded` in the one JSON line
                    res["degraded"] = "rccl_timeout"
                    res.setdefault("extra", {})["dist_commit"] = {"status": "skipped: the RCCL arm did not return"}
                    if rank == 0:
                        print(json.dumps(res))
                        sys.stdout.flush()
                    os._exit(0)
    n_local = res["config"]["num_vars_per_gpu"]
    if world > 1 and not args.no_extra:
        # the multi-rank commitment as an extra, under a watchdog: a collective that never returns must not take the headline along —
        # rank 0 prints the line it has, and the process exits non-zero
        import threading

        box = {}

        def run_extra():
            try:
                box["r"] = dist_commit_extra()
            except Exception as e:  # noqa: BLE001
                box["r"] = {"status": f"failed: {type(e).__name__}: {e}"}

        th = threading.Thread(target=run_extra, daemon=True)
        th.start()
        th.join(float(os.environ.get("CENO_BENCH_DIST_COMMIT_TIMEOUT_S", "120")))
        if th.is_alive():
            res["degraded"] = "dist_commit_timeout"
            res.setdefault("extra", {})["dist_commit"] = {"status": "timeout: the multi-rank commit did not return (reported, process exits non-zero)"}
            if rank == 0:
                print(json.dumps(res))
                sys.stdout.flush()
            os._exit(4)
        res.setdefault("extra", {})["dist_commit"] = box["r"]
        # the row-sharded chip proof under the same kind of watchdog: a rank that fails leaves the others waiting in an exchange — the line the
        # bench has must still come out
        box2 = {}

        def run_extra2():
            try:
                box2["r"] = dist_chip_proof_extra()
            except Exception as e:  # noqa: BLE001
                box2["r"] = {"status": f"failed: {type(e).__name__}: {e}"}

        th2 = threading.Thread(target=run_extra2, daemon=True)
        th2.start()
        th2.join(float(os.environ.get("CENO_BENCH_DIST_CHIP_TIMEOUT_S", "180")))
        if th2.is_alive():
            res["degraded"] = "dist_chip_proof_timeout"
            res["extra"]["dist_chip_proof"] = {"status": "timeout: the row-sharded chip proof did not return (reported, process exits non-zero)"}
            if rank == 0:
                print(json.dumps(res))
                sys.stdout.flush()
            os._exit(5)
        res["extra"]["dist_chip_proof"] = box2["r"]
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline leg must not take the bench line down
                res["cpu_baseline"] = {"value": None, "unit": "ext-mults/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure
        # comes from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
        # (profiles/, tools/pmc_summary.py); null when no matching profile is committed
        try:
            if world == 1 and n_local == 26:
                import glob

                cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sumcheck_nv26_pmc_traffic.json")))
                if cands:
                    pm = json.load(open(cands[-1]))
                    # HBM bytes of ONE sumcheck (= one step) by the PMC counters; `achieved` is that step's algorithmic bytes over its kernel time
                    res["roofline"]["traffic"] = pm["hbm_bytes_per_sumcheck"]
                    res["roofline"]["traffic_unit"] = "HBM bytes per sumcheck (one step), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE"
                    res["roofline"]["traffic_per_sumcheck"] = pm["hbm_bytes_per_sumcheck"]
                    res["roofline"]["traffic_source"] = os.path.relpath(cands[-1], ROOT)
        except Exception:
            pass
        if world == 1 and not args.no_extra:
            try:
                res["extra"] = extra_measurements(dev, prover, factories[args.transcript], args.transcript)
            except Exception as e:  # the extras must not take the headline down
                res["extra"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        for c in comms.values():
            try:
                c.close()
            except Exception:
                pass
        dist.destroy_process_group()
    dev.close()


if __name__ == "__main__":
    main()
