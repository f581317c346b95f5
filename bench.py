#!/usr/bin/env python3
"""bench.py — sumcheck + MLE-fold throughput (Goldilocks-ext mults/sec) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched through
torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.

A "step" = one complete sumcheck of prod_{k<3} f_k over synthetic GoldilocksExt2 tables already
resident in HBM: begin, n rounds (each: fused fold+accumulate pass, message to the host transcript,
challenge back), finish.  N=1 workload: BASELINE.json config "sumcheck, 3 MLEs, Goldilocks-ext2" at the
size the metric is quoted on, nv=26 (3 x 2^26 x 16 B = 3.2 GB).  N>1: weak scaling — every rank keeps a
2^26 shard of a 2^(26+log2 N) hypercube (top-bit sharding, one all-gather of partials per round).

Algorithmic work: ext mults = d^2 (2^n - 1), d = 3; bytes = 3*d*16*2^n.
"""
from __future__ import annotations

import argparse
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

    def roof(alg_bytes, ms, note=None):
        gbps = alg_bytes / (ms * 1e-3) / 1e9
        r = {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0, "algorithmic_bytes": alg_bytes}
        if note:
            r["note"] = note
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
    # config #4 shape on one GPU: batched main-constraint sumcheck, 24 chips, max_nv = 24
    jobs, elems = synthetic.batched_jobs(dev, 24, 12)
    mj = prover.MainJobs(jobs)  # the C view of the job list, marshalled once (a Rust caller hands the structs over directly)
    ms = best_of(lambda: prover.prove_batched_main_constraints(dev, mj, [(11, 22), (33, 44)], new_transcript()))
    out["batched_main"] = {"workload": "config #4 shape: prove_batched_main_constraints, 24 chips of 14..24 variables, 12 base columns + "
                                       "selector, 16 terms of degree <= 4 each", "ms": ms, "table_elements": elems,
                           "roofline": roof(synthetic.batched_algorithmic_bytes(24, 12), ms, "integer-ALU bound: DESIGN.md section 3")}
    for j in jobs:
        for m in j["mles"]:
            if m is not None:
                m.free()
    # config #3: ADD-shaped chip, 2^20 rows x 22 columns, commit -> chip proof -> main constraints -> open
    flow = synthetic.ChipFlow(dev, prover, 20, 22)
    best = None
    for _ in range(reps):
        r = flow.run(new_transcript)
        if best is None or r["total_ms"] < best["total_ms"]:
            best = r
    ab = flow.algorithmic_bytes()
    flow.close()
    best["workload"] = "config #3: ADD-shaped chip, 2^20 rows x 22 base columns, blow-up 2, 100 queries, 16-bit proof of work"
    best["roofline"] = roof(sum(ab.values()), best["total_ms"], "commit is integer-ALU bound (Poseidon2): DESIGN.md section 3")
    best["roofline_by_phase"] = {k: roof(v, best[f"{k}_ms"]) for k, v in ab.items()}
    out["chip_flow"] = best
    # metric M2 shape: one synthetic shard of 2^20 cycles through the whole create_proof flow
    shard = synthetic.ShardFlow(dev, prover)
    fork = (lambda: prover.Transcript.poseidon2(b"fork")) if transcript_name == "poseidon2" else (lambda: prover.Transcript.stub(0xF0))
    bs, by_lanes = None, {}
    for lanes in (1, 4):  # chip proofs serially, then four at a time (one host thread + one HIP stream per lane: the chip scheduler)
        for _ in range(reps):
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
    out["chip_flow_ms"], out["batched_main_ms"], out["nv22_ms"] = best["total_ms"], out["batched_main"]["ms"], out["nv22"]["ms"]
    out["shard_e2e_sec"] = bs["e2e_prover_sec_for_2p20_cycles"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nv", type=int, default=26, help="variables per GPU shard")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the config #2 / #3 / #4 shaped measurements reported under `extra`")
    ap.add_argument("--transcript", choices=["poseidon2", "stub"], default="poseidon2",
                    help="Fiat-Shamir transcript of the timed steps (the other one is timed too and reported beside it)")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    else:
        torch.cuda.set_device(local_rank)

    from ceno_amd import Device
    from ceno_amd import dist as cdist
    from ceno_amd import prover

    dev = Device(local_rank)
    from ceno_amd import goldens
    poseidon2_pinned = goldens.install(dev)  # reference constants when tests/golden/ref_goldens.json exists (README "Closing parity")
    n_local = args.nv
    log_w = world.bit_length() - 1
    n_total = n_local + log_w
    # shard `rank` of table j: words [rank * 2 * 2^n_local, ...) of the SplitMix stream seeded SEED0 + j
    mles = [dev.synthetic(n_local, True, SEED0 + j, word_offset=rank * 2 * (1 << n_local)) for j in range(K)]
    dev.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dev.sync()

    ONE = np.array([[1, 0]], dtype=np.uint64)
    TERMS = [list(range(K))]
    # Fiat-Shamir on the host between every two rounds: the Poseidon2 duplex challenger (what the reference's BasicTranscript
    # is; 2 permutations per round on the critical path) by default, the SplitMix stub for comparison
    factories = {"poseidon2": lambda: prover.Transcript.poseidon2(b"riscv"), "stub": lambda: prover.Transcript.stub(TR_SEED)}
    new_transcript = factories[args.transcript]
    collective = "none"
    comm = stream = None
    comms, collective_ms = {}, {}

    def step_torch():
        # per-round all-gather issued from Python through torch.distributed (RCCL underneath)
        eng = cdist.HipShardEngine(dev, mles)
        return cdist.sharded_sumcheck_prove(eng, n_total, K, new_transcript(), dist=dist, world=world, rank=rank)

    def step_native():
        # the same protocol driven from C++ with ncclAllGather on the kernels' HIP stream (host/dist.cpp)
        return prover.dist_sumcheck_prove(dev, comm, mles, ONE, TERMS, n_total, K, new_transcript(), stream)

    if world > 1:
        # Three drivers of the same protocol, fastest first; each candidate must reproduce the torch.distributed path's
        # proof on every rank before it is used for the timed steps:
        #   shm  : C++ loop, per-round partials exchanged through host shared memory (the messages are in host memory
        #          anyway for the transcript) — no device collective on the round path
        #   rccl : C++ loop, ncclAllGather on the kernels' HIP stream + early gather of small shards
        #   torch: Python loop over torch.distributed all_gather (RCCL underneath)
        collective = "torch.distributed all_gather (python loop)"
        step_fn = step_torch
        stream = dev.stream_create()
        try:
            reference = step_torch()
        except Exception as e:  # without it a candidate is accepted when it runs on every rank (its proof is replicated by construction)
            print(f"bench.py: torch.distributed reference path failed on rank {rank}: {e}", file=sys.stderr)
            reference = None
        # every exchange that works on all ranks and reproduces the reference proof is kept: each is timed below and reported
        # under `collective_ms`; the fastest one carries `value`
        order = [x for x in os.environ.get("CENO_BENCH_EXCHANGE", "shm,rccl").split(",") if x]
        comms = {}
        for kind in order:
            ok = 1
            try:
                comm = prover.ShmComm(world, rank, dist) if kind == "shm" else prover.RcclComm(world, rank, dist)
                got = step_native()
                ok = 1 if reference is None or all(np.array_equal(x, y) for x, y in zip(got, reference)) else 0
            except Exception as e:  # keep looking
                print(f"bench.py: {kind} exchange unavailable on rank {rank}: {e}", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                comms[kind] = comm
            comm = None
        names = {"shm": "host shared-memory exchange of the d partial evaluations per round from the C++ host loop",
                 "rccl": "ncclAllGather of the d partial evaluations per round on the kernels' stream from the C++ host loop (RCCL over xGMI)"}
        collective_ms = {}

        def time_steps(fn, n):
            for _ in range(min(2, args.warmup + 1)):
                fn()
            barrier()
            t_ = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            d_ = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device=f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(d_, op=dist.ReduceOp.MAX)
            return float(d_.item()) / n * 1e3

        for kind, c in comms.items():
            comm = c
            collective_ms[kind] = time_steps(step_native, max(2, args.steps // 2))
        if not comms:
            collective_ms["torch"] = time_steps(step_torch, max(2, args.steps // 2))
        else:
            best = min(comms, key=lambda k_: collective_ms[k_])
            comm = comms[best]
            step_fn = step_native
            collective = names[best] + (" (checked against the torch.distributed path)" if reference is not None else " (unchecked: reference path failed)")

    def step():
        if world == 1:
            return prover.sumcheck_prove(dev, mles, ONE, TERMS, n_total, K, new_transcript())
        return step_fn()

    for _ in range(args.warmup):
        step()
    # timed region: exactly `steps` steps, no profiling hooks active
    barrier()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    # second, untimed pass of the same steps with HIP events around every launch of the dominant kernel
    # (events recorded on the library's launch stream) -> roofline numbers
    dev.prof_enable(True)
    dev.prof_reset()
    for _ in range(args.steps):
        step()
    barrier()
    kernel_ms, launches, prof_bytes = dev.prof_get()
    dev.prof_enable(False)
    # the same steps with the other transcript (untimed for `value`; reported as ms_per_step_<name>)
    other = "stub" if args.transcript == "poseidon2" else "poseidon2"
    new_transcript = factories[other]
    step()
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt_other = time.perf_counter() - t1
    new_transcript = factories[args.transcript]

    # max over ranks
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    mults_per_step = K * K * ((1 << n_total) - 1)
    value = mults_per_step * args.steps / dt
    alg_bytes_per_step = 3 * K * 16 * (1 << n_local)          # 3*d*s*2^n per sumcheck (per GPU)
    sched_bytes = prof_bytes                                     # bytes the fused schedule itself must move
    achieved = alg_bytes_per_step * args.steps / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    peak = 8000.0
    res = {
        "metric": f"Goldilocks-ext mults/sec in sumcheck nv={n_local}",  # BASELINE.json metric at the default --nv 26
        "value": value,
        "unit": "ext-mults/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        f"ms_per_step_{args.transcript}": dt / args.steps * 1e3,
        f"ms_per_step_{other}": dt_other / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "poseidon2_constants": "reference" if poseidon2_pinned else "placeholder (PARITY UNPINNED)",
        "config": {
            "workload": f"single sumcheck instance, {K} MLEs x nv={n_local} per GPU, Goldilocks-ext2 (16 B/elem), "
                        f"degree {K}, {args.transcript} Fiat-Shamir transcript on host, inputs resident in HBM",
            "global_num_vars": n_total,
            "sharding": "none" if world == 1 else f"top-{log_w}-bits over {world} GPUs, all-gather of partials per round",
            "collective": collective,
        },
        **({"collective_ms": collective_ms, "exchanges_validated": sorted(comms)} if world > 1 else {}),
        "roofline": {
            "bound": "hbm",
            "kernel": "k_dense<3,*> (fused fold + round-polynomial accumulate)",
            "achieved": achieved,
            "peak": peak,
            "unit": "GB/s",
            "frac": achieved / peak,
            "traffic": None,
            "launches": int(launches),
            "avg_launch_ms": kernel_ms / launches if launches else None,
            "algorithmic_bytes_per_launch": alg_bytes_per_step * args.steps / launches if launches else None,
            "schedule_bytes_per_step": sched_bytes / args.steps if args.steps else None,
            "schedule_gbps": (sched_bytes / (kernel_ms * 1e-3) / 1e9) if kernel_ms > 0 else None,
        },
    }
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
                    res["roofline"]["traffic"] = pm["hbm_bytes_per_launch"]
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
        dist.destroy_process_group()
    for m in mles:
        m.free()
    dev.close()


if __name__ == "__main__":
    main()
