#!/usr/bin/env python3
"""bench.py's chip flow and shard flow after the 13.6 GB batched main sumcheck (the order of bench.extra_measurements), with the phase
breakdown: does the big batch leave the pool in a state that slows the flows down?"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover, synthetic

dev = Device(0)
new_tr = lambda: prover.Transcript.poseidon2(b"bench")
fork = lambda: prover.Transcript.poseidon2(b"fork")
def flows(tag):
    flow = synthetic.ChipFlow(dev, prover, 20, 22)
    best = None
    for _ in range(3):
        r = flow.run(new_tr)
        if best is None or r["total_ms"] < best["total_ms"]:
            best = r
    flow.close()
    print(tag, "chip", {k: round(v, 2) for k, v in best.items() if k.endswith("_ms")})
    shard = synthetic.ShardFlow(dev, prover)
    for lanes in (1, 4):
        b = None
        for _ in range(3):
            r = shard.run(new_tr, fork, lanes=lanes)
            if b is None or r["total_ms"] < b["total_ms"]:
                b = r
        print(tag, "shard lanes", lanes, {k: round(v, 2) for k, v in b.items() if k.endswith("_ms")})
    shard.close()
    print(tag, "pool", dev.mem_info() if hasattr(dev, "mem_info") else "")
if os.environ.get("BEFORE", "1") == "1":
    flows("fresh ")
jobs, elems = synthetic.batched_jobs(dev, int(os.environ.get("MAXNV", "26")), 12)
mj = prover.MainJobs(jobs)
prover.prove_batched_main_constraints(dev, mj, [(11, 22), (33, 44)], new_tr())
for j in jobs:
    for m in j["mles"]:
        if m is not None:
            m.free()
del mj, jobs
flows("after ")
