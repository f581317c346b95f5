#!/usr/bin/env python3
"""the chip flow right after the 13.6 GB batch, as bench.py's extras run it: does the pool's state slow it down?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover, synthetic
dev = Device(0)
tr = lambda: prover.Transcript.poseidon2(b"riscv")
def chip(tag):
    flow = synthetic.ChipFlow(dev, prover, 20, 22)
    for i in range(4):
        r = flow.run(tr)
        print(tag, i, {k: round(v, 2) for k, v in r.items() if k.endswith("_ms")}, {k: v >> 20 for k, v in dev.mem_info().items()}, flush=True)
    flow.close()
if os.environ.get("FIRST", "1") == "1":
    chip("fresh")
for max_nv in (26, 24):
    jobs, elems = synthetic.batched_jobs(dev, max_nv, 12)
    mj = prover.MainJobs(jobs)
    for _ in range(2):
        t0 = time.perf_counter(); prover.prove_batched_main_constraints(dev, mj, [(11, 22), (33, 44)], tr()); dev.sync()
        print("batched", max_nv, round((time.perf_counter() - t0) * 1e3, 2), {k: v >> 20 for k, v in dev.mem_info().items()}, flush=True)
    for j in jobs:
        for m in j["mles"]:
            if m is not None:
                m.free()
    del mj, jobs
chip("after")
