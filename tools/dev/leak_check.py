#!/usr/bin/env python3
"""repeat the wide batched main sumcheck and watch the pool: pool_used must return to the tables' bytes after every proof"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover, synthetic

dev = Device(0)
jobs, chips, _ = synthetic.wide_batched_jobs(dev, 16)
mj = prover.MainJobs(jobs)
base = None
for i in range(150):
    prover.prove_batched_main_constraints(dev, mj, [(11, 22), (33, 44)], prover.Transcript.stub(5))
    dev.sync()
    info = dev.mem_info()
    if base is None:
        base = info["pool_used"]
    if i % 50 == 49 or info["pool_used"] != base:
        print(i, info["pool_used"], info["pool_cached"], "base", base, flush=True)
        if info["pool_used"] != base:
            sys.exit("pool_used drifts: a block is not returned")
print("ok: pool_used constant over 150 proofs")
dev.close()
