#!/usr/bin/env python3
"""ONE chip's main-constraint sumcheck (2^nv rows x 22 columns, the chip flow's plan) through prove_batched_main_constraints: time and the round
kernels it was given.  usage: single_chip_main.py [nv]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("CENO_HIP_PLAN_REPORT", "1")
import numpy as np
from ceno_amd import Device, prover, synthetic

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = 22
dev = Device(0)
cols = [dev.synthetic(nv, False, 1000 + j) for j in range(w)]
P = synthetic.P if hasattr(synthetic, "P") else 0xFFFFFFFF00000001
point = np.array([[(i * 7919 + 13) % P, (i * 104729 + 17) % P] for i in range(nv)], dtype=np.uint64)
sel = (1, 0, (1 << nv) - 5, 0, (), 0, point)
terms, scalars = synthetic.main_plan(w, w)
job = dict(num_vars=nv, mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2, max_degree=4, terms=terms, scalars=scalars)
mj = prover.MainJobs([job])
runs = []
for _ in range(6):
    dev.sync()
    t0 = time.perf_counter()
    prover.prove_batched_main_constraints(dev, mj, [(11, 22), (33, 44)], prover.Transcript.poseidon2(b"riscv"))
    dev.sync()
    runs.append((time.perf_counter() - t0) * 1e3)
print("ms:", [round(x, 3) for x in sorted(runs)[:4]], "report:", dev.L.ceno_hip_plan_report(dev.h).decode()[:400])
dev.close()
