#!/usr/bin/env python3
"""repeat the wide batched main sumcheck (small sizes) and compare every run with the first: a timing-dependent difference shows up as a mismatch.
usage: stress_wide.py [max_nv] [reps]  (environment switches apply)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover, synthetic

max_nv = int(sys.argv[1]) if len(sys.argv) > 1 else 14
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = Device(0)
n_chips = int(sys.argv[3]) if len(sys.argv) > 3 else 48
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
jobs, chips, _ = synthetic.wide_batched_jobs(dev, max_nv, n_chips)
jobs, chips = jobs[first:], chips[first:]
mj = prover.MainJobs(jobs)
gch = [(11, 22), (33, 44)]
ref = None
bad = 0
for it in range(reps):
    claimed, msgs, rt, evals = prover.prove_batched_main_constraints(dev, mj, gch, prover.Transcript.stub(5))
    if ref is None:
        ref = (msgs.copy(), rt.copy(), evals.copy())
        continue
    if not (np.array_equal(msgs, ref[0]) and np.array_equal(evals, ref[2])):
        bad += 1
        d = np.argwhere(msgs != ref[0])
        print(f"run {it}: MISMATCH, first differing message words (round, point, word): {d[:4].tolist()}", flush=True)
print(f"max_nv {max_nv} chips [{first}, {n_chips}): {reps} runs, {bad} mismatches")
