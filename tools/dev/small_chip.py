#!/usr/bin/env python3
"""one small chip proof (2^LOG rows x 22 columns, ADD-shaped records) timed back to back: where do its ~2 ms go?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover, synthetic

log_rows = int(os.environ.get("LOG", "13"))
d = Device(0)
w = 22
cols = [d.synthetic(log_rows, False, 300 + j) for j in range(w)]
alpha, beta = (5, 6), (7, 8)
coeffs, terms, out_terms = synthetic.record_plan(w, 16, alpha, beta)
task = dict(mles=cols, n_witin=w, n_fixed=0, n_structural=0, num_instances=(1 << log_rows) - 3, log2_num_instances=log_rows, num_reads=4,
            num_writes=4, num_lk_tables=0, num_lk=8, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
for _ in range(3):
    prover.create_chip_proof(d, task, [alpha, beta], prover.Transcript.poseidon2(b"x"))
d.sync()
best = 1e9
for _ in range(10):
    t0 = time.perf_counter()
    prover.create_chip_proof(d, task, [alpha, beta], prover.Transcript.poseidon2(b"x"))
    d.sync()
    best = min(best, (time.perf_counter() - t0) * 1e3)
print(f"chip proof 2^{log_rows} rows: {best:.3f} ms")
