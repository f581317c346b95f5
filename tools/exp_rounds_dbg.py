#!/usr/bin/env python3
"""Device-side timeline of the rounds of one tower-layer-shaped generic sumcheck (CENO_HIP_DEBUG=1 prints it at free)."""
import os, sys, time
os.environ["CENO_HIP_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ceno_amd import Device, prover

dev = Device(0)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mles = [dev.synthetic(nv, True, 100 + j) for j in range(9)]
terms = [[0, 1, 2], [0, 3, 4], [0, 5, 8], [0, 6, 7], [0, 7, 8]]
coeffs = np.array([[3 + i, 5 * i + 1] for i in range(len(terms))], dtype=np.uint64)
for rep in range(3):
    dev.sync(); t0 = time.perf_counter()
    prover.sumcheck_prove(dev, mles, coeffs, terms, nv, 3, prover.Transcript.stub(7))
    dev.sync(); print("wall ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
