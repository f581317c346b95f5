#!/usr/bin/env python3
"""Experiment: per-round host waits of the ADD-shaped main sumcheck (CENO_HIP_DEBUG=1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ceno_amd import Device, api, prover

dev = Device(0)
n, w = int(sys.argv[1]) if len(sys.argv) > 1 else 20, 22
rows = 1 << n
P = api.P
cols = [dev.synthetic(n, False, 10 + j) for j in range(w)]
pt_ = np.array([[(i * 7919 + 13) % P, (i * 104729 + 17) % P] for i in range(n)], dtype=np.uint64)
sel = dev.selector_build(1, pt_, 0, rows - 3)
mles = cols + [sel]
s_idx = len(cols)
mterms = [[j, (j + 1) % w] for j in range(w)] + [[j, (j + 3) % w, (j + 5) % w] for j in range(0, w, 2)]
mcoeffs = np.array([[(3 + 5 * i) % P, (11 * i + 1) % P] for i in range(len(mterms))], dtype=np.uint64)
groups = [([s_idx], list(range(len(mterms))))]
for _ in range(3):
    dev.sync(); t0 = time.perf_counter()
    prover.sumcheck_prove(dev, mles, mcoeffs, mterms, n, 4, prover.Transcript.stub(2), groups=groups)
    dev.sync(); print("total ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
