#!/usr/bin/env python3
"""Experiment: Basefold open alone (for kernel traces of the commit phase)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ceno_amd import Device, api, prover

dev = Device(0)
n, w = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 22
P = api.P
stream = dev.stream_create()
host = (np.random.default_rng(1).integers(0, 1 << 62, size=(1 << n, w), dtype=np.uint64)) % np.uint64(P)
pcs = prover.PcsData(dev, [host], 1, stream)
pt = np.array([[(i * 7919 + 13) % P, (i * 104729 + 17) % P] for i in range(n)], dtype=np.uint64)
evals = np.zeros((w, 2), dtype=np.uint64)
for c in range(w):
    evals[c] = pcs.witness_mle(0, c).evaluate(pt)
for _ in range(3):
    dev.sync(); t0 = time.perf_counter()
    pcs.basefold_open([pt], [evals], 100, 16, prover.Transcript.stub(3))
    print("open ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
