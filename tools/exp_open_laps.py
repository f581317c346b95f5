#!/usr/bin/env python3
"""Phase laps of the Basefold open at chip-flow size (CENO_HIP_DEBUG=1 makes the host layer print them)."""
import os, sys
os.environ["CENO_HIP_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ceno_amd import Device, api, prover
dev = Device(0)
n, w = 20, 22
P = api.P
rows = 1 << n
stream = dev.stream_create()
host = (np.random.default_rng(1).integers(0, 1 << 62, size=(rows, w), dtype=np.uint64)) % np.uint64(P)
pcs = prover.PcsData(dev, [host], 1, stream)
pt = np.array([[(i * 7919 + 13) % P, (i * 104729 + 17) % P] for i in range(n)], dtype=np.uint64)
evals = np.zeros((w, 2), dtype=np.uint64)
for c in range(w):
    evals[c] = pcs.witness_mle(0, c).evaluate(pt)
for rep in range(2):
    print("---- open", rep, file=sys.stderr)
    pcs.basefold_open([pt], [evals], 100, 16, prover.Transcript.stub(3))
