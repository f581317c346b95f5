#!/usr/bin/env python3
"""the Basefold opening of the config-#3 chip (2^20 x 22, blow-up 2, 100 queries, 16-bit PoW), with the per-round host timeline
(CENO_HIP_DEBUG=1 prints it)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ceno_amd import Device, prover, synthetic

d = Device(0)
w, n = 22, int(os.environ.get("LOG", "20"))
rows = 1 << n
trace = d.synthetic((rows * w - 1).bit_length(), False, 0xADD)
st = d.stream_create()
pcs = prover.PcsData(d, None, 1, st, device_ptrs=[(trace.device_ptr, rows, w)])
tr = prover.Transcript.poseidon2(b"x")
rt = np.array([[(i * 7919 + 13) % synthetic.P, (i * 104729 + 17) % synthetic.P] for i in range(n)], dtype=np.uint64)
cols = [pcs.witness_mle(0, c) for c in range(w)]
evals = np.array([list(c.evaluate(rt)) for c in cols], dtype=np.uint64)
best = 1e9
for _ in range(5):
    t = prover.Transcript.poseidon2(b"x")
    d.sync()
    t0 = time.perf_counter()
    pcs.basefold_open([rt], [evals], 100, 16, t)
    d.sync()
    best = min(best, (time.perf_counter() - t0) * 1e3)
print(f"open 2^{n} x {w}: {best:.3f} ms")
