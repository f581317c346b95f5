#!/usr/bin/env python3
"""config #3's commitment alone (2^20 rows x 22 base columns, blow-up 2): `reps` times commit_traces — for a --pmc pass over the commit kernels"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = Device(0)
rows, w = 1 << 20, 22
trace = dev.synthetic((rows * w - 1).bit_length(), False, 0xADD)
st = dev.stream_create()
best = 1e9
for _ in range(reps):
    dev.sync(); t0 = time.perf_counter()
    pcs = prover.PcsData(dev, None, 1, st, device_ptrs=[(trace.device_ptr, rows, w)])
    dev.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
    pcs.free()
print(json.dumps({"commit_ms": best, "reps": reps}))
