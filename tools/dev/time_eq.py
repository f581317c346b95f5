#!/usr/bin/env python3
"""wall time of eq_build (launch to completion, best of 20) for n = 12 .. 24"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ceno_amd import Device
dev = Device(0)
P = (1 << 64) - (1 << 32) + 1
out = {}
for n in range(12, 25):
    pt = np.array([[(i * 7919 + 13) % P, (i * 104729 + 17) % P] for i in range(n)], dtype=np.uint64)
    best = 1e9
    for _ in range(20):
        dev.sync()
        t0 = time.perf_counter()
        m = dev.eq_build(pt)
        dev.sync()
        best = min(best, (time.perf_counter() - t0) * 1e6)
        m.free()
    out[n] = round(best, 1)
print(out)
