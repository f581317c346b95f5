#!/usr/bin/env python3
"""Experiment: does the generic accumulate kernel slow down when its tables sit at power-of-two strides
(limb-major tower layers) compared with separately placed tables?  Times round 0 of the logup-layer plan."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ceno_amd import Device, api

dev = Device(0)
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << nv
P = api.P
terms = [[1, 4], [2, 3], [3, 4]]
coeffs = np.array([[3, 5], [7, 11], [13, 17]], dtype=np.uint64)
groups = [([0], [0, 1, 2])]

def run(mles, label):
    best = 1e9
    for _ in range(4):
        sc = api.Sumcheck(dev, mles, coeffs, terms, nv, 3, groups=groups)
        torch.cuda.synchronize(); dev.sync()
        t0 = time.perf_counter()
        sc.round(None)
        dt = (time.perf_counter() - t0) * 1e6
        best = min(best, dt)
        t0 = time.perf_counter()
        sc.round((5, 7))
        dt1 = (time.perf_counter() - t0) * 1e6
        sc.free()
    print(f"{label:40s} round0 {best:8.1f} us   round1 {dt1:8.1f} us")

for pad in (0, 16, 256, 4096 + 16):
    stride = n + pad  # ext elements
    buf = torch.empty(5 * stride * 2 + 64, dtype=torch.int64, device="cuda:0")
    big = dev.synthetic(nv + 3, True, 1)  # source of random canonical words
    # fill: copy canonical data into the strided slots
    src = torch.empty(0)
    mles = []
    for j in range(5):
        ptr = buf.data_ptr() + 16 * stride * j
        # device-to-device copy through torch views
        s = big.device_ptr + 16 * n * j
        dst_view = buf[2 * stride * j: 2 * stride * j + 2 * n]
        import ctypes
        src_t = torch.empty(0)
        # use hipMemcpy via torch: build a tensor alias of the source with from_blob-like trick
        dst_view.copy_(torch.tensor([], dtype=torch.int64, device="cuda:0").new_empty(2 * n).set_(torch.cuda.LongStorage.from_buffer if False else dst_view.untyped_storage(), dst_view.storage_offset(), (2 * n,))) if False else None
        mles.append((ptr, s))
    # simple path: launch the library fill on each slot instead of copying
    handles = []
    for j, (ptr, s) in enumerate(mles):
        h = dev.wrap(ptr, nv, True)
        dev.check(dev.L.ceno_hip_mle_fill_splitmix(dev.h, h.h, 100 + j, 0, None))
        handles.append(h)
    dev.sync()
    run(handles, f"one buffer, limb stride 2^{nv} + {pad} elems")
sep = [dev.synthetic(nv, True, 100 + j) for j in range(5)]
run(sep, "five pool allocations")
