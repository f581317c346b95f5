#!/usr/bin/env python3
"""Merkle commit timing of a 2^21 x 22 codeword (chip-flow size): best of N, leaf hash + tree."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ceno_amd import Device, api

dev = Device(0)
log_rows, w = 21, 22
cw = dev.synthetic((( w << log_rows) - 1).bit_length(), False, 0xC0DE)
best = 1e9
root = None
for _ in range(6):
    dev.sync(); t0 = time.perf_counter()
    t = api.Merkle(dev, cw.device_ptr, log_rows, w)
    dev.sync(); best = min(best, (time.perf_counter() - t0) * 1e3)
    root = t.root(); t.free()
print(json.dumps({"merkle_ms": best, "root": [int(x) for x in root], "canonical": os.environ.get("CENO_HIP_P2_CANONICAL", "0")}))
