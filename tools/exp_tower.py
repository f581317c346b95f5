#!/usr/bin/env python3
"""Experiment: per-layer timing of a tower proof (CENO_HIP_DEBUG=1 prints begin / rounds / free per layer)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ceno_amd import Device, api, prover

dev = Device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "logup"
rows = 1 << n
recs = [dev.synthetic(n, True, 50 + j) for j in range(8)]
alpha = (0x1234567, 0x89abcde)
for rep in range(2):
    if which == "logup":
        pt, lt = [], [prover.Tower.build_logup(dev, None, recs, rows, alpha)]
    else:
        pt, lt = [prover.Tower.build_prod(dev, recs[:4], rows, (1, 0))], []
    dev.sync()
    t0 = time.perf_counter()
    prover.prove_tower_relation(dev, pt, lt, prover.Transcript.stub(1))
    dev.sync()
    print("total ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
    for x in pt + lt:
        x.free()
