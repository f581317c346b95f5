#!/usr/bin/env python3
"""Fixed cost of one dense sumcheck (begin + rounds + finish + free) at tiny sizes: what every tower layer pays."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ceno_amd import Device, prover
dev = Device(0)
for nv in (1, 2, 4, 8):
    mles = [dev.synthetic(nv, True, 100 + j) for j in range(3)]
    one = np.array([[1, 0]], dtype=np.uint64)
    for rep in range(3):
        dev.sync(); t0 = time.perf_counter()
        for _ in range(50):
            prover.sumcheck_prove(dev, mles, one, [[0, 1, 2]], nv, 3, prover.Transcript.stub(7))
        dev.sync(); dt = (time.perf_counter() - t0) / 50
    print(f"nv={nv}: {dt * 1e6:.1f} us per sumcheck = {dt * 1e6 / nv:.1f} us per round")
