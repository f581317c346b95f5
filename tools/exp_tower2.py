#!/usr/bin/env python3
"""Experiment: round 0 of the top logup layer — library plan (tower_layer_sumcheck_begin) vs a hand-made plan over the
same layer tables vs the same plan over freshly filled tables."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ceno_amd import Device, api, prover
from ceno_amd.api import Mle, Sumcheck, _p

dev = Device(0)
n = 20
rows = 1 << n
recs = [dev.synthetic(n, True, 50 + j) for j in range(8)]
alpha = (0x1234567, 0x89abcde)
t = prover.Tower.build_logup(dev, None, recs, rows, alpha)
L = t.num_vars - 1  # top layer index
print("tower nv", t.num_vars, "top layer", L)
P = api.P
rt = np.array([[(i * 7919 + 13) % P, (i * 104729 + 17) % P] for i in range(L)], dtype=np.uint64)
alphas = np.array([[3, 5], [7, 11]], dtype=np.uint64)

def time_round0(make, label):
    best = 1e9
    for _ in range(3):
        sc = make()
        dev.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc.round(None)
        best = min(best, (time.perf_counter() - t0) * 1e6)
        sc.free()
    print(f"{label:50s} round0 {best:8.1f} us")

def lib_plan():
    h = C.c_void_p()
    arr = (C.c_void_p * 1)(t.h)
    dev.check(dev.L.ceno_hip_tower_layer_sumcheck_begin(dev.h, None, 0, arr, 1, L, _p(rt), _p(alphas), None, C.byref(h)))
    return Sumcheck(dev, [], None, [], L, 3, _handle=h)

time_round0(lib_plan, "library layer plan")
limbs = []
for s in range(4):
    h = C.c_void_p()
    dev.check(dev.L.ceno_hip_tower_layer(dev.h, t.h, L, s, C.byref(h)))
    limbs.append(Mle(dev, h))
eq = dev.eq_build(rt)
terms = [[1, 4], [2, 3], [3, 4]]
coeffs = np.array([[3, 5], [3, 5], [7, 11]], dtype=np.uint64)
groups = [([0], [0, 1, 2])]
time_round0(lambda: Sumcheck(dev, [eq] + limbs, coeffs, terms, L, 3, groups=groups), "hand plan over the tower layer tables")
fresh = [dev.synthetic(L, True, 900 + j) for j in range(4)]
time_round0(lambda: Sumcheck(dev, [eq] + fresh, coeffs, terms, L, 3, groups=groups), "hand plan over fresh random tables")
ones = dev.upload(np.tile(np.array([[1, 0]], dtype=np.uint64), (1 << L, 1)))
time_round0(lambda: Sumcheck(dev, [eq, ones, ones, fresh[2], fresh[3]], coeffs, terms, L, 3, groups=groups), "hand plan, p1 = p2 = table of ones")
