#!/usr/bin/env python3
"""where the chip-proof phase of the shard flow spends its wall time: forks (python), the C++ call, proof conversion (python)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover, synthetic
import ceno_amd.prover as P

dev = Device(0)
flow = synthetic.ShardFlow(dev, prover)
new_tr = lambda: prover.Transcript.poseidon2(b"riscv")
fork = lambda: prover.Transcript.poseidon2(b"fork")
orig = P.create_chip_proofs
T = {}
def timed_ccp(dev_, tasks, ch, forks, lanes):
    t0 = time.perf_counter()
    L = P.plib()
    import ctypes as C, numpy as np
    ct = tasks
    chs = np.array([[int(c[0]), int(c[1])] for c in ch], dtype=np.uint64)
    trs = (C.c_void_p * ct.n)(*[t.h for t in forks])
    outs = (P.ChipProofC * ct.n)()
    status = (C.c_int * ct.n)()
    t1 = time.perf_counter()
    rc = L.ceno_prover_create_chip_proofs(dev_.h, ct.arr, ct.n, P._p(chs), trs, lanes, outs, status)
    t2 = time.perf_counter()
    res = [P.ChipProof(outs[i]) for i in range(ct.n)]
    for i in range(ct.n):
        L.ceno_chip_proof_free(C.byref(outs[i]))
    t3 = time.perf_counter()
    T["marshal"], T["cpp"], T["convert"] = (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3
    return res
P.create_chip_proofs = timed_ccp
prover.create_chip_proofs = timed_ccp
for _ in range(4):
    r = flow.run(new_tr, fork, lanes=4)
    print({k: round(v, 3) for k, v in r.items() if k.endswith("_ms")}, {k: round(v, 3) for k, v in T.items()})
flow.close()
