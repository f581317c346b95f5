#!/usr/bin/env python3
"""nv = 26 dense sumchecks, one / two / three independent instances in flight (one host thread + stream each): aggregate ext-mults/s"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from ceno_amd import Device, prover

dev = Device(0)
K, NV = 3, int(os.environ.get("NV", "26"))
one = np.array([[1, 0]], dtype=np.uint64)
for n_inst in (1, 2, 3):
    insts = [[dev.synthetic(NV, True, 77 + 10 * t + j) for j in range(K)] for t in range(n_inst)]
    streams = [dev.stream_create() for _ in range(n_inst)]
    reps = 6
    def work(t):
        for _ in range(reps):
            prover.sumcheck_prove(dev, insts[t], one, [list(range(K))], NV, K, prover.Transcript.poseidon2(b"bench"), stream=streams[t])
    work_threads = lambda: [threading.Thread(target=work, args=(t,)) for t in range(n_inst)]
    for th in work_threads()[:1]:
        th.start(); th.join()
    dev.sync()
    ths = work_threads()
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dev.sync()
    dt = time.perf_counter() - t0
    mults = K * K * ((1 << NV) - 1) * reps * n_inst
    print(f"{n_inst} in flight: {dt / (reps * n_inst) * 1e3:.3f} ms per sumcheck, {mults / dt:.4g} ext-mults/s")
    for row in insts:
        for m in row: m.free()
    for s in streams: dev.stream_destroy(s)
