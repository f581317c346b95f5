#!/usr/bin/env python3
"""independent dense sumchecks in flight (one host thread + stream each): ms per sumcheck for 1 / 2 / 3 instances, under the switches given in the
environment (CENO_HIP_DENSE_LADDER, CENO_HIP_NO_PIPELINE, CENO_HIP_MID_W, ...).  usage: flight_ab.py [nv] [stub|poseidon2] [per]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 26
trn = sys.argv[2] if len(sys.argv) > 2 else "poseidon2"
per = int(sys.argv[3]) if len(sys.argv) > 3 else 4
K = 3
dev = Device(0)
one = np.array([[1, 0]], dtype=np.uint64)
new_tr = (lambda: prover.Transcript.poseidon2(b"riscv")) if trn == "poseidon2" else (lambda: prover.Transcript.stub(5))
if os.environ.get("FLIGHT_PREFILL") == "1":
    # what bench.py has done to the pool before its in-flight extra: the headline at nv = 26 on the default stream, then config #2
    for nvp in (26, 22):
        tabs = [dev.synthetic(nvp, True, 0xCE10 + j) for j in range(K)]
        for _ in range(4):
            prover.sumcheck_prove(dev, tabs, one, [list(range(K))], nvp, K, new_tr())
        for m in tabs:
            m.free()
    print("after prefill:", dev.mem_info(), flush=True)
out = {}
for n_inst in (1, 2, 3):
    insts = [[dev.synthetic(nv, True, 0xCE10 + 100 * (t + 1) + j) for j in range(K)] for t in range(n_inst)]
    streams = [dev.stream_create() for _ in range(n_inst)]
    times = [[] for _ in range(n_inst)]

    def work(t):
        for _ in range(per):
            t0 = time.perf_counter()
            prover.sumcheck_prove(dev, insts[t], one, [list(range(K))], nv, K, new_tr(), stream=streams[t])
            times[t].append((time.perf_counter() - t0) * 1e3)

    work(0)
    times[0].clear()
    dev.sync()
    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_inst)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    dev.sync()
    dt = time.perf_counter() - t0
    out[n_inst] = dt / (per * n_inst) * 1e3
    print(f"nv={nv} {trn} in_flight={n_inst}: {out[n_inst]:.3f} ms per sumcheck (aggregate); per-call ms by thread: " +
          " | ".join(",".join(f"{x:.2f}" for x in ts) for ts in times), flush=True)
    print("   pool:", dev.mem_info(), flush=True)
    for row in insts:
        for m in row:
            m.free()
    for s in streams:
        dev.stream_destroy(s)
dev.close()
