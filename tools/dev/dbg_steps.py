"""per-step wall time of back-to-back sumchecks on the same tables: python tools/dev/dbg_steps.py [nv] [steps]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ceno_amd import Device, prover
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 22
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = Device(0)
mles = [dev.synthetic(nv, True, 0xCE10 + j) for j in range(3)]
dev.sync()
ONE = np.array([[1, 0]], dtype=np.uint64)
ts = []
for i in range(steps):
    t0 = time.perf_counter()
    prover.sumcheck_prove(dev, mles, ONE, [[0, 1, 2]], nv, 3, prover.Transcript.poseidon2(b"riscv"))
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.2f}" for t in ts))
print("pool:", dev.mem_stats() if hasattr(dev, "mem_stats") else "")
