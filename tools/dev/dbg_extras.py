"""bench.py's extra measurements one at a time, each in a try block, to locate a failing flow"""
import os, sys, traceback
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ceno_amd import Device, prover, synthetic
dev = Device(0)
tn = os.environ.get("TR", "poseidon2")
new_tr = (lambda: prover.Transcript.poseidon2(b"riscv")) if tn == "poseidon2" else (lambda: prover.Transcript.stub(7))
fork = (lambda: prover.Transcript.poseidon2(b"fork")) if tn == "poseidon2" else (lambda: prover.Transcript.stub(0xF0))
steps = os.environ.get("STEPS", "nv22,batched,chip,shard").split(",")
for name in steps:
    try:
        if name == "nv22":
            m22 = [dev.synthetic(22, True, 5 + j) for j in range(3)]
            prover.sumcheck_prove(dev, m22, np.array([[1, 0]], dtype=np.uint64), [[0, 1, 2]], 22, 3, new_tr())
            for m in m22: m.free()
        elif name == "batched":
            jobs, elems = synthetic.batched_jobs(dev, 24, 12)
            mj = prover.MainJobs(jobs)
            prover.prove_batched_main_constraints(dev, mj, [(11, 22), (33, 44)], new_tr())
            for j in jobs:
                for m in j["mles"]:
                    if m is not None: m.free()
        elif name == "chip":
            flow = synthetic.ChipFlow(dev, prover, 20, 22)
            for _ in range(2): r = flow.run(new_tr)
            flow.close()
        elif name == "shard":
            shard = synthetic.ShardFlow(dev, prover)
            for lanes in (1, 4):
                for _ in range(2): r = shard.run(new_tr, fork, lanes=lanes)
            shard.close()
        print(name, "ok")
    except Exception as e:
        print(name, "FAILED", repr(e)[:200])
