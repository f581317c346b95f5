#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover, synthetic
dev = Device(0)
tr = lambda: prover.Transcript.poseidon2(b"riscv")
flow = synthetic.ChipFlow(dev, prover, 20, 22)
for i in range(5):
    sys.stderr.write(f"=== run {i}\n"); sys.stderr.flush()
    r = flow.run(tr)
    print(i, {k: v >> 20 for k, v in dev.mem_info().items()}, flush=True)
flow.close()
