"""The opening's per-phase and per-round host times on the wide shard (CENO_HIP_DEBUG=1 prints them from basefold_open)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ceno_amd import Device, prover, synthetic
dev = Device(0)
flow = synthetic.ShardFlowWide(dev, prover, log_cycles=int(os.environ.get("LOG_CYCLES", "20")))
new_tr = lambda: prover.Transcript.poseidon2(b"riscv")
fork = lambda: prover.Transcript.poseidon2(b"fork")
for _ in range(3):
    flow.run(new_tr, fork, lanes=8)
os.environ["CENO_HIP_DEBUG"] = "1"
sys.stderr.write("==== debug run\n")
r = flow.run(new_tr, fork, lanes=8)
os.environ.pop("CENO_HIP_DEBUG")
print(r)
flow.close()
