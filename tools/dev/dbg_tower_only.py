"""one tower proof (chip-flow shape, 2^LOG rows) for kernel-trace timelines: python tools/dev/dbg_tower_only.py [log_rows]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ceno_amd import Device, prover, synthetic
dev = Device(0)
log_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12
flow = synthetic.ChipFlow(dev, prover, log_rows=log_rows)
for _ in range(3):
    r = flow.run(lambda: prover.Transcript.stub(1))
print({k: round(v, 3) for k, v in r.items() if k.endswith("_ms")})
flow.close()
