"""per-round host-observed latency of the DENSE path (one product of three ext tables) at NV variables, CENO_HIP_DEBUG=1"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ceno_amd import Device, prover
dev = Device(0)
nv = int(os.environ.get("NV", "18"))
tabs = [dev.synthetic(nv, True, j) for j in range(3)]
one = np.array([[1, 0]], dtype=np.uint64)
tr = (lambda k: prover.Transcript.poseidon2(b"x")) if os.environ.get("TR", "stub") == "poseidon2" else (lambda k: prover.Transcript.stub(k))
for k in range(3):
    sys.stderr.write(f"--- sumcheck {k}\n")
    prover.sumcheck_prove(dev, tabs, one, [[0, 1, 2]], nv, 3, tr(k))
