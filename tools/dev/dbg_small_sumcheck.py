import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ceno_amd import Device, prover
dev = Device(0)
nv = int(os.environ.get("NV", "7"))
tabs = [dev.synthetic(nv, True, j) for j in range(4)]
coeffs = np.array([[3, 1], [5, 2]], dtype=np.uint64)
terms = [[0, 1, 2], [1, 2, 3]]
s = dev.stream_create()
for k in range(4):
    sys.stderr.write(f"--- sumcheck {k}\n")
    prover.sumcheck_prove(dev, tabs, coeffs, terms, nv, 3, prover.Transcript.stub(k), stream=s)
