"""chip flow, then the shard flow: digests of every artifact (to diff a failing configuration against a passing one)"""
import hashlib, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ceno_amd import Device, prover, synthetic
dev = Device(0)
new_tr = lambda: prover.Transcript.stub(7)
fork = lambda: prover.Transcript.stub(0xF0)
def dg(x):
    return hashlib.sha1(np.ascontiguousarray(np.asarray(x, dtype=np.uint64)).tobytes()).hexdigest()[:10]
if os.environ.get("PRE", "1") == "1":
    flow = synthetic.ChipFlow(dev, prover, 20, 22)
    for _ in range(2): flow.run(new_tr)
    flow.close()
shard = synthetic.ShardFlow(dev, prover)
lanes = int(os.environ.get("LANES", "1"))
for rep in range(2):
    try:
        shard.run(new_tr, fork, lanes=lanes)
        st = "ok"
    except Exception as e:
        st = "FAILED"
    a = getattr(shard, "pre_open", None)
    print("rep", rep, st)
    if a:
        print(" roots", dg(np.array(a["roots"])), "alpha", a["alpha"])
        for i, p in enumerate(a["chip_proofs"]):
            print("  chip", i, "tower_msgs", dg(p.tower_msgs), "prod_evals", dg(p.tower_prod_evals), "logup", dg(p.tower_logup_evals), "rt", dg(p.rt_main))
        print(" main msgs", dg(a["msgs"]), "evals", dg(a["evals"]), "rt", dg(a["rt"]))
