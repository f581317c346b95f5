"""pool growth over repeated shard flows with 1 / 4 / 8 lanes (a block last used on a busy stream is not handed to another one:
does the cache keep growing?)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ceno_amd import Device, prover, synthetic
dev = Device(0)
new_tr = lambda: prover.Transcript.stub(7)
fork = lambda: prover.Transcript.stub(0xF0)
flow = synthetic.ShardFlow(dev, prover)
for it in range(24):
    lanes = (1, 4, 8)[it % 3]
    flow.run(new_tr, fork, lanes=lanes)
    if it % 3 == 2:
        mi = dev.mem_info()
        print(it, "lanes", lanes, "pool_used MB", mi["pool_used"] >> 20, "pool_cached MB", mi["pool_cached"] >> 20)
flow.close()
