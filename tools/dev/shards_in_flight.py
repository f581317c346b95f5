#!/usr/bin/env python3
"""K synthetic shards (M2 shape) proved concurrently on ONE GPU, each by its own host thread on its own context (pool, lane streams):
the compute-bound commit of one shard overlaps the latency-bound chip proofs / opening of another.  Prints ms per shard (wall / K)."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover, synthetic

tname = os.environ.get("TR", "stub")
new_tr = (lambda: prover.Transcript.poseidon2(b"riscv")) if tname == "poseidon2" else (lambda: prover.Transcript.stub(0x5A))
fork = (lambda: prover.Transcript.poseidon2(b"fork")) if tname == "poseidon2" else (lambda: prover.Transcript.stub(0xF0))
lanes = int(os.environ.get("LANES", "4"))
reps = int(os.environ.get("REPS", "6"))
for K in [int(x) for x in os.environ.get("KS", "1,2,3").split(",")]:
    devs = [Device(0) for _ in range(K)]
    flows = [synthetic.ShardFlow(d, prover) for d in devs]
    for f in flows:
        f.run(new_tr, fork, lanes=lanes)
    best = None
    for _ in range(reps):
        bar = threading.Barrier(K + 1)
        def work(f):
            bar.wait()
            for _ in range(2):
                f.run(new_tr, fork, lanes=lanes)
        ts = [threading.Thread(target=work, args=(f,)) for f in flows]
        for t in ts: t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in ts: t.join()
        dt = (time.perf_counter() - t0) * 1e3 / (2 * K)
        best = dt if best is None else min(best, dt)
    print(json.dumps({"shards_in_flight": K, "lanes_each": lanes, "transcript": tname, "ms_per_shard": round(best, 3)}), flush=True)
    for f in flows: f.close()
    for d in devs: d.close()
