#!/usr/bin/env python3
"""Determinism under concurrency: the whole shard flow with 1 lane gives the reference digests of every artifact; N runs with
4 and 8 chip-proof lanes must reproduce them bit for bit (forked transcripts make the chip proofs independent of the
interleaving).  usage: python tools/stress_lanes.py [runs_per_lane_count]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ceno_amd import Device, prover, synthetic

dev = Device(0)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
new_tr = lambda: prover.Transcript.poseidon2(b"riscv")
fork = lambda: prover.Transcript.poseidon2(b"fork")


def dg(x):
    return hashlib.sha1(np.ascontiguousarray(np.asarray(x, dtype=np.uint64)).tobytes()).hexdigest()[:12]


def digest(a):
    parts = [dg(np.array(a["roots"]))]
    for p in a["chip_proofs"]:
        parts += [dg(p.tower_msgs), dg(p.tower_prod_evals), dg(p.tower_logup_evals), dg(p.rt_main)]
    parts += [dg(a["msgs"]), dg(a["evals"]), dg(a["rt"]), dg(a["open_proof"])]
    return parts


flow = synthetic.ShardFlow(dev, prover)
flow.run(new_tr, fork, lanes=1)
ref = digest(flow.artifacts)
bad = 0
for lanes in (1, 4, 8):
    for k in range(runs):
        flow.run(new_tr, fork, lanes=lanes)
        d = digest(flow.artifacts)
        if d != ref:
            bad += 1
            print("MISMATCH lanes", lanes, "run", k, [i for i, (x, y) in enumerate(zip(d, ref)) if x != y])
print("runs per lane count", runs, "mismatches", bad)
flow.close()
sys.exit(1 if bad else 0)
