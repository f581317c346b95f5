#!/usr/bin/env python3
"""N identical small chips on 1 / N lanes: do latency-bound chip proofs slow each other down when no large kernel is around?
LOGS="13,13,13,13" LANES="1,4" python tools/dev/lanes_interference.py   (CENO_LANES_TRACE=1 for per-task times)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ceno_amd import Device, prover, synthetic

dev = Device(0)
logs = [int(x) for x in os.environ.get("LOGS", "13,13,13,13").split(",")]
flow = synthetic.ShardFlow(dev, prover, log_rows=logs)
new_tr = lambda: prover.Transcript.stub(0x5A)
fork = lambda: prover.Transcript.stub(0xF0)
for lanes in [int(x) for x in os.environ.get("LANES", "1,4").split(",")]:
    best = None
    for rep in range(int(os.environ.get("REPS", "4"))):
        print(f"=== lanes {lanes} rep {rep}", file=sys.stderr, flush=True)
        r = flow.run(new_tr, fork, lanes=lanes)
        if best is None or r["chip_proofs_ms"] < best["chip_proofs_ms"]:
            best = r
    print(json.dumps({"logs": logs, "lanes": lanes, "chip_proofs_ms": round(best["chip_proofs_ms"], 3)}), flush=True)
flow.close()
