#!/usr/bin/env python3
"""Metric M2 shape: synthetic shard of 2^20 cycles through the whole create_proof flow (ceno_amd/synthetic.py ShardFlow) with
1 / 2 / 4 / 8 concurrent chip-proof lanes.  Prints one JSON line per lane count."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ceno_amd import Device, prover, synthetic

    dev = Device(0)
    tname = sys.argv[1] if len(sys.argv) > 1 else "poseidon2"
    new_tr = (lambda: prover.Transcript.poseidon2(b"riscv")) if tname == "poseidon2" else (lambda: prover.Transcript.stub(0x5A))
    fork = (lambda: prover.Transcript.poseidon2(b"fork")) if tname == "poseidon2" else (lambda: prover.Transcript.stub(0xF0))
    flow = synthetic.ShardFlow(dev, prover)
    for lanes in [int(x) for x in os.environ.get("LANES", "1,2,4,8").split(",")]:
        best = None
        for _ in range(int(os.environ.get("REPS", "3"))):
            r = flow.run(new_tr, fork, lanes=lanes)
            if best is None or r["total_ms"] < best["total_ms"]:
                best = r
        print(json.dumps({"lanes": lanes, "transcript": tname, **{k: round(v, 3) for k, v in best.items() if k.endswith("_ms")}}))
    flow.close()


if __name__ == "__main__":
    main()
