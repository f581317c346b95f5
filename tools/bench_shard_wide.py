#!/usr/bin/env python3
"""Metric M2 on a shard with the reference's population (ceno_amd/synthetic.py ShardFlowWide): 2^LOG_CYCLES cycles over 45 opcode circuits
with on-device witness generation, two wide circuits, seven table circuits and a fixed commitment, with LANES chip-proof lanes
(CENO_HIP_MAX_LANES caps what really runs at once).  Prints one JSON line per lane count: per-phase ms, pool high-water vs booking."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ceno_amd import Device, prover, synthetic

    dev = Device(0)
    tname = sys.argv[1] if len(sys.argv) > 1 else "poseidon2"
    new_tr = (lambda: prover.Transcript.poseidon2(b"riscv")) if tname == "poseidon2" else (lambda: prover.Transcript.stub(0x5A))
    fork = (lambda: prover.Transcript.poseidon2(b"fork")) if tname == "poseidon2" else (lambda: prover.Transcript.stub(0xF0))
    flow = synthetic.ShardFlowWide(dev, prover, log_cycles=int(os.environ.get("LOG_CYCLES", "20")))
    print(json.dumps({"population": flow.population()}))
    for lanes in [int(x) for x in os.environ.get("LANES", "1,4,8,16").split(",")]:
        best = None
        for _ in range(int(os.environ.get("REPS", "3"))):
            r = flow.run(new_tr, fork, lanes=lanes)
            dev.L.ceno_hip_host_timing_dump(b"one run of the wide shard")  # (prints only under CENO_HIP_HOST_TIMING=1)
            if best is None or r["total_ms"] < best["total_ms"]:
                best = r
        flow.free_last()
        print(json.dumps({"lanes": lanes, "max_lanes": os.environ.get("CENO_HIP_MAX_LANES", "adaptive: 8 for >= 24 tasks, else 4"), "transcript": tname,
                          **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in best.items()}}), flush=True)
    flow.close()


if __name__ == "__main__":
    main()
