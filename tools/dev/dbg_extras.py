#!/usr/bin/env python3
"""Reproduce bench.py's extra measurements alone (same order: nv22, batched main 26 / 24, chip flow, shard lanes) with the library's
own error text and the lanes trace on stderr."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("CENO_LANES_TRACE", "1")
import bench
from ceno_amd import Device, prover

dev = Device(0)
new_tr = lambda: prover.Transcript.poseidon2(b"bench")
try:
    out = bench.extra_measurements(dev, prover, new_tr, "poseidon2", reps=int(os.environ.get("REPS", "2")))
    print({k: (v.get("ms") or v.get("total_ms")) if isinstance(v, dict) else v for k, v in out.items()})
except Exception:
    traceback.print_exc()
    from ceno_amd import _lib
    try:
        print("hip last error:", _lib.lib().ceno_hip_last_error(dev.h))
    except Exception as e:
        print("no error text:", e)
