#!/usr/bin/env python3
"""Where do concurrent lanes stop scaling?  T host threads (one stream each) run independent small generic sumchecks through
the C++ host loop (one C call per sumcheck, so the GIL is out of the picture).  nv = 7 runs entirely inside ONE persistent
k_tail launch (no kernel boundary per round); nv = 12 / 16 add 5 / 9 k_tile rounds = kernel boundaries.  Prints the wall time
of a fixed number of sumchecks for 1, 2, 4, 8 lanes."""
import json, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ceno_amd import Device, prover
    dev = Device(0)
    total = int(os.environ.get("TOTAL", "64"))
    out = {}
    for nv in [int(x) for x in os.environ.get("NVS", "7,12,16").split(",")]:
        tabs = [[dev.synthetic(nv, True, 100 * l + j) for j in range(4)] for l in range(8)]
        coeffs = np.array([[3, 1], [5, 2]], dtype=np.uint64)
        terms = [[0, 1, 2], [1, 2, 3]]
        dev.sync()
        row = {}
        for lanes in (1, 2, 4, 8):
            streams = [dev.stream_create_lane(i) for i in range(lanes)]
            best = None
            for rep in range(3):
                errs = []

                def worker(li):
                    try:
                        for k in range(total // lanes):
                            prover.sumcheck_prove(dev, tabs[li], coeffs, terms, nv, 3, prover.Transcript.stub(k), stream=streams[li])
                    except Exception as e:  # noqa: BLE001
                        errs.append(repr(e))

                ts = [threading.Thread(target=worker, args=(i,)) for i in range(lanes)]
                t0 = time.perf_counter()
                for t in ts:
                    t.start()
                for t in ts:
                    t.join()
                dt = (time.perf_counter() - t0) * 1e3
                assert not errs, errs
                best = dt if best is None else min(best, dt)
            for s in streams:
                dev.stream_destroy(s)
            row[lanes] = round(best, 2)
        out[f"nv{nv}_ms_for_{total}_sumchecks"] = row
        out[f"nv{nv}_us_per_round_one_lane"] = round(row[1] * 1e3 / total / nv, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
