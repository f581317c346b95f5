#!/usr/bin/env python3
"""Concurrent chip proving on HIP streams ("lanes", reference: ceno_zkvm/src/scheme/scheduler.rs:231-336,
docs/src/concurrent-chip-proving.md): T host threads, each with its own stream, prove independent ADD-shaped chips
(record inference -> tower build -> tower proof -> main sumcheck).  Small rounds are latency chains, so chips on
different streams overlap; prints the throughput against the one-lane run."""
import argparse, json, os, sys, threading, time
# HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4); two lanes on one queue run their
# kernels back to back — and a pipelined round kernel waiting for its challenge holds the queue meanwhile.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-rows", type=int, default=18)
    ap.add_argument("--chips", type=int, default=8)
    ap.add_argument("--lanes", type=str, default="1,2,4,8")
    ap.add_argument("--stream-step", type=int, default=1, help="use every k-th of the created streams (queue-mapping experiments)")
    args = ap.parse_args()
    import torch
    from ceno_amd import Device, api, prover

    if os.environ.get("CENO_SWITCH_INTERVAL"):
        sys.setswitchinterval(float(os.environ["CENO_SWITCH_INTERVAL"]))
    dev = Device(0)
    P = api.P
    n, w = args.log_rows, 22
    rows = 1 << n
    alpha, beta = (0x1234567, 0x89abcde), (0x13579b, 0x2468ac)
    b2 = ((beta[0] * beta[0] + 7 * beta[1] * beta[1]) % P, (2 * beta[0] * beta[1]) % P)
    terms, coeffs, out_terms = [], [], []
    for k in range(16):
        base = len(terms)
        terms += [[(2 * k) % w], [(2 * k + 1) % w], [(3 * k + 5) % w, (k + 7) % w]]
        coeffs += [beta, b2, alpha]
        out_terms.append([base, base + 1, base + 2])
    coeffs = np.array(coeffs, dtype=np.uint64)
    pt_ = np.array([[(i * 7919 + 13) % P, (i * 104729 + 17) % P] for i in range(n)], dtype=np.uint64)
    mterms = [[j, (j + 1) % w] for j in range(w)] + [[j, (j + 3) % w, (j + 5) % w] for j in range(0, w, 2)]
    mcoeffs = np.array([[(3 + 5 * i) % P, (11 * i + 1) % P] for i in range(len(mterms))], dtype=np.uint64)
    chips = [[dev.synthetic(n, False, 7000 + 100 * c + j) for j in range(w)] for c in range(args.chips)]
    dev.sync()

    def prove_chip(cols, stream, seed):
        recs = dev.wit_infer(cols, coeffs, terms, out_terms, n, stream=stream)
        reads, writes, lks = recs[:4], recs[4:8], recs[8:16]
        pt = [prover.Tower.build_prod(dev, reads, rows, (1, 0), stream=stream), prover.Tower.build_prod(dev, writes, rows, (1, 0), stream=stream)]
        lt = [prover.Tower.build_logup(dev, None, lks, rows, alpha, stream=stream)]
        prover.prove_tower_relation(dev, pt, lt, prover.Transcript.stub(seed), stream=stream)
        sel = dev.selector_build(1, pt_, 0, rows - 3, stream=stream)
        groups = [([w], list(range(len(mterms))))]
        prover.sumcheck_prove(dev, cols + [sel], mcoeffs, mterms, n, 4, prover.Transcript.stub(seed + 1), groups=groups, stream=stream)
        for x in pt + lt + recs:
            x.free()

    res = {"log_rows": n, "chips": args.chips}
    base_ms = None
    for lanes in [int(x) for x in args.lanes.split(",")]:
        pool_ = [dev.stream_create_lane(i) if os.environ.get("CENO_LANE_PRIORITIES", "1") != "0" else dev.stream_create()
                 for i in range(lanes * args.stream_step)]
        streams = pool_[::args.stream_step]
        for rep in range(2):
            todo = list(range(args.chips))
            lock = threading.Lock()
            errs = []

            def worker(li):
                try:
                    while True:
                        with lock:
                            if not todo:
                                return
                            c = todo.pop()
                        prove_chip(chips[c], streams[li], 10 * c)
                except Exception as e:  # noqa: BLE001
                    errs.append(repr(e))

            torch.cuda.synchronize(); dev.sync()
            t0 = time.perf_counter()
            ths = [threading.Thread(target=worker, args=(i,)) for i in range(lanes)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            for s_ in streams:
                dev.sync(s_)
            ms = (time.perf_counter() - t0) * 1e3
            assert not errs, errs
        res[f"lanes_{lanes}_ms"] = ms
        if base_ms is None:
            base_ms = ms
        res[f"lanes_{lanes}_speedup"] = base_ms / ms
    print(json.dumps(res))


if __name__ == "__main__":
    main()
