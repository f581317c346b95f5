#!/usr/bin/env python3
"""S-batched — BASELINE.json config #4 shape on ONE MI355X: one batched main-constraint sumcheck over
24 "chips" of mixed sizes (front-load rule), ADD-shaped plans (base witness columns, a Prefix selector per chip,
degree <= 4).  Prints wall time and the ext-mult count per the SURVEY formula (per chip: pairs x sum of (deg_t + 1) points).
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-nv", type=int, default=24)
    ap.add_argument("--width", type=int, default=12)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from ceno_amd import Device, api, prover

    dev = Device(0)
    P = api.P
    sizes = [args.max_nv, args.max_nv - 2, args.max_nv - 2] + [args.max_nv - 4] * 5 + [args.max_nv - 6] * 8 + [args.max_nv - 10] * 8
    w = args.width
    jobs, total_elems = [], 0
    for c, nv in enumerate(sizes):
        cols = [dev.synthetic(nv, False, 1000 + 50 * c + j) for j in range(w)]
        point = np.array([[(i * 7919 + 13 + c) % P, (i * 104729 + 17) % P] for i in range(nv)], dtype=np.uint64)
        sel = (1, 0, (1 << nv) - 5 - c, 0, (), 0, point)  # Prefix selector
        s_id = w
        terms = [[s_id, j, (j + 1) % w] for j in range(w)] + [[s_id, j, (j + 3) % w, (j + 5) % w] for j in range(0, w, 3)]
        scalars = [[((3 + t, 1), [2 + (t % 2)])] for t in range(len(terms))]
        jobs.append(dict(num_vars=nv, mles=cols + [None], n_witin=w, n_fixed=0, n_structural=1, selectors=[sel], n_exprs=2,
                         max_degree=4, terms=terms, scalars=scalars))
        total_elems += (w + 1) << nv
    gch = [(11, 22), (33, 44)]
    jobs = prover.MainJobs(jobs)  # marshalled once: the timing below is the library's, not the binding's
    best = 1e9
    for _ in range(args.reps):
        torch.cuda.synchronize(); dev.sync()
        t0 = time.perf_counter()
        prover.prove_batched_main_constraints(dev, jobs, gch, prover.Transcript.stub(5))
        dev.sync()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    from ceno_amd import synthetic
    plain = [dict(num_vars=nv, mles=[None] * (w + 1), n_witin=w, n_fixed=0, n_structural=1,
                  terms=[[w, j, (j + 1) % w] for j in range(w)] + [[w, j, (j + 3) % w, (j + 5) % w] for j in range(0, w, 3)]) for nv in sizes]
    print(json.dumps({"ext_mult_equivalents": synthetic.eq_form_mult_equivalents(plain, 4), "chips": len(sizes), "num_vars": sizes, "width": w, "table_elements": total_elems, "batched_main_sumcheck_ms": best}))


if __name__ == "__main__":
    main()
