#!/usr/bin/env python3
"""Config #4 at the reference's plan statistics on ONE MI355X: the batched main-constraint sumcheck over 48 chips of 22..96 base columns,
1..3 Prefix selectors and 60..250 monomials each (ceno_amd/synthetic.py wide_batched_jobs; shapes: gkr_iop/src/gkr/layer/zerocheck_layer.rs:86-207,
ceno_zkvm/src/instructions.rs:48-83, ceno_zkvm/src/scheme/gpu/mod.rs:2811-2982).  Prints ONE JSON line: wall time, which round kernels every size
class was given (CENO_HIP_PLAN_REPORT), eq-factored launches, term / table counts."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CENO_HIP_PLAN_REPORT", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-nv", type=int, default=24)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--transcript", choices=["stub", "poseidon2"], default="poseidon2")
    args = ap.parse_args()
    from ceno_amd import Device, prover, synthetic

    dev = Device(0)
    jobs, chips, elems = synthetic.wide_batched_jobs(dev, args.max_nv)
    mj = prover.MainJobs(jobs)
    new_tr = (lambda: prover.Transcript.poseidon2(b"riscv")) if args.transcript == "poseidon2" else (lambda: prover.Transcript.stub(5))
    gch = [(11, 22), (33, 44)]
    best, runs = 1e9, []
    before = dev.L.ceno_hip_stat_eq_launches(dev.h)
    for _ in range(args.reps):
        dev.sync()
        t0 = time.perf_counter()
        prover.prove_batched_main_constraints(dev, mj, gch, new_tr())
        dev.sync()
        runs.append((time.perf_counter() - t0) * 1e3)
    best = min(runs)
    launches = (dev.L.ceno_hip_stat_eq_launches(dev.h) - before) // args.reps
    report = json.loads(dev.L.ceno_hip_plan_report(dev.h).decode() or "[]")
    n_terms = sum(len(j["terms"]) for j in jobs)
    lin = sum(sum(1 for t in j["terms"] if len(t) == 2) for j in jobs)
    out = {"workload": f"prove_batched_main_constraints, 48 chips of 2^{max(1, args.max_nv - 12)}..2^{args.max_nv} rows x 22..96 base columns, 1..3 Prefix "
                       "selectors, 42..250 monomials per chip (selector x column for every column, selector x constant, products of 2..4 columns)",
           "chips": len(jobs), "tables": sum(len(j["mles"]) for j in jobs), "monomials": n_terms, "selector_x_column_monomials": lin,
           "max_degree": max(j["max_degree"] for j in jobs), "table_elements": elems, "ms": best, "runs_ms": runs,
           "eq_launches_per_sumcheck": int(launches), "ext_mult_equivalents": synthetic.eq_form_mult_equivalents(jobs, max(j["max_degree"] for j in jobs)), "algorithmic_bytes": synthetic.wide_algorithmic_bytes(args.max_nv),
           "classes": report}
    print(json.dumps(out))
    dev.close()


if __name__ == "__main__":
    main()
