"""ONE 2^20-row chip (config #3's: 22 columns, 4 + 4 + 8 records) through the plural entry: the per-chip tower prover (CENO_TOWER_COHORT_MIN_TASKS unset)
against cohorts for its layers 6 .. CENO_TOWER_COHORT_LAYERS (CENO_TOWER_COHORT_MIN_TASKS=1).  Prints the best of REPS and checks the proofs are equal."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ceno_amd import Device, prover, synthetic

dev = Device(0)
flow = synthetic.ChipFlow(dev, prover, log_rows=int(os.environ.get("LOG_ROWS", "20")), w=22)
rows, w = flow.rows, flow.w
n = rows.bit_length() - 1
pcs = prover.PcsData(dev, None, flow.log_blowup, flow.stream, device_ptrs=[(flow.trace.device_ptr, rows, w)])
alpha, beta = (3, 5), (7, 11)
cols = [pcs.witness_mle(0, c) for c in range(w)]
coeffs, terms, out_terms = synthetic.record_plan(w, 16, alpha, beta)
task = dict(mles=cols, n_witin=w, n_fixed=0, n_structural=0, num_instances=rows - 3, log2_num_instances=n, num_reads=4, num_writes=4,
            num_lk_tables=0, num_lk=8, record_coeffs=coeffs, record_terms=terms, record_out_terms=out_terms)
ct = prover.ChipTasks([task])
ref = None
for setting in os.environ.get("SETTINGS", "0:0,1:11,1:13,1:15,1:19").split(","):
    mn, last = setting.split(":")
    if mn == "1":
        os.environ["CENO_TOWER_COHORT_MIN_TASKS"] = "1"
        os.environ["CENO_TOWER_COHORT_LAYERS"] = last
    else:
        os.environ.pop("CENO_TOWER_COHORT_MIN_TASKS", None)
        os.environ.pop("CENO_TOWER_COHORT_LAYERS", None)
    best = None
    for _ in range(int(os.environ.get("REPS", "6"))):
        dev.sync()
        t0 = time.perf_counter()
        proofs = prover.create_chip_proofs(dev, ct, [alpha, beta], [prover.Transcript.poseidon2(b"fork")], 1)
        dev.sync()
        ms = (time.perf_counter() - t0) * 1e3
        best = ms if best is None or ms < best else best
    words = np.concatenate([np.asarray(proofs[0].tower_msgs).ravel(), np.asarray(proofs[0].rt_main).ravel()])
    if ref is None:
        ref = words
    print(json.dumps({"cohorts": mn == "1", "last_cohort_layer": int(last), "chip_proof_ms": round(best, 3), "same_proof": bool(np.array_equal(ref, words))}), flush=True)
