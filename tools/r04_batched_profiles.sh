#!/bin/bash
# round-4 evidence for the batched main-constraint sumcheck (config #4 shape) on ONE box: wall times with the eq-factored rounds on / off
# (CENO_HIP_GEN_EQF) at max_nv 24 and 26, rocprofv3 kernel stats, and SQ counters per round kernel launch for both forms (separate
# --pmc passes, no trace domains).  Outputs: gpurun_out/r04/r04_batched_main.json + r04_batched_main_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o; rm -rf $o/bm_*
for rep in 1 2 3; do
  python3 tools/bench_batched.py --reps 4 2>/dev/null | tail -1 >> $o/bm_wall_eqf1.jsonl
  CENO_HIP_GEN_EQF=0 python3 tools/bench_batched.py --reps 4 2>/dev/null | tail -1 >> $o/bm_wall_eqf0.jsonl
done
python3 tools/bench_batched.py --max-nv 26 --reps 3 2>/dev/null | tail -1 > $o/bm_wall26_eqf1.jsonl
CENO_HIP_GEN_EQF=0 python3 tools/bench_batched.py --max-nv 26 --reps 3 2>/dev/null | tail -1 > $o/bm_wall26_eqf0.jsonl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/bm_kt -- python3 tools/bench_batched.py --reps 3 > $o/bm_kt.log 2>&1
cp $(ls $o/bm_kt/*/*kernel_stats.csv | head -1) $o/r04_batched_main_kernel_stats.csv
python3 tools/round_trace.py $o/bm_kt 36 > $o/bm_rounds.txt
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $o/bm_pmc1 -- python3 tools/bench_batched.py --reps 1 > $o/bm_pmc1.log 2>&1
CENO_HIP_GEN_EQF=0 timeout 300 rocprofv3 --pmc $C --output-format csv -d $o/bm_pmc0 -- python3 tools/bench_batched.py --reps 1 > $o/bm_pmc0.log 2>&1
python3 - $o <<'PY'
import csv, sys, json, glob, collections
o = sys.argv[1]
def walls(f):
    return [json.loads(l)["batched_main_sumcheck_ms"] for l in open(f) if l.strip()]
def pmc(d):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(x in k for x in ("k_gen", "k_eq_base0", "k_accum_base0", "k_tile")): continue
        per.setdefault((int(r["Dispatch_Id"]), k), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    out = []
    for (disp, k), v in sorted(per.items()):
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        out.append({"kernel": k, "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"), "SQ_WAVES": v.get("SQ_WAVES"), "SQ_BUSY_CYCLES": v.get("SQ_BUSY_CYCLES"),
                    "valu_active_frac_of_wave_cycles": round(v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4),
                    "wait_inst_frac": round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 4), "wait_any_frac": round(v.get("SQ_WAIT_ANY", 0) / wc, 4)})
    return out
p1, p0 = pmc(o + "/bm_pmc1"), pmc(o + "/bm_pmc0")
tot = lambda p: sum(x["SQ_INSTS_VALU"] or 0 for x in p)
res = {
 "workload": "config #4 shape (tools/bench_batched.py): prove_batched_main_constraints, 24 chips of 14..24 variables (max_nv 26: 16..26), 12 base columns + Prefix selector, 16 terms of degree <= 4 per chip, stub transcript, one MI355X",
 "wall_ms_max_nv24": {"eq_factored (default)": walls(o + "/bm_wall_eqf1.jsonl"), "CENO_HIP_GEN_EQF=0 (generic rounds, occupancy-aware grid)": walls(o + "/bm_wall_eqf0.jsonl")},
 "wall_ms_max_nv26": {"eq_factored (default)": walls(o + "/bm_wall26_eqf1.jsonl"), "CENO_HIP_GEN_EQF=0": walls(o + "/bm_wall26_eqf0.jsonl")},
 "SQ_INSTS_VALU_total_one_sumcheck": {"eq_factored": tot(p1), "generic": tot(p0), "ratio": round(tot(p1) / max(tot(p0), 1), 4)},
 "SQ_INSTS_VALU_per_chip_pair_round1": "round 1 = 8,159,232 pairs over the 24 chips (2^22 + 2*2^20 + 5*2^18 + 8*2^16 + 8*2^12): SQ_INSTS_VALU is per wave (64 pairs)",
 "per_launch_eq_factored (first 10 round kernels)": p1[:10],
 "per_launch_generic (first 10 round kernels)": p0[:10],
 "round_durations_us (rocprofv3 --kernel-trace, last sumcheck)": [l.rstrip() for l in open(o + "/bm_rounds.txt")][:30],
}
json.dump(res, open(o + "/r04_batched_main.json", "w"), indent=1)
print(json.dumps({k: res[k] for k in ("wall_ms_max_nv24", "wall_ms_max_nv26", "SQ_INSTS_VALU_total_one_sumcheck")}))
PY
