#!/bin/bash
# round-5 evidence for the batched main-constraint sumcheck at the reference's plan statistics (tools/bench_batched_wide.py) on ONE box:
# wall times with / without the column-block split, rocprofv3 kernel stats + the per-launch trace of the last sumcheck, SQ counters per launch
# (separate --pmc pass, no trace domains), and the 12-column plan beside it.  Outputs under gpurun_out/r05w/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r05w; mkdir -p $o; rm -rf $o/*
python3 tools/bench_batched_wide.py --reps 4 2>/dev/null | tail -1 > $o/wide_split.json
CENO_HIP_GEN_SPLIT=0 python3 tools/bench_batched_wide.py --reps 4 2>/dev/null | tail -1 > $o/wide_nosplit.json
python3 tools/bench_batched.py --reps 4 2>/dev/null | tail -1 > $o/narrow.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -- python3 tools/bench_batched_wide.py --reps 2 > $o/kt.log 2>&1
cp $(ls $o/kt/*/*kernel_stats.csv | head -1) $o/r05_batched_main_wide_kernel_stats.csv
python3 tools/round_trace.py $o/kt 30 > $o/wide_rounds.txt
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $o/pmc_wide -- python3 tools/bench_batched_wide.py --reps 1 > $o/pmc_wide.log 2>&1
timeout 300 rocprofv3 --pmc $C --output-format csv -d $o/pmc_narrow -- python3 tools/bench_batched.py --reps 1 > $o/pmc_narrow.log 2>&1
python3 - $o <<'PY'
import csv, sys, json, glob, collections
o = sys.argv[1]
def pmc(d):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not any(x in k for x in ("k_gen", "k_eq_base0", "k_accum", "k_tile", "k_fold_batch")): continue
        per.setdefault((int(r["Dispatch_Id"]), k), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    out = []
    for (disp, k), v in sorted(per.items()):
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        out.append({"kernel": k, "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"), "SQ_WAVES": v.get("SQ_WAVES"), "SQ_BUSY_CYCLES": v.get("SQ_BUSY_CYCLES"),
                    "valu_active_frac_of_wave_cycles": round(v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4),
                    "wait_inst_frac": round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 4), "wait_any_frac": round(v.get("SQ_WAIT_ANY", 0) / wc, 4)})
    return out
res = {"wide": json.load(open(o + "/wide_split.json")), "wide_no_column_blocks": json.load(open(o + "/wide_nosplit.json")),
       "narrow_12_columns": json.load(open(o + "/narrow.json")), "pmc_wide_per_launch": pmc(o + "/pmc_wide"), "pmc_narrow_per_launch": pmc(o + "/pmc_narrow")}
for k in ("pmc_wide_per_launch", "pmc_narrow_per_launch"):
    res[k + "_total_SQ_INSTS_VALU"] = sum(x["SQ_INSTS_VALU"] or 0 for x in res[k])
# VALU lane-instructions per extension-multiplication equivalent (ceno_amd/synthetic.py eq_form_mult_equivalents): the same kernel family on both
# plans, so equal efficiency = equal numbers
res["valu_lane_insts_per_mult_equivalent"] = {
    "wide": 64 * res["pmc_wide_per_launch_total_SQ_INSTS_VALU"] / res["wide"]["ext_mult_equivalents"],
    "narrow_12_columns": 64 * res["pmc_narrow_per_launch_total_SQ_INSTS_VALU"] / res["narrow_12_columns"]["ext_mult_equivalents"]}
res["mult_equivalents_per_second"] = {"wide": res["wide"]["ext_mult_equivalents"] / (res["wide"]["ms"] * 1e-3),
                                      "narrow_12_columns": res["narrow_12_columns"]["ext_mult_equivalents"] / (res["narrow_12_columns"]["batched_main_sumcheck_ms"] * 1e-3)}
print(res["valu_lane_insts_per_mult_equivalent"], res["mult_equivalents_per_second"])
json.dump(res, open(o + "/r05_batched_main_wide.json", "w"), indent=1)
print("wide", res["wide"]["ms"], "nosplit", res["wide_no_column_blocks"]["ms"], "narrow", res["narrow_12_columns"]["batched_main_sumcheck_ms"])
print("VALU insts wide", res["pmc_wide_per_launch_total_SQ_INSTS_VALU"], "narrow", res["pmc_narrow_per_launch_total_SQ_INSTS_VALU"])
PY
cat $o/wide_rounds.txt
head -12 $o/r05_batched_main_wide_kernel_stats.csv
