#!/bin/bash
# SQ_INSTS_VALU of one run of every VALU-bound extra of bench.py (separate --pmc passes, no trace domains) -> gpurun_out/r06v/r06_valu_counters.json,
# which bench.py reads (from profiles/) to state each extra's VALU-issue fraction: insts x 4.3 cycles / (1024 SIMDs x 2.4 GHz x time)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r06v; mkdir -p $o; rm -rf $o/*
C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES"
timeout 300 rocprofv3 --pmc $C --output-format csv -d $o/narrow24 -- python3 tools/bench_batched.py --reps 1 > $o/narrow24.log 2>&1
timeout 300 rocprofv3 --pmc $C --output-format csv -d $o/narrow26 -- python3 tools/bench_batched.py --max-nv 26 --reps 1 > $o/narrow26.log 2>&1
timeout 300 rocprofv3 --pmc $C --output-format csv -d $o/wide24 -- python3 tools/bench_batched_wide.py --reps 1 > $o/wide24.log 2>&1
timeout 300 rocprofv3 --pmc $C --output-format csv -d $o/commit -- python3 tools/dev/commit_only.py 1 > $o/commit.log 2>&1
timeout 300 rocprofv3 --pmc $C --output-format csv -d $o/dense26 -- python3 bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $o/dense26.log 2>&1
LANES=8 REPS=1 timeout 200 rocprofv3 --pmc $C --output-format csv -d $o/widesh -- python3 tools/bench_shard_wide.py poseidon2 > $o/widesh.log 2>&1
python3 - $o <<'PY'
import csv, sys, json, glob, collections
o = sys.argv[1]
def total(d):
    f = sorted(glob.glob(o + "/" + d + "/**/*counter_collection.csv", recursive=True))[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if "k_fill_splitmix" in k: continue
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    tot = collections.defaultdict(float)
    for k, v in per.items():
        for c, x in v.items(): tot[c] += x
    top = {k: v["SQ_INSTS_VALU"] for k, v in sorted(per.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:10]}
    # rocprof's VALUBusy over the kernels' own active time: 4 x SQ_ACTIVE_INST_VALU (quad-cycles, MI355X_MICROARCH.md) / (SIMDs x GRBM_GUI_ACTIVE)
    busy = 4.0 * tot["SQ_ACTIVE_INST_VALU"] / (1024.0 * tot["GRBM_GUI_ACTIVE"]) if tot["GRBM_GUI_ACTIVE"] else None
    return dict(tot), top, busy
res = {"what": "wave-level VALU instructions (SQ_INSTS_VALU, rocprofv3 --pmc) of ONE run of each workload, input generation excluded",
       "cycles_per_valu_inst": 4.3, "cycles_per_valu_inst_source": "profiles/r01_valu_issue_rates.txt (v_mad_u64_u32, carries, cndmask: 4.3 cycles per wave64 instruction)",
       "simds": 1024, "clock_ghz": 2.4}
for key, d in (("batched_main", "narrow24"), ("batched_main_nv26", "narrow26"), ("batched_main_wide", "wide24"), ("chip_flow.commit", "commit"), ("shard_e2e_wide", "widesh")):
    try:
        t, top, busy = total(d)
    except Exception as e:  # (a pass that did not finish must not take the others down)
        res[key] = {"error": repr(e)}
        continue
    res[key] = {"SQ_INSTS_VALU": t["SQ_INSTS_VALU"], "SQ_ACTIVE_INST_VALU_quad_cycles": t["SQ_ACTIVE_INST_VALU"], "GRBM_GUI_ACTIVE_cycles": t["GRBM_GUI_ACTIVE"],
                "VALUBusy_over_kernel_time": busy, "largest_kernels": top}
# the headline's kernel: wave-level VALU instructions of ONE nv = 26 sumcheck = all k_dense dispatches / the number of round-0 dispatches
f = sorted(glob.glob(o + "/dense26/**/*counter_collection.csv", recursive=True))[0]
tot, r0, fold, r0v = 0.0, set(), 0.0, 0.0
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not k.startswith("void k_dense") or r["Counter_Name"] != "SQ_INSTS_VALU": continue
    tot += float(r["Counter_Value"])
    if k.startswith("void k_dense<3, 0"):
        r0.add(r["Dispatch_Id"]); r0v += float(r["Counter_Value"])
    else:
        fold += float(r["Counter_Value"])
if r0:
    res["sumcheck_nv26"] = {"SQ_INSTS_VALU": tot / len(r0), "round0_SQ_INSTS_VALU": r0v / len(r0), "fold_rounds_SQ_INSTS_VALU": fold / len(r0), "sumchecks_counted": len(r0),
                            "fold_rounds_valu_issue_frac": 0.86, "fold_rounds_valu_issue_frac_source": "profiles/r05_dense_overlap.json (kernel unchanged since)"}
import hashlib, os
h = hashlib.sha256()
root = os.environ.get("GRAFT_REPO_ROOT", ".")
for fn in sorted(glob.glob(root + "/ceno_amd/csrc/*")):
    if os.path.isfile(fn):
        h.update(os.path.basename(fn).encode()); h.update(open(fn, "rb").read())
res["csrc_sha16"] = h.hexdigest()[:16]   # bench.py uses the counts only while the tree's kernel sources hash to this
json.dump(res, open(o + "/r06_valu_counters.json", "w"), indent=1)
print(json.dumps({k: (v.get("SQ_INSTS_VALU"), v.get("VALUBusy_over_kernel_time")) for k, v in res.items() if isinstance(v, dict)}))
PY
