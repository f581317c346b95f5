#!/bin/bash
# SQ counters of the generic round kernel (k_gen) on the batched main sumcheck, per dispatch (round): separate --pmc passes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/pmc_gen; rm -rf $o $o.a $o.b
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $o.a -- python3 tools/bench_batched.py --reps 1 > $o.a.log 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $o.b -- python3 tools/bench_batched.py --reps 1 > $o.b.log 2>&1
python3 - $o.a $o.b <<'PY'
import csv, sys, json, collections, glob
out = {}
for d in sys.argv[1:]:
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    per = collections.OrderedDict()
    for r in rows:
        if "k_gen" not in r["Kernel_Name"]: continue
        per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for i, (k, v) in enumerate(sorted(per.items())):
        out.setdefault(i, {}).update(v)
for i in sorted(out)[:8]:
    v = out[i]; wc = v.get("SQ_WAVE_CYCLES", 1)
    v["valu_active_frac"] = v.get("SQ_ACTIVE_INST_VALU", 0) / wc
    v["wait_inst_frac"] = v.get("SQ_WAIT_INST_ANY", 0) / wc
    v["wait_any_frac"] = v.get("SQ_WAIT_ANY", 0) / wc
    print(json.dumps({"round": i, **v}))
PY
