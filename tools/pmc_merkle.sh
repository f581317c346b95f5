#!/bin/bash
# SQ counters of the Merkle leaf hash (is it VALU-issue bound?): one --pmc pass, no trace domains
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/pmc_merkle; rm -rf $o
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $o -- python3 tools/bench_merkle.py > $o.log 2>&1
f=$(ls $o/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, json, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "leaf_hash" in k or k.startswith("k_compress"):
        d = dict(v)
        wc = d.get("SQ_WAVE_CYCLES", 0)
        out = {"kernel": k, **{a: b for a, b in d.items()}}
        if wc:
            out["valu_active_fraction_of_wave_cycles"] = d.get("SQ_ACTIVE_INST_VALU", 0) / wc
            out["wait_inst_any_fraction"] = d.get("SQ_WAIT_INST_ANY", 0) / wc
            out["wait_any_fraction"] = d.get("SQ_WAIT_ANY", 0) / wc
        print(json.dumps(out))
PY
