#!/bin/bash
# one rocprofv3 --pmc pass (counters in $1, comma separated) over a python tool; prints per-kernel sums for kernels matching $2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ctrs=$1; pat=$2; shift 2
out=gpurun_out/pmc; rm -rf $out
timeout 600 rocprofv3 --pmc ${ctrs//,/ } --output-format csv -d $out -- python3 "$@" > gpurun_out/pmc.log 2>&1
f=$(ls $out/*/*counter_collection.csv | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        key = (r["Kernel_Name"][:40], r["Dispatch_Id"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
for (k, d), v in sorted(acc.items(), key=lambda kv: int(kv[0][1])):
    print(k, d, {n: int(x) for n, x in v.items()})
PY
