#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cohort_sweep6.log
: > $out
run() { echo "== $*" >> $out; env "$@" LANES=8 REPS=6 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep -v "population\|WARNING" | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print({k: r[k] for k in ('witgen_ms','commit_ms','chip_proofs_ms','chip_proofs_native_ms','batched_main_ms','open_ms','total_ms')})
" >> $out; }
run CENO_TOWER_VIRTUAL_RECORDS=1
run CENO_TOWER_VIRTUAL_RECORDS=0
run CENO_TOWER_VIRTUAL_RECORDS=1
run CENO_TOWER_VIRTUAL_RECORDS=0
cat $out
