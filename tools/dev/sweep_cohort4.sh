#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cohort_sweep4.log
: > $out
run() { echo "== $*" >> $out; env "$@" LANES=8 REPS=4 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep -v "population\|WARNING" | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print({k: r[k] for k in ('lanes','witgen_ms','commit_ms','chip_proofs_ms','batched_main_ms','open_ms','total_ms')})
" >> $out; }
run CENO_TOWER_COHORT_LAYERS=0
run CENO_TOWER_COHORT_LAYERS=16
run CENO_TOWER_COHORT_LAYERS=16 CENO_COHORT_THREADS=16
run CENO_TOWER_COHORT_LAYERS=16 CENO_COHORT_THREADS=16 CENO_TOWER_HOST_LAYERS=6
echo "== trace" >> $out
CENO_COHORT_TRACE=1 LANES=8 REPS=2 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep "chip proofs in" | tail -4 >> $out
cat $out
