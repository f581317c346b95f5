#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cohort_sweep2.log
: > $out
run() { echo "== $*" >> $out; env "$@" LANES=8 REPS=4 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep -v "population\|WARNING" | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print({k: r[k] for k in ('lanes','witgen_ms','commit_ms','chip_proofs_ms','batched_main_ms','open_ms','total_ms')})
" >> $out; }
run CENO_COHORT_THREADS=4
run CENO_COHORT_THREADS=12
run CENO_COHORT_THREADS=16
run CENO_COHORT_THREADS=16 CENO_HIP_MAX_LANES=12
run CENO_COHORT_THREADS=16 CENO_TOWER_HOST_LAYERS=6
run CENO_COHORT_THREADS=16 CENO_TOWER_HOST_LAYERS=10
echo "== chip trace" >> $out
CENO_PROVER_CHIP_TRACE=1 CENO_COHORT_TRACE=1 LANES=8 REPS=2 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep "chip 2\|chip proofs in" | tail -56 >> $out
cat $out
