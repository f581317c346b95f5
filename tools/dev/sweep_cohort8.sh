#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cohort_sweep8.log
: > $out
run() { echo "== $*" >> $out; env "$@" LANES=8 REPS=8 timeout 100 python tools/bench_shard_wide.py 2>&1 | grep -v "population\|WARNING" | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print({k: r[k] for k in ('chip_proofs_ms','chip_proofs_native_ms','total_ms')})
" >> $out; }
for L in 18 19 20 22 18 19 20 22; do run CENO_TOWER_COHORT_LAYERS=$L; done
cat $out
