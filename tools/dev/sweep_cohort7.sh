#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cohort_sweep7.log
: > $out
run() { echo "== $*" >> $out; env "$@" LANES=8 REPS=6 timeout 100 python tools/bench_shard_wide.py 2>&1 | grep -v "population\|WARNING" | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print({k: r[k] for k in ('chip_proofs_ms','chip_proofs_native_ms','total_ms')})
" >> $out; }
for H in 8 6 5 4 3 2 8 5 3; do run CENO_TOWER_COHORT_HOST_LAYERS=$H; done
cat $out
