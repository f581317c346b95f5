#!/bin/bash
# the chip-proof phase of the wide shard with the cohort layers off / to 13 / 16 / 18, and the phase trace of one run
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cohort_last_layer.log
: > $out
for L in ${LAYER_LIST:-0 13 16 17 18}; do
  echo "== CENO_TOWER_COHORT_LAYERS=$L" >> $out
  CENO_TOWER_COHORT_LAYERS=$L LANES=8 REPS=4 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep -v population | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: print(l.rstrip()); continue
    print({k: r[k] for k in ('lanes','witgen_ms','commit_ms','chip_proofs_ms','batched_main_ms','open_ms','total_ms')})
" >> $out
done
echo "== trace (16)" >> $out
CENO_COHORT_TRACE=1 LANES=8 REPS=2 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep -i "cohort" | tail -40 >> $out
cat $out
