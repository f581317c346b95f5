#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cohort_trace.log
: > $out
for SUB in 0; do
echo "== CENO_TOWER_COHORT_SUB=$SUB" >> $out
CENO_TOWER_COHORT_SUB=$SUB CENO_COHORT_TRACE=1 LANES=8 REPS=3 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep "cohort" | tail -18 | cut -c1-150 >> $out
done
cat $out
