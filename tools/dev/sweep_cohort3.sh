#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cohort_sweep3.log
: > $out
CENO_COHORT_TRACE=1 LANES=8 REPS=2 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep "cohort" | tail -19 >> $out
cat $out
