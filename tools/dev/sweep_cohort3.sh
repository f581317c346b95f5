#!/bin/bash
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cohort_sweep3.log
: > $out
for T in 8 16; do
echo "== threads $T" >> $out
CENO_COHORT_THREADS=$T CENO_COHORT_TRACE=1 LANES=8 REPS=2 timeout 300 python tools/bench_shard_wide.py 2>&1 | grep "cohort" | tail -17 >> $out
done
nproc >> $out; cat /sys/fs/cgroup/cpu.max >> $out 2>&1
cat $out
