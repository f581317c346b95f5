#!/bin/bash
# kernel counts and durations of the shard flow at 1 and 4 lanes (do lanes push sumchecks off the persistent k_mid ladder?)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for L in 1 4; do
  rm -rf gpurun_out/lk_$L
  LANES=$L REPS=3 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lk_$L -- python3 tools/bench_shard.py stub > gpurun_out/lk_$L.log 2>&1
  echo "== lanes $L rc $?"
  f=$(find gpurun_out/lk_$L -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep "k_mid\|k_tile\|k_tail\|k_tower<\|k_accum\|k_fold_batch\|k_gen" $f | awk -F'","' '{printf "%-60s calls %5s total %8.2f ms avg %8.1f us\n", substr($1,2,58), $2, $3/1e6, $4/1e3}'
done
