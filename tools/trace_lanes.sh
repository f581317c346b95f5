#!/bin/bash
# HIP API statistics of the shard flow with 1 and 8 chip-proof lanes: which runtime calls inflate under concurrency?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for L in 1 8; do
  o=gpurun_out/tr_lanes$L; rm -rf $o
  LANES=$L timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $o -- python3 tools/bench_shard.py stub > $o.log 2>&1
  f=$(ls $o/*/*hip_api_stats.csv 2>/dev/null | head -1)
  echo "== lanes $L"; tail -2 $o.log | cut -c1-200
  [ -n "$f" ] && head -14 "$f" | cut -d, -f1-7
done
