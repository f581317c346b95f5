#!/bin/bash
# kernel-trace bench runs under tuning variants and print the first rounds of the last sumcheck (GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export $v
  rm -rf gpurun_out/kt
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 2 --warmup 1 > gpurun_out/kt.log 2>&1
  echo "== $v"; python tools/round_trace.py gpurun_out/kt 37 | sed -n 8,13p | cut -c1-20,70-110
  timeout 200 python bench.py --steps 10 2>&1 | tail -1 | cut -c50-110,180-215
done
