#!/bin/bash
# A/B: the round size at which a dense sumcheck hands over from k_dense launches to the persistent ladder (CENO_HIP_DENSE_LADDER_LOG)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 15 16 17 18; do
  echo -n "ladder from 2^$v pairs: nv22 "; CENO_HIP_DENSE_LADDER_LOG=$v python3 bench.py --nv 22 --steps 30 --warmup 5 --no-extra 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],4))"
  echo -n "                         nv26 "; CENO_HIP_DENSE_LADDER_LOG=$v python3 bench.py --steps 10 --warmup 3 --no-extra 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],4))"
done; done
