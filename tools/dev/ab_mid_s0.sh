#!/bin/bash
# A/B: slices of the persistent mid rounds (CENO_HIP_MID_S0 pairs per workgroup, 256 workgroups) together with the round size at which a dense
# sumcheck hands over to them (CENO_HIP_DENSE_LADDER_LOG): can k_mid take the rounds of 2^16 / 2^17 pairs from the per-round launches?
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "15 128" "16 256" "17 512" "16 512"; do
  set -- $cfg
  echo -n "ladder from 2^$1 pairs, slices of $2: nv22 "; CENO_HIP_DENSE_LADDER_LOG=$1 CENO_HIP_MID_S0=$2 timeout 120 python3 bench.py --nv 22 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],4))"
  echo -n "                                          nv26 "; CENO_HIP_DENSE_LADDER_LOG=$1 CENO_HIP_MID_S0=$2 timeout 120 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],4))"
done; done
