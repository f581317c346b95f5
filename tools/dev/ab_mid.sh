#!/bin/bash
# sweep of the persistent mid-round geometry on one box: W (workgroups) x S0 cap (pairs per workgroup in the first round)
cd $GRAFT_REPO_ROOT
for cfg in "0 128" "64 128" "128 64" "256 32" "256 128" "128 128" "64 64" "0 128" "64 128"; do
  set -- $cfg
  CENO_HIP_MID_W=$1 CENO_HIP_MID_S0=$2 python tools/bench_chip.py 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('W=$1 S0=$2', {k: round(v,3) for k,v in r.items() if k in ('tower_prove_ms','main_sumcheck_ms','open_ms','total_ms')})"
done
for cfg in "0 128" "64 128" "256 32"; do
  set -- $cfg
  CENO_HIP_MID_W=$1 CENO_HIP_MID_S0=$2 NVS=12,16 python tools/bench_lane_rounds.py 2>/dev/null | tail -1 | cut -c1-400
done
