#!/bin/bash
# A/B: workgroups of the persistent mid rounds (CENO_HIP_MID_W) on the chip flow, the shard and the dense sumchecks
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for w in 256 128; do
  echo -n "MID_W=$w: chip (tower main open total) "; CENO_HIP_MID_W=$w timeout 100 python tools/bench_chip.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['tower_prove_ms'],3), round(d['main_sumcheck_ms'],3), round(d['open_ms'],3), round(d['total_ms'],3), end='  ')"
  echo -n "shard "; CENO_HIP_MID_W=$w LANES=4 REPS=6 timeout 100 python tools/bench_shard.py stub 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['chip_proofs_ms'], d['total_ms'], end='  ')"
  echo -n "nv22 "; CENO_HIP_MID_W=$w timeout 120 python3 bench.py --nv 22 --steps 30 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],4), end='  ')"
  echo -n "nv26 "; CENO_HIP_MID_W=$w timeout 120 python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_step'],4))"
done; done
