#!/bin/bash
# A/B sweep of environment switches on the wide shard flow: one line per setting (chip-proof phase and total, ms)
run() { echo -n "$* : "; env "$@" REPS=${REPS:-4} timeout 80 python tools/bench_shard_wide.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lanes'], round(d['chip_proofs_ms'],2), round(d['total_ms'],2))"; }
for L in 4 8; do
run CENO_HIP_MAX_LANES=$L LANES=$L X=1
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_HIP_NO_PIPELINE=1
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_HIP_PIPE_LOOKAHEAD=1
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_HIP_MID_W=0
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_HIP_MID_W=64
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_TOWER_HOST_LAYERS=10
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_TOWER_HOST_LAYERS=11
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_HIP_VRAM_MAILBOX=0
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_HIP_TAIL_PAIRS=512
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_HIP_TOWER_FAST_MIN_LOG=13
done
