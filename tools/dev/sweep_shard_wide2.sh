#!/bin/bash
run() { echo -n "$* : "; env "$@" REPS=${REPS:-5} timeout 80 python tools/bench_shard_wide.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lanes'], round(d['chip_proofs_ms'],2), round(d['total_ms'],2))"; }
for H in 8 10; do for L in 4 5 6 8 10 12; do
run CENO_HIP_MAX_LANES=$L LANES=$L CENO_TOWER_HOST_LAYERS=$H
done; done
