#!/bin/bash
# the opening's sumcheck: one launch per round for all matrices (default) against one handle per height group (CENO_BASEFOLD_OPEN_ROUNDS=0),
# same box, alternating; the parity tests first
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 280 python -m pytest tests/test_gpu_basefold.py tests/test_gpu_shard_wide.py tests/test_gpu_flows.py -x -q -m gpu --timeout 250 --timeout-method thread < /dev/null 2>&1 | tail -5
for i in 1 2 3; do
  for g in 0 1; do
    echo "== CENO_BASEFOLD_OPEN_ROUNDS=$g"
    CENO_BASEFOLD_OPEN_ROUNDS=$g LANES=8 REPS=5 timeout 120 python tools/bench_shard_wide.py < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in d if k.endswith('_ms')})"
  done
done
timeout 100 python tools/dev/open_rounds.py 2>&1 < /dev/null | grep -a "basefold_open"
