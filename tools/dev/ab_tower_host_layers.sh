#!/bin/bash
# tower proof / chip flow against the number of host-proved layers, and a per-layer trace
cd $GRAFT_REPO_ROOT
for h in 8 9 10 11; do echo "== CENO_TOWER_HOST_LAYERS=$h"; for i in 1 2; do CENO_TOWER_HOST_LAYERS=$h python3 tools/bench_chip.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ('tower_prove_ms','main_sumcheck_ms','open_ms','commit_ms','total_ms')})"; done; done
CENO_PROVER_LAYER_TRACE=1 python3 tools/bench_chip.py 2>&1 | grep "tower layer" | tail -24
