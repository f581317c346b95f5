#!/bin/bash
cd $GRAFT_REPO_ROOT
f='import json,sys; d=json.loads(sys.stdin.read()); print({k: round(d[k],3) for k in ("tower_prove_ms","main_sumcheck_ms","open_ms","commit_ms","total_ms")})'
for i in 1 2; do
echo "== default"; python tools/bench_chip.py 2>/dev/null | tail -1 | python3 -c "$f"
echo "== GEN_PIPE_MIN_LOG=13"; CENO_HIP_GEN_PIPE_MIN_LOG=13 python tools/bench_chip.py 2>/dev/null | tail -1 | python3 -c "$f"
echo "== GEN_PIPE_MIN_LOG=16"; CENO_HIP_GEN_PIPE_MIN_LOG=16 python tools/bench_chip.py 2>/dev/null | tail -1 | python3 -c "$f"
echo "== NO_PIPELINE + GEN (eq-factored, launch per round)"; CENO_HIP_GEN_PIPE_MIN_LOG=13 CENO_HIP_NO_PIPELINE=1 python tools/bench_chip.py 2>/dev/null | tail -1 | python3 -c "$f"
done
