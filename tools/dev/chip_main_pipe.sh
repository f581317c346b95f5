# the single chip's main-constraint sumcheck (config #3: 2^20 rows x 22 columns, 33 terms of degree <= 4): component tables + eq-factored rounds
# (CENO_HIP_GEN_PIPE_MIN_LOG) against the two-kernel pipelined path
for m in none 20 16 14 12; do
  echo -n "GEN_PIPE_MIN_LOG=$m: "
  if [ $m = none ]; then python3 tools/bench_chip.py 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['main_sumcheck_ms'], r['tower_prove_ms'], r['total_ms'])"
  else CENO_HIP_GEN_PIPE_MIN_LOG=$m python3 tools/bench_chip.py 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['main_sumcheck_ms'], r['tower_prove_ms'], r['total_ms'])"; fi
done
