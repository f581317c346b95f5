# A/B of the per-degree launches of the large eq-factored rounds (CENO_HIP_GEN_BY_DEGREE): wide batch, 12-column batch, the 2^20-cycle shard
mkdir -p gpurun_out/bydeg
for m in 1 0; do
  echo "== BY_DEGREE=$m: wide"
  CENO_HIP_GEN_BY_DEGREE=$m CENO_HIP_PLAN_REPORT=2 python3 tools/bench_batched_wide.py --reps 4 2>gpurun_out/bydeg/wide_$m.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print([round(x,2) for x in r['runs_ms']], r['eq_launches_per_sumcheck'])"
  grep "eq launch of round 1" gpurun_out/bydeg/wide_$m.err
  echo "== BY_DEGREE=$m: 12-column"
  CENO_HIP_GEN_BY_DEGREE=$m python3 tools/bench_batched.py --reps 4 2>&1 | tail -1 | cut -c1-400
done
