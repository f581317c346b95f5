# per-degree launches x column-block size: does a third workgroup per CU pay on the kernels of length 3 / 4 (168 / 164 registers: three waves per SIMD)?
mkdir -p gpurun_out/bydeg
for m in 20 19 17 15 13; do
  echo "== SPLIT_MLES=$m"
  CENO_HIP_GEN_SPLIT_MLES=$m CENO_HIP_PLAN_REPORT=2 python3 tools/bench_batched_wide.py --reps 4 2>gpurun_out/bydeg/split_$m.err | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print([round(x,2) for x in r['runs_ms']], r['eq_launches_per_sumcheck'])"
  grep "eq launch of round 1" gpurun_out/bydeg/split_$m.err | head -3
done
