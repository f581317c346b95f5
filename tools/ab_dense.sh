#!/bin/bash
# A/B of the dense sumcheck kernel variants on one box: CENO_HIP_DENSE_PF = 0 (compiler-placed loads) / 1 / 2 / 3 (software-pipelined)
for pf in 0 3 0 3 1 2; do
  echo "== CENO_HIP_DENSE_PF=$pf"
  CENO_HIP_DENSE_PF=$pf python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'ms_per_step': round(r['ms_per_step'],4), 'stub_ms': round(r['ms_per_step_stub'],4), 'kernel_ms_per_step': round(r['roofline']['avg_launch_ms']*r['roofline']['launches']/r['steps'],4), 'frac': round(r['roofline']['frac'],4)}))"
done
