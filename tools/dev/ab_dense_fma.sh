#!/bin/bash
# A/B on one box: the dense kernel's fold as a0 + r (a1 - a0) with a separate modular add (shipped) against the fused multiply-add
cd $GRAFT_REPO_ROOT
trap 'env -u CENO_HIP_EXTRA_FLAGS python -m ceno_amd.build --force > gpurun_out/ab_restore.log 2>&1' EXIT
run() {
  echo "== $1"
  for i in 1 2 3; do python bench.py --no-extra --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'ms_per_step': round(r['ms_per_step'],4), 'stub_ms': round(r['ms_per_step_stub'],4), 'kernel_ms_per_step': round(r['roofline']['kernel_ms_per_sumcheck'],4), 'frac': round(r['roofline']['frac'],4)}))"; done
}
run "shipped"
CENO_HIP_EXTRA_FLAGS="-DCENO_DENSE_FMA=1" python -m ceno_amd.build --force > gpurun_out/ab_build.log 2>&1 || tail -5 gpurun_out/ab_build.log
run "CENO_DENSE_FMA=1"
env -u CENO_HIP_EXTRA_FLAGS python -m ceno_amd.build --force > gpurun_out/ab_build2.log 2>&1
run "shipped again"
