#!/bin/bash
# A/B of the two field-arithmetic formulations on ONE box: the shipped build (GL_ARITH64=1), then a rebuild with the
# 32-bit carry chains (GL_ARITH64=0).  Prints merkle / bench / batched / chip-flow numbers for both.
cd $GRAFT_REPO_ROOT
# whatever happens, leave the shipped (default-flag) build behind: later tests / benches / profiles must not run a variant
trap 'env -u CENO_HIP_EXTRA_FLAGS python -m ceno_amd.build --force > gpurun_out/ab_restore.log 2>&1' EXIT
run() {
  echo "== $1"
  python tools/bench_merkle.py | cut -c1-40
  python bench.py --steps 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms', r['ms_per_step'], 'frac', r['roofline']['frac'])"
  python tools/bench_batched.py 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('batched ms', r['batched_main_sumcheck_ms'])"
  python tools/bench_chip.py 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print({k: round(v,3) for k,v in r.items() if k.endswith('_ms')})"
}
run "shipped build"
CENO_HIP_EXTRA_FLAGS="${AB_FLAGS:--DGL_ARITH64=0}" python -m ceno_amd.build --force > gpurun_out/ab_build.log 2>&1 || tail -5 gpurun_out/ab_build.log
run "rebuilt with ${AB_FLAGS:--DGL_ARITH64=0}"
