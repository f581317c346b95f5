#!/bin/bash
# A/B of bench_batched on ONE box: shipped build vs a rebuild with $AB_FLAGS
cd $GRAFT_REPO_ROOT
# whatever happens, leave the shipped (default-flag) build behind: later tests / benches / profiles must not run a variant
trap 'env -u CENO_HIP_EXTRA_FLAGS python -m ceno_amd.build --force > gpurun_out/ab_restore.log 2>&1' EXIT
run() { echo "== $1"; for i in 1 2 3; do python tools/bench_batched.py 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('batched ms', r['batched_main_sumcheck_ms'])"; done; }
run "shipped build"
CENO_HIP_EXTRA_FLAGS="$AB_FLAGS" python -m ceno_amd.build --force > gpurun_out/ab_build.log 2>&1 || tail -5 gpurun_out/ab_build.log
run "rebuilt with $AB_FLAGS"
