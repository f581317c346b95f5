#!/bin/bash
# A/B of the merged k_gen launch's grid (CENO_HIP_GEN_MAXB) on the config-#4-shaped batch
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for g in 1024 768 512 640 896; do
  echo "== CENO_HIP_GEN_MAXB=$g"; CENO_HIP_GEN_MAXB=$g python3 tools/bench_batched.py --reps 4 2>/dev/null | tail -1
done; done
