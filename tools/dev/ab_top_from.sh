#!/bin/bash
# A/B of the Merkle level at which the 8-lanes-per-permutation kernel takes over (CENO_HIP_MERKLE_TOP_FROM_LOG): tree build alone and the chip flow
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in 14 15 16 17; do
  echo "== top_from $v"
  CENO_HIP_MERKLE_TOP_FROM_LOG=$v python3 tools/bench_merkle.py 2>/dev/null | cut -c1-40
  CENO_HIP_MERKLE_TOP_FROM_LOG=$v python3 tools/dev/dbg_tower_only.py 20 2>/dev/null | tail -1
done
done
