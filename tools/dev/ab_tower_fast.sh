#!/bin/bash
# A/B of the fused tower-layer rounds (sumcheck_tower.hip): chip flow at 2^20 rows, hand-over points 15..17, and the generic rounds
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== generic"; CENO_HIP_TOWER_FAST=0 python3 tools/dev/dbg_tower_only.py 20 2>/dev/null | tail -1
  for v in 14 15 16 17; do echo "== fused from 2^$v pairs"; CENO_HIP_TOWER_FAST_MIN_LOG=$v python3 tools/dev/dbg_tower_only.py 20 2>/dev/null | tail -1; done
done
CENO_PROVER_LAYER_TRACE=1 python3 tools/dev/dbg_tower_only.py 20 2>&1 | grep "tower layer" | tail -14
