#!/bin/bash
# A/B: k_gen for the largest rounds of a single chip's main sumcheck (CENO_HIP_GEN_PIPE_MIN_LOG = rounds with at least 2^v pairs)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "== two-kernel"; python3 tools/dev/dbg_tower_only.py 20 2>/dev/null | tail -1
  for v in 19 18 17 16; do echo "== k_gen from 2^$v pairs"; CENO_HIP_GEN_PIPE_MIN_LOG=$v python3 tools/dev/dbg_tower_only.py 20 2>/dev/null | tail -1; done
done
