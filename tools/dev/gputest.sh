#!/bin/bash
# pytest -m gpu with the noise of RCCL's banner filtered out: prints failures and the summary
cd $GRAFT_REPO_ROOT
python -m pytest -m gpu -x -q "$@" 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -${TAIL:-12} | cut -c1-260
