#!/bin/bash
# host-tail budget sweep after the grouped host evaluation: chip flow at 2^20 rows
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "host_finished or tower or sumcheck" 2>&1 | tail -2
CENO_HIP_DEBUG=1 python3 tools/dev/dbg_tower_only.py 14 2>&1 | grep "host round" | tail -12
for rep in 1 2; do for ns in 10500 14000 18000 24000; do echo -n "budget $ns ns: "; CENO_HIP_HOST_TAIL_NS=$ns python3 tools/dev/dbg_tower_only.py 20 2>/dev/null | tail -1; done; done
