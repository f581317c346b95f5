#!/bin/bash
# k_gen vs the two-kernel generic path on one box: parity tests, then the batched main sumcheck and the chip flow both ways
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_flows.py -m gpu -x -q -k "lds_blocked or random_plans or batched or generic or frontload or common_factor or config" 2>&1 | tail -3
for i in 1 2; do python3 tools/bench_batched.py | tail -1 | cut -c 200-; done
CENO_HIP_NO_GEN=1 python3 tools/bench_batched.py | tail -1 | cut -c 200-
python3 tools/bench_chip.py | tail -1 | cut -c 150-420
CENO_HIP_NO_GEN=1 python3 tools/bench_chip.py | tail -1 | cut -c 150-420
rm -rf gpurun_out/tr_batched2
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_batched2 -- python3 tools/bench_batched.py --reps 2 > gpurun_out/tr_batched2.log 2>&1; tail -1 gpurun_out/tr_batched2.log | cut -c 200-
