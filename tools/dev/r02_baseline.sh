#!/bin/bash
# round-2 baseline on one box: GPU tests, then kernel traces (CSV kept) of the batched main sumcheck and the chip flow
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest0.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_pytest0.log
rm -rf gpurun_out/tr_batched gpurun_out/tr_chip
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_batched -- python3 tools/bench_batched.py --reps 2 > gpurun_out/tr_batched.log 2>&1; tail -1 gpurun_out/tr_batched.log
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_chip -- python3 tools/bench_chip.py --reps 2 > gpurun_out/tr_chip.log 2>&1; tail -1 gpurun_out/tr_chip.log
python3 tools/bench_batched.py | tail -1
python3 tools/bench_chip.py | tail -1
