#!/bin/bash
# round-2 profile set on ONE box: bench line (with extras), rocprofv3 kernel stats + FETCH/WRITE PMC of the headline, per-round
# SQ counters of the generic kernel on the batched main sumcheck (k_gen) and the same with the two-kernel path
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1; tail -3 gpurun_out/refresh.log | cut -c1-300
bash tools/pmc_gen.sh > gpurun_out/pmc_gen.json 2> gpurun_out/pmc_gen.err
CENO_HIP_NO_GEN=1 bash tools/pmc_batched.sh > gpurun_out/pmc_batched_legacy.json 2> gpurun_out/pmc_batched_legacy.err
for i in 1 2 3; do python3 tools/bench_batched.py | tail -1; done > gpurun_out/batched_gen.txt
for i in 1 2 3; do CENO_HIP_NO_GEN=1 python3 tools/bench_batched.py | tail -1; done > gpurun_out/batched_legacy.txt
rm -rf gpurun_out/tr_batched3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_batched3 -- python3 tools/bench_batched.py --reps 3 > gpurun_out/tr_batched3.log 2>&1
cat gpurun_out/batched_gen.txt | cut -c 200-; cat gpurun_out/batched_legacy.txt | cut -c 200-
