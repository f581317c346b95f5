#!/bin/bash
# round-3 profile set on ONE box: bench line (with extras), rocprofv3 kernel stats + FETCH/WRITE PMC of the headline, kernel stats
# of the chip flow and of the shard flow (four lanes on the C++ scheduler), per-primitive roofline table
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1; tail -3 gpurun_out/refresh.log | cut -c1-300
rm -rf gpurun_out/r03_chip_kt gpurun_out/r03_shard_kt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_chip_kt -- python3 tools/bench_chip.py > gpurun_out/r03_chip_kt.log 2>&1
LANES=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_shard_kt -- python3 tools/bench_shard.py poseidon2 > gpurun_out/r03_shard_kt.log 2>&1
cp $(ls gpurun_out/r03_chip_kt/*/*kernel_stats.csv | head -1) gpurun_out/refresh/out/r03_chip_flow_kernel_stats.csv
cp $(ls gpurun_out/r03_shard_kt/*/*kernel_stats.csv | head -1) gpurun_out/refresh/out/r03_shard_flow_kernel_stats.csv
python3 tools/roofline_table.py > gpurun_out/refresh/out/r03_kernel_roofline_table.json 2> gpurun_out/roofline_table.err
LANES=1,2,4,8 python3 tools/bench_shard.py poseidon2 2>/dev/null | grep lanes > gpurun_out/refresh/out/r03_shard_lanes.jsonl
python3 tools/bench_chip.py 2>/dev/null | tail -1 > gpurun_out/refresh/out/r03_chip_flow.json
ls -la gpurun_out/refresh/out/
