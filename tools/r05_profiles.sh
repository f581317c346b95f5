#!/bin/bash
# round-5 profile set on ONE box: the bench line (with extras), rocprofv3 kernel stats + FETCH / WRITE PMC of the headline, kernel stats of the
# chip flow and of the shard flow, the wide batched main sumcheck (tools/r05_wide_profiles.sh) and the VALU counters of the VALU-bound extras
# (tools/r05_valu_counters.sh).  Outputs: gpurun_out/r05/out/r05_*
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r05; mkdir -p $o/out; rm -rf $o/trace $o/pmc_fetch $o/pmc_write $o/chip_kt $o/shard_kt
bash tools/r05_valu_counters.sh > $o/valu.log 2>&1
cp gpurun_out/r05v/r05_valu_counters.json $o/out/
cp gpurun_out/r05v/r05_valu_counters.json profiles/r05_valu_counters.json   # (bench.py reads it from profiles/ on this box too)
timeout 500 python3 bench.py > $o/bench_n1.json 2> $o/bench.err
cp $o/bench_n1.json $o/out/r05_bench_n1.json
for i in 1 2 3; do timeout 300 python3 bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null >> $o/out/r05_bench_runs.jsonl; done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $o/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $o/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $o/pmc_write.log 2>&1
python3 tools/pmc_summary.py $(dirname $(ls $o/trace/*/*kernel_stats.csv | head -1)) $(dirname $(ls $o/pmc_fetch/*/*counter_collection.csv | head -1)) $(dirname $(ls $o/pmc_write/*/*counter_collection.csv | head -1)) 0 $o/out/r05_sumcheck_nv26
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/chip_kt -- python3 tools/bench_chip.py > $o/chip_kt.log 2>&1
LANES=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/shard_kt -- python3 tools/bench_shard.py poseidon2 > $o/shard_kt.log 2>&1
cp $(ls $o/chip_kt/*/*kernel_stats.csv | head -1) $o/out/r05_chip_flow_kernel_stats.csv
cp $(ls $o/shard_kt/*/*kernel_stats.csv | head -1) $o/out/r05_shard_flow_kernel_stats.csv
LANES=1,2,4,8 python3 tools/bench_shard.py poseidon2 2>/dev/null | grep lanes > $o/out/r05_shard_lanes.jsonl
python3 tools/bench_chip.py 2>/dev/null | tail -1 > $o/out/r05_chip_flow.json
bash tools/r05_wide_profiles.sh > $o/wide.log 2>&1
cp gpurun_out/r05w/r05_batched_main_wide.json gpurun_out/r05w/r05_batched_main_wide_kernel_stats.csv $o/out/
cp gpurun_out/r05w/wide_rounds.txt $o/out/r05_batched_main_wide_rounds.txt
ls -la $o/out/
