#!/bin/bash
# round-6 profile set on ONE box: the bench line (with extras), rocprofv3 kernel stats + FETCH / WRITE PMC of the headline, kernel stats of the
# chip flow, of the eight-chip shard and of the WIDE shard (the reference's population: tools/bench_shard_wide.py) with its lane sweep, the wide
# batched main sumcheck's kernel stats, and the VALU counters stamped with the kernel sources' hash (tools/r06_valu_counters.sh).
# Outputs: gpurun_out/r06/out/r06_*  (copy to profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r06; mkdir -p $o/out; rm -rf $o/trace $o/pmc_fetch $o/pmc_write $o/chip_kt $o/shard_kt $o/wide_kt $o/bmw_kt
bash tools/r06_valu_counters.sh > $o/valu.log 2>&1
cp gpurun_out/r06v/r06_valu_counters.json $o/out/
cp gpurun_out/r06v/r06_valu_counters.json profiles/r06_valu_counters.json   # (bench.py reads it from profiles/ on this box too)
timeout 500 python3 bench.py > $o/bench_n1.json 2> $o/bench.err
cp $o/bench_n1.json $o/out/r06_bench_n1.json
for i in 1 2 3; do timeout 300 python3 bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null >> $o/out/r06_bench_runs.jsonl; done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $o/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $o/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $o/pmc_write.log 2>&1
python3 tools/pmc_summary.py $(dirname $(ls $o/trace/*/*kernel_stats.csv | head -1)) $(dirname $(ls $o/pmc_fetch/*/*counter_collection.csv | head -1)) $(dirname $(ls $o/pmc_write/*/*counter_collection.csv | head -1)) 0 $o/out/r06_sumcheck_nv26
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/chip_kt -- python3 tools/bench_chip.py > $o/chip_kt.log 2>&1
LANES=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/shard_kt -- python3 tools/bench_shard.py poseidon2 > $o/shard_kt.log 2>&1
LANES=8 REPS=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/wide_kt -- python3 tools/bench_shard_wide.py poseidon2 > $o/wide_kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/bmw_kt -- python3 tools/bench_batched_wide.py --reps 2 > $o/bmw_kt.log 2>&1
cp $(ls $o/chip_kt/*/*kernel_stats.csv | head -1) $o/out/r06_chip_flow_kernel_stats.csv
cp $(ls $o/shard_kt/*/*kernel_stats.csv | head -1) $o/out/r06_shard_flow_kernel_stats.csv
cp $(ls $o/wide_kt/*/*kernel_stats.csv | head -1) $o/out/r06_shard_wide_kernel_stats.csv
cp $(ls $o/bmw_kt/*/*kernel_stats.csv | head -1) $o/out/r06_batched_main_wide_kernel_stats.csv
LANES=1,2,4,8 python3 tools/bench_shard.py poseidon2 2>/dev/null | grep lanes > $o/out/r06_shard_lanes.jsonl
LANES=1,4,8,16 REPS=5 python3 tools/bench_shard_wide.py poseidon2 2>/dev/null > $o/out/r06_shard_wide_lanes.jsonl
# the per-chip tower prover (round 5's path, CENO_TOWER_COHORT_LAYERS=0): lanes, the lane cap, where a chip's time goes
rm -f $o/out/r06_shard_wide_lane_cap_sweep.jsonl $o/out/r06_cohort_last_layer.txt
CENO_TOWER_COHORT_LAYERS=0 LANES=1,4,8,16 REPS=5 python3 tools/bench_shard_wide.py poseidon2 2>/dev/null > $o/out/r06_shard_wide_lanes_per_chip_prover.jsonl
for L in 4 6 8 10 12; do CENO_TOWER_COHORT_LAYERS=0 CENO_HIP_MAX_LANES=$L LANES=$L REPS=5 python3 tools/bench_shard_wide.py poseidon2 2>/dev/null | tail -1 >> $o/out/r06_shard_wide_lane_cap_sweep.jsonl; done
CENO_TOWER_COHORT_LAYERS=0 CENO_LANES_TRACE=1 CENO_PROVER_CHIP_TRACE=1 LANES=1 REPS=2 python3 tools/bench_shard_wide.py poseidon2 2>&1 >/dev/null | grep "lanes trace\|chip 2" | tail -108 > $o/out/r06_shard_wide_chip_trace_1lane.txt
CENO_TOWER_COHORT_LAYERS=0 CENO_LANES_TRACE=1 LANES=8 REPS=2 python3 tools/bench_shard_wide.py poseidon2 2>&1 >/dev/null | grep "lanes trace" | tail -54 > $o/out/r06_shard_wide_chip_trace_8lanes.txt
# the cohort layers: the phase's own account (records / towers / host layers / cohort layers / the rest), device-side round times of the first
# and the last chip of every layer, the last cohort layer swept, the sub-cube cut fixed at 2^13 against the adaptive one
CENO_COHORT_TRACE=1 CENO_COHORT_TIMES=1 LANES=8 REPS=3 python3 tools/bench_shard_wide.py poseidon2 2>&1 >/dev/null | grep "cohort" | tail -34 | cut -c1-1200 > $o/out/r06_cohort_trace.txt
for L in 0 13 16 17 18 19 20 22; do echo "CENO_TOWER_COHORT_LAYERS=$L $(CENO_TOWER_COHORT_LAYERS=$L LANES=8 REPS=5 python3 tools/bench_shard_wide.py poseidon2 2>/dev/null | tail -1)" >> $o/out/r06_cohort_last_layer.txt; done
echo "CENO_TOWER_COHORT_SUB=13 $(CENO_TOWER_COHORT_SUB=13 LANES=8 REPS=5 python3 tools/bench_shard_wide.py poseidon2 2>/dev/null | tail -1)" >> $o/out/r06_cohort_last_layer.txt
echo "CENO_TOWER_VIRTUAL_RECORDS=0 $(CENO_TOWER_VIRTUAL_RECORDS=0 LANES=8 REPS=5 python3 tools/bench_shard_wide.py poseidon2 2>/dev/null | tail -1)" >> $o/out/r06_cohort_last_layer.txt
python3 tools/bench_chip.py 2>/dev/null | tail -1 > $o/out/r06_chip_flow.json
python3 tools/bench_batched_wide.py --reps 4 2>/dev/null | tail -1 > $o/out/r06_batched_main_wide.json
ls -la $o/out/
