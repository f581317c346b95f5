#!/bin/bash
# round-4 profile set on ONE box: the bench line (with extras), rocprofv3 kernel stats + FETCH / WRITE PMC of the headline, kernel stats of
# the chip flow and of the shard flow (four lanes on the C++ scheduler), the batched main sumcheck (tools/r04_batched_profiles.sh), per-round
# host timeline of one pipelined sumcheck (round latency), shard lanes, small chip proofs.  Outputs: gpurun_out/r04/out/r04_*
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o/out; rm -rf $o/trace $o/pmc_fetch $o/pmc_write $o/chip_kt $o/shard_kt
timeout 400 python3 bench.py > $o/bench_n1.json 2> $o/bench.err
cp $o/bench_n1.json $o/out/r04_bench_n1.json
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $o/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $o/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $o/pmc_write.log 2>&1
python3 tools/pmc_summary.py $(dirname $(ls $o/trace/*/*kernel_stats.csv | head -1)) $(dirname $(ls $o/pmc_fetch/*/*counter_collection.csv | head -1)) $(dirname $(ls $o/pmc_write/*/*counter_collection.csv | head -1)) 0 $o/out/r04_sumcheck_nv26
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/chip_kt -- python3 tools/bench_chip.py > $o/chip_kt.log 2>&1
LANES=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/shard_kt -- python3 tools/bench_shard.py poseidon2 > $o/shard_kt.log 2>&1
cp $(ls $o/chip_kt/*/*kernel_stats.csv | head -1) $o/out/r04_chip_flow_kernel_stats.csv
cp $(ls $o/shard_kt/*/*kernel_stats.csv | head -1) $o/out/r04_shard_flow_kernel_stats.csv
LANES=1,2,4,8 python3 tools/bench_shard.py poseidon2 2>/dev/null | grep lanes > $o/out/r04_shard_lanes.jsonl
python3 tools/bench_chip.py 2>/dev/null | tail -1 > $o/out/r04_chip_flow.json
# round latency: host-observed duration of every round of one pipelined generic sumcheck (nv = 18) and of a dense one (nv = 22), tower layers
NV=18 CENO_PROVER_ROUND_TRACE=1 python3 tools/dev/dbg_small_sumcheck.py 2>&1 | grep "sumcheck of 18 variables" | tail -18 > $o/round_generic.txt
CENO_PROVER_LAYER_TRACE=1 python3 tools/bench_chip.py 2>&1 | grep "tower layer" | tail -14 > $o/tower_layers.txt
python3 tools/dev/small_chip.py 2>/dev/null | tail -1 > $o/small_chip.txt
LOG=16 python3 tools/dev/small_chip.py 2>/dev/null | tail -1 >> $o/small_chip.txt
python3 - $o <<'PY'
import json, re, sys
o = sys.argv[1]
rounds = []
for l in open(o + "/round_generic.txt"):
    m = re.search(r"round (\d+): ([\d.]+) us", l)
    if m: rounds.append({"round": int(m.group(1)), "pairs": 1 << (17 - int(m.group(1))), "us": float(m.group(2))})
layers = []
for l in open(o + "/tower_layers.txt"):
    m = re.search(r"tower layer (\d+): begin (\d+) us, rounds (\d+) us, free (\d+) us", l)
    if m: layers.append({"layer": int(m.group(1)), "begin_us": int(m.group(2)), "rounds_us": int(m.group(3)), "free_us": int(m.group(4))})
res = {"what": "host-observed wall time of every round of ONE pipelined generic sumcheck (4 ext MLEs, 2 degree-3 terms, nv = 18, stub transcript; CENO_PROVER_ROUND_TRACE=1, tools/dev/dbg_small_sumcheck.py) and of every device-proved tower layer of the config-#3 chip (2^20 rows: towers of 22 / 22 / 23 variables; CENO_PROVER_LAYER_TRACE=1, tools/bench_chip.py)",
       "generic_sumcheck_rounds": rounds, "tower_layers": layers,
       "small_chip_proofs": [l.strip() for l in open(o + "/small_chip.txt")]}
json.dump(res, open(o + "/out/r04_round_latency.json", "w"), indent=1)
PY
bash tools/r04_batched_profiles.sh > $o/batched.log 2>&1
cp $o/r04_batched_main.json $o/r04_batched_main_kernel_stats.csv $o/out/
ls -la $o/out/
