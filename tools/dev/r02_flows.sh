#!/bin/bash
# round-2 flow numbers on ONE box: chip flow (config #3), shard flow with 1/2/4/8 lanes (metric M2 shape, both transcripts),
# small-sumcheck ladder, per-primitive roofline table, and the kernel statistics of the chip flow
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/r02_flows; rm -rf $o; mkdir -p $o
for i in 1 2 3; do timeout 200 python3 tools/bench_chip.py 2>/dev/null | tail -1; done > $o/chip.txt
timeout 300 python3 tools/bench_shard.py poseidon2 2>/dev/null | tail -4 > $o/shard_poseidon2.txt
timeout 300 python3 tools/bench_shard.py stub 2>/dev/null | tail -4 > $o/shard_stub.txt
NVS=7,12,16 timeout 200 python3 tools/bench_lane_rounds.py 2>/dev/null | tail -1 > $o/lane_rounds.txt
CENO_HIP_MID_W=0 NVS=7,12,16 timeout 200 python3 tools/bench_lane_rounds.py 2>/dev/null | tail -1 > $o/lane_rounds_no_mid.txt
timeout 300 python3 tools/roofline_table.py 2>/dev/null > $o/roofline_table.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/tr_chip -- python3 tools/bench_chip.py > $o/tr_chip.log 2>&1
cp $(ls $o/tr_chip/*/*kernel_stats.csv | head -1) $o/chip_kernel_stats.csv
head -c 600 $o/chip.txt; echo; cat $o/shard_poseidon2.txt; cat $o/lane_rounds.txt | cut -c1-300
