#!/bin/bash
# Regenerate profiles/r03_* on the GPU box: kernel stats, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, no
# trace domains combined with --pmc), and the bench line itself.  Outputs land in gpurun_out/refresh/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/refresh; rm -rf $o; mkdir -p $o
timeout 300 python3 bench.py > $o/bench_n1.json 2> $o/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $o/trace -- python3 bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $o/trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $o/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-extra --no-cpu-baseline > $o/pmc_write.log 2>&1
mkdir -p $o/out
python3 tools/pmc_summary.py $(dirname $(ls $o/trace/*/*kernel_stats.csv | head -1)) $(dirname $(ls $o/pmc_fetch/*/*counter_collection.csv | head -1)) $(dirname $(ls $o/pmc_write/*/*counter_collection.csv | head -1)) 0 $o/out/r03_sumcheck_nv26
cp $o/bench_n1.json $o/out/r03_bench_n1.json
tail -1 $o/bench_n1.json | cut -c1-400
head -8 $o/out/r03_sumcheck_nv26_kernel_stats.csv | cut -c1-150
