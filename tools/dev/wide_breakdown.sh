# where the wide batch's time goes: host phases (CENO_HIP_DEBUG) and kernel statistics
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/wideb; mkdir -p $o; rm -rf $o/*
CENO_HIP_DEBUG=1 python3 tools/bench_batched_wide.py --reps 4 2>&1 | grep "batched main:" | tail -4
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -- python3 tools/bench_batched_wide.py --reps 3 > $o/kt.log 2>&1
head -16 $(ls $o/kt/*/*kernel_stats.csv | head -1) | cut -c1-200
