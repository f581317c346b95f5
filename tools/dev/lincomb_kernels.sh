# kernel times of the combination / evaluation passes of the wide batch + parity of the two entries
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/lck; mkdir -p $o; rm -rf $o/*
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "prefix_batch or lincomb" 2>&1 | tail -2
timeout 600 python3 -m pytest tests/test_gpu_flows.py -x -q -k "linear_only or wide_batched_main_constraints" 2>&1 | tail -2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/kt -- python3 tools/bench_batched_wide.py --reps 3 > $o/kt.log 2>&1
grep -E "k_lincomb|k_eval_cols" $(ls $o/kt/*/*kernel_stats.csv | head -1) | cut -c1-160
python3 tools/bench_batched_wide.py --reps 5 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(sorted(round(x,2) for x in r['runs_ms'])[:3])"
