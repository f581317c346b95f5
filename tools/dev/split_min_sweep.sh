for m in 0 14 16 18 20 22 30; do
  echo -n "split_min_log $m: wide "
  CENO_HIP_GEN_SPLIT_MIN_LOG=$m python3 tools/bench_batched_wide.py --reps 3 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms'],2), end=' ')"
  echo -n " shard(4 lanes) "
  CENO_HIP_GEN_SPLIT_MIN_LOG=$m LANES=4 python3 tools/bench_shard.py poseidon2 2>/dev/null | grep lanes | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['batched_main_ms'], r['total_ms'], end=' ')"
  echo -n " chip main "
  CENO_HIP_GEN_SPLIT_MIN_LOG=$m python3 tools/bench_chip.py 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['main_sumcheck_ms'])"
done
