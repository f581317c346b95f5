for e in "X=1" "CENO_HIP_GEN_OVERSUB=1"; do
  echo -n "$e: "; for i in 1 2; do env $e LANES=4 python3 tools/bench_shard.py poseidon2 2>/dev/null | grep lanes | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['batched_main_ms'], r['total_ms'], end='  ')"; done; echo
done
python3 tools/bench_batched_wide.py --reps 3 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('wide', round(r['ms'],2))"
python3 tools/bench_batched.py --reps 4 2>/dev/null | tail -1 | python3 -c "import json,sys; print('narrow24', json.loads(sys.stdin.read())['batched_main_sumcheck_ms'])"
python3 tools/bench_batched.py --max-nv 26 --reps 3 2>/dev/null | tail -1 | python3 -c "import json,sys; print('narrow26', json.loads(sys.stdin.read())['batched_main_sumcheck_ms'])"
