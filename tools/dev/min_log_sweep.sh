# batches of SMALL chips: from how many rows on should a size class go through the component tables (CENO_HIP_GEN_MIN_LOG)?
for nv in 14 16 20 24; do
  for m in 13 10 8 6 4; do
    echo -n "wide max_nv $nv MIN_LOG=$m: "
    CENO_HIP_GEN_MIN_LOG=$m python3 tools/bench_batched_wide.py --max-nv $nv --reps 6 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(sorted(round(x,3) for x in r['runs_ms'])[:3])"
  done
done
for m in 13 10 8 6 4; do
  echo -n "shard flow MIN_LOG=$m: "
  CENO_HIP_GEN_MIN_LOG=$m LANES=4 python3 tools/bench_shard.py poseidon2 2>/dev/null | grep lanes | cut -c1-200
done
