# sumchecks SMALLER than 2^13: do batches of many small chips want the component tables too?
for nv in 8 10 12 13; do
  for m in 13 4; do
    echo -n "wide max_nv $nv MIN_LOG=$m: "
    CENO_HIP_GEN_MIN_LOG=$m python3 tools/bench_batched_wide.py --max-nv $nv --reps 8 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(sorted(round(x,3) for x in r['runs_ms'])[:3])"
  done
done
echo "default thresholds:"
for nv in 14 20 24; do
  echo -n "wide max_nv $nv: "
  python3 tools/bench_batched_wide.py --max-nv $nv --reps 6 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(sorted(round(x,3) for x in r['runs_ms'])[:3])"
done
python3 tools/bench_batched.py --reps 4 2>&1 | tail -1 | cut -c1-300
