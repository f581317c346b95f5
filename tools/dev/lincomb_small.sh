# does combining the linear-only columns cost a batch of SMALL chips anything?  (wide batch at max_nv 14 / 16 / 18: chips of 2^2 .. 2^18 rows)
for nv in 14 16 18 20; do
  for m in 3 0; do
    echo -n "max_nv $nv LINCOMB=$m: "
    CENO_PROVER_MAIN_LINCOMB=$m python3 tools/bench_batched_wide.py --max-nv $nv --reps 8 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(sorted(round(x,3) for x in r['runs_ms'])[:4])"
  done
done
