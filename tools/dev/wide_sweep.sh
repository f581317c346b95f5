for kb in 31 28 24 20 16; do
  echo -n "stage KB $kb: "
  CENO_HIP_PLAN_REPORT=2 CENO_HIP_GEN_OVERSUB=16 CENO_HIP_GEN_STAGE_KB=$kb python3 tools/bench_batched_wide.py --reps 3 2>/tmp/err.txt | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms'],2), sum(c['tables_staged'] for c in r['classes']), sum(c['components'] for c in r['classes']))"
  grep "eq launch of round 1" /tmp/err.txt | head -1
done
