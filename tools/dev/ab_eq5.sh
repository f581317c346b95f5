#!/bin/bash
# A/B of the degree-5 eq kernel on ONE box: the one-pass form at two waves per SIMD (round 5) against the two-pass form at three
mkdir -p gpurun_out/r06
out=gpurun_out/r06/ab_eq5.log; : > $out
for F in "-DGEN_EQ5_SPLIT=0 -DGEN_EQ5_WAVES=1" "-DGEN_EQ5_SPLIT=1 -DGEN_EQ5_WAVES=3" "-DGEN_EQ5_SPLIT=1 -DGEN_EQ5_WAVES=1" "-DGEN_EQ5_SPLIT=0 -DGEN_EQ5_WAVES=1" "-DGEN_EQ5_SPLIT=1 -DGEN_EQ5_WAVES=3"; do
  CENO_HIP_EXTRA_FLAGS="$F" timeout 400 python -m ceno_amd.build > /dev/null 2>&1
  echo "== $F" >> $out
  timeout 120 python tools/bench_batched_wide.py --reps 6 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('wide batch ms', round(r['ms'],3), [round(x,2) for x in r.get('runs_ms',[])])" >> $out
  LANES=8 REPS=5 timeout 120 python tools/bench_shard_wide.py 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('shard main ms', r['batched_main_ms'], 'total', r['total_ms'])" >> $out
done
timeout 400 python -m ceno_amd.build > /dev/null 2>&1
cat $out
