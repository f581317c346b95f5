#!/bin/bash
# same-box A/B of the batched main sumcheck (tools/bench_batched.py) between the working tree and the build of an older revision under _old/
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for d in . _old; do
  (cd $d && echo -n "$d: " && timeout 120 python tools/bench_batched.py --reps ${REPS:-6} ${ARGS:-} 2>/dev/null | tail -1 | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['batched_main_sumcheck_ms'],3))")
done
done
