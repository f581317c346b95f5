#!/bin/bash
# same-box A/B of bench.py headline between the working tree and the build of an older revision kept under _old/
# (prepare with: git worktree add -f _old <rev> && (cd _old && python -m ceno_amd.build); remove the worktree afterwards)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for d in . _old . _old; do
  (cd $d && python bench.py --steps 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$d', 'poseidon', round(r['ms_per_step'],4), 'stub', round(r.get('ms_per_step_stub',0),4), 'frac', round(r['roofline']['frac'],4))")
done
done
