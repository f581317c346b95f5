#!/bin/bash
# same-box A/B of the shard flow (tools/bench_shard.py) and the small chip proofs between the working tree and the build of an older
# revision kept under _old/ (prepare: git worktree add -f _old <rev> && (cd _old && python -m ceno_amd.build))
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for d in . _old; do
  (cd $d && echo "== $d" && LANES=${LANES:-1,4} python tools/bench_shard.py ${TR:-poseidon2} 2>/dev/null | tail -2 && python tools/dev/small_chip.py 2>/dev/null | tail -1)
done
done
