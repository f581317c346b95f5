#!/bin/bash
# A/B: host-finished tree tops with eight permutations per pass (AVX-512) against one by one; host levels 6 (default), 7, 8
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import ctypes as C, time, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ceno_amd import prover
from oracle import pyoracle as po
L = prover.plib()
L.ceno_prover_test_poseidon2_permute_many.restype = C.c_int
L.ceno_prover_test_poseidon2_permute_many.argtypes = [po.u64p, C.c_size_t]
L.ceno_prover_test_poseidon2_chain.argtypes = [po.u64p, C.c_int, C.c_int]
st = np.arange(64 * 8, dtype=np.uint64)
for n in (8, 16, 64):
    t0 = time.perf_counter()
    for _ in range(4000): L.ceno_prover_test_poseidon2_permute_many(po._p(st), n)
    print(f"permute_many({n}): {(time.perf_counter() - t0) / 4000 * 1e6:.2f} us per call")
s8 = np.arange(8, dtype=np.uint64)
t0 = time.perf_counter(); L.ceno_prover_test_poseidon2_chain(po._p(s8), 100000, 1)
print(f"scalar: {(time.perf_counter() - t0) / 100000 * 1e6:.3f} us per permutation")
PY
for rep in 1 2; do
  echo -n "scalar tops, 6 levels: "; CENO_HIP_HOST_AVX512=0 python3 tools/dev/dbg_tower_only.py 20 2>/dev/null | tail -1
  for lv in 6 7 8; do echo -n "avx512 tops, $lv levels: "; CENO_HIP_HOST_TOP=$lv python3 tools/dev/dbg_tower_only.py 20 2>/dev/null | tail -1; done
done
