#!/usr/bin/env python3
"""Host side of tools/ubench_p2lat.hip: microseconds per Poseidon2 permutation of the transcript's host implementation (a chain of
dependent permutations through ceno_prover_test_poseidon2_chain), and what a degree-3 round costs a host challenger: 2 permutations +
one PCIe round trip of the message / challenge (measured inside the round loops: profiles/r02_round_latency.json)."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from ceno_amd import prover

L = prover.plib()
L.ceno_prover_test_poseidon2_chain.restype = None
L.ceno_prover_test_poseidon2_chain.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int]
st = np.arange(8, dtype=np.uint64)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    L.ceno_prover_test_poseidon2_chain(st.ctypes.data_as(C.POINTER(C.c_uint64)), 20000, 1)
    best = min(best, (time.perf_counter() - t0) / 20000 * 1e6)
print(json.dumps({"host_permutation_us": round(best, 3)}))
