"""Experiment: one nv=28 sumcheck (13 GB of tables) verified through independent kernel paths."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceno_amd import Device, prover
from oracle import pyoracle as po
P = po.P
dev = Device(0)
nv, k = 28, 3
mles = [dev.synthetic(nv, True, 0xCE10 + j) for j in range(k)]
dev.sync(); t0 = time.perf_counter()
msgs, chal, fin = prover.sumcheck_prove(dev, mles, po.ext([1]), [[0, 1, 2]], nv, k, prover.Transcript.stub(0xF5))
dt = time.perf_counter() - t0
prod = dev.wit_infer(mles, po.ext([1]), [[0, 1, 2]], [[0]], nv)[0]
half = np.tile(np.array([[(P + 1) // 2, 0]], dtype=np.uint64), (nv, 1))
claim = po.e2_mul(prod.evaluate(half), (pow(2, nv, P), 0))
point, expected = po.sumcheck_verify(claim, msgs, po.StubTranscript(0xF5))
want = (1, 0)
for j in range(k):
    assert mles[j].evaluate(chal) == (int(fin[j][0]), int(fin[j][1]))
    want = po.e2_mul(want, (int(fin[j][0]), int(fin[j][1])))
assert expected == want and np.array_equal(point, chal)
print("nv=28 ok, %.2f ms, %.3e ext-mults/s" % (dt * 1e3, 9 * ((1 << nv) - 1) / dt))
