"""Oracle self-checks: C field arithmetic vs an independent Python big-int model."""
import random

import numpy as np

from oracle import pyoracle as po

P = po.P


def test_gl_mul_inv_vs_bigint():
    rng = random.Random(1)
    lib = po.lib()
    edge = [0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 32, (1 << 32) - 1]
    vals = edge + [rng.randrange(P) for _ in range(200)]
    for a in vals:
        for b in vals[:20]:
            assert lib.orc_gl_mul(a, b) == a * b % P
            assert lib.orc_gl_mul_div(a, b) == a * b % P
        if a:
            assert lib.orc_gl_inv(a) * a % P == 1


def test_ext2_mul_inv_vs_bigint():
    rng = random.Random(2)
    lib = po.lib()
    for _ in range(300):
        a = (rng.randrange(P), rng.randrange(P))
        b = (rng.randrange(P), rng.randrange(P))
        o = np.zeros(2, dtype=np.uint64)
        lib.orc_e2_mul(po._p(po.ext([a]).reshape(2)), po._p(po.ext([b]).reshape(2)), po._p(o))
        assert (int(o[0]), int(o[1])) == po.e2_mul(a, b)
        lib.orc_e2_inv(po._p(po.ext([a]).reshape(2)), po._p(o))
        assert po.e2_mul((int(o[0]), int(o[1])), a) == (1, 0)


def test_w_is_nonresidue():
    # X^2 - 7 irreducible over F_p  <=>  7^((p-1)/2) == -1
    assert pow(po.W, (P - 1) // 2, P) == P - 1


def test_splitmix_matches_python():
    a = po.fill_splitmix(64, 0xCE10, 5)
    for i in range(64):
        assert int(a[i]) == po.splitmix_gl(0xCE10, 5 + i)
    assert all(int(x) < P for x in a)


def test_stub_transcript_is_data_dependent():
    t1, t2 = po.StubTranscript(7), po.StubTranscript(7)
    t1.append_ext((1, 2))
    t2.append_ext((1, 3))
    assert t1.sample_ext() != t2.sample_ext()
    t3, t4 = po.StubTranscript(7), po.StubTranscript(7)
    t3.append_label(b"merge")
    t4.append_label(b"merge")
    assert t3.sample_ext() == t4.sample_ext()
