"""Device field arithmetic on CHOSEN inputs (ceno_hip_selftest_field) against Python integers.

The proof-level parity tests exercise the arithmetic on random data, where the rare paths of the reduction (a borrow, a carry
and a borrow together, maximal limbs) occur with probability ~2^-32; here they are constructed.  Reference semantics:
Goldilocks p = 2^64 - 2^32 + 1 and GoldilocksExt2 = F[X]/(X^2 - 7) (p3-goldilocks, used throughout gkr_iop / sumcheck)."""
import ctypes as C
import itertools

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
P = po.P
W = 7
M32, M64 = (1 << 32) - 1, (1 << 64) - 1


@pytest.fixture(scope="module")
def dev():
    from ceno_amd import Device

    d = Device(0)
    yield d
    d.close()


def _run(dev, which, arr, out_words):
    out = np.zeros(out_words, dtype=np.uint64)
    n = {0: arr.size // 5, 1: arr.size // 2, 2: arr.size // 4}[which]
    dev.check(dev.L.ceno_hip_selftest_field(dev.h, which, arr.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.POINTER(C.c_uint64))))
    return out


def test_reduction_every_correction_path(dev):
    edge32 = [0, 1, 2, 3, 0x7FFFFFFF, 0x80000000, 0x80000001, M32 - 2, M32 - 1, M32]
    recs = [(w0, w1, w2, w3, c) for w0, w1, w2, w3 in itertools.product(edge32, repeat=4) for c in (0, 1, 15)]
    # carry out of w2 * (2^32 - 1) + (w1:w0) followed by a borrow from - (c:w3): the two corrections must cancel
    for w2 in (2, 3, 0x10000, M32):
        for small in (0, 1, 5, 1000):
            lo = ((1 << 64) + small - w2 * M32) & M64        # (w1:w0) + w2 * EPS = 2^64 + small
            if lo + w2 * M32 >= (1 << 64):
                for w3 in (small + 1, small + 7, M32):
                    if w3 <= M32:
                        for c in (0, 1, 15):
                            recs.append((lo & M32, lo >> 32, w2, w3, c))
    rng = np.random.default_rng(5)
    recs += [tuple(int(x) for x in rng.integers(0, 1 << 32, 4)) + (int(rng.integers(0, 16)),) for _ in range(20000)]
    arr = np.array(recs, dtype=np.uint32).reshape(-1)
    out = _run(dev, 0, arr, 2 * len(recs))
    for k, (w0, w1, w2, w3, c) in enumerate(recs):
        x128 = w0 + (w1 << 32) + (w2 << 64) + (w3 << 96)
        assert int(out[2 * k]) == x128 % P, (k, recs[k])
        assert int(out[2 * k + 1]) == (x128 + (c << 128)) % P, (k, recs[k])


def test_base_field_operations_on_edge_words(dev):
    edge = [0, 1, 2, M32 - 1, M32, M32 + 1, M32 + 2, 1 << 33, (1 << 63) - 1, 1 << 63, P - 2, P - 1, P, P + 1, M64 - M32, M64 - 1, M64,
            0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF, 0x00000001FFFFFFFF, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000002]
    pairs = list(itertools.product(edge, repeat=2))
    rng = np.random.default_rng(6)
    pairs += [(int(a), int(b)) for a, b in rng.integers(0, 1 << 64, (20000, 2), dtype=np.uint64)]
    arr = np.array(pairs, dtype=np.uint64).reshape(-1)
    out = _run(dev, 1, arr, 6 * len(pairs)).reshape(-1, 6)
    for k, (a, b) in enumerate(pairs):
        ac, bc = a % P, b % P
        want = [ac * bc % P, a * b % P, (ac + bc) % P, (ac - bc) % P, (a + bc) % P, (a * bc + ac) % P]
        assert [int(x) for x in out[k]] == want, (hex(a), hex(b))


def test_extension_multiply_and_unreduced_accumulator(dev):
    edge = [0, 1, 2, M32, M32 + 1, P - 1, P - 2, (1 << 63), 0xFFFFFFFF00000000 % P, 7]
    els = list(itertools.product(edge, repeat=2))
    pairs = list(itertools.product(els[::3], els[::2]))
    rng = np.random.default_rng(7)
    pairs += [((int(x[0]) % P, int(x[1]) % P), (int(x[2]) % P, int(x[3]) % P)) for x in rng.integers(0, 1 << 64, (20000, 4), dtype=np.uint64)]
    arr = np.array([[a[0], a[1], b[0], b[1]] for a, b in pairs], dtype=np.uint64).reshape(-1)
    out = _run(dev, 2, arr, 4 * len(pairs)).reshape(-1, 4)

    def mul(a, b):
        return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    for k, (a, b) in enumerate(pairs):
        p = mul(a, b)
        aa = mul(a, a)
        q = ((2 * p[0] + aa[0]) % P, (2 * p[1] + aa[1]) % P)
        assert (int(out[k][0]), int(out[k][1])) == p, (a, b)
        assert (int(out[k][2]), int(out[k][3])) == q, (a, b)
