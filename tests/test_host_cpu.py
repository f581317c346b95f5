"""CPU-side checks of the product's host code (no GPU needed):
 - the C ABI library loads and exports every symbol include/ceno_hip.h declares;
 - the field arithmetic shared between device kernels and the host layer (csrc/gl64.hpp, compiled for
   the host inside libceno_prover.so) matches the oracle / big-int model bit for bit;
 - the product's stub transcript equals the oracle's stub transcript.
"""
import os
import random

import numpy as np
import pytest

from oracle import pyoracle as po

P = po.P


@pytest.fixture(scope="module")
def built():
    from ceno_amd import build

    build.build_all()
    from ceno_amd import _lib, prover

    return _lib, prover


def test_c_abi_exports_every_declared_symbol(built):
    _lib, _ = built
    L = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) > 40
    missing = [s for s in declared if not hasattr(L, s)]
    assert missing == [], f"declared in include/ceno_hip.h but not exported: {missing}"
    assert L._ceno_missing == []
    assert set(L._ceno_sig) == set(declared), sorted(set(L._ceno_sig) ^ set(declared))
    assert b"gfx950" in L.ceno_hip_version()


def test_prover_library_exports_every_symbol_of_its_header(built):
    """include/ceno_prover.h (host layer: transcript, round loops, tower, rotation, main constraints, commit, open,
    sharded driver) against libceno_prover.so"""
    import re

    _, prover = built
    L = prover.plib()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "ceno_prover.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(ceno_[a-z0-9_]+)\s*\(", text)))
    assert len(names) > 30
    missing = [n for n in names if not hasattr(L, n)]
    assert missing == [], f"declared in include/ceno_prover.h but not exported: {missing}"


def test_init_without_gpu_fails_loudly(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ceno_amd import CenoHipError, Device

    with pytest.raises(CenoHipError):
        Device(0)


def test_device_field_source_on_host_matches_bigint(built):
    _, prover = built
    L = prover.plib()
    rng = random.Random(11)
    edge = [0, 1, 2, 7, P - 1, P - 2, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 32, (1 << 32) - 1, (1 << 63), P - (1 << 32)]
    vals = edge + [rng.randrange(P) for _ in range(300)]
    for a in vals:
        for b in vals[:24]:
            assert L.ceno_prover_test_gl_mul(a, b) == a * b % P
            assert L.ceno_prover_test_gl_mul_ref(a, b) == a * b % P
            assert L.ceno_prover_test_gl_mul_add2(a, b, b, a) == 2 * a * b % P
            assert L.ceno_prover_test_gl_mul_add2(a, b, P - 1, P - 1) == (a * b + 1) % P
            assert L.ceno_prover_test_gl_mul_nc(a, b) == a * b % P
            assert L.ceno_prover_test_gl_mul_nc(a + (P if a < (1 << 64) - P else 0), b) == a * b % P  # non-canonical operand
            assert L.ceno_prover_test_gl_mul_add(a, b, P - 1 - (a % 3)) == (a * b + P - 1 - (a % 3)) % P
            assert L.ceno_prover_test_gl_add(a, b) == (a + b) % P
            assert L.ceno_prover_test_gl_sub(a, b) == (a - b) % P
        for c in (0, 1, 7, 0xFFFFFFFF):
            assert L.ceno_prover_test_gl_mul_small(a, c) == a * c % P
    o = np.zeros(2, dtype=np.uint64)
    for _ in range(500):
        a = (rng.choice(vals), rng.choice(vals))
        b = (rng.choice(vals), rng.choice(vals))
        L.ceno_prover_test_e2_mul(po._p(po.ext([a]).reshape(2)), po._p(po.ext([b]).reshape(2)), po._p(o))
        assert (int(o[0]), int(o[1])) == po.e2_mul(a, b)
        L.ceno_prover_test_e2_mul_ref(po._p(po.ext([a]).reshape(2)), po._p(po.ext([b]).reshape(2)), po._p(o))
        assert (int(o[0]), int(o[1])) == po.e2_mul(a, b)
        L.ceno_prover_test_e2_mul_pre(po._p(po.ext([a]).reshape(2)), po._p(po.ext([b]).reshape(2)), po._p(o))
        assert (int(o[0]), int(o[1])) == po.e2_mul(a, b)
        # non-canonical operands (any 64-bit pattern) are legal inputs of the unreduced forms
        an = (a[0] + (P if a[0] < (1 << 64) - P and rng.random() < 0.5 else 0), a[1])
        L.ceno_prover_test_e2_mul_nc(po._p(np.array(an, dtype=np.uint64)), po._p(po.ext([b]).reshape(2)), po._p(o))
        assert (int(o[0]), int(o[1])) == po.e2_mul(a, b)
        c = (rng.choice(vals), rng.choice(vals))
        L.ceno_prover_test_e2_fma_pre(po._p(po.ext([a]).reshape(2)), po._p(po.ext([b]).reshape(2)), po._p(po.ext([c]).reshape(2)), po._p(o))
        assert (int(o[0]), int(o[1])) == po.e2_add(po.e2_mul(a, b), c)
        if a != (0, 0):
            L.ceno_prover_test_e2_inv(po._p(po.ext([a]).reshape(2)), po._p(o))
            assert po.e2_mul((int(o[0]), int(o[1])), a) == (1, 0)


def test_unreduced_ext_accumulator_matches_bigint(built):
    """gl::E2Acc (ceno_amd/csrc/gl64.hpp): sums of ext products kept in 160-bit limbs, reduced once"""
    _, prover = built
    L = prover.plib()
    rng = random.Random(5)
    o = np.zeros(2, dtype=np.uint64)
    worst = [(P - 1, P - 1)] * 16  # maximal products: the top limb must absorb the carries
    for n, reps, gen in ((16, 4096, lambda: worst), (37, 1, None), (1, 1, None), (64, 300, None)):
        a = gen() if gen else [(rng.randrange(1 << 64), rng.randrange(1 << 64)) for _ in range(n)]  # non-canonical allowed
        b = gen() if gen else [(rng.randrange(P), rng.randrange(P)) for _ in range(n)]
        A = np.array(a, dtype=np.uint64).reshape(-1)
        B = np.array(b, dtype=np.uint64).reshape(-1)
        L.ceno_prover_test_e2_acc(po._p(A), po._p(B), n, reps, po._p(o))
        want = (0, 0)
        for x, y in zip(a, b):
            want = po.e2_add(want, po.e2_mul((x[0] % P, x[1] % P), y))
        want = po.e2_mul(want, (reps % P, 0))
        assert (int(o[0]), int(o[1])) == want


def test_stub_transcript_equals_oracle_stub(built):
    _, prover = built
    t1, t2 = prover.Transcript.stub(0xF5), po.StubTranscript(0xF5)
    script = [("l", b"combine subset evals"), ("e", (5, 9)), ("s",), ("l", b""), ("l", (26).to_bytes(8, "little")),
              ("e", (P - 1, 0)), ("s",), ("s",), ("l", b"Internal round"), ("s",)]
    for step in script:
        if step[0] == "l":
            t1.append_label(step[1])
            t2.append_label(step[1])
        elif step[0] == "e":
            t1.append_ext(step[1])
            t2.append_ext(step[1])
        else:
            assert t1.sample_ext() == t2.sample_ext()


def test_poseidon2_host_source_matches_oracle_and_transcript_is_deterministic(built):
    _, prover = built
    import ctypes as C

    L = prover.plib()
    L.ceno_prover_test_poseidon2_permute.restype = None
    L.ceno_prover_test_poseidon2_permute.argtypes = [po.u64p]
    L.ceno_prover_test_poseidon2_permute_fast.restype = None
    L.ceno_prover_test_poseidon2_permute_fast.argtypes = [po.u64p]
    rng = random.Random(3)
    # the shipped permutation keeps its state non-canonical with lazily reduced linear layers: stress the carry paths
    edge = [[P - 1] * 8, [0] * 8, [P - 1, 0] * 4, [0xFFFFFFFF00000000] * 8, [0xFFFFFFFF] * 8, [P - 1, 1, P - 2, 2, 1 << 63, (1 << 63) - 1, 1 << 32, P - (1 << 32)]]
    for k in range(80 + len(edge)):
        st = np.array(edge[k] if k < len(edge) else [rng.randrange(P) for _ in range(8)], dtype=np.uint64)
        exp = po.poseidon2_permute(st)
        got = st.copy()
        L.ceno_prover_test_poseidon2_permute(po._p(got))
        assert np.array_equal(got, exp)
        fast = st.copy()  # the host-only 128-bit form the transcript runs on (host/transcript.cpp p2host)
        L.ceno_prover_test_poseidon2_permute_fast(po._p(fast))
        assert np.array_equal(fast, exp)
    # eight permutations at once (AVX-512, the host-finished tree tops): every lane position, odd counts, the same edge states in every lane
    L.ceno_prover_test_poseidon2_permute_many.restype = C.c_int
    L.ceno_prover_test_poseidon2_permute_many.argtypes = [po.u64p, C.c_size_t]
    for n in (1, 7, 8, 9, 16, 29, 64):
        states = np.array([[rng.randrange(1 << 64) if (k + j) % 5 == 0 else rng.randrange(P) for j in range(8)] for k in range(n)], dtype=np.uint64)
        states[:min(n, len(edge))] = np.array(edge[:min(n, len(edge))], dtype=np.uint64)
        states %= np.uint64(P)     # inputs are canonical in the product (digests); the non-canonical draws above become other values
        exp = np.stack([po.poseidon2_permute(states[k].copy()) for k in range(n)])
        got = states.copy()
        vec = L.ceno_prover_test_poseidon2_permute_many(po._p(got), n)
        assert np.array_equal(got, exp), n
    if "avx512f" in open("/proc/cpuinfo").read():
        assert vec == 1
    # duplex challenger: same script -> same challenges; one changed element -> different challenges
    def run(x):
        t = prover.Transcript.poseidon2(b"riscv")
        t.append_label(b"combine subset evals")
        a = t.sample_ext()
        t.append_ext((x, 7))
        t.append_ext((1, 2))
        t.append_label(b"Internal round")
        return a, t.sample_ext(), t.sample_ext()

    r1, r2, r3 = run(5), run(5), run(6)
    assert r1 == r2 and r1[0] == r3[0] and r1[1] != r3[1]
    assert all(0 <= c < P for pair in r1 for c in pair)
    # manual model of the duplex rules (overwrite absorb at rate 4, squeeze pops from the back)
    state = np.zeros(8, dtype=np.uint64)
    t = prover.Transcript.poseidon2(b"")
    t.append_ext((11, 22))
    got = t.sample_ext()
    state[0], state[1] = 11, 22
    state = po.poseidon2_permute(state)
    assert got == (int(state[3]), int(state[2]))


def test_entry_points_resolve_their_stream_before_they_allocate():
    """The pool tags a block with the stream the calling thread resolved last (ctx_stream) and hands it to another stream only
    once that one has drained.  An entry point that allocates BEFORE it resolves its own stream argument would take blocks
    under the previous call's tag (a real bug once: the pool-ordering GPU test).  Static check over the C-ABI sources."""
    import glob
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for f in sorted(glob.glob(os.path.join(root, "ceno_amd", "csrc", "*.hip"))):
        s = open(f).read()
        for m in re.finditer(r'\n(?:extern "C" )?int (ceno_hip_\w+)\(([^)]*)\)\s*\{', s):
            if "ceno_hip_stream" not in m.group(2):
                continue
            body = s[m.end(): s.find("\n}\n", m.end())]
            allocs = [body.find(x) for x in ("ctx_alloc(", "ceno_hip_mle_alloc(", "merkle_alloc(", "tower_alloc(") if body.find(x) >= 0]
            st = body.find("ctx_stream(")
            if allocs and (st < 0 or min(allocs) < st):
                offenders.append((os.path.basename(f), m.group(1)))
    assert offenders == [], offenders


def test_eight_lane_host_arithmetic_matches_the_scalar_operators(built):
    """csrc/e2_host_avx512.hpp (the host-finished sumcheck tails and tower layers): add / sub / mul / lane sum / fold on eight lanes equal the oracle's
    extension arithmetic word for word, on canonical inputs including the carry-heavy ones"""
    _, prover = built
    import ctypes as C

    L = prover.plib()
    L.ceno_prover_test_e2v.restype = C.c_int
    L.ceno_prover_test_e2v.argtypes = [po.u64p] * 5
    rng = random.Random(11)
    edge = [P - 1, 0, 1, P - 2, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 63, (1 << 32) + 1]
    ran = 0
    for it in range(40):
        pick = (lambda: rng.choice(edge)) if it < 12 else (lambda: rng.randrange(P))
        a = np.array([[pick(), pick()] for _ in range(8)], dtype=np.uint64)
        b = np.array([[pick(), pick()] for _ in range(8)], dtype=np.uint64)
        tab = np.array([[pick(), pick()] for _ in range(16)], dtype=np.uint64)
        r = np.array([pick(), pick()], dtype=np.uint64)
        out = np.zeros((25, 2), dtype=np.uint64)
        io = tab.copy()
        if not L.ceno_prover_test_e2v(po._p(a), po._p(b), po._p(out), po._p(io), po._p(r)):
            pytest.skip("no AVX-512 on this CPU")
        ran += 1
        t = lambda x: (int(x[0]), int(x[1]))
        for k in range(8):
            assert t(out[k]) == po.e2_add(t(a[k]), t(b[k]))
            assert t(out[8 + k]) == po.e2_sub(t(a[k]), t(b[k]))
            assert t(out[16 + k]) == po.e2_mul(t(a[k]), t(b[k]))
            lo, hi = t(tab[2 * k]), t(tab[2 * k + 1])
            assert t(io[k]) == po.e2_add(lo, po.e2_mul(t(r), po.e2_sub(hi, lo)))
        s = (0, 0)
        for k in range(8):
            s = po.e2_add(s, t(a[k]))
        assert t(out[24]) == s
    assert ran == 40


def test_shard_population_is_the_references():
    """45 opcode circuits (every RV32IM opcode circuit but ECALL), instance counts that add up to the cycle count, seven table circuits"""
    from ceno_amd import synthetic

    names = [k[0] for k in synthetic.OPCODE_KINDS]
    assert len(set(names)) == 45
    for lc in (10, 12, 20):
        c = synthetic.opcode_counts(lc)
        assert sum(c) == 1 << lc and min(c) >= 1
    c = synthetic.opcode_counts(20)
    assert max(c) > 100 * min(c) and sum(1 for x in c if x & (x - 1)) > 40   # a few hot opcodes, a long tail, not powers of two
    assert [t[0] for t in synthetic.TABLE_KINDS] == ["DynamicRange", "DoubleU8", "AndTable", "OrTable", "XorTable", "LtuTable", "Program"]
