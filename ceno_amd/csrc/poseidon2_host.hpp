// Poseidon2 permutation for the HOST side of the product: the Fiat-Shamir challenger (host/transcript.cpp) and the top levels of
// a Merkle tree (poseidon2.hip merkle_finish_host).  Same permutation as p2::permute (csrc/poseidon2.hpp), different arithmetic.
#pragma once
#include "poseidon2.hpp"

// The challenger sits on the critical path of every sumcheck round (two permutations between a message and its challenge) and a
// tree top is a chain of dependent levels; the shared source (csrc/poseidon2.hpp) is written for the GPU's
// 32-bit multiplier.  Here products and the sums of the linear layers are plain 128-bit integers, reduced once per word
// with 2^64 = 2^32 - 1, 2^96 = -1 (mod p); bit-identical to p2::permute (tests/test_host_cpu.py), ~3x faster on x86-64.
namespace p2host {
typedef unsigned __int128 u128;
static inline uint64_t red128(u128 x) {  // residue of x in [0, 2^64), not necessarily canonical
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    const uint64_t hh = hi >> 32, hl = hi & gl::EPS;
    uint64_t t0, r;
    // branch-free corrections: the carries are data dependent coin flips, a mispredicted branch costs more than the reduction
    const uint64_t b = __builtin_sub_overflow(lo, hh, &t0);
    t0 -= (0 - b) & gl::EPS;
    const uint64_t t1 = (hl << 32) - hl;
    const uint64_t c = __builtin_add_overflow(t0, t1, &r);
    return r + ((0 - c) & gl::EPS);
}
static inline uint64_t mulnc(uint64_t a, uint64_t b) { return red128((u128)a * b); }
static inline uint64_t sbox7(uint64_t x) {
    const uint64_t x2 = mulnc(x, x), x3 = mulnc(x2, x), x4 = mulnc(x2, x2);
    return mulnc(x4, x3);
}
static inline uint64_t addnc(uint64_t a, uint64_t b) {
    uint64_t s;
    const uint64_t c = __builtin_add_overflow(a, b, &s);
    return s + ((0 - c) & gl::EPS);
}
static inline void mat4(const uint64_t* x, u128* n) {  // rows [2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]
    const u128 t01 = (u128)x[0] + x[1], t23 = (u128)x[2] + x[3], t0123 = t01 + t23;
    const u128 t01123 = t0123 + x[1], t01233 = t0123 + x[3];
    n[3] = t01233 + x[0] + x[0];
    n[1] = t01123 + x[2] + x[2];
    n[0] = t01123 + t01;
    n[2] = t01233 + t23;
}
static inline void external(uint64_t* s) {
    u128 n[8];
    mat4(s, n);
    mat4(s + 4, n + 4);
    for (int i = 0; i < 4; i++) {
        const u128 sum = n[i] + n[i + 4];
        s[i] = red128(n[i] + sum);
        s[i + 4] = red128(n[i + 4] + sum);
    }
}
static inline void permute(uint64_t* s, const p2::Params& p) {
    external(s);
    for (int r = 0; r < p2::ROUNDS_F / 2; r++) {
        for (int i = 0; i < 8; i++) s[i] = sbox7(addnc(s[i], p.ext_rc[r][i]));
        external(s);
    }
    for (int r = 0; r < p2::ROUNDS_P; r++) {
        s[0] = sbox7(addnc(s[0], p.int_rc[r]));
        u128 sum = 0;
        for (int i = 0; i < 8; i++) sum += s[i];
        for (int i = 0; i < 8; i++) s[i] = red128((u128)s[i] * p.int_diag[i] + sum);  // < 2^128: diag is canonical, sum < 2^67
    }
    for (int r = p2::ROUNDS_F / 2; r < p2::ROUNDS_F; r++) {
        for (int i = 0; i < 8; i++) s[i] = sbox7(addnc(s[i], p.ext_rc[r][i]));
        external(s);
    }
    for (int i = 0; i < 8; i++) s[i] = gl::canon(s[i]);
}
}  // namespace p2host
