// Poseidon2 permutation for the HOST side of the product: the Fiat-Shamir challenger (host/transcript.cpp) and the top levels of
// a Merkle tree (poseidon2.hip merkle_finish_host).  Same permutation as p2::permute (csrc/poseidon2.hpp), different arithmetic.
#pragma once
#include "poseidon2.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cstdlib>

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

// ---- eight permutations at once (AVX-512: one 64-bit lane per permutation, the state's eight words in eight registers) ------------------
// A tree level finished on the host is 2^k INDEPENDENT permutations: the scalar code above runs them one after the other (0.54 us each on
// an EPYC 9575F), the vector code runs eight in the time of roughly one and a half.  Same arithmetic: residues in [0, 2^64), one reduction
// per product, canonical at the end — bit-identical to permute() (tests/test_host_cpu.py).  Used only where the CPU has AVX-512F/DQ.
#if defined(__x86_64__)
#define P2_AVX512 __attribute__((target("avx512f,avx512dq"), always_inline)) static inline
static inline bool have_avx512() {
    static const bool v = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
                          !(getenv("CENO_HIP_HOST_AVX512") && atoi(getenv("CENO_HIP_HOST_AVX512")) == 0);  // A/B switch
    return v;
}
P2_AVX512 __m512i v_eps() { return _mm512_set1_epi64((long long)gl::EPS); }
// a + b mod p for ANY residues a, b < 2^64 (two conditional corrections: the first may overflow again)
P2_AVX512 __m512i v_add(__m512i a, __m512i b) {
    const __m512i s = _mm512_add_epi64(a, b);
    const __mmask8 c = _mm512_cmplt_epu64_mask(s, a);
    const __m512i s2 = _mm512_mask_add_epi64(s, c, s, v_eps());
    const __mmask8 c2 = _mm512_mask_cmplt_epu64_mask(c, s2, s);
    return _mm512_mask_add_epi64(s2, c2, s2, v_eps());
}
// (hi, lo) -> residue in [0, 2^64): lo - (hi >> 32) + (hi & EPS) * EPS  (2^64 = EPS, 2^96 = -1 mod p), as red128 above
P2_AVX512 __m512i v_red(__m512i hi, __m512i lo) {
    const __m512i hh = _mm512_srli_epi64(hi, 32), hl = _mm512_and_si512(hi, v_eps());
    __m512i t0 = _mm512_sub_epi64(lo, hh);
    const __mmask8 b = _mm512_cmplt_epu64_mask(lo, hh);
    t0 = _mm512_mask_sub_epi64(t0, b, t0, v_eps());
    const __m512i t1 = _mm512_sub_epi64(_mm512_slli_epi64(hl, 32), hl);
    const __m512i r = _mm512_add_epi64(t0, t1);
    const __mmask8 c = _mm512_cmplt_epu64_mask(r, t0);
    return _mm512_mask_add_epi64(r, c, r, v_eps());
}
// 64 x 64 -> 128 from four 32 x 32 -> 64 products
P2_AVX512 void v_mul_wide(__m512i a, __m512i b, __m512i& hi, __m512i& lo) {
    const __m512i m32 = v_eps();
    const __m512i ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
    const __m512i ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b), hh = _mm512_mul_epu32(ah, bh);
    const __m512i t0 = _mm512_add_epi64(hl, _mm512_srli_epi64(ll, 32));                 // < 2^64
    const __m512i t1 = _mm512_add_epi64(lh, _mm512_and_si512(t0, m32));                 // < 2^64
    hi = _mm512_add_epi64(_mm512_add_epi64(hh, _mm512_srli_epi64(t0, 32)), _mm512_srli_epi64(t1, 32));
    lo = _mm512_or_si512(_mm512_slli_epi64(t1, 32), _mm512_and_si512(ll, m32));
}
P2_AVX512 __m512i v_mul(__m512i a, __m512i b) {
    __m512i hi, lo;
    v_mul_wide(a, b, hi, lo);
    return v_red(hi, lo);
}
P2_AVX512 __m512i v_sbox7(__m512i x) {
    const __m512i x2 = v_mul(x, x), x3 = v_mul(x2, x), x4 = v_mul(x2, x2);
    return v_mul(x4, x3);
}
P2_AVX512 void v_mat4(const __m512i* x, __m512i* n) {  // rows [2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]
    const __m512i t01 = v_add(x[0], x[1]), t23 = v_add(x[2], x[3]), t0123 = v_add(t01, t23);
    const __m512i t01123 = v_add(t0123, x[1]), t01233 = v_add(t0123, x[3]);
    n[3] = v_add(t01233, v_add(x[0], x[0]));
    n[1] = v_add(t01123, v_add(x[2], x[2]));
    n[0] = v_add(t01123, t01);
    n[2] = v_add(t01233, t23);
}
P2_AVX512 void v_external(__m512i* s) {
    __m512i n[8];
    v_mat4(s, n);
    v_mat4(s + 4, n + 4);
    for (int i = 0; i < 4; i++) {
        const __m512i sum = v_add(n[i], n[i + 4]);
        s[i] = v_add(n[i], sum);
        s[i + 4] = v_add(n[i + 4], sum);
    }
}
// states: B x eight states of eight words each, state after state; permuted in place.  B = 2 runs two independent groups of eight through
// every step side by side: the S-box is a chain of four dependent multiplications (~30 cycles each), two chains fill the pipes better than one.
template <int B>
__attribute__((target("avx512f,avx512dq"))) static inline void permute8xB(uint64_t* states, const p2::Params& p) {
    __m512i s[B][8];
    const __m512i idx = _mm512_setr_epi64(0, 8, 16, 24, 32, 40, 48, 56);
    // word i of every state -> register i (a gather per word: the load is a one-off next to ~500 vector multiplications)
    for (int b = 0; b < B; b++)
        for (int i = 0; i < 8; i++) s[b][i] = _mm512_i64gather_epi64(idx, (const long long*)states + 64 * b + i, 8);
    for (int b = 0; b < B; b++) v_external(s[b]);
    for (int r = 0; r < p2::ROUNDS_F / 2; r++) {
        for (int i = 0; i < 8; i++) {
            const __m512i rc = _mm512_set1_epi64((long long)p.ext_rc[r][i]);
            for (int b = 0; b < B; b++) s[b][i] = v_sbox7(v_add(s[b][i], rc));
        }
        for (int b = 0; b < B; b++) v_external(s[b]);
    }
    for (int r = 0; r < p2::ROUNDS_P; r++) {
        const __m512i rc = _mm512_set1_epi64((long long)p.int_rc[r]);
        __m512i sum[B];
        for (int b = 0; b < B; b++) s[b][0] = v_sbox7(v_add(s[b][0], rc));
        for (int b = 0; b < B; b++)
            sum[b] = v_add(v_add(v_add(s[b][0], s[b][1]), v_add(s[b][2], s[b][3])), v_add(v_add(s[b][4], s[b][5]), v_add(s[b][6], s[b][7])));
        for (int i = 0; i < 8; i++) {
            const __m512i dg = _mm512_set1_epi64((long long)p.int_diag[i]);
            for (int b = 0; b < B; b++) s[b][i] = v_add(v_mul(s[b][i], dg), sum[b]);
        }
    }
    for (int r = p2::ROUNDS_F / 2; r < p2::ROUNDS_F; r++) {
        for (int i = 0; i < 8; i++) {
            const __m512i rc = _mm512_set1_epi64((long long)p.ext_rc[r][i]);
            for (int b = 0; b < B; b++) s[b][i] = v_sbox7(v_add(s[b][i], rc));
        }
        for (int b = 0; b < B; b++) v_external(s[b]);
    }
    const __m512i pp = _mm512_set1_epi64((long long)gl::P);
    for (int b = 0; b < B; b++)
        for (int i = 0; i < 8; i++) {
            const __mmask8 ge = _mm512_cmpge_epu64_mask(s[b][i], pp);  // canonical form: residues in [p, 2^64) drop by p
            _mm512_i64scatter_epi64((long long*)states + 64 * b + i, idx, _mm512_mask_sub_epi64(s[b][i], ge, s[b][i], pp), 8);
        }
}
static inline void permute8(uint64_t* states, const p2::Params& p) { permute8xB<1>(states, p); }
static inline void permute16(uint64_t* states, const p2::Params& p) { permute8xB<2>(states, p); }
#else
static inline bool have_avx512() { return false; }
static inline void permute8(uint64_t*, const p2::Params&) {}
static inline void permute16(uint64_t*, const p2::Params&) {}
#endif
// n independent states, state after state: eight at a time where the CPU can, one by one otherwise
static inline void permute_many(uint64_t* states, size_t n, const p2::Params& p) {
    size_t i = 0;
    if (have_avx512()) {
        for (; i + 16 <= n; i += 16) permute16(states + 8 * i, p);
        for (; i + 8 <= n; i += 8) permute8(states + 8 * i, p);
    }
    for (; i < n; i++) p2host::permute(states + 8 * i, p);
}
}  // namespace p2host
