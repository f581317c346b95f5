/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may build, load or call it.
 *
 * Goldilocks base field F_p, p = 2^64 - 2^32 + 1, and its quadratic extension
 * F_p[X]/(X^2 - 7) ("GoldilocksExt2").
 *
 * The reference does not hold this arithmetic in-tree: it is `ff_ext::GoldilocksExt2`
 * from scroll-tech/gkr-backend @ v1.0.0-alpha.35 (reference Cargo.toml:30-40), which
 * wraps crates.io p3-goldilocks 0.4.3 `BinomialExtensionField<Goldilocks, 2>`
 * (reference Cargo.lock:4142-4375).  The published p3-goldilocks definition is
 * `impl BinomiallyExtendable<2> for Goldilocks { const W = 7; }` — W is kept as the
 * single named constant GL_W below (not stated anywhere under
 * /root/reference; every in-tree golden vector uses embedded base-field values only,
 * so none of them exercises W).
 *
 * This file deliberately uses unsigned __int128 so that it shares no code with the
 * 32-bit-limb device implementation in ceno_amd/csrc/gl64.hpp.
 */
#ifndef CENO_ORACLE_GL64_H
#define CENO_ORACLE_GL64_H

#include <stdint.h>
#include <stddef.h>

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_W 7ULL

typedef unsigned __int128 u128;

typedef struct { uint64_t c[2]; } ext2; /* c[0] + c[1]*X, canonical limbs */

static inline uint64_t gl_reduce(uint64_t x) { return x >= GL_P ? x - GL_P : x; }

static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    u128 s = (u128)a + b;
    if (s >= GL_P) s -= GL_P;
    return (uint64_t)s;
}
static inline uint64_t gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (GL_P - b); }
static inline uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }
/* definition by division — kept as the cross-check of the fast form below (tests/test_oracle_field.py) */
static inline uint64_t gl_mul_div(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % GL_P); }
/* x = hi*2^64 + lo = lo + hi_lo*(2^32-1) - hi_hi  (2^64 = 2^32-1, 2^96 = -1 mod p), all in u128/i128 */
static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
    u128 x = (u128)a * b;
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t hi_lo = hi & 0xFFFFFFFFULL, hi_hi = hi >> 32;
    /* lo + hi_lo*EPS < 2^64 + 2^64 ; subtract hi_hi after adding p to stay non-negative */
    u128 t = (u128)lo + (u128)hi_lo * 0xFFFFFFFFULL + GL_P - hi_hi;  /* < 3 * 2^64 */
    uint64_t tl = (uint64_t)t, th = (uint64_t)(t >> 64);               /* th in {0,1,2} */
    u128 r = (u128)tl + (u128)th * 0xFFFFFFFFULL;                     /* 2^64 = EPS */
    if (r >> 64) r = (uint64_t)r + (u128)0xFFFFFFFFULL;
    uint64_t v = (uint64_t)r;
    if (r >> 64) v += 0xFFFFFFFFULL; /* cannot happen twice; kept for safety */
    return v >= GL_P ? v - GL_P : v;
}
static inline uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) { if (e & 1) r = gl_mul(r, a); a = gl_mul(a, a); e >>= 1; }
    return r;
}
static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }

static inline ext2 e2_zero(void) { ext2 r = {{0, 0}}; return r; }
static inline ext2 e2_one(void) { ext2 r = {{1, 0}}; return r; }
static inline ext2 e2_from_u64(uint64_t v) { ext2 r = {{v % GL_P, 0}}; return r; }
static inline ext2 e2_from_base(uint64_t v) { ext2 r = {{v, 0}}; return r; }
static inline int e2_eq(ext2 a, ext2 b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1]; }
static inline int e2_is_zero(ext2 a) { return a.c[0] == 0 && a.c[1] == 0; }
static inline ext2 e2_add(ext2 a, ext2 b) { ext2 r = {{gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1])}}; return r; }
static inline ext2 e2_sub(ext2 a, ext2 b) { ext2 r = {{gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1])}}; return r; }
static inline ext2 e2_neg(ext2 a) { ext2 r = {{gl_neg(a.c[0]), gl_neg(a.c[1])}}; return r; }
static inline ext2 e2_mul(ext2 a, ext2 b) {
    /* schoolbook: (a0 b0 + W a1 b1) + (a0 b1 + a1 b0) X */
    ext2 r;
    r.c[0] = gl_add(gl_mul(a.c[0], b.c[0]), gl_mul(GL_W, gl_mul(a.c[1], b.c[1])));
    r.c[1] = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    return r;
}
static inline ext2 e2_mul_base(ext2 a, uint64_t b) { ext2 r = {{gl_mul(a.c[0], b), gl_mul(a.c[1], b)}}; return r; }
static inline ext2 e2_inv(ext2 a) {
    /* 1/(a0 + a1 X) = (a0 - a1 X)/(a0^2 - W a1^2) */
    uint64_t n = gl_sub(gl_mul(a.c[0], a.c[0]), gl_mul(GL_W, gl_mul(a.c[1], a.c[1])));
    uint64_t ni = gl_inv(n);
    ext2 r = {{gl_mul(a.c[0], ni), gl_mul(gl_neg(a.c[1]), ni)}};
    return r;
}

/* SplitMix64 stream: i-th output of the generator seeded with `seed` (BASELINE.md
 * "Synthetic inputs").  Values are reduced mod p once (x >= p ? x - p : x). */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t splitmix_gl(uint64_t seed, uint64_t i) { return gl_reduce(splitmix64_at(seed, i)); }

#endif
