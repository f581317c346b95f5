// Goldilocks (p = 2^64 - 2^32 + 1) and GoldilocksExt2 = F_p[X]/(X^2 - 7) for gfx950.
//
// Device counterpart of the reference's `ff_ext::GoldilocksExt2` (EXT crate: scroll-tech/gkr-backend
// v1.0.0-alpha.35 over p3-goldilocks 0.4.3 `BinomialExtensionField<Goldilocks, 2>`, W = 7; reference
// Cargo.toml:30-40).  All values held in memory are canonical ([0,p)), so results can be compared
// bit-for-bit with the CPU prover's.
//
// 64-bit modular multiply = one 64x64->128 product (4 x v_mad_u64_u32 on CDNA4) + a reduction
// using 2^64 = 2^32 - 1 and 2^96 = -1 (mod p): no division, no MFMA (not a dense fp contraction).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else  // plain C++ host translation units (ceno_amd/host)
#define GL_HD inline
#endif

namespace gl {

constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t EPS = 0xFFFFFFFFULL;  // 2^64 mod p
constexpr uint64_t W = 7;                // X^2 = W

GL_HD uint64_t add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    uint64_t t = s + EPS;  // s - p (mod 2^64)
    return (s < a || s >= P) ? t : s;
}
GL_HD uint64_t sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    return (a < b) ? d - EPS : d;  // + p (mod 2^64)
}
GL_HD uint64_t neg(uint64_t a) { return a ? P - a : 0; }
GL_HD uint64_t dbl(uint64_t a) { return add(a, a); }

GL_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// reduce hi*2^64 + lo (any 128-bit value) to canonical form
GL_HD uint64_t reduce128(uint64_t lo, uint64_t hi) {
    uint64_t hi_hi = hi >> 32, hi_lo = hi & EPS;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= EPS;
    uint64_t t1 = (hi_lo << 32) - hi_lo;  // hi_lo * EPS
    uint64_t r = t0 + t1;
    if (r < t1) r += EPS;
    return r >= P ? r - P : r;
}
GL_HD uint64_t mul(uint64_t a, uint64_t b) { return reduce128(a * b, mulhi64(a, b)); }
GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }
// small-constant multiply (c < 2^32): product fits 96 bits
GL_HD uint64_t mul_small(uint64_t a, uint32_t c) {
    uint64_t lo = a * (uint64_t)c, hi = mulhi64(a, (uint64_t)c);  // hi < 2^32
    uint64_t t1 = (hi << 32) - hi;
    uint64_t r = lo + t1;
    if (r < t1) r += EPS;
    return r >= P ? r - P : r;
}
GL_HD uint64_t pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = mul(a, a);
        e >>= 1;
    }
    return r;
}
GL_HD uint64_t inv(uint64_t a) { return pow(a, P - 2); }

struct alignas(16) E2 {
    uint64_t c0, c1;
};

GL_HD E2 e2(uint64_t c0, uint64_t c1) { return E2{c0, c1}; }
GL_HD E2 e2_zero() { return E2{0, 0}; }
GL_HD E2 e2_one() { return E2{1, 0}; }
GL_HD E2 e2_from_base(uint64_t b) { return E2{b, 0}; }
GL_HD bool e2_eq(E2 a, E2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
GL_HD E2 operator+(E2 a, E2 b) { return E2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
GL_HD E2 operator-(E2 a, E2 b) { return E2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
GL_HD E2 e2_neg(E2 a) { return E2{neg(a.c0), neg(a.c1)}; }
GL_HD E2 e2_dbl(E2 a) { return E2{dbl(a.c0), dbl(a.c1)}; }

// (a0 + a1 X)(b0 + b1 X) = (a0 b0 + 7 a1 b1) + ((a0+a1)(b0+b1) - a0 b0 - a1 b1) X   [Karatsuba, 3 mults]
GL_HD E2 operator*(E2 a, E2 b) {
    uint64_t m0 = mul(a.c0, b.c0);
    uint64_t m1 = mul(a.c1, b.c1);
    // (a0+a1) and (b0+b1) may exceed 64 bits: reduce first (canonical add)
    uint64_t m2 = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    uint64_t w = mul_small(m1, (uint32_t)W);
    return E2{add(m0, w), sub(sub(m2, m0), m1)};
}
GL_HD E2 e2_mul_base(E2 a, uint64_t b) { return E2{mul(a.c0, b), mul(a.c1, b)}; }
GL_HD E2 e2_sqr(E2 a) {
    uint64_t m0 = mul(a.c0, a.c0), m1 = mul(a.c1, a.c1), m2 = mul(a.c0, a.c1);
    return E2{add(m0, mul_small(m1, (uint32_t)W)), dbl(m2)};
}
GL_HD E2 e2_inv(E2 a) {
    uint64_t n = sub(mul(a.c0, a.c0), mul_small(mul(a.c1, a.c1), (uint32_t)W));
    uint64_t ni = inv(n);
    return E2{mul(a.c0, ni), mul(neg(a.c1), ni)};
}

// SplitMix64 stream used for synthetic inputs (BASELINE.md "Synthetic inputs")
GL_HD uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
GL_HD uint64_t splitmix_gl(uint64_t seed, uint64_t i) {
    uint64_t z = splitmix64_at(seed, i);
    return z >= P ? z - P : z;
}

}  // namespace gl
