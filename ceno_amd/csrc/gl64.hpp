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

// ---- 32-bit add/sub with carry: clang lowers the builtins to v_add_co/v_addc_co chains (carry kept
// in VCC / an SGPR pair, no compare+select), g++ gets portable equivalents for the host build ----
#if defined(__clang__)
GL_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t cin, uint32_t& cout) { return __builtin_addc(a, b, cin, &cout); }
GL_HD uint32_t subc32(uint32_t a, uint32_t b, uint32_t bin, uint32_t& bout) { return __builtin_subc(a, b, bin, &bout); }
#else
GL_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t cin, uint32_t& cout) {
    uint64_t s = (uint64_t)a + b + cin;
    cout = (uint32_t)(s >> 32);
    return (uint32_t)s;
}
GL_HD uint32_t subc32(uint32_t a, uint32_t b, uint32_t bin, uint32_t& bout) {
    uint64_t d = (uint64_t)a - b - bin;
    bout = (uint32_t)(d >> 63);
    return (uint32_t)d;
}
#endif
GL_HD uint64_t join(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// Two formulations of the conditional corrections (identical results): GL_ARITH64 = 1 writes them with 64-bit overflow
// builtins (the compiler picks v_lshl_add_u64 / v_cmp_*_u64 / v_cndmask: fewer VALU instructions and fewer VCC hazards;
// measured on MI355X with tools/ubench_red.hip: 2.15e12 vs 1.75e12 modular multiplications per second), GL_ARITH64 = 0
// keeps the explicit 32-bit carry chains.
#ifndef GL_ARITH64
#define GL_ARITH64 1
#endif
#if GL_ARITH64
GL_HD uint64_t add(uint64_t a, uint64_t b) {
    // s = a + b (65 bit); u = s - p = s + EPS (mod 2^64); take u when s >= p
    uint64_t s, u;
    const bool c = __builtin_add_overflow(a, b, &s);
    const bool c2 = __builtin_add_overflow(s, EPS, &u);
    return (c | c2) ? u : s;
}
GL_HD uint64_t sub(uint64_t a, uint64_t b) {
    uint64_t d;
    const bool bw = __builtin_sub_overflow(a, b, &d);
    return d - (bw ? EPS : 0);  // borrow: add p = subtract EPS
}
#else
GL_HD uint64_t add(uint64_t a, uint64_t b) {
    // s = a + b (65 bit); u = s - p = s + EPS (mod 2^64); take u when s >= p
    uint32_t c, c2;
    const uint32_t s0 = addc32((uint32_t)a, (uint32_t)b, 0u, c);
    const uint32_t s1 = addc32((uint32_t)(a >> 32), (uint32_t)(b >> 32), c, c);
    const uint32_t u0 = addc32(s0, 0xFFFFFFFFu, 0u, c2);
    const uint32_t u1 = addc32(s1, 0u, c2, c2);
    const bool take = (c | c2) != 0;  // overflowed 2^64, or s + EPS overflowed <=> s >= p
    return take ? join(u0, u1) : join(s0, s1);
}
GL_HD uint64_t sub(uint64_t a, uint64_t b) {
    uint32_t bw, b2;
    uint32_t d0 = subc32((uint32_t)a, (uint32_t)b, 0u, bw);
    uint32_t d1 = subc32((uint32_t)(a >> 32), (uint32_t)(b >> 32), bw, bw);
    const uint32_t m = 0u - bw;  // borrow: add p = subtract EPS
    d0 = subc32(d0, m, 0u, b2);
    d1 = subc32(d1, 0u, b2, b2);
    return join(d0, d1);
}
#endif
GL_HD uint64_t neg(uint64_t a) { return a ? P - a : 0; }
GL_HD uint64_t dbl(uint64_t a) { return add(a, a); }

// ---- 64x64 -> 128 product as four 32-bit limbs, from four 32x32+64 multiply-adds (v_mad_u64_u32) ----
struct L4 {
    uint32_t w0, w1, w2, w3;
};
GL_HD L4 mul_wide(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t p00 = (uint64_t)a0 * b0;
    const uint64_t t = (uint64_t)a0 * b1 + (p00 >> 32);              // <= (2^32-1)^2 + 2^32-1 < 2^64
    const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t;              // same bound
    const uint64_t hi = (uint64_t)a1 * b1 + (t >> 32) + (t2 >> 32);   // <= 2^64 - 1
    return L4{(uint32_t)p00, (uint32_t)t2, (uint32_t)hi, (uint32_t)(hi >> 32)};
}

// Reduce  (w1:w0) + w2*2^64 + w3*2^96 + c*2^128  (c < 16) to canonical form with
// 2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32 (mod p):   x = (w1:w0) + w2*(2^32-1) - (c:w3).
// Same value modulo p as reduce_limbs but only guaranteed to lie in [0, 2^64): enough for anything that
// is next fed to mul_wide (whose inputs are plain 64-bit integers) and four instructions shorter.
#ifndef GL_REDUCE_ASM
#define GL_REDUCE_ASM 1
#endif
#if GL_REDUCE_ASM && defined(__HIP_DEVICE_COMPILE__)
// Device form, 9 VALU instructions: t = w2 * (2^32 - 1) + (w1:w0) in ONE v_mad_u64_u32 (its 64-bit addend is free and the
// carry lands in an SGPR pair), t -= (c:w3) with borrow, then + (carry - borrow) * EPS.  Bounds as in the portable form
// below: after a carry t < 2^64 - 2^33, after a borrow t >= 2^64 - 2^36, and both together cancel.  The compiler does not
// fold the addition into the multiply-add nor reuse the carry flags (it re-derives them with 64-bit compares), hence the
// assembly; s_nop = the two wait states gfx950 wants between a VALU write of VCC / an SGPR and a VALU read of it as
// carry or mask.  tools/ubench_red.hip: 2.35e12 modular multiplications per second against 2.10e12 (builtins), 1.89e12 (limbs).
GL_HD uint64_t reduce_limbs_nc(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t c) {
    uint64_t t = join(w0, w1), cy;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %0\n\ts_nop 1" : "+v"(t), "=s"(cy) : "v"(w2));
    uint32_t r0 = (uint32_t)t, r1 = (uint32_t)(t >> 32), a, b;
    asm("v_sub_co_u32 %0, vcc, %0, %5\n\t"
        "v_cndmask_b32 %2, 0, -1, %4\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32 %1, vcc, %1, %6, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %3, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %3\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(r0), "+v"(r1), "=&v"(a), "=&v"(b)
        : "s"(cy), "v"(w3), "v"(c)
        : "vcc");
    return join(r0, r1);
}
// 128-bit input (no fifth limb): the borrow step subtracts with an inline zero
GL_HD uint64_t reduce128_nc(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    uint64_t t = join(w0, w1), cy;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %0\n\ts_nop 1" : "+v"(t), "=s"(cy) : "v"(w2));
    uint32_t r0 = (uint32_t)t, r1 = (uint32_t)(t >> 32), a, b;
    asm("v_sub_co_u32 %0, vcc, %0, %5\n\t"
        "v_cndmask_b32 %2, 0, -1, %4\n\t"
        "s_nop 0\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32 %3, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_sub_co_u32 %0, vcc, %0, %3\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(r0), "+v"(r1), "=&v"(a), "=&v"(b)
        : "s"(cy), "v"(w3)
        : "vcc");
    return join(r0, r1);
}
// The same reduction in 8 VALU instructions: carry (SGPR pair) and borrow (VCC) are combined on the SCALAR unit into "carry
// only" / "borrow only" masks (both together cancel), turned into ONE 64-bit correction D in {+EPS, 0, -EPS} = (a - b : b) with
// a = -[carry only], b = -[borrow only], and added with a single add / add-with-carry pair.  Measured on one box
// (tools/ab_arith.sh, AB_FLAGS=-DGL_REDUCE_MERGED=0): the Poseidon2 leaf hash gains 6 % (Merkle commit 4.34 -> 4.07 ms), the
// sumcheck kernels LOSE 2 % (two more scalar instructions per reduction compete with their address and loop arithmetic, and
// tools/ubench_red.hip's bare multiply chains lose 9 %), so only the hash (mul_ncm, mul_add_s96_ncm) uses this form.
#ifndef GL_REDUCE_MERGED
#define GL_REDUCE_MERGED 1
#endif
#if GL_REDUCE_MERGED
GL_HD uint64_t reduce128_ncm(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    uint64_t t = join(w0, w1), cy, ma, mb;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %0" : "+v"(t), "=s"(cy) : "v"(w2));
    uint32_t r0 = (uint32_t)t, r1 = (uint32_t)(t >> 32), a, b;
    asm("v_sub_co_u32 %0, vcc, %0, %7\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "s_andn2_b64 %4, %6, vcc\n\t"
        "s_andn2_b64 %5, vcc, %6\n\t"
        "v_cndmask_b32 %2, 0, -1, %4\n\t"
        "v_cndmask_b32 %3, 0, -1, %5\n\t"
        "v_sub_u32 %2, %2, %3\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, %1, %3, vcc"
        : "+v"(r0), "+v"(r1), "=&v"(a), "=&v"(b), "=&s"(ma), "=&s"(mb)
        : "s"(cy), "v"(w3)
        : "vcc", "scc");
    return join(r0, r1);
}
#else
GL_HD uint64_t reduce128_ncm(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) { return reduce128_nc(w0, w1, w2, w3); }
#endif
// 96-bit input (w2 * 2^64 + (w1:w0)): nothing to subtract, 4 instructions
#define GL_HAVE_REDUCE96 1
GL_HD uint64_t reduce96_nc(uint32_t w0, uint32_t w1, uint32_t w2) {
    uint64_t t = join(w0, w1), cy;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %0\n\ts_nop 1" : "+v"(t), "=s"(cy) : "v"(w2));
    uint32_t r0 = (uint32_t)t, r1 = (uint32_t)(t >> 32), a;
    asm("v_cndmask_b32 %2, 0, -1, %3\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(r0), "+v"(r1), "=&v"(a)
        : "s"(cy)
        : "vcc");
    return join(r0, r1);
}
GL_HD uint64_t canon(uint64_t r) {  // r + EPS overflows <=> r >= p
    uint64_t u;
    return __builtin_add_overflow(r, EPS, &u) ? u : r;
}
#elif GL_ARITH64
GL_HD uint64_t reduce_limbs_nc(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t c) {
    uint64_t t0, r;
    const bool bw = __builtin_sub_overflow(join(w0, w1), join(w3, c), &t0);
    t0 -= bw ? EPS : 0;              // wrapped by 2^64 = EPS (mod p); t0 >= 2^64 - 2^36 then, so no second wrap (c < 16)
    const uint64_t t1 = ((uint64_t)w2 << 32) - w2;  // w2 * (2^32 - 1) <= 2^64 - 2^33 + 1
    const bool cy = __builtin_add_overflow(t0, t1, &r);
    return r + (cy ? EPS : 0);       // wrapped again: add EPS; cannot wrap a third time
}
GL_HD uint64_t canon(uint64_t r) {  // r + EPS overflows <=> r >= p
    uint64_t u;
    return __builtin_add_overflow(r, EPS, &u) ? u : r;
}
#else
GL_HD uint64_t reduce_limbs_nc(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t c) {
    uint32_t bw, b2, cy, c2;
    uint32_t r0 = subc32(w0, w3, 0u, bw);
    uint32_t r1 = subc32(w1, c, bw, bw);
    const uint32_t m = 0u - bw;      // wrapped by 2^64 = EPS (mod p): subtract EPS; r >= 2^64 - 2^36 so no second wrap (c < 16)
    r0 = subc32(r0, m, 0u, b2);
    r1 = subc32(r1, 0u, b2, b2);
    // w2 * (2^32 - 1) without a multiply: low word = -w2, high word = w2 - (w2 != 0)
    const uint32_t t1l = subc32(0u, w2, 0u, b2);
    const uint32_t t1h = subc32(w2, 0u, b2, b2);
    r0 = addc32(r0, t1l, 0u, cy);
    r1 = addc32(r1, t1h, cy, cy);
    const uint32_t m2 = 0u - cy;     // wrapped again: add EPS; cannot wrap a third time (t1 <= 2^64 - 2^33 + 1)
    r0 = addc32(r0, m2, 0u, c2);
    r1 = addc32(r1, 0u, c2, c2);
    return join(r0, r1);
}
GL_HD uint64_t canon(uint64_t r) {  // r + EPS overflows <=> r >= p
    uint32_t cy;
    const uint32_t u0 = addc32((uint32_t)r, 0xFFFFFFFFu, 0u, cy);
    const uint32_t u1 = addc32((uint32_t)(r >> 32), 0u, cy, cy);
    return cy ? join(u0, u1) : r;
}
#endif
#ifndef GL_HAVE_REDUCE96
GL_HD uint64_t reduce96_nc(uint32_t w0, uint32_t w1, uint32_t w2) { return reduce_limbs_nc(w0, w1, w2, 0u, 0u); }
GL_HD uint64_t reduce128_nc(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) { return reduce_limbs_nc(w0, w1, w2, w3, 0u); }
GL_HD uint64_t reduce128_ncm(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) { return reduce_limbs_nc(w0, w1, w2, w3, 0u); }
#endif
GL_HD uint64_t reduce_limbs(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t c) {
    return canon(reduce_limbs_nc(w0, w1, w2, w3, c));
}
GL_HD uint64_t reduce128(uint64_t lo, uint64_t hi) {
    return canon(reduce128_nc((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)));
}
// Host code (the C++ prover layer: transcripts, host-side rounds, small tower layers) multiplies through the native
// 64 x 64 -> 128 product instead of the device's four 32-bit multiply-adds: ~3x fewer instructions, identical results.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
#define GL_HOST_INT128 1
inline uint64_t host_red128(unsigned __int128 x) {  // canonical residue of a 128-bit value: 2^64 = 2^32 - 1, 2^96 = -1
    const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    uint64_t t0, r;
    if (__builtin_sub_overflow(lo, hi >> 32, &t0)) t0 -= EPS;
    if (__builtin_add_overflow(t0, (hi & EPS) * EPS, &r)) r += EPS;
    return canon(r);
}
#endif
GL_HD uint64_t mul(uint64_t a, uint64_t b) {
#ifdef GL_HOST_INT128
    return host_red128((unsigned __int128)a * b);
#else
    const L4 p = mul_wide(a, b);
    return canon(reduce128_nc(p.w0, p.w1, p.w2, p.w3));
#endif
}
// a*b + c*d with a single reduction (129-bit sum)
GL_HD uint64_t mul_add2(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
#ifdef GL_HOST_INT128
    const unsigned __int128 p = (unsigned __int128)a * b, q = (unsigned __int128)c * d, s = p + q;
    const uint64_t r = host_red128(s);
    return s < p ? sub(r, (uint64_t)1 << 32) : r;  // the carry out of bit 128: 2^128 = -2^32 (mod p)
#else
    const L4 p = mul_wide(a, b), q = mul_wide(c, d);
    uint32_t cy;
    const uint32_t s0 = addc32(p.w0, q.w0, 0u, cy);
    const uint32_t s1 = addc32(p.w1, q.w1, cy, cy);
    const uint32_t s2 = addc32(p.w2, q.w2, cy, cy);
    const uint32_t s3 = addc32(p.w3, q.w3, cy, cy);
    return reduce_limbs(s0, s1, s2, s3, cy);
#endif
}
// product that is only multiplied again (any 64-bit inputs, result in [0, 2^64) not canonical)
GL_HD uint64_t mul_nc(uint64_t a, uint64_t b) {
    const L4 p = mul_wide(a, b);
    return reduce128_nc(p.w0, p.w1, p.w2, p.w3);
}
// the same product through the 8-instruction reduction (hash kernels, see reduce128_ncm)
GL_HD uint64_t mul_ncm(uint64_t a, uint64_t b) {
    const L4 p = mul_wide(a, b);
    return reduce128_ncm(p.w0, p.w1, p.w2, p.w3);
}
// a*b + c with one reduction (c < 2^64: the sum stays below 2^128), canonical result
GL_HD uint64_t mul_add(uint64_t a, uint64_t b, uint64_t c) {
    const L4 p = mul_wide(a, b);
    uint32_t cy;
    const uint32_t s0 = addc32(p.w0, (uint32_t)c, 0u, cy);
    const uint32_t s1 = addc32(p.w1, (uint32_t)(c >> 32), cy, cy);
    const uint32_t s2 = addc32(p.w2, 0u, cy, cy);
    const uint32_t s3 = addc32(p.w3, 0u, cy, cy);
    return canon(reduce128_nc(s0, s1, s2, s3));
}
GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }
// ---- lazy sums for hash linear layers: a few 64-bit values added without reduction ----
// a (any 64-bit value) + b (canonical): same residue, result in [0, 2^64) — one wrap at most since a + b - 2^64 <= p - 2
#if GL_ARITH64
GL_HD uint64_t add_nc(uint64_t a, uint64_t b) {
    uint64_t s;
    const bool c = __builtin_add_overflow(a, b, &s);
    return s + (c ? EPS : 0);
}
#else
GL_HD uint64_t add_nc(uint64_t a, uint64_t b) {
    uint32_t c, c2;
    uint32_t s0 = addc32((uint32_t)a, (uint32_t)b, 0u, c);
    uint32_t s1 = addc32((uint32_t)(a >> 32), (uint32_t)(b >> 32), c, c);
    const uint32_t m = 0u - c;
    s0 = addc32(s0, m, 0u, c2);
    s1 = addc32(s1, 0u, c2, c2);
    return join(s0, s1);
}
#endif
// 96-bit plain integer sum of up to 2^32 arbitrary 64-bit values
struct S96 {
    uint32_t w0, w1, w2;
};
GL_HD S96 s96(uint64_t a) { return S96{(uint32_t)a, (uint32_t)(a >> 32), 0u}; }
GL_HD S96 operator+(S96 a, uint64_t b) {
    uint32_t c;
    S96 r;
    r.w0 = addc32(a.w0, (uint32_t)b, 0u, c);
    r.w1 = addc32(a.w1, (uint32_t)(b >> 32), c, c);
    r.w2 = addc32(a.w2, 0u, c, c);  // stays in the carry chain (v_addc_co), no carry -> integer round trip
    return r;
}
GL_HD S96 operator+(S96 a, S96 b) {
    uint32_t c;
    S96 r;
    r.w0 = addc32(a.w0, b.w0, 0u, c);
    r.w1 = addc32(a.w1, b.w1, c, c);
    r.w2 = addc32(a.w2, b.w2, c, c);
    return r;
}
GL_HD S96 s96_sum(uint64_t a, uint64_t b) {
    uint32_t c;
    S96 r;
    r.w0 = addc32((uint32_t)a, (uint32_t)b, 0u, c);
    r.w1 = addc32((uint32_t)(a >> 32), (uint32_t)(b >> 32), c, c);
    r.w2 = addc32(0u, 0u, c, c);
    return r;
}
// residue of a 96-bit sum in [0, 2^64) (w2 * 2^64 = w2 * (2^32 - 1))
GL_HD uint64_t s96_reduce_nc(S96 a) { return reduce96_nc(a.w0, a.w1, a.w2); }
// a*b + s for a < 2^64, b canonical, s < 2^67: the sum stays below 2^128; result in [0, 2^64)
GL_HD uint64_t mul_add_s96_nc(uint64_t a, uint64_t b, S96 s) {
    const L4 p = mul_wide(a, b);
    uint32_t cy;
    const uint32_t s0 = addc32(p.w0, s.w0, 0u, cy);
    const uint32_t s1 = addc32(p.w1, s.w1, cy, cy);
    const uint32_t s2 = addc32(p.w2, s.w2, cy, cy);
    const uint32_t s3 = addc32(p.w3, 0u, cy, cy);
    return reduce128_nc(s0, s1, s2, s3);
}
GL_HD uint64_t mul_add_s96_ncm(uint64_t a, uint64_t b, S96 s) {
    const L4 p = mul_wide(a, b);
    uint32_t cy;
    const uint32_t s0 = addc32(p.w0, s.w0, 0u, cy);
    const uint32_t s1 = addc32(p.w1, s.w1, cy, cy);
    const uint32_t s2 = addc32(p.w2, s.w2, cy, cy);
    const uint32_t s3 = addc32(p.w3, 0u, cy, cy);
    return reduce128_ncm(s0, s1, s2, s3);
}
// small-constant multiply (c < 2^32): the product has 96 bits
GL_HD uint64_t mul_small(uint64_t a, uint32_t c) {
#ifdef GL_HOST_INT128
    return host_red128((unsigned __int128)a * c);
#else
    const uint64_t p0 = (uint64_t)(uint32_t)a * c;
    const uint64_t p1 = (uint64_t)(uint32_t)(a >> 32) * c + (p0 >> 32);
    return canon(reduce96_nc((uint32_t)p0, (uint32_t)p1, (uint32_t)(p1 >> 32)));
#endif
}
// reference forms kept for cross-checks (tests/test_host_cpu.py)
GL_HD uint64_t mul_ref(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t lo = a * b, hi = __umul64hi(a, b);
#else
    unsigned __int128 w = (unsigned __int128)a * b;
    uint64_t lo = (uint64_t)w, hi = (uint64_t)(w >> 64);
#endif
    uint64_t hi_hi = hi >> 32, hi_lo = hi & EPS;
    uint64_t t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= EPS;
    uint64_t t1 = (hi_lo << 32) - hi_lo;
    uint64_t r = t0 + t1;
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

// (a0 + a1 X)(b0 + b1 X) = (a0 b0 + (7 a1) b1) + (a0 b1 + a1 b0) X
// schoolbook with lazy reduction: four 128-bit products, two 129-bit sums, two reductions.
GL_HD E2 operator*(E2 a, E2 b) {
    const uint64_t a1w = mul_small(a.c1, (uint32_t)W);
    return E2{mul_add2(a.c0, b.c0, a1w, b.c1), mul_add2(a.c0, b.c1, a.c1, b.c0)};
}
// Products whose result is only used as a multiplicand again: components in [0, 2^64), not canonical.
GL_HD uint64_t mul_add2_nc(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    const L4 p = mul_wide(a, b), q = mul_wide(c, d);
    uint32_t cy;
    const uint32_t s0 = addc32(p.w0, q.w0, 0u, cy);
    const uint32_t s1 = addc32(p.w1, q.w1, cy, cy);
    const uint32_t s2 = addc32(p.w2, q.w2, cy, cy);
    const uint32_t s3 = addc32(p.w3, q.w3, cy, cy);
    return reduce_limbs_nc(s0, s1, s2, s3, cy);
}
GL_HD E2 e2_mul_nc(E2 a, E2 b) {
    const uint64_t p0 = (uint64_t)(uint32_t)a.c1 * (uint32_t)W;
    const uint64_t p1 = (uint64_t)(uint32_t)(a.c1 >> 32) * (uint32_t)W + (p0 >> 32);
    const uint64_t a1w = reduce96_nc((uint32_t)p0, (uint32_t)p1, (uint32_t)(p1 >> 32));
    return E2{mul_add2_nc(a.c0, b.c0, a1w, b.c1), mul_add2_nc(a.c0, b.c1, a.c1, b.c0)};
}
// a*b + c*d + e with a single reduction (e < 2^64)
GL_HD uint64_t mul_add2_plus(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e) {
    const L4 p = mul_wide(a, b), q = mul_wide(c, d);
    uint32_t cy, c2, top;
    uint32_t s0 = addc32(p.w0, q.w0, 0u, cy);
    uint32_t s1 = addc32(p.w1, q.w1, cy, cy);
    uint32_t s2 = addc32(p.w2, q.w2, cy, cy);
    uint32_t s3 = addc32(p.w3, q.w3, cy, cy);
    s0 = addc32(s0, (uint32_t)e, 0u, c2);
    s1 = addc32(s1, (uint32_t)(e >> 32), c2, c2);
    s2 = addc32(s2, 0u, c2, c2);
    s3 = addc32(s3, 0u, c2, c2);
    top = cy + c2;
    return reduce_limbs(s0, s1, s2, s3, top);
}
// multiply by a fixed element whose W*c1 is precomputed (sumcheck challenge r)
struct E2Pre {
    uint64_t c0, c1, c1w;
};
GL_HD E2Pre e2_pre(E2 r) { return E2Pre{r.c0, r.c1, mul_small(r.c1, (uint32_t)W)}; }
GL_HD E2 e2_mul_pre(E2Pre r, E2 b) { return E2{mul_add2(r.c0, b.c0, r.c1w, b.c1), mul_add2(r.c0, b.c1, r.c1, b.c0)}; }
// a + r * b for the fixed element r (the MLE fold lo + r (hi - lo)), canonical result
GL_HD E2 e2_fma_pre(E2Pre r, E2 b, E2 a) {
    return E2{mul_add2_plus(r.c0, b.c0, r.c1w, b.c1, a.c0), mul_add2_plus(r.c0, b.c1, r.c1, b.c0, a.c1)};
}
// Karatsuba form with three full multiplications, kept as an independent cross-check
GL_HD E2 e2_mul_ref(E2 a, E2 b) {
    uint64_t m0 = mul_ref(a.c0, b.c0);
    uint64_t m1 = mul_ref(a.c1, b.c1);
    uint64_t m2 = mul_ref(add(a.c0, a.c1), add(b.c0, b.c1));
    uint64_t w = mul_ref(m1, W);
    return E2{add(m0, w), sub(sub(m2, m0), m1)};
}
// ---- unreduced accumulation of ext products (sumcheck inner loops) ----
// Sum_i a_i * b_i over many i without reducing each product: the four 128-bit partial products of an
// ext multiply are added limb-wise into 160-bit accumulators (room for 2^32 products) and reduced once.
//   c0 = S(a0 b0) + W * S(a1 b1),   c1 = S(a0 b1) + S(a1 b0)
struct Acc5 {
    uint32_t w0, w1, w2, w3, w4;
};
GL_HD void acc5_add(Acc5& a, const L4& p) {
    uint32_t cy;
    a.w0 = addc32(a.w0, p.w0, 0u, cy);
    a.w1 = addc32(a.w1, p.w1, cy, cy);
    a.w2 = addc32(a.w2, p.w2, cy, cy);
    a.w3 = addc32(a.w3, p.w3, cy, cy);
    a.w4 = addc32(a.w4, 0u, cy, cy);
}
// w4 * 2^128 = -(w4 << 32) (mod p), and (w4 << 32) <= p - 1 is canonical
GL_HD uint64_t acc5_reduce(const Acc5& a) {
    return sub(canon(reduce128_nc(a.w0, a.w1, a.w2, a.w3)), (uint64_t)a.w4 << 32);
}
struct E2Acc {
    Acc5 s00, s11, s01;
};
GL_HD E2Acc e2acc_zero() { return E2Acc{Acc5{0, 0, 0, 0, 0}, Acc5{0, 0, 0, 0, 0}, Acc5{0, 0, 0, 0, 0}}; }
GL_HD void e2acc_mac(E2Acc& acc, E2 a, E2 b) {
    acc5_add(acc.s00, mul_wide(a.c0, b.c0));
    acc5_add(acc.s11, mul_wide(a.c1, b.c1));
    acc5_add(acc.s01, mul_wide(a.c0, b.c1));
    acc5_add(acc.s01, mul_wide(a.c1, b.c0));
}
GL_HD E2 e2acc_reduce(const E2Acc& acc) {
    return E2{add(acc5_reduce(acc.s00), mul_small(acc5_reduce(acc.s11), (uint32_t)W)), acc5_reduce(acc.s01)};
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
