// Extension-field arithmetic for the HOST-finished sumcheck tails, eight pairs per AVX-512 register set (sumcheck.hip: sc_host_round, host_fold).
// Built on the vector Goldilocks arithmetic of poseidon2_host.hpp: residues in [0, 2^64), one reduction per product; values are made canonical
// before they are stored or summed, so every word the host publishes equals the scalar code's (and the device's).
#pragma once
#include "poseidon2_host.hpp"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
namespace e2v {
using p2host::v_add;
using p2host::v_eps;
using p2host::v_mul;
#define E2V __attribute__((target("avx512f,avx512dq"), always_inline)) static inline
// a - b mod p for any residues a, b < 2^64
E2V __m512i v_sub(__m512i a, __m512i b) {
    const __m512i d = _mm512_sub_epi64(a, b);
    const __mmask8 c = _mm512_cmplt_epu64_mask(a, b);  // wrapped: d = a - b + 2^64, and 2^64 = EPS (mod p)
    const __m512i d2 = _mm512_mask_sub_epi64(d, c, d, v_eps());
    const __mmask8 c2 = _mm512_mask_cmplt_epu64_mask(c, d, v_eps());  // the correction itself wrapped
    return _mm512_mask_sub_epi64(d2, c2, d2, v_eps());
}
E2V __m512i v_canon(__m512i x) {
    const __m512i pp = _mm512_set1_epi64((long long)gl::P);
    return _mm512_mask_sub_epi64(x, _mm512_cmpge_epu64_mask(x, pp), x, pp);
}
struct VE2 {
    __m512i c0, c1;
};
E2V VE2 bcast(gl::E2 x) { return VE2{_mm512_set1_epi64((long long)x.c0), _mm512_set1_epi64((long long)x.c1)}; }
E2V VE2 add(VE2 a, VE2 b) { return VE2{v_add(a.c0, b.c0), v_add(a.c1, b.c1)}; }
E2V VE2 sub(VE2 a, VE2 b) { return VE2{v_sub(a.c0, b.c0), v_sub(a.c1, b.c1)}; }
// (a0 + a1 X)(b0 + b1 X) with X^2 = W = 7: three products (Karatsuba) and the multiplication by 7
E2V VE2 mul(VE2 a, VE2 b) {
    const __m512i m0 = v_mul(a.c0, b.c0), m1 = v_mul(a.c1, b.c1), m2 = v_mul(v_add(a.c0, a.c1), v_add(b.c0, b.c1));
    const __m512i w1 = v_mul(m1, _mm512_set1_epi64((long long)gl::W));
    return VE2{v_add(m0, w1), v_sub(v_sub(m2, m0), m1)};
}
// entries first, first + stride, ... (E2 units) of a table
E2V VE2 load(const gl::E2* t, size_t first, size_t stride) {
    const __m512i idx = _mm512_mullo_epi64(_mm512_setr_epi64(0, 1, 2, 3, 4, 5, 6, 7), _mm512_set1_epi64((long long)(2 * stride)));
    const long long* base = reinterpret_cast<const long long*>(t + first);
    return VE2{_mm512_i64gather_epi64(idx, base, 8), _mm512_i64gather_epi64(idx, base + 1, 8)};
}
// eight consecutive entries, canonical
E2V void store(gl::E2* t, size_t first, VE2 v) {
    const __m512i idx = _mm512_setr_epi64(0, 2, 4, 6, 8, 10, 12, 14);
    long long* base = reinterpret_cast<long long*>(t + first);
    _mm512_i64scatter_epi64(base, idx, v_canon(v.c0), 8);
    _mm512_i64scatter_epi64(base + 1, idx, v_canon(v.c1), 8);
}
// the sum of the eight lanes, canonical
__attribute__((target("avx512f,avx512dq"))) static inline gl::E2 hsum(VE2 v) {
    alignas(64) uint64_t a[8], b[8];
    _mm512_store_si512((__m512i*)a, v_canon(v.c0));
    _mm512_store_si512((__m512i*)b, v_canon(v.c1));
    gl::E2 s = gl::e2_zero();
    for (int i = 0; i < 8; i++) s = s + gl::E2{a[i], b[i]};
    return s;
}
#undef E2V
}  // namespace e2v
#endif
