/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/).  Nothing under oracle/ is part of the
 * shipped product path.
 *
 * AVX-512 form of orc_sumcheck_dense_mt (oracle.c): the same fused schedule (round 0 accumulates over the inputs; round
 * i > 0 folds the table of round i - 1 with r_{i-1}, writes the half-size table and accumulates the message of round i in
 * the same pass), eight pairs per vector iteration, OpenMP over the iterations.  Why it exists: the reference's CPU prover
 * (IOPProverState::prove under rayon, call site ceno_zkvm/src/scheme/cpu/mod.rs:490-493) runs on p3-goldilocks' PACKED
 * field (PackedGoldilocksAVX512, crates.io p3-goldilocks 0.4.3, reference Cargo.lock:4142-4375) — a scalar port is a
 * strawman beside it.  This is NOT that crate: it is an independent statement of 64 x 64 -> 128-bit products from four
 * 32 x 32 multiplies (vpmuludq) and the 2^64 = 2^32 - 1, 2^96 = -1 reduction, validated word for word against the scalar
 * restatement (tests/test_oracle_golden.py::test_dense_avx512_equals_scalar).
 *
 * Values are residues in [0, 2^64) between operations and made canonical before they are stored or summed, so every
 * message, folded table and final evaluation equals the scalar code's.
 */
#include <immintrin.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "gl64.h"
#include "oracle.h"

#define AVX __attribute__((target("avx512f,avx512dq"), always_inline)) static inline
#define EPS 0xFFFFFFFFULL

int orc_have_avx512(void) { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"); }

AVX __m512i v_eps(void) { return _mm512_set1_epi64((long long)EPS); }
/* a + b mod p for any residues a, b < 2^64 */
AVX __m512i v_add(__m512i a, __m512i b) {
    const __m512i s = _mm512_add_epi64(a, b);
    const __mmask8 c = _mm512_cmplt_epu64_mask(s, a);        /* wrapped: 2^64 = EPS */
    const __m512i s2 = _mm512_mask_add_epi64(s, c, s, v_eps());
    const __mmask8 c2 = _mm512_mask_cmplt_epu64_mask(c, s2, s);
    return _mm512_mask_add_epi64(s2, c2, s2, v_eps());
}
AVX __m512i v_sub(__m512i a, __m512i b) {
    const __m512i d = _mm512_sub_epi64(a, b);
    const __mmask8 c = _mm512_cmplt_epu64_mask(a, b);        /* wrapped: d = a - b + 2^64 */
    const __m512i d2 = _mm512_mask_sub_epi64(d, c, d, v_eps());
    const __mmask8 c2 = _mm512_mask_cmplt_epu64_mask(c, d, v_eps());
    return _mm512_mask_sub_epi64(d2, c2, d2, v_eps());
}
AVX __m512i v_canon(__m512i x) {
    const __m512i pp = _mm512_set1_epi64((long long)GL_P);
    return _mm512_mask_sub_epi64(x, _mm512_cmpge_epu64_mask(x, pp), x, pp);
}
/* 64 x 64 -> 128 from four 32 x 32 -> 64 products, then lo - (hi >> 32) + (hi & EPS) * EPS */
AVX __m512i v_mul(__m512i a, __m512i b) {
    const __m512i m32 = v_eps();
    const __m512i ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
    const __m512i ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b), hh = _mm512_mul_epu32(ah, bh);
    const __m512i t0 = _mm512_add_epi64(hl, _mm512_srli_epi64(ll, 32));
    const __m512i t1 = _mm512_add_epi64(lh, _mm512_and_si512(t0, m32));
    const __m512i hi = _mm512_add_epi64(_mm512_add_epi64(hh, _mm512_srli_epi64(t0, 32)), _mm512_srli_epi64(t1, 32));
    const __m512i lo = _mm512_or_si512(_mm512_slli_epi64(t1, 32), _mm512_and_si512(ll, m32));
    const __m512i hhi = _mm512_srli_epi64(hi, 32), hlo = _mm512_and_si512(hi, m32);
    __m512i x = _mm512_sub_epi64(lo, hhi);
    const __mmask8 bw = _mm512_cmplt_epu64_mask(lo, hhi);
    x = _mm512_mask_sub_epi64(x, bw, x, m32);
    const __m512i y = _mm512_sub_epi64(_mm512_slli_epi64(hlo, 32), hlo);   /* hlo * EPS */
    const __m512i r = _mm512_add_epi64(x, y);
    const __mmask8 c = _mm512_cmplt_epu64_mask(r, x);
    return _mm512_mask_add_epi64(r, c, r, m32);
}
typedef struct { __m512i c0, c1; } ve2;
AVX ve2 e_add(ve2 a, ve2 b) { ve2 r = {v_add(a.c0, b.c0), v_add(a.c1, b.c1)}; return r; }
AVX ve2 e_sub(ve2 a, ve2 b) { ve2 r = {v_sub(a.c0, b.c0), v_sub(a.c1, b.c1)}; return r; }
/* (a0 + a1 X)(b0 + b1 X), X^2 = W: three products (Karatsuba) and the multiplication by W */
AVX ve2 e_mul(ve2 a, ve2 b) {
    const __m512i m0 = v_mul(a.c0, b.c0), m1 = v_mul(a.c1, b.c1), m2 = v_mul(v_add(a.c0, a.c1), v_add(b.c0, b.c1));
    const __m512i w1 = v_mul(m1, _mm512_set1_epi64((long long)GL_W));
    ve2 r = {v_add(m0, w1), v_sub(v_sub(m2, m0), m1)};
    return r;
}
AVX ve2 e_bcast(ext2 x) { ve2 r = {_mm512_set1_epi64((long long)x.c[0]), _mm512_set1_epi64((long long)x.c[1])}; return r; }
/* ext element `k` of eight consecutive groups of `stride` words starting at base */
AVX ve2 e_gather(const uint64_t* base, int stride_words, int k) {
    const __m512i idx = _mm512_mullo_epi64(_mm512_setr_epi64(0, 1, 2, 3, 4, 5, 6, 7), _mm512_set1_epi64(stride_words));
    const long long* b = (const long long*)base + 2 * k;
    ve2 r = {_mm512_i64gather_epi64(idx, b, 8), _mm512_i64gather_epi64(idx, b + 1, 8)};
    return r;
}
AVX void e_scatter(uint64_t* base, int stride_words, int k, ve2 v) {
    const __m512i idx = _mm512_mullo_epi64(_mm512_setr_epi64(0, 1, 2, 3, 4, 5, 6, 7), _mm512_set1_epi64(stride_words));
    long long* b = (long long*)base + 2 * k;
    _mm512_i64scatter_epi64(b, idx, v_canon(v.c0), 8);
    _mm512_i64scatter_epi64(b + 1, idx, v_canon(v.c1), 8);
}
__attribute__((target("avx512f,avx512dq"))) static ext2 e_hsum(ve2 v) {
    uint64_t a[8] __attribute__((aligned(64))), b[8] __attribute__((aligned(64)));
    _mm512_store_si512((__m512i*)a, v_canon(v.c0));
    _mm512_store_si512((__m512i*)b, v_canon(v.c1));
    ext2 s = e2_zero();
    for (int i = 0; i < 8; i++) { ext2 t = {{a[i], b[i]}}; s = e2_add(s, t); }
    return s;
}

static inline ext2 ld2(const uint64_t* p) { ext2 r = {{p[0], p[1]}}; return r; }
static inline void st2(uint64_t* p, ext2 v) { p[0] = v.c[0]; p[1] = v.c[1]; }

/* one round over the pairs [0, pairs): fold != 0 reads groups of four (a0, a1, a2, a3), folds with r and writes (lo, hi) to nxt */
__attribute__((target("avx512f,avx512dq"))) static void round_avx512(const uint64_t* const* cur, uint64_t* const* nxt, int k, size_t pairs, int fold, ext2 r,
                                                                       ext2* msg) {
    const int d = k;
    const ve2 vr = e_bcast(r);
    const size_t blocks = pairs / 8;
#pragma omp parallel
    {
        ve2 acc[8];
        ext2 loc[8];
        for (int t = 0; t < d; t++) { acc[t] = e_bcast(e2_zero()); loc[t] = e2_zero(); }
#pragma omp for schedule(static) nowait
        for (size_t b = 0; b < blocks; b++) {
            ve2 prod[8];
            for (int m = 0; m < k; m++) {
                ve2 lo, hi;
                if (!fold) {
                    const uint64_t* src = cur[m] + 32 * b;                 /* eight pairs of (lo, hi) */
                    lo = e_gather(src, 4, 0);
                    hi = e_gather(src, 4, 1);
                } else {
                    const uint64_t* src = cur[m] + 64 * b;                 /* eight groups of (a0, a1, a2, a3) */
                    const ve2 a0 = e_gather(src, 8, 0), a1 = e_gather(src, 8, 1), a2 = e_gather(src, 8, 2), a3 = e_gather(src, 8, 3);
                    lo = e_add(a0, e_mul(vr, e_sub(a1, a0)));
                    hi = e_add(a2, e_mul(vr, e_sub(a3, a2)));
                    /* (the scalar code evaluates on the canonical folded values: canonical form is unique, so are the products) */
                    e_scatter(nxt[m] + 32 * b, 4, 0, lo);
                    e_scatter(nxt[m] + 32 * b, 4, 1, hi);
                }
                const ve2 delta = e_sub(hi, lo);
                ve2 v = hi;
                for (int t = 0; t < d; t++) {
                    prod[t] = m == 0 ? v : e_mul(prod[t], v);
                    v = e_add(v, delta);
                }
            }
            for (int t = 0; t < d; t++) acc[t] = e_add(acc[t], prod[t]);
        }
        /* the pairs that do not fill a vector (rounds of fewer than eight pairs: all of them), by one thread, scalar */
#pragma omp single nowait
        for (size_t p = blocks * 8; p < pairs; p++) {
            ext2 prod[8];
            for (int t = 0; t < d; t++) prod[t] = e2_one();
            for (int m = 0; m < k; m++) {
                ext2 lo, hi;
                if (!fold) {
                    lo = ld2(cur[m] + 4 * p);
                    hi = ld2(cur[m] + 4 * p + 2);
                } else {
                    const uint64_t* src = cur[m] + 8 * p;
                    const ext2 a0 = ld2(src), a1 = ld2(src + 2), a2 = ld2(src + 4), a3 = ld2(src + 6);
                    lo = e2_add(a0, e2_mul(r, e2_sub(a1, a0)));
                    hi = e2_add(a2, e2_mul(r, e2_sub(a3, a2)));
                    st2(nxt[m] + 4 * p, lo);
                    st2(nxt[m] + 4 * p + 2, hi);
                }
                ext2 delta = e2_sub(hi, lo), v = hi;
                for (int t = 0; t < d; t++) { prod[t] = e2_mul(prod[t], v); v = e2_add(v, delta); }
            }
            for (int t = 0; t < d; t++) loc[t] = e2_add(loc[t], prod[t]);
        }
        for (int t = 0; t < d; t++) loc[t] = e2_add(loc[t], e_hsum(acc[t]));
#pragma omp critical
        { for (int t = 0; t < d; t++) msg[t] = e2_add(msg[t], loc[t]); }
    }
}

/* same contract as orc_sumcheck_dense_mt (oracle.c); -2 when the CPU has no AVX-512 F + DQ */
int orc_sumcheck_dense_mt_avx512(uint64_t** bufs, int k, int num_vars, const uint64_t* challenges, int threads, uint64_t* out_msgs,
                                 uint64_t* out_final_evals) {
    if (k < 1 || k > 8 || num_vars < 1) return -1;
    if (!orc_have_avx512()) return -2;
    const int d = k;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    const uint64_t* cur[8];
    uint64_t* nxt[8];
    for (int m = 0; m < k; m++) { cur[m] = bufs[m]; nxt[m] = NULL; }
    size_t cur_len = (size_t)1 << num_vars;
    int which = 0;
    for (int round = 0; round < num_vars; round++) {
        ext2 msg[8];
        for (int t = 0; t < d; t++) msg[t] = e2_zero();
        if (round == 0) {
            round_avx512(cur, nxt, k, cur_len / 2, 0, e2_zero(), msg);
        } else {
            const ext2 r = ld2(challenges + 2 * (round - 1));
            const size_t new_len = cur_len / 2;
            for (int m = 0; m < k; m++) nxt[m] = bufs[(which ? 2 * k : k) + m];
            round_avx512(cur, nxt, k, new_len / 2, 1, r, msg);
            for (int m = 0; m < k; m++) cur[m] = nxt[m];
            cur_len = new_len;
            which ^= 1;
        }
        for (int t = 0; t < d; t++) st2(out_msgs + 2 * ((size_t)round * d + t), msg[t]);
    }
    const ext2 r = ld2(challenges + 2 * (num_vars - 1));
    for (int m = 0; m < k; m++) {
        const ext2 a0 = ld2(cur[m]), a1 = ld2(cur[m] + 2);
        st2(out_final_evals + 2 * m, e2_add(a0, e2_mul(r, e2_sub(a1, a0))));
    }
    return 0;
}
