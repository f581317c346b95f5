/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h / gl64.h for scope and parity status).
 * Plain-C restatement of the reference's CPU algorithms for the GKR/sumcheck hot path.
 * Each function cites the reference file:line whose behaviour it follows.
 */
#include "oracle.h"
#include "gl64.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline ext2 ld2(const uint64_t* p) { ext2 r = {{p[0], p[1]}}; return r; }
static inline void st2(uint64_t* p, ext2 v) { p[0] = v.c[0]; p[1] = v.c[1]; }
static inline ext2 ld_mle(const uint64_t* d, int is_ext, size_t i) {
    if (is_ext) return ld2(d + 2 * i);
    return e2_from_base(d[i]);
}

/* ------------------------------------------------------------------------------------------
 * stub transcript
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
void orc_stub_init(orc_stub_state* st, uint64_t seed) { st->s = mix64(seed); }
static void stub_absorb(orc_stub_state* st, uint64_t w) { st->s = mix64(st->s ^ w); }
void orc_stub_append_label(orc_stub_state* st, const uint8_t* bytes, size_t n) {
    stub_absorb(st, 0x4c41424c00000000ULL | (uint64_t)n); /* "LABL" | len */
    for (size_t i = 0; i < n; i += 8) {
        uint64_t w = 0;
        for (size_t k = 0; k < 8 && i + k < n; k++) w |= (uint64_t)bytes[i + k] << (8 * k);
        stub_absorb(st, w);
    }
}
void orc_stub_append_ext(orc_stub_state* st, const uint64_t* e2) { stub_absorb(st, e2[0]); stub_absorb(st, e2[1]); }
void orc_stub_sample_ext(orc_stub_state* st, uint64_t* out2) {
    st->s = mix64(st->s); out2[0] = gl_reduce(st->s);
    st->s = mix64(st->s); out2[1] = gl_reduce(st->s);
}
void orc_stub_append_base(orc_stub_state* st, uint64_t v) {
    stub_absorb(st, 0x4241534500000000ULL); /* "BASE" */
    stub_absorb(st, v);
}
uint64_t orc_stub_sample_bits(orc_stub_state* st, int bits) {
    st->s = mix64(st->s);
    const uint64_t v = gl_reduce(st->s);
    return bits >= 64 ? v : v & (((uint64_t)1 << bits) - 1);
}
static void stub_al(void* s, const uint8_t* b, size_t n) { orc_stub_append_label((orc_stub_state*)s, b, n); }
static void stub_ae(void* s, const uint64_t* e) { orc_stub_append_ext((orc_stub_state*)s, e); }
static void stub_se(void* s, uint64_t* o) { orc_stub_sample_ext((orc_stub_state*)s, o); }
static void stub_ab(void* s, uint64_t v) { orc_stub_append_base((orc_stub_state*)s, v); }
static uint64_t stub_sb(void* s, int bits) { return orc_stub_sample_bits((orc_stub_state*)s, bits); }
static void* stub_fork(void* s) {
    orc_stub_state* c = malloc(sizeof(*c));
    *c = *(orc_stub_state*)s;
    return c;
}
void orc_stub_bind(orc_transcript* t, orc_stub_state* st) {
    t->append_label = stub_al; t->append_ext = stub_ae; t->sample_ext = stub_se; t->self = st;
    t->append_base = stub_ab; t->sample_bits = stub_sb; t->fork = stub_fork; t->fork_free = free;
}

static void tr_label(orc_transcript* t, const char* s) { t->append_label(t->self, (const uint8_t*)s, strlen(s)); }
static void tr_usize(orc_transcript* t, uint64_t v) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i)); /* usize::to_le_bytes */
    t->append_label(t->self, b, 8);
}
static void tr_ext(orc_transcript* t, ext2 e) { t->append_ext(t->self, e.c); }
static ext2 tr_sample(orc_transcript* t) { ext2 r; t->sample_ext(t->self, r.c); return r; }
/* get_challenge_pows: label b"combine subset evals", one sample, powers
 * (ceno_recursion_v2/src/main/mod.rs:3459-3468; tower/mod.rs:1560-1565) */
__attribute__((unused)) static void tr_challenge_pows(orc_transcript* t, int n, ext2* out) {
    tr_label(t, "combine subset evals");
    ext2 a = tr_sample(t);
    ext2 acc = e2_one();
    for (int i = 0; i < n; i++) { out[i] = acc; acc = e2_mul(acc, a); }
}

/* ------------------------------------------------------------------------------------------
 * exported field helpers
 * ---------------------------------------------------------------------------------------- */
uint64_t orc_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a, b); }
uint64_t orc_gl_mul_div(uint64_t a, uint64_t b) { return gl_mul_div(a, b); }
uint64_t orc_gl_inv(uint64_t a) { return gl_inv(a); }
void orc_e2_mul(const uint64_t* a, const uint64_t* b, uint64_t* out) { st2(out, e2_mul(ld2(a), ld2(b))); }
void orc_e2_inv(const uint64_t* a, uint64_t* out) { st2(out, e2_inv(ld2(a))); }
void orc_fill_splitmix(uint64_t* out, size_t n_words, uint64_t seed, uint64_t word_offset) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n_words; i++) out[i] = splitmix_gl(seed, word_offset + i);
}

/* ------------------------------------------------------------------------------------------
 * MLE primitives
 * ---------------------------------------------------------------------------------------- */
/* build_eq_x_r_vec (EXT multilinear_extensions::virtual_poly; call sites
 * gkr_iop/src/selector.rs:140,152,166,194).  eq[i] = prod_k (bit_k(i) ? r_k : 1 - r_k),
 * bit k of the index <-> variable k (gkr_iop/src/utils.rs:215-232). */
void orc_build_eq_x_r_vec(const uint64_t* point, int n, uint64_t* out) {
    st2(out, e2_one());
    size_t len = 1;
    for (int k = 0; k < n; k++) {
        ext2 r = ld2(point + 2 * k);
        for (size_t i = 0; i < len; i++) {
            ext2 v = ld2(out + 2 * i);
            ext2 hi = e2_mul(v, r);
            st2(out + 2 * (i + len), hi);
            st2(out + 2 * i, e2_sub(v, hi));
        }
        len <<= 1;
    }
}
void orc_eq_eval(const uint64_t* a, const uint64_t* b, int n, uint64_t* out2) {
    ext2 acc = e2_one();
    for (int i = 0; i < n; i++) {
        ext2 x = ld2(a + 2 * i), y = ld2(b + 2 * i);
        ext2 xy = e2_mul(x, y);
        /* xy + (1-x)(1-y) = 2xy - x - y + 1 */
        ext2 t = e2_add(e2_sub(e2_sub(e2_add(xy, xy), x), y), e2_one());
        acc = e2_mul(acc, t);
    }
    st2(out2, acc);
}
void orc_mle_fix_variable(const uint64_t* evals, int is_ext, int num_vars, const uint64_t* r2, uint64_t* out) {
    size_t half = (size_t)1 << (num_vars - 1);
    ext2 r = ld2(r2);
    for (size_t j = 0; j < half; j++) {
        ext2 lo = ld_mle(evals, is_ext, 2 * j), hi = ld_mle(evals, is_ext, 2 * j + 1);
        st2(out + 2 * j, e2_add(lo, e2_mul(r, e2_sub(hi, lo))));
    }
}
void orc_mle_evaluate(const uint64_t* evals, int is_ext, int num_vars, const uint64_t* point, uint64_t* out2) {
    size_t len = (size_t)1 << num_vars;
    ext2* buf = (ext2*)malloc(sizeof(ext2) * len);
    for (size_t i = 0; i < len; i++) buf[i] = ld_mle(evals, is_ext, i);
    for (int k = 0; k < num_vars; k++) {
        ext2 r = ld2(point + 2 * k);
        len >>= 1;
        for (size_t j = 0; j < len; j++) buf[j] = e2_add(buf[2 * j], e2_mul(r, e2_sub(buf[2 * j + 1], buf[2 * j])));
    }
    st2(out2, buf[0]);
    free(buf);
}
/* extrapolate_uni_poly: value at x of the degree-d polynomial with p(0)=p0, p(i)=evals[i-1]
 * (EXT sumcheck::util; use in ceno_recursion_v2/src/main/mod.rs:3525-3526) */
static ext2 extrapolate(ext2 p0, const ext2* ev, int d, ext2 x) {
    ext2 acc = e2_zero();
    for (int i = 0; i <= d; i++) {
        ext2 yi = i == 0 ? p0 : ev[i - 1];
        ext2 num = e2_one();
        uint64_t den = 1;
        for (int j = 0; j <= d; j++) {
            if (j == i) continue;
            num = e2_mul(num, e2_sub(x, e2_from_u64((uint64_t)j)));
            uint64_t dij = i > j ? (uint64_t)(i - j) : gl_neg((uint64_t)(j - i));
            den = gl_mul(den, dij);
        }
        acc = e2_add(acc, e2_mul(yi, e2_mul_base(num, gl_inv(den))));
    }
    return acc;
}
void orc_extrapolate_uni_poly(const uint64_t* p0, const uint64_t* evals, int d, const uint64_t* x, uint64_t* out2) {
    ext2 ev[16];
    for (int i = 0; i < d; i++) ev[i] = ld2(evals + 2 * i);
    st2(out2, extrapolate(ld2(p0), ev, d, ld2(x)));
}

/* ------------------------------------------------------------------------------------------
 * succinct evaluators — gkr_iop/src/utils.rs
 * ---------------------------------------------------------------------------------------- */
/* eq_eval_less_or_equal_than, gkr_iop/src/utils.rs:166-207 */
static ext2 eq_le(uint64_t max_idx, const ext2* a, int na, const ext2* b, int nb) {
    ext2* rp = (ext2*)malloc(sizeof(ext2) * (nb + 1));
    ext2* rp2 = (ext2*)malloc(sizeof(ext2) * (nb + 1));
    ext2 one = e2_one();
    rp[0] = one;
    for (int i = 0; i < nb; i++)
        rp[i + 1] = e2_mul(rp[i], e2_add(e2_mul(a[i], b[i]), e2_mul(e2_sub(one, a[i]), e2_sub(one, b[i]))));
    rp2[nb] = one;
    for (int i = nb - 1; i >= 0; i--) {
        ext2 bit = e2_from_u64((max_idx >> i) & 1);
        ext2 t = e2_add(e2_mul(e2_mul(a[i], b[i]), bit),
                        e2_mul(e2_mul(e2_sub(one, a[i]), e2_sub(one, b[i])), e2_sub(one, bit)));
        rp2[i] = e2_mul(rp2[i + 1], t);
    }
    ext2 ans = rp[nb];
    for (int i = 0; i < nb; i++) {
        if ((max_idx >> i) & 1) continue;
        ans = e2_sub(ans, e2_mul(e2_mul(e2_mul(rp[i], rp2[i + 1]), a[i]), b[i]));
    }
    for (int i = nb; i < na; i++) ans = e2_mul(ans, e2_sub(one, a[i]));
    free(rp); free(rp2);
    return ans;
}
void orc_eq_eval_less_or_equal_than(uint64_t max_idx, const uint64_t* a, int na, const uint64_t* b, int nb, uint64_t* out2) {
    st2(out2, eq_le(max_idx, (const ext2*)a, na, (const ext2*)b, nb));
}
/* eval_wellform_address_vec, gkr_iop/src/utils.rs:215-232 */
static ext2 wellform(uint64_t offset, uint64_t scaled, const ext2* r, int n, int descending) {
    ext2 sum = e2_zero(), st = e2_one(), two = e2_from_u64(2);
    for (int i = 0; i < n; i++) { sum = e2_add(sum, e2_mul(r[i], st)); st = e2_mul(st, two); }
    ext2 tmp = e2_mul(e2_from_u64(scaled), sum);
    if (descending) tmp = e2_neg(tmp);
    return e2_add(e2_from_u64(offset), tmp);
}
void orc_eval_wellform_address_vec(uint64_t offset, uint64_t scaled, const uint64_t* r, int n, int descending, uint64_t* out2) {
    st2(out2, wellform(offset, scaled, (const ext2*)r, n, descending));
}
/* eval_stacked_wellform_address_vec, gkr_iop/src/utils.rs:256-266 */
void orc_eval_stacked_wellform_address_vec(const uint64_t* rw, int n, uint64_t* out2) {
    const ext2* r = (const ext2*)rw;
    ext2 res = e2_zero(), one = e2_one();
    for (int i = 1; i < n; i++) res = e2_add(e2_mul(res, e2_sub(one, r[i])), e2_mul(wellform(0, 1, r, i, 0), r[i]));
    st2(out2, res);
}
/* eval_stacked_constant_vec, gkr_iop/src/utils.rs:279-289 */
void orc_eval_stacked_constant_vec(const uint64_t* rw, int n, uint64_t* out2) {
    const ext2* r = (const ext2*)rw;
    ext2 res = e2_zero(), one = e2_one();
    for (int i = 1; i < n; i++) res = e2_add(e2_mul(res, e2_sub(one, r[i])), e2_mul(e2_from_u64((uint64_t)i), r[i]));
    st2(out2, res);
}

/* ------------------------------------------------------------------------------------------
 * selectors — gkr_iop/src/selector.rs:131-363
 * ---------------------------------------------------------------------------------------- */
int orc_selector_compute(int kind, const uint64_t* out_point, int num_vars, size_t offset, size_t num_instances,
                         const uint32_t* sparse_indices, int n_sparse, int sparse_num_vars, uint64_t* out) {
    size_t len = (size_t)1 << num_vars;
    orc_build_eq_x_r_vec(out_point, num_vars, out);
    ext2* sel = (ext2*)out;
    switch (kind) {
    case ORC_SEL_WHOLE: return 0; /* selector.rs:140 */
    case ORC_SEL_PREFIX: {        /* selector.rs:141-156 */
        size_t start = offset, end = offset + num_instances;
        if (end > len) return -1;
        for (size_t i = 0; i < start; i++) sel[i] = e2_zero();
        for (size_t i = end; i < len; i++) sel[i] = e2_zero();
        return 0;
    }
    case ORC_SEL_ORDERED_SPARSE: { /* selector.rs:158-189 */
        size_t chunk = (size_t)1 << sparse_num_vars;
        for (size_t c = 0; c < len / chunk; c++) {
            ext2* ch = sel + c * chunk;
            if (c >= num_instances) { for (size_t i = 0; i < chunk; i++) ch[i] = e2_zero(); continue; }
            int it = 0;
            for (size_t i = 0; i < chunk; i++) {
                if (it < n_sparse && sparse_indices[it] == i) it++;
                else ch[i] = e2_zero();
            }
        }
        return 0;
    }
    case ORC_SEL_QUARK_LT: { /* selector.rs:191-243 */
        if (offset != 0) return -1;
        size_t n_inst = num_instances;
        size_t start = 0, chunk_len = len / 2;
        int i = 0;
        while (chunk_len > 0) {
            size_t cur = 0;
            if (i < num_vars) { cur = n_inst / 2; n_inst = (n_inst + 1) / 2; }
            size_t zero_start = cur < chunk_len ? cur : chunk_len;
            for (size_t x = zero_start; x < chunk_len; x++) sel[start + x] = e2_zero();
            start += chunk_len; chunk_len /= 2; i++;
        }
        sel[len - 1] = e2_zero();
        return 0;
    }
    }
    return -2;
}

int orc_selector_evaluate(int kind, const uint64_t* out_point_w, const uint64_t* in_point_w, int num_vars, size_t offset,
                          size_t num_instances, const uint32_t* sparse_indices, int n_sparse, int sparse_num_vars,
                          uint64_t* out2) {
    const ext2* op = (const ext2*)out_point_w;
    const ext2* ip = (const ext2*)in_point_w;
    ext2 one = e2_one();
    switch (kind) {
    case ORC_SEL_WHOLE: orc_eq_eval(out_point_w, in_point_w, num_vars, out2); return 0; /* selector.rs:258-261 */
    case ORC_SEL_PREFIX: { /* selector.rs:262-287 */
        size_t start = offset, end = offset + num_instances;
        if (end == 0) { st2(out2, e2_zero()); return 0; }
        ext2 eq_end = eq_le(end - 1, op, num_vars, ip, num_vars);
        if (start > 0) eq_end = e2_sub(eq_end, eq_le(start - 1, op, num_vars, ip, num_vars));
        st2(out2, eq_end);
        return 0;
    }
    case ORC_SEL_ORDERED_SPARSE: { /* selector.rs:289-306 */
        size_t sl = (size_t)1 << sparse_num_vars;
        uint64_t* oe = (uint64_t*)malloc(16 * sl);
        uint64_t* ie = (uint64_t*)malloc(16 * sl);
        orc_build_eq_x_r_vec(out_point_w, sparse_num_vars, oe);
        orc_build_eq_x_r_vec(in_point_w, sparse_num_vars, ie);
        ext2 ev = e2_zero();
        for (int k = 0; k < n_sparse; k++) ev = e2_add(ev, e2_mul(ld2(oe + 2 * sparse_indices[k]), ld2(ie + 2 * sparse_indices[k])));
        free(oe); free(ie);
        ext2 s = eq_le(num_instances - 1, op + sparse_num_vars, num_vars - sparse_num_vars, ip + sparse_num_vars,
                       num_vars - sparse_num_vars);
        st2(out2, e2_mul(ev, s));
        return 0;
    }
    case ORC_SEL_QUARK_LT: { /* selector.rs:307-357 */
        if (num_instances == 0 || num_vars == 0) return -1;
        size_t* seq = (size_t*)malloc(sizeof(size_t) * num_vars);
        size_t n_inst = num_instances;
        for (int i = 0; i < num_vars; i++) { seq[i] = n_inst / 2; n_inst = (n_inst + 1) / 2; }
        /* reverse */
        for (int i = 0; i < num_vars / 2; i++) { size_t t = seq[i]; seq[i] = seq[num_vars - 1 - i]; seq[num_vars - 1 - i] = t; }
        ext2 res = seq[0] == 0 ? e2_zero() : e2_mul(e2_sub(one, op[0]), e2_sub(one, ip[0]));
        for (int i = 1; i < num_vars; i++) {
            ext2 lhs = e2_zero();
            if (seq[i] != 0)
                lhs = e2_mul(e2_mul(e2_sub(one, op[i]), e2_sub(one, ip[i])), eq_le(seq[i] - 1, op, i, ip, i));
            ext2 rhs = e2_mul(e2_mul(op[i], ip[i]), res);
            res = e2_add(lhs, rhs);
        }
        free(seq);
        st2(out2, res);
        return 0;
    }
    }
    return -2;
}

/* ------------------------------------------------------------------------------------------
 * generic sumcheck prover (a1)
 * ---------------------------------------------------------------------------------------- */
int orc_sumcheck_prove(const orc_mle* mles, int num_mles, const uint64_t* term_coeffs, const uint32_t* term_offsets,
                       const uint32_t* term_mle_idx, int num_terms, int n, int d, orc_transcript* tr,
                       uint64_t* out_msgs, uint64_t* out_challenges, uint64_t* out_final_evals) {
    if (d < 1 || d > 15) return -1;
    /* every term: non-empty, all factors share num_vars (gkr_iop/src/gkr/layer/gpu/utils.rs:54-63) */
    for (int t = 0; t < num_terms; t++) {
        uint32_t b = term_offsets[t], e = term_offsets[t + 1];
        if (e <= b || (int)(e - b) > d) return -2;
        for (uint32_t k = b; k < e; k++) {
            if ((int)term_mle_idx[k] >= num_mles) return -3;
            if (mles[term_mle_idx[k]].num_vars != mles[term_mle_idx[b]].num_vars) return -4;
            if (mles[term_mle_idx[k]].num_vars > n) return -5;
        }
    }
    ext2** tab = (ext2**)calloc(num_mles, sizeof(ext2*));
    size_t* len = (size_t*)calloc(num_mles, sizeof(size_t));
    ext2* tail = (ext2*)malloc(sizeof(ext2) * num_mles); /* prod of challenges bound after exhaustion */
    for (int j = 0; j < num_mles; j++) {
        len[j] = (size_t)1 << mles[j].num_vars;
        tab[j] = (ext2*)malloc(sizeof(ext2) * len[j]);
        for (size_t i = 0; i < len[j]; i++) tab[j][i] = ld_mle(mles[j].data, mles[j].is_ext, i);
        tail[j] = e2_one();
    }
    /* transcript prologue: ceno_recursion_v2/src/main/mod.rs:3503-3504 */
    tr_usize(tr, (uint64_t)n);
    tr_usize(tr, (uint64_t)d);
    for (int round = 0; round < n; round++) {
        ext2 msg[16];
        for (int t = 0; t < d; t++) msg[t] = e2_zero();
        for (int tm = 0; tm < num_terms; tm++) {
            uint32_t b = term_offsets[tm], e = term_offsets[tm + 1];
            ext2 c = ld2(term_coeffs + 2 * tm);
            int nv = mles[term_mle_idx[b]].num_vars;
            ext2 acc[16];
            for (int t = 0; t < d; t++) acc[t] = e2_zero();
            if (nv > round) {
                size_t pairs = len[term_mle_idx[b]] / 2;
                for (size_t p = 0; p < pairs; p++) {
                    ext2 prod[16];
                    for (int t = 0; t < d; t++) prod[t] = e2_one();
                    for (uint32_t k = b; k < e; k++) {
                        const ext2* f = tab[term_mle_idx[k]];
                        ext2 lo = f[2 * p], hi = f[2 * p + 1];
                        ext2 delta = e2_sub(hi, lo);
                        ext2 v = hi; /* f(1) */
                        for (int t = 0; t < d; t++) { prod[t] = e2_mul(prod[t], v); v = e2_add(v, delta); }
                    }
                    for (int t = 0; t < d; t++) acc[t] = e2_add(acc[t], prod[t]);
                }
            } else {
                /* front-loaded term: each factor is value * x_nv * ... (scheme/verifier.rs:233-237) */
                for (int t = 0; t < d; t++) {
                    ext2 prod = e2_one();
                    ext2 tt = e2_from_u64((uint64_t)(t + 1));
                    for (uint32_t k = b; k < e; k++) {
                        int j = term_mle_idx[k];
                        prod = e2_mul(prod, e2_mul(e2_mul(tab[j][0], tail[j]), tt));
                    }
                    acc[t] = prod;
                }
            }
            for (int t = 0; t < d; t++) msg[t] = e2_add(msg[t], e2_mul(c, acc[t]));
        }
        for (int t = 0; t < d; t++) { st2(out_msgs + 2 * ((size_t)round * d + t), msg[t]); tr_ext(tr, msg[t]); }
        tr_label(tr, "Internal round");
        ext2 r = tr_sample(tr);
        st2(out_challenges + 2 * round, r);
        for (int j = 0; j < num_mles; j++) {
            if (mles[j].num_vars > round) {
                size_t half = len[j] / 2;
                for (size_t p = 0; p < half; p++)
                    tab[j][p] = e2_add(tab[j][2 * p], e2_mul(r, e2_sub(tab[j][2 * p + 1], tab[j][2 * p])));
                len[j] = half;
            } else {
                tail[j] = e2_mul(tail[j], r);
            }
        }
    }
    for (int j = 0; j < num_mles; j++) { st2(out_final_evals + 2 * j, tab[j][0]); free(tab[j]); }
    free(tab); free(len); free(tail);
    return 0;
}

int orc_sumcheck_verify(const uint64_t* claimed_sum, const uint64_t* msgs, int n, int d, orc_transcript* tr,
                        uint64_t* out_point, uint64_t* out_expected2) {
    ext2 expected = ld2(claimed_sum);
    tr_usize(tr, (uint64_t)n);
    tr_usize(tr, (uint64_t)d);
    for (int round = 0; round < n; round++) {
        ext2 ev[16];
        for (int t = 0; t < d; t++) { ev[t] = ld2(msgs + 2 * ((size_t)round * d + t)); tr_ext(tr, ev[t]); }
        tr_label(tr, "Internal round");
        ext2 r = tr_sample(tr);
        ext2 p0 = e2_sub(expected, ev[0]);
        expected = extrapolate(p0, ev, d, r);
        st2(out_point + 2 * round, r);
    }
    st2(out_expected2, expected);
    return 0;
}

void orc_sumcheck_expected_from_evals(const int* mle_num_vars, int num_mles, const uint64_t* term_coeffs,
                                      const uint32_t* term_offsets, const uint32_t* term_mle_idx, int num_terms,
                                      int n, const uint64_t* point, const uint64_t* final_evals, uint64_t* out2) {
    (void)num_mles;
    ext2 acc = e2_zero();
    for (int t = 0; t < num_terms; t++) {
        ext2 v = ld2(term_coeffs + 2 * t);
        for (uint32_t k = term_offsets[t]; k < term_offsets[t + 1]; k++) {
            int j = term_mle_idx[k];
            v = e2_mul(v, ld2(final_evals + 2 * j));
            for (int i = mle_num_vars[j]; i < n; i++) v = e2_mul(v, ld2(point + 2 * i));
        }
        acc = e2_add(acc, v);
    }
    st2(out2, acc);
}

void orc_recover_claim_from_final(const uint64_t* final_claim, const uint64_t* msgs, const uint64_t* challenges, int n,
                                  int d, uint64_t* out2) {
    ext2 expected = ld2(final_claim);
    ext2 zeros[16];
    for (int i = 0; i < d; i++) zeros[i] = e2_zero();
    for (int round = n - 1; round >= 0; round--) {
        ext2 ev[16];
        for (int t = 0; t < d; t++) ev[t] = ld2(msgs + 2 * ((size_t)round * d + t));
        ext2 r = ld2(challenges + 2 * round);
        ext2 hidden = extrapolate(e2_one(), zeros, d, r);
        ext2 without = extrapolate(e2_neg(ev[0]), ev, d, r);
        expected = e2_mul(e2_sub(expected, without), e2_inv(hidden));
    }
    st2(out2, expected);
}

/* ------------------------------------------------------------------------------------------
 * multi-threaded fused dense sumcheck (cpu_baseline)
 *
 * Same schedule as the device path: round 0 accumulates over the input tables; round i>0
 * reads the table of round i-1, folds it with r_{i-1}, writes the half-size table and
 * accumulates the message of round i on the folded values in the same pass.
 * bufs holds 3*k pointers: [0,k) inputs (2^n ext, never written), [k,2k) ping buffers of
 * 2^(n-1) ext, [2k,3k) pong buffers of 2^(n-2) ext (may be NULL when n < 2).
 * ---------------------------------------------------------------------------------------- */
int orc_sumcheck_dense_mt(uint64_t** bufs, int k, int num_vars, const uint64_t* challenges, int threads,
                          uint64_t* out_msgs, uint64_t* out_final_evals) {
    if (k < 1 || k > 8 || num_vars < 1) return -1;
    const int d = k;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    const uint64_t* cur[8];
    uint64_t* nxt[8];
    for (int m = 0; m < k; m++) cur[m] = bufs[m];
    size_t cur_len = (size_t)1 << num_vars;
    int which = 0; /* next write target: 0 -> ping, 1 -> pong */
    for (int round = 0; round < num_vars; round++) {
        ext2 msg[8];
        for (int t = 0; t < d; t++) msg[t] = e2_zero();
        if (round == 0) {
            size_t pairs = cur_len / 2;
#pragma omp parallel
            {
                ext2 loc[8];
                for (int t = 0; t < d; t++) loc[t] = e2_zero();
#pragma omp for schedule(static)
                for (size_t p = 0; p < pairs; p++) {
                    ext2 prod[8];
                    for (int t = 0; t < d; t++) prod[t] = e2_one();
                    for (int m = 0; m < k; m++) {
                        ext2 lo = ld2(cur[m] + 4 * p), hi = ld2(cur[m] + 4 * p + 2);
                        ext2 delta = e2_sub(hi, lo), v = hi;
                        for (int t = 0; t < d; t++) { prod[t] = e2_mul(prod[t], v); v = e2_add(v, delta); }
                    }
                    for (int t = 0; t < d; t++) loc[t] = e2_add(loc[t], prod[t]);
                }
#pragma omp critical
                { for (int t = 0; t < d; t++) msg[t] = e2_add(msg[t], loc[t]); }
            }
        } else {
            ext2 r = ld2(challenges + 2 * (round - 1));
            size_t new_len = cur_len / 2, pairs = new_len / 2;
            for (int m = 0; m < k; m++) nxt[m] = bufs[(which ? 2 * k : k) + m];
#pragma omp parallel
            {
                ext2 loc[8];
                for (int t = 0; t < d; t++) loc[t] = e2_zero();
#pragma omp for schedule(static)
                for (size_t p = 0; p < pairs; p++) {
                    ext2 prod[8];
                    for (int t = 0; t < d; t++) prod[t] = e2_one();
                    for (int m = 0; m < k; m++) {
                        const uint64_t* src = cur[m] + 8 * p;
                        ext2 a0 = ld2(src), a1 = ld2(src + 2), a2 = ld2(src + 4), a3 = ld2(src + 6);
                        ext2 lo = e2_add(a0, e2_mul(r, e2_sub(a1, a0)));
                        ext2 hi = e2_add(a2, e2_mul(r, e2_sub(a3, a2)));
                        st2(nxt[m] + 4 * p, lo);
                        st2(nxt[m] + 4 * p + 2, hi);
                        ext2 delta = e2_sub(hi, lo), v = hi;
                        for (int t = 0; t < d; t++) { prod[t] = e2_mul(prod[t], v); v = e2_add(v, delta); }
                    }
                    for (int t = 0; t < d; t++) loc[t] = e2_add(loc[t], prod[t]);
                }
#pragma omp critical
                { for (int t = 0; t < d; t++) msg[t] = e2_add(msg[t], loc[t]); }
            }
            for (int m = 0; m < k; m++) cur[m] = nxt[m];
            cur_len = new_len;
            which ^= 1;
        }
        for (int t = 0; t < d; t++) st2(out_msgs + 2 * ((size_t)round * d + t), msg[t]);
    }
    /* cur_len == 2: bind the last variable */
    ext2 r = ld2(challenges + 2 * (num_vars - 1));
    for (int m = 0; m < k; m++) {
        ext2 a0 = ld2(cur[m]), a1 = ld2(cur[m] + 2);
        st2(out_final_evals + 2 * m, e2_add(a0, e2_mul(r, e2_sub(a1, a0))));
    }
    return 0;
}
