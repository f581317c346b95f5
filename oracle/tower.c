/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  Tower witness inference, tower
 * prover and tower verifier, restated from ceno_zkvm/src/scheme/{utils,cpu/mod,verifier}.rs.
 */
#include "oracle.h"
#include "gl64.h"
#include <stdlib.h>
#include <string.h>

static inline ext2 ld2(const uint64_t* p) { ext2 r = {{p[0], p[1]}}; return r; }
static inline void st2(uint64_t* p, ext2 v) { p[0] = v.c[0]; p[1] = v.c[1]; }
static inline ext2 ld_mle(const uint64_t* d, int is_ext, size_t i) {
    if (is_ext) return ld2(d + 2 * i);
    return e2_from_base(d[i]);
}
static int ceil_log2(size_t x) { int l = 0; while (((size_t)1 << l) < x) l++; return l; }
static size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }
/* witness::next_pow2_instance_padding: max(next_pow2(n), 2) (ceno_zkvm/src/scheme/hal.rs:127-128) */
static size_t next_pow2_instance_padding(size_t n) { size_t p = next_pow2(n); return p < 2 ? 2 : p; }

/* ------------------------------------------------------------------------------------------
 * wit_infer_by_monomial_expr (a5): out[x] = sum_t c_t prod_j f_j[x]
 * (gkr_iop/src/cpu/mod.rs:119-176; GPU call gkr_iop/src/gpu/mod.rs:599-609)
 * ---------------------------------------------------------------------------------------- */
int orc_wit_infer(const orc_mle* mles, int num_mles, const uint64_t* term_coeffs, const uint32_t* term_offsets,
                  const uint32_t* term_mle_idx, int num_terms, int num_vars, uint64_t* out) {
    size_t len = (size_t)1 << num_vars;
    for (int t = 0; t < num_terms; t++)
        for (uint32_t k = term_offsets[t]; k < term_offsets[t + 1]; k++) {
            if ((int)term_mle_idx[k] >= num_mles) return -1;
            if (mles[term_mle_idx[k]].num_vars != num_vars) return -2;
        }
#pragma omp parallel for schedule(static)
    for (size_t x = 0; x < len; x++) {
        ext2 acc = e2_zero();
        for (int t = 0; t < num_terms; t++) {
            ext2 v = ld2(term_coeffs + 2 * t);
            for (uint32_t k = term_offsets[t]; k < term_offsets[t + 1]; k++) {
                const orc_mle* m = &mles[term_mle_idx[k]];
                v = e2_mul(v, ld_mle(m->data, m->is_ext, x));
            }
            acc = e2_add(acc, v);
        }
        st2(out + 2 * x, acc);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * interleaving_mles_to_mles — ceno_zkvm/src/scheme/utils.rs:402-462
 * ---------------------------------------------------------------------------------------- */
size_t orc_interleave_out_len(int num_mles, size_t num_instances, int num_limbs) {
    int log2_num_instances = ceil_log2(next_pow2_instance_padding(num_instances));
    int log2_mle_size = ceil_log2((size_t)num_mles);
    int log2_num_limbs = ceil_log2((size_t)num_limbs);
    int sh = log2_num_instances - log2_num_limbs;
    if (sh < 0) sh = 0; /* saturating_sub */
    return (size_t)1 << (log2_mle_size + sh);
}
int orc_interleaving_mles_to_mles(const orc_mle* mles, int num_mles, size_t num_instances, int num_limbs,
                                  const uint64_t* default2, uint64_t** out_limbs) {
    if (num_mles <= 0 || (num_limbs & (num_limbs - 1))) return -1;
    size_t np2 = next_pow2_instance_padding(num_instances);
    for (int i = 0; i < num_mles; i++)
        if (((size_t)1 << mles[i].num_vars) > np2) return -2;
    size_t mle0_len = (size_t)1 << mles[0].num_vars;
    size_t per_fanin_len = mle0_len / (size_t)num_limbs;
    if (per_fanin_len < 1) per_fanin_len = 1;
    int log2_mle_size = ceil_log2((size_t)num_mles);
    size_t per_instance_size = (size_t)1 << log2_mle_size;
    size_t out_len = orc_interleave_out_len(num_mles, num_instances, num_limbs);
    ext2 dflt = ld2(default2);
    for (int limb = 0; limb < num_limbs; limb++) {
        ext2* ev = (ext2*)out_limbs[limb];
        for (size_t i = 0; i < out_len; i++) ev[i] = dflt;
        size_t start = per_fanin_len * (size_t)limb;
        if (start >= num_instances) continue;
        size_t valid = num_instances - start;
        if (valid > per_fanin_len) valid = per_fanin_len;
        size_t n_chunks = out_len / per_instance_size;
        for (int i = 0; i < num_mles; i++) {
            size_t len_i = (size_t)1 << mles[i].num_vars;
            /* Ext arm: slice start..start+valid ; Base arm: start..start+per_fanin_len (utils.rs:436-455);
             * `.get(range).unwrap_or(&[])` yields nothing when the range exceeds the vector */
            size_t cnt = mles[i].is_ext ? valid : per_fanin_len;
            if (start + cnt > len_i) continue;
            if (cnt > n_chunks) cnt = n_chunks; /* zip with chunks */
            for (size_t r = 0; r < cnt; r++) ev[r * per_instance_size + (size_t)i] = ld_mle(mles[i].data, mles[i].is_ext, start + r);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * infer_tower_product_witness — ceno_zkvm/src/scheme/utils.rs:588-659
 * layers[2*l + s] (l = 0..num_vars-1) receives 2^l ext elements; the last layer is copied in.
 * ---------------------------------------------------------------------------------------- */
int orc_infer_tower_product_witness(int num_vars, const uint64_t* last0, const uint64_t* last1, uint64_t** layers) {
    if (num_vars < 1) return -1;
    size_t len = (size_t)1 << (num_vars - 1);
    memcpy(layers[2 * (num_vars - 1)], last0, 16 * len);
    memcpy(layers[2 * (num_vars - 1) + 1], last1, 16 * len);
    for (int l = num_vars - 2; l >= 0; l--) {
        const ext2* f1 = (const ext2*)layers[2 * (l + 1)];
        const ext2* f2 = (const ext2*)layers[2 * (l + 1) + 1];
        size_t out_len = ((size_t)1 << (l + 1)) / 2;
        for (int index = 0; index < 2; index++) {
            ext2* o = (ext2*)layers[2 * l + index];
            size_t start = (size_t)index * out_len;
            for (size_t j = 0; j < out_len; j++) o[j] = e2_mul(f1[start + j], f2[start + j]);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * infer_tower_logup_witness — ceno_zkvm/src/scheme/utils.rs:488-582 (+ tower_mle_4 :464-479)
 * limb_num_vars = log2 of q limb length; produces limb_num_vars+1 layers, layer l limbs have
 * 2^l elements: layers[4*l + {0:p1, 1:p2, 2:q1, 3:q2}].  p0/p1 NULL => numerators all ONE at
 * the input layer and (q1+q2) one layer up.
 * ---------------------------------------------------------------------------------------- */
int orc_infer_tower_logup_witness(int nv, const uint64_t* p0, const uint64_t* p1, const uint64_t* q0,
                                  const uint64_t* q1, uint64_t** layers) {
    size_t len = (size_t)1 << nv;
    ext2* L = (ext2*)layers[4 * nv + 0];
    ext2* M = (ext2*)layers[4 * nv + 1];
    if (p0 && p1) { memcpy(L, p0, 16 * len); memcpy(M, p1, 16 * len); }
    else { for (size_t i = 0; i < len; i++) { L[i] = e2_one(); M[i] = e2_one(); } }
    memcpy(layers[4 * nv + 2], q0, 16 * len);
    memcpy(layers[4 * nv + 3], q1, 16 * len);
    int have_p = (p0 && p1);
    for (int l = nv - 1; l >= 0; l--) {
        const ext2* P1 = (const ext2*)layers[4 * (l + 1) + 0];
        const ext2* P2 = (const ext2*)layers[4 * (l + 1) + 1];
        const ext2* Q1 = (const ext2*)layers[4 * (l + 1) + 2];
        const ext2* Q2 = (const ext2*)layers[4 * (l + 1) + 3];
        size_t cur_len = ((size_t)1 << (l + 1)) / 2;
        for (int index = 0; index < 2; index++) {
            size_t start = cur_len * (size_t)index;
            ext2* po = (ext2*)layers[4 * l + index];
            ext2* qo = (ext2*)layers[4 * l + 2 + index];
            for (size_t j = 0; j < cur_len; j++) {
                ext2 a = Q1[start + j], b = Q2[start + j];
                if (have_p || l < nv - 1) /* after the first fold p always exists */
                    po[j] = e2_add(e2_mul(a, P2[start + j]), e2_mul(b, P1[start + j]));
                else
                    po[j] = e2_add(a, b);
                qo[j] = e2_mul(a, b);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * tower prover — CpuTowerProver::create_proof, ceno_zkvm/src/scheme/cpu/mod.rs:346-554
 * ---------------------------------------------------------------------------------------- */
static void tr_label(orc_transcript* t, const char* s) { t->append_label(t->self, (const uint8_t*)s, strlen(s)); }
static void tr_ext(orc_transcript* t, ext2 e) { t->append_ext(t->self, e.c); }
static ext2 tr_sample(orc_transcript* t) { ext2 r; t->sample_ext(t->self, r.c); return r; }
static void tr_challenge_pows(orc_transcript* t, int n, ext2* out) {
    tr_label(t, "combine subset evals");
    ext2 a = tr_sample(t);
    ext2 acc = e2_one();
    for (int i = 0; i < n; i++) { out[i] = acc; acc = e2_mul(acc, a); }
}

size_t orc_tower_msgs_words(int max_nv) {
    /* round r (1..max_nv-1) contributes r rounds x 3 evals */
    size_t tot = 0;
    for (int r = 1; r < max_nv; r++) tot += (size_t)r * 3 * 2;
    return tot;
}

int orc_tower_prove(const orc_tower_spec* prod, int n_prod, const orc_tower_spec* logup, int n_logup,
                    orc_transcript* tr, orc_tower_proof* out) {
    int max_nv = 0;
    for (int i = 0; i < n_prod; i++) if (prod[i].num_vars > max_nv) max_nv = prod[i].num_vars;
    for (int i = 0; i < n_logup; i++) if (logup[i].num_vars > max_nv) max_nv = logup[i].num_vars;
    if (max_nv < 1) return -1;
    int n_alpha = n_prod + 2 * n_logup;
    ext2* alpha = (ext2*)malloc(sizeof(ext2) * (n_alpha ? n_alpha : 1));
    tr_challenge_pows(tr, n_alpha, alpha);                /* cpu/mod.rs:375-380 */
    tr_label(tr, "product_sum");                          /* cpu/mod.rs:381 */
    ext2* out_rt = (ext2*)malloc(sizeof(ext2) * (max_nv + 1));
    out_rt[0] = tr_sample(tr);
    int rt_len = 1;
    out->num_rounds = max_nv - 1;
    size_t msg_off = 0;
    int max_mles = 1 + 2 * n_prod + 4 * n_logup;
    orc_mle* mles = (orc_mle*)malloc(sizeof(orc_mle) * max_mles);
    uint64_t* coeffs = (uint64_t*)malloc(16 * (n_prod + 3 * n_logup + 1));
    uint32_t* toff = (uint32_t*)malloc(4 * (n_prod + 3 * n_logup + 2));
    uint32_t* tidx = (uint32_t*)malloc(4 * 3 * (n_prod + 3 * n_logup + 1));
    int R = out->num_rounds;
    for (int round = 1; round <= R; round++) {            /* cpu/mod.rs:409 (skip(1)) */
        size_t len = (size_t)1 << round;
        uint64_t* eq = (uint64_t*)malloc(16 * len);
        orc_build_eq_x_r_vec((const uint64_t*)out_rt, rt_len, eq);   /* cpu/mod.rs:417 */
        int nm = 0, nt = 0, ni = 0;
        mles[nm].data = eq; mles[nm].is_ext = 1; mles[nm].num_vars = round; nm++;
        int* prod_idx = (int*)malloc(sizeof(int) * (n_prod + 1));
        int* logup_idx = (int*)malloc(sizeof(int) * (n_logup + 1));
        toff[0] = 0;
        for (int i = 0; i < n_prod; i++) {
            prod_idx[i] = -1;
            if (prod[i].num_vars <= round) continue;       /* spec has no layer `round` */
            prod_idx[i] = nm;
            for (int s = 0; s < 2; s++) { mles[nm].data = prod[i].layers[2 * round + s]; mles[nm].is_ext = 1; mles[nm].num_vars = round; nm++; }
            st2(coeffs + 2 * nt, alpha[i]);
            tidx[ni++] = 0; tidx[ni++] = prod_idx[i]; tidx[ni++] = prod_idx[i] + 1;
            toff[++nt] = ni;                               /* eq * alpha^i * a * b, cpu/mod.rs:442 */
        }
        for (int i = 0; i < n_logup; i++) {
            logup_idx[i] = -1;
            if (logup[i].num_vars <= round) continue;
            logup_idx[i] = nm;
            for (int s = 0; s < 4; s++) { mles[nm].data = logup[i].layers[4 * round + s]; mles[nm].is_ext = 1; mles[nm].num_vars = round; nm++; }
            int p1 = logup_idx[i], p2 = p1 + 1, q1 = p1 + 2, q2 = p1 + 3;
            ext2 an = alpha[n_prod + 2 * i], ad = alpha[n_prod + 2 * i + 1];
            /* eq * (an * (p1*q2 + p2*q1) + ad * q1*q2), cpu/mod.rs:480-484 */
            st2(coeffs + 2 * nt, an); tidx[ni++] = 0; tidx[ni++] = p1; tidx[ni++] = q2; toff[++nt] = ni;
            st2(coeffs + 2 * nt, an); tidx[ni++] = 0; tidx[ni++] = p2; tidx[ni++] = q1; toff[++nt] = ni;
            st2(coeffs + 2 * nt, ad); tidx[ni++] = 0; tidx[ni++] = q1; tidx[ni++] = q2; toff[++nt] = ni;
        }
        uint64_t* chal = (uint64_t*)malloc(16 * round);
        uint64_t* fin = (uint64_t*)malloc(16 * nm);
        int rc = orc_sumcheck_prove(mles, nm, coeffs, toff, tidx, nt, round, 3, tr, out->msgs + msg_off, chal, fin);
        if (rc) return rc;
        msg_off += (size_t)round * 3 * 2;
        /* evals appended to the transcript per active spec, prod first — cpu/mod.rs:498-531 */
        for (int i = 0; i < n_prod; i++) {
            uint64_t* dst = out->prod_evals + 2 * ((size_t)(i * R + (round - 1)) * 2);
            if (prod_idx[i] < 0) { memset(dst, 0, 32); continue; }
            for (int s = 0; s < 2; s++) { ext2 e = ld2(fin + 2 * (prod_idx[i] + s)); st2(dst + 2 * s, e); tr_ext(tr, e); }
        }
        for (int i = 0; i < n_logup; i++) {
            uint64_t* dst = out->logup_evals + 2 * ((size_t)(i * R + (round - 1)) * 4);
            if (logup_idx[i] < 0) { memset(dst, 0, 64); continue; }
            for (int s = 0; s < 4; s++) { ext2 e = ld2(fin + 2 * (logup_idx[i] + s)); st2(dst + 2 * s, e); tr_ext(tr, e); }
        }
        tr_label(tr, "merge");                             /* cpu/mod.rs:534 */
        ext2 r_merge = tr_sample(tr);
        for (int k = 0; k < round; k++) out_rt[k] = ld2(chal + 2 * k);   /* rt' = challenges || r_merge, :535 */
        out_rt[round] = r_merge;
        rt_len = round + 1;
        tr_challenge_pows(tr, n_alpha, alpha);             /* cpu/mod.rs:538-541 */
        free(eq); free(chal); free(fin); free(prod_idx); free(logup_idx);
    }
    for (int k = 0; k < rt_len; k++) st2(out->point + 2 * k, out_rt[k]);
    free(alpha); free(out_rt); free(mles); free(coeffs); free(toff); free(tidx);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * tower verifier — TowerVerify::verify, ceno_zkvm/src/scheme/verifier.rs:1372-1709
 * ---------------------------------------------------------------------------------------- */
static ext2 eval2(ext2 a, ext2 b, ext2 r) { return e2_add(a, e2_mul(r, e2_sub(b, a))); } /* vec![a,b].into_mle().evaluate(&[r]) */

int orc_tower_verify(const uint64_t* prod_out_evals, const uint64_t* logup_out_evals, const int* num_variables,
                     int n_prod, int n_logup, const orc_tower_proof* proof, orc_transcript* tr, uint64_t* out_point,
                     uint64_t* out_prod_claims, uint64_t* out_logup_p_claims, uint64_t* out_logup_q_claims) {
    int n_alpha = n_prod + 2 * n_logup;
    int max_nv = 0;
    for (int i = 0; i < n_prod + n_logup; i++) if (num_variables[i] > max_nv) max_nv = num_variables[i];
    if (max_nv < 1) return -1;
    int R = proof->num_rounds;
    if (R < max_nv - 1) return -2;
    ext2* alpha = (ext2*)malloc(sizeof(ext2) * (n_alpha ? n_alpha : 1));
    ext2* nalpha = (ext2*)malloc(sizeof(ext2) * (n_alpha ? n_alpha : 1));
    tr_challenge_pows(tr, n_alpha, alpha);                 /* verifier.rs:1429-1432 */
    tr_label(tr, "product_sum");
    ext2* rt = (ext2*)malloc(sizeof(ext2) * (max_nv + 1));
    rt[0] = tr_sample(tr);
    int rt_len = 1;
    ext2* pc = (ext2*)malloc(sizeof(ext2) * (n_prod + 1));
    ext2* lp = (ext2*)malloc(sizeof(ext2) * (n_logup + 1));
    ext2* lq = (ext2*)malloc(sizeof(ext2) * (n_logup + 1));
    ext2 claim = e2_zero();
    for (int i = 0; i < n_prod; i++) {                      /* verifier.rs:1441-1474 */
        pc[i] = eval2(ld2(prod_out_evals + 4 * i), ld2(prod_out_evals + 4 * i + 2), rt[0]);
        claim = e2_add(claim, e2_mul(pc[i], alpha[i]));
    }
    for (int i = 0; i < n_logup; i++) {
        const uint64_t* e = logup_out_evals + 8 * i;
        lp[i] = eval2(ld2(e), ld2(e + 2), rt[0]);
        lq[i] = eval2(ld2(e + 4), ld2(e + 6), rt[0]);
        claim = e2_add(claim, e2_add(e2_mul(lp[i], alpha[n_prod + 2 * i]), e2_mul(lq[i], alpha[n_prod + 2 * i + 1])));
    }
    size_t msg_off = 0;
    int rc = 0;
    uint64_t* pt = (uint64_t*)malloc(16 * (max_nv + 1));
    for (int round = 0; round < max_nv - 1 && rc == 0; round++) {
        int nvars = round + 1;
        uint64_t exp2[2];
        orc_sumcheck_verify(claim.c, proof->msgs + msg_off, nvars, 3, tr, pt, exp2);  /* verifier.rs:1555-1566 */
        msg_off += (size_t)nvars * 3 * 2;
        uint64_t eqv[2];
        orc_eq_eval((const uint64_t*)rt, pt, nvars, eqv);      /* verifier.rs:1570 */
        /* bind_active_tower_eval_round: append evals of active specs (verifier.rs:1572-1578) */
        for (int i = 0; i < n_prod; i++)
            if (round < num_variables[i] - 1)
                for (int s = 0; s < 2; s++) tr_ext(tr, ld2(proof->prod_evals + 2 * ((size_t)(i * R + round) * 2 + s)));
        for (int i = 0; i < n_logup; i++)
            if (round < num_variables[n_prod + i] - 1)
                for (int s = 0; s < 4; s++) tr_ext(tr, ld2(proof->logup_evals + 2 * ((size_t)(i * R + round) * 4 + s)));
        ext2 fold = e2_zero();
        for (int i = 0; i < n_prod; i++) {
            if (round < num_variables[i] - 1) {
                const uint64_t* e = proof->prod_evals + 2 * ((size_t)(i * R + round) * 2);
                fold = e2_add(fold, e2_mul(alpha[i], e2_mul(ld2(e), ld2(e + 2))));
            }
        }
        for (int i = 0; i < n_logup; i++) {
            if (round < num_variables[n_prod + i] - 1) {
                const uint64_t* e = proof->logup_evals + 2 * ((size_t)(i * R + round) * 4);
                ext2 p1 = ld2(e), p2 = ld2(e + 2), q1 = ld2(e + 4), q2 = ld2(e + 6);
                fold = e2_add(fold, e2_add(e2_mul(alpha[n_prod + 2 * i], e2_add(e2_mul(p1, q2), e2_mul(p2, q1))),
                                           e2_mul(alpha[n_prod + 2 * i + 1], e2_mul(q1, q2))));
            }
        }
        ext2 expected = e2_mul(ld2(eqv), fold);
        if (!e2_eq(expected, ld2(exp2))) { rc = -10 - round; break; }   /* "mismatch tower evaluation" */
        tr_label(tr, "merge");
        ext2 r_merge = tr_sample(tr);
        for (int k = 0; k < nvars; k++) rt[k] = ld2(pt + 2 * k);
        rt[nvars] = r_merge;
        rt_len = nvars + 1;
        tr_challenge_pows(tr, n_alpha, nalpha);
        int next_round = round + 1;
        ext2 next = e2_zero();
        for (int i = 0; i < n_prod; i++) {
            int mr = num_variables[i];
            if (round < mr - 1) {
                const uint64_t* e = proof->prod_evals + 2 * ((size_t)(i * R + round) * 2);
                ext2 ev = eval2(ld2(e), ld2(e + 2), r_merge);   /* sum_b eq(r_merge,b) e_b */
                pc[i] = ev;
                if (next_round < mr - 1) next = e2_add(next, e2_mul(nalpha[i], ev));
            }
        }
        for (int i = 0; i < n_logup; i++) {
            int mr = num_variables[n_prod + i];
            if (round < mr - 1) {
                const uint64_t* e = proof->logup_evals + 2 * ((size_t)(i * R + round) * 4);
                ext2 pe = eval2(ld2(e), ld2(e + 2), r_merge);
                ext2 qe = eval2(ld2(e + 4), ld2(e + 6), r_merge);
                lp[i] = pe; lq[i] = qe;
                if (next_round < mr - 1)
                    next = e2_add(next, e2_add(e2_mul(nalpha[n_prod + 2 * i], pe), e2_mul(nalpha[n_prod + 2 * i + 1], qe)));
            }
        }
        claim = next;
        memcpy(alpha, nalpha, sizeof(ext2) * (n_alpha ? n_alpha : 1));
    }
    if (rc == 0) {
        for (int k = 0; k < rt_len; k++) st2(out_point + 2 * k, rt[k]);
        for (int i = 0; i < n_prod; i++) st2(out_prod_claims + 2 * i, pc[i]);
        for (int i = 0; i < n_logup; i++) { st2(out_logup_p_claims + 2 * i, lp[i]); st2(out_logup_q_claims + 2 * i, lq[i]); }
    }
    free(alpha); free(nalpha); free(rt); free(pc); free(lp); free(lq); free(pt);
    return rc;
}
