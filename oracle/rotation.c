/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  Rotation argument of the keccak-style chips (a11):
 * BooleanHypercube cyclic tables (gkr_iop/src/gkr/booleanhypercube.rs:10-113: x^i in GF(2)[X]/(X^5+X^2+1)
 * resp. /(X^6+X+1), 2^k entries, the last one wrapping to 1), rotation_next_base_mle / rotation_selector
 * (gkr_iop/src/utils.rs:19-76), get_rotation_points / get_rotation_right_eval_from_left
 * (booleanhypercube.rs:117-193) and prove_rotation (gkr_iop/src/gkr/layer/cpu/mod.rs:249-389).
 */
#include "oracle.h"
#include "gl64.h"
#include <stdlib.h>
#include <string.h>

static inline ext2 ld2(const uint64_t* p) { ext2 r = {{p[0], p[1]}}; return r; }
static inline void st2(uint64_t* p, ext2 v) { p[0] = v.c[0]; p[1] = v.c[1]; }

/* out[i] = x^i for i in [0, 2^log2) */
int orc_cyclic_table(int log2, uint32_t* out) {
    uint32_t modulus;
    if (log2 == 5) modulus = 0x25;       /* X^5 + X^2 + 1 */
    else if (log2 == 6) modulus = 0x43;  /* X^6 + X + 1 */
    else return -1;
    uint32_t cur = 1;
    for (int i = 0; i < (1 << log2); i++) {
        out[i] = cur;
        cur <<= 1;
        if (cur & (1u << log2)) cur ^= modulus;
    }
    return 0;
}

/* rotation_next_base_mle, gkr_iop/src/utils.rs:19-52 (base-field tables) */
int orc_rotation_next_base_mle(const uint64_t* in, int num_vars, int log2, uint64_t* out) {
    uint32_t r[64];
    if (orc_cyclic_table(log2, r)) return -1;
    size_t g = (size_t)1 << log2, len = (size_t)1 << num_vars;
    memset(out, 0, 8 * len);
    for (size_t c = 0; c < len / g; c++) {
        const uint64_t* o = in + c * g;
        uint64_t* ro = out + c * g;
        uint32_t first = r[0], last = r[g - 1];
        if (first == last) ro[last] = o[first];
        ro[0] = o[0];
        for (int i = (int)g - 2; i >= 0; i--) ro[r[i]] = o[r[i + 1]];
    }
    return 0;
}

/* rotation_selector, gkr_iop/src/utils.rs:54-76 */
int orc_rotation_selector(const uint64_t* eq, int num_vars, int subgroup_size, int log2, uint64_t* out) {
    uint32_t r[64];
    if (orc_cyclic_table(log2, r)) return -1;
    size_t g = (size_t)1 << log2, len = (size_t)1 << num_vars;
    if ((size_t)subgroup_size > g) return -2;
    memset(out, 0, 16 * len);
    for (size_t c = 0; c < len / g; c++)
        for (int i = subgroup_size - 1; i >= 0; i--) memcpy(out + 2 * (c * g + r[i]), eq + 2 * (c * g + r[i]), 16);
    return 0;
}

/* get_rotation_points, booleanhypercube.rs:117-168 */
int orc_rotation_points(const uint64_t* point, int n, int log2, uint64_t* left, uint64_t* right) {
    ext2 one = e2_one(), zero = e2_zero();
    if (n < log2 + 0 || (log2 != 5 && log2 != 6)) return -1;
    const ext2* p = (const ext2*)point;
    ext2* l = (ext2*)left;
    ext2* r = (ext2*)right;
    /* left: (0, p[0..log2-1), p[log2..]) truncated to n */
    l[0] = zero;
    for (int i = 1; i < n; i++) l[i] = (i <= log2 - 1) ? p[i - 1] : p[i];
    r[0] = one;
    if (log2 == 5) {
        /* (1, r0, 1-r1, r2, r3, r5, ...) */
        for (int i = 1; i < n; i++) r[i] = (i <= 4) ? p[i - 1] : p[i];
        if (n > 2) r[2] = e2_sub(one, p[1]);
    } else {
        /* (1, 1-r0, r1, r2, r3, r4, r6, ...) */
        for (int i = 1; i < n; i++) r[i] = (i <= 5) ? p[i - 1] : p[i];
        if (n > 1) r[1] = e2_sub(one, p[0]);
    }
    return 0;
}

/* prove_rotation: pairs (source_j, target_j) of base-field witness tables; returns the sumcheck messages
 * (n rounds x 2 ext), evals (left, right, target per pair) and the three points. */
int orc_prove_rotation(const orc_mle* wit, int n_wit, const int* src, const int* tgt, int n_pairs, int subgroup_size, int log2,
                       const uint64_t* rt, int n, orc_transcript* tr, uint64_t* out_msgs, uint64_t* out_evals, uint64_t* out_origin,
                       uint64_t* out_left, uint64_t* out_right) {
    (void)n_wit;
    size_t len = (size_t)1 << n;
    uint64_t* eq = (uint64_t*)malloc(16 * len);
    uint64_t* sel = (uint64_t*)malloc(16 * len);
    orc_build_eq_x_r_vec(rt, n, eq);
    if (orc_rotation_selector(eq, n, subgroup_size, log2, sel)) return -1;
    /* get_challenge_pows(n_pairs) */
    static const char lbl[] = "combine subset evals";
    tr->append_label(tr->self, (const uint8_t*)lbl, sizeof(lbl) - 1);
    ext2 alpha; tr->sample_ext(tr->self, alpha.c);
    int n_mles = 2 * n_pairs + 1;
    orc_mle* mles = (orc_mle*)malloc(sizeof(orc_mle) * n_mles);
    uint64_t** rot = (uint64_t**)malloc(sizeof(uint64_t*) * n_pairs);
    uint64_t* coeffs = (uint64_t*)malloc(16 * 2 * n_pairs);
    uint32_t* toff = (uint32_t*)malloc(4 * (2 * n_pairs + 1));
    uint32_t* tidx = (uint32_t*)malloc(4 * 4 * n_pairs);
    ext2 a = e2_one();
    toff[0] = 0;
    for (int j = 0; j < n_pairs; j++) {
        rot[j] = (uint64_t*)malloc(8 * len);
        if (wit[src[j]].is_ext || wit[src[j]].num_vars != n) return -2;
        orc_rotation_next_base_mle(wit[src[j]].data, n, log2, rot[j]);
        mles[2 * j].data = rot[j]; mles[2 * j].is_ext = 0; mles[2 * j].num_vars = n;
        mles[2 * j + 1] = wit[tgt[j]];
        /* sel * alpha^j * (rot_j - tgt_j) */
        st2(coeffs + 2 * (2 * j), a);
        st2(coeffs + 2 * (2 * j + 1), e2_neg(a));
        tidx[4 * j] = 2 * n_pairs; tidx[4 * j + 1] = 2 * j; toff[2 * j + 1] = 4 * j + 2;
        tidx[4 * j + 2] = 2 * n_pairs; tidx[4 * j + 3] = 2 * j + 1; toff[2 * j + 2] = 4 * j + 4;
        a = e2_mul(a, alpha);
    }
    mles[2 * n_pairs].data = sel; mles[2 * n_pairs].is_ext = 1; mles[2 * n_pairs].num_vars = n;
    uint64_t* fin = (uint64_t*)malloc(16 * n_mles);
    int rc = orc_sumcheck_prove(mles, n_mles, coeffs, toff, tidx, 2 * n_pairs, n, 2, tr, out_msgs, out_origin, fin);
    if (rc) return rc;
    orc_rotation_points(out_origin, n, log2, out_left, out_right);
    ext2 rk = ld2(out_origin + 2 * (log2 - 1)); /* point[4] resp. point[5] */
    ext2 rk_inv = e2_inv(rk);
    for (int j = 0; j < n_pairs; j++) {
        uint64_t le[2];
        orc_mle_evaluate(wit[src[j]].data, 0, n, out_left, le);
        ext2 left = ld2(le), rotated = ld2(fin + 2 * (2 * j)), target = ld2(fin + 2 * (2 * j + 1));
        /* right = (rotated - (1 - r_k) * left) / r_k   (booleanhypercube.rs:170-186) */
        ext2 right = e2_mul(e2_sub(rotated, e2_mul(e2_sub(e2_one(), rk), left)), rk_inv);
        st2(out_evals + 2 * (3 * j), left);
        st2(out_evals + 2 * (3 * j + 1), right);
        st2(out_evals + 2 * (3 * j + 2), target);
    }
    for (int j = 0; j < 3 * n_pairs; j++) tr->append_ext(tr->self, out_evals + 2 * j);
    for (int j = 0; j < n_pairs; j++) free(rot[j]);
    free(rot); free(eq); free(sel); free(mles); free(coeffs); free(toff); free(tidx); free(fin);
    return 0;
}
