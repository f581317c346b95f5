/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  Basefold batch OPEN and its verifier.
 *
 * PARITY UNPINNED.  The reference's implementation is `mpcs::Basefold::batch_open` in an EXT crate
 * (scroll-tech/gkr-backend v1.0.0-alpha.35, reference Cargo.toml:30-40; call site
 * ceno_zkvm/src/scheme/cpu/mod.rs:1418-1457).  The only in-tree statement of the protocol is the verifier
 * replay of the recursion circuit, which this file follows step by step:
 *   ceno_recursion_v2/src/pcs/mod.rs:1111-1316  replay_basefold: transcript script, initial claim
 *                                               sum coeff * eval * 2^(max_nv - nv), degree-2 rounds [p(1), p(2)]
 *   ceno_recursion_v2/src/pcs/mod.rs:444-580    final claim = sum_g eq(point_g, last nv_g challenges) * final_message[g]
 *   ceno_recursion_v2/src/pcs/mod.rs:7494-7720  query phase: input openings reduced per height with the batch
 *                                               coefficients, joined to the running fold when the height matches,
 *                                               sibling + Merkle path per commit round, final constant codeword
 *   ceno_recursion_v2/src/pcs/mod.rs:7765-7781  fold: lo=(a+b)/2, hi=(a-b) g^-bitrev(i) /2, lo + r (hi - lo)
 * with basecode_msg_size_log = 0 (the only shape that verifier supports, pcs/mod.rs:8330-8338).
 *   ceno_recursion_v2/src/pcs/mod.rs:8125-8155  proof of work: check_witness = observe(witness), sample_bits(bits) == 0
 *   ceno_recursion_v2/src/pcs/mod.rs:1253-1266  query index = sample_bits(max_num_var + rate_log), ONE base sample each
 *   ceno_recursion_v2/src/pcs/mod.rs:7547-7565  one input opening per COMMITMENT (`rounds`): reduced_index = query >>
 *                                               (log2_max_codeword_size - commit.log2_max_codeword_size), opened rows of
 *                                               all its matrices + one MMCS path (commit.c: orc_mmcs_*)
 * Unpinned remainder: the Goldilocks field/hash instances (commit.c: Poseidon2 width 8 constants) and
 * the byte packing of labels.  A digest is observed as its four base elements (written here as two ext appends: the same
 * absorb sequence).  p3's grinding takes ANY valid witness (`find_any`, parallel); this restatement takes the least.
 */
#include "oracle.h"
#include "gl64.h"
#include <stdlib.h>
#include <string.h>

static inline ext2 ld2(const uint64_t* p) { ext2 r = {{p[0], p[1]}}; return r; }
static inline void st2(uint64_t* p, ext2 v) { p[0] = v.c[0]; p[1] = v.c[1]; }
static void tr_label(orc_transcript* t, const char* s) { t->append_label(t->self, (const uint8_t*)s, strlen(s)); }
static void tr_ext(orc_transcript* t, ext2 e) { t->append_ext(t->self, e.c); }
static ext2 tr_sample(orc_transcript* t) { ext2 r; t->sample_ext(t->self, r.c); return r; }
static void tr_digest(orc_transcript* t, const uint64_t* d4) { t->append_ext(t->self, d4); t->append_ext(t->self, d4 + 2); }
static ext2 e2_scale(ext2 a, uint64_t b) { ext2 r = {{gl_mul(a.c[0], b), gl_mul(a.c[1], b)}}; return r; }

static unsigned bitrev_u(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* radix-2 decimation-in-frequency transform, natural in -> bit-reversed out (same map as orc_dft_bitrev) */
void orc_fft_bitrev(uint64_t* a, int log_n) {
    size_t n = (size_t)1 << log_n;
    for (int s = log_n; s >= 1; s--) {
        size_t half = (size_t)1 << (s - 1);
        uint64_t w = orc_two_adic_generator(s);
        for (size_t base = 0; base < n; base += 2 * half) {
            uint64_t x = 1;
            for (size_t j = 0; j < half; j++) {
                uint64_t u = a[base + j], v = a[base + j + half];
                a[base + j] = gl_add(u, v);
                a[base + j + half] = gl_mul(gl_sub(u, v), x);
                x = gl_mul(x, w);
            }
        }
    }
}

/* Reed-Solomon codeword of one column: evaluations-as-coefficients, zero extended, bit-reversed order */
static void rs_encode_col(const uint64_t* col, int nv, int rate_log, uint64_t* out) {
    size_t n = (size_t)1 << nv, N = (size_t)1 << (nv + rate_log);
    memcpy(out, col, n * 8);
    memset(out + n, 0, (N - n) * 8);
    orc_fft_bitrev(out, nv + rate_log);
}

static int total_mats(int n_commits, const int* commit_sizes) {
    int t = 0;
    for (int c = 0; c < n_commits; c++) t += commit_sizes[c];
    return t;
}
size_t orc_basefold_query_words(int n_commits, const int* commit_sizes, const int* nv, const int* width, int rate_log) {
    const int n_mats = total_mats(n_commits, commit_sizes);
    int n = 0;
    for (int m = 0; m < n_mats; m++) if (nv[m] > n) n = nv[m];
    size_t w = 1;
    for (int c = 0, m0 = 0; c < n_commits; m0 += commit_sizes[c], c++) {
        int hc = 0;
        for (int m = m0; m < m0 + commit_sizes[c]; m++) {
            w += (size_t)width[m];
            if (nv[m] > hc) hc = nv[m];
        }
        w += 4 * (size_t)(hc + rate_log);
    }
    for (int r = 0; r < n; r++) w += 2 + 4 * (size_t)(n + rate_log - r - 1);
    return w;
}
/* [msgs 4n][commits 4n][final 2*n_mats][pow 1][n_queries * query_words] */
size_t orc_basefold_proof_words(int n_commits, const int* commit_sizes, const int* nv, const int* width, int rate_log, int n_queries) {
    const int n_mats = total_mats(n_commits, commit_sizes);
    int n = 0;
    for (int m = 0; m < n_mats; m++) if (nv[m] > n) n = nv[m];
    return 8 * (size_t)n + 2 * (size_t)n_mats + 1 +
           (size_t)n_queries * orc_basefold_query_words(n_commits, commit_sizes, nv, width, rate_log);
}

static void hash_pair(ext2 a, ext2 b, const uint64_t* params, uint64_t* digest4) {
    uint64_t s[8] = {a.c[0], a.c[1], b.c[0], b.c[1], 0, 0, 0, 0};
    orc_poseidon2_permute(s, params);
    memcpy(digest4, s, 32);
}
static void compress(const uint64_t* l4, const uint64_t* r4, const uint64_t* params, uint64_t* out4) {
    uint64_t s[8];
    memcpy(s, l4, 32);
    memcpy(s + 4, r4, 32);
    orc_poseidon2_permute(s, params);
    memcpy(out4, s, 32);
}
/* levels over `n_leaf` (power of two) digests already stored at out[0 .. 4 n_leaf) */
static void tree_from_leaves(uint64_t* out, size_t n_leaf, const uint64_t* params) {
    uint64_t* child = out;
    size_t n = n_leaf;
    while (n > 1) {
        uint64_t* parent = child + 4 * n;
        for (size_t i = 0; i < n / 2; i++) compress(child + 8 * i, child + 8 * i + 4, params, parent + 4 * i);
        child = parent;
        n /= 2;
    }
}
static const uint64_t* tree_level(const uint64_t* levels, size_t n_leaf, int l) {
    const uint64_t* p = levels;
    size_t n = n_leaf;
    for (int i = 0; i < l; i++) { p += 4 * n; n /= 2; }
    return p;
}
static void tree_path(const uint64_t* levels, size_t n_leaf, int depth, size_t idx, uint64_t* path) {
    for (int l = 0; l < depth; l++) {
        memcpy(path + 4 * l, tree_level(levels, n_leaf, l) + 4 * (idx ^ 1), 32);
        idx >>= 1;
    }
}
static void path_root(const uint64_t* leaf4, size_t idx, const uint64_t* path, int depth, const uint64_t* params, uint64_t* out4) {
    uint64_t cur[4];
    memcpy(cur, leaf4, 32);
    for (int l = 0; l < depth; l++) {
        uint64_t nxt[4];
        if (idx & 1) compress(path + 4 * l, cur, params, nxt);
        else compress(cur, path + 4 * l, params, nxt);
        memcpy(cur, nxt, 32);
        idx >>= 1;
    }
    memcpy(out4, cur, 32);
}

/* fold coefficient of pair `leaf_idx` at codeword height log2_height (pcs/mod.rs:7765-7769) */
static uint64_t folding_coeff(int log2_height, size_t leaf_idx) {
    uint64_t g_inv = gl_inv(orc_two_adic_generator(log2_height));
    unsigned e = bitrev_u((unsigned)leaf_idx, log2_height - 1);
    return gl_mul(gl_pow(g_inv, e), gl_inv(2));
}
static ext2 fold_pair(ext2 a, ext2 b, ext2 r, uint64_t coeff) {
    ext2 lo = e2_scale(e2_add(a, b), gl_inv(2));
    ext2 hi = e2_scale(e2_sub(a, b), coeff);
    return e2_add(lo, e2_mul(r, e2_sub(hi, lo)));
}

int orc_basefold_open(int n_commits, const int* commit_sizes, const int* nv, const int* width, const uint64_t* const* traces,
                      const uint64_t* const* points, const uint64_t* const* evals, int rate_log, int n_queries, int pow_bits,
                      const uint64_t* params, orc_transcript* tr, uint64_t* proof) {
    int n = 0, total_cols = 0;
    const int n_mats = total_mats(n_commits, commit_sizes);
    if (n_commits < 1 || n_mats < 1 || n_mats > 4096) return -1;
    if (!tr->append_base || !tr->sample_bits || !tr->fork) return -1;
    for (int m = 0; m < n_mats; m++) {
        if (nv[m] < 1 || width[m] < 1) return -1;
        if (nv[m] > n) n = nv[m];
        total_cols += width[m];
    }
    const int H = n + rate_log;
    int rc = 0;
    /* input codewords + ONE mixed-height tree per commitment (what commit_traces produced) */
    uint64_t** cw = calloc(n_mats, sizeof(*cw));
    uint64_t** in_tree = calloc(n_commits, sizeof(*in_tree));
    int* log_h = calloc(n_mats, sizeof(int));
    for (int m = 0; m < n_mats; m++) {
        size_t rows = (size_t)1 << nv[m], N = rows << rate_log;
        cw[m] = malloc(N * width[m] * 8);
        for (int c = 0; c < width[m]; c++) rs_encode_col(traces[m] + (size_t)c * rows, nv[m], rate_log, cw[m] + (size_t)c * N);
        log_h[m] = nv[m] + rate_log;
    }
    for (int c = 0, m0 = 0; c < n_commits; m0 += commit_sizes[c], c++) {
        const int hc = orc_mmcs_log_max(commit_sizes[c], log_h + m0);
        in_tree[c] = malloc(4 * (((size_t)2 << hc) - 1) * 8);
        orc_mmcs_commit(commit_sizes[c], log_h + m0, width + m0, (const uint64_t* const*)cw + m0, params, in_tree[c]);
    }
    /* batch coefficients */
    tr_label(tr, "batch coeffs");
    ext2 alpha = tr_sample(tr);
    ext2* coeff = malloc(sizeof(ext2) * total_cols);
    {
        ext2 c = e2_one();
        for (int i = 0; i < total_cols; i++) { coeff[i] = c; c = e2_mul(c, alpha); }
    }
    /* per height: batched codeword B[h]; per matrix: F_m, E_m, S_m */
    ext2** B = calloc(H + 1, sizeof(*B));
    ext2** F = calloc(n_mats, sizeof(*F));
    ext2** E = calloc(n_mats, sizeof(*E));
    ext2* S = calloc(n_mats, sizeof(ext2));
    int* live_nv = calloc(n_mats, sizeof(int)); /* remaining variables of the live tables */
    {
        int ci = 0;
        for (int m = 0; m < n_mats; m++) {
            int h = nv[m] + rate_log;
            size_t rows = (size_t)1 << nv[m], N = rows << rate_log;
            if (!B[h]) B[h] = calloc(N, sizeof(ext2));
            F[m] = calloc(rows, sizeof(ext2));
            E[m] = malloc(rows * sizeof(ext2));
            orc_build_eq_x_r_vec(points[m], nv[m], (uint64_t*)E[m]);
            S[m] = e2_zero();
            for (int c = 0; c < width[m]; c++, ci++) {
                for (size_t i = 0; i < N; i++) B[h][i] = e2_add(B[h][i], e2_scale(coeff[ci], cw[m][(size_t)c * N + i]));
                for (size_t i = 0; i < rows; i++) F[m][i] = e2_add(F[m][i], e2_scale(coeff[ci], traces[m][(size_t)c * rows + i]));
                S[m] = e2_add(S[m], e2_mul(coeff[ci], ld2(evals[m] + 2 * c)));
            }
            live_nv[m] = nv[m];
        }
    }
    uint64_t* msgs = proof;
    uint64_t* commits = proof + 4 * (size_t)n;
    uint64_t* finalm = proof + 8 * (size_t)n;
    uint64_t* powp = finalm + 2 * (size_t)n_mats;
    uint64_t* qbase = powp + 1;
    const size_t qwords = orc_basefold_query_words(n_commits, commit_sizes, nv, width, rate_log);

    /* commit phase */
    ext2** C = calloc(n + 1, sizeof(*C));          /* running codeword before the fold of round r */
    uint64_t** ctree = calloc(n, sizeof(*ctree));  /* tree of its pairs */
    ext2* chal = malloc(sizeof(ext2) * (n ? n : 1));
    C[0] = malloc(sizeof(ext2) << H);
    memcpy(C[0], B[H], sizeof(ext2) << H);
    for (int r = 0; r < n; r++) {
        const int h = H - r;
        ext2 p1 = e2_zero(), p2 = e2_zero();
        for (int m = 0; m < n_mats; m++) {
            const int s_m = n - nv[m];
            if (s_m > r) { /* joins later: constant in this variable (pcs/mod.rs:1153-1156 scale factor) */
                ext2 c = e2_scale(S[m], gl_pow(2, (uint64_t)(s_m - r - 1)));
                p1 = e2_add(p1, c);
                p2 = e2_add(p2, c);
                continue;
            }
            size_t half = (size_t)1 << (live_nv[m] - 1);
            for (size_t j = 0; j < half; j++) {
                ext2 e0 = E[m][2 * j], e1 = E[m][2 * j + 1], f0 = F[m][2 * j], f1 = F[m][2 * j + 1];
                ext2 e2v = e2_add(e1, e2_sub(e1, e0)), f2v = e2_add(f1, e2_sub(f1, f0));
                p1 = e2_add(p1, e2_mul(e1, f1));
                p2 = e2_add(p2, e2_mul(e2v, f2v));
            }
        }
        st2(msgs + 4 * r, p1);
        st2(msgs + 4 * r + 2, p2);
        tr_ext(tr, p1);
        tr_ext(tr, p2);
        tr_label(tr, "commit round");
        ext2 c = tr_sample(tr);
        chal[r] = c;
        /* commit the running codeword (pairs are the leaves) */
        size_t n_leaf = (size_t)1 << (h - 1);
        ctree[r] = malloc(4 * (2 * n_leaf - 1) * 8);
        for (size_t j = 0; j < n_leaf; j++) hash_pair(C[r][2 * j], C[r][2 * j + 1], params, ctree[r] + 4 * j);
        tree_from_leaves(ctree[r], n_leaf, params);
        memcpy(commits + 4 * r, tree_level(ctree[r], n_leaf, h - 1), 32);
        tr_digest(tr, commits + 4 * r);
        /* fold, then join the codewords of the next height */
        C[r + 1] = malloc(sizeof(ext2) * n_leaf);
        for (size_t j = 0; j < n_leaf; j++) {
            ext2 v = fold_pair(C[r][2 * j], C[r][2 * j + 1], c, folding_coeff(h, j));
            if (B[h - 1]) v = e2_add(v, B[h - 1][j]);
            C[r + 1][j] = v;
        }
        for (int m = 0; m < n_mats; m++) {
            if (n - nv[m] > r) continue;
            size_t half = (size_t)1 << (live_nv[m] - 1);
            for (size_t j = 0; j < half; j++) {
                F[m][j] = e2_add(F[m][2 * j], e2_mul(c, e2_sub(F[m][2 * j + 1], F[m][2 * j])));
                E[m][j] = e2_add(E[m][2 * j], e2_mul(c, e2_sub(E[m][2 * j + 1], E[m][2 * j])));
            }
            live_nv[m]--;
        }
    }
    /* final message: one row (width 1) per opening point, in order */
    ext2 total = e2_zero();
    for (int m = 0; m < n_mats; m++) {
        st2(finalm + 2 * m, F[m][0]);
        total = e2_add(total, F[m][0]);
        tr_ext(tr, F[m][0]);
    }
    for (size_t i = 0; i < ((size_t)1 << rate_log); i++)
        if (!e2_eq(C[n][i], total)) rc = -2; /* the final codeword must be the constant codeword of the message */
    /* proof of work (pcs/mod.rs:1248-1251, 8125-8155) */
    *powp = 0;
    if (pow_bits > 0) *powp = orc_tr_grind(tr, pow_bits);
    /* queries (pcs/mod.rs:1252-1266) */
    tr_label(tr, "query indices");
    for (int q = 0; q < n_queries; q++) {
        uint64_t* out = qbase + (size_t)q * qwords;
        size_t query = (size_t)orc_tr_sample_bits(tr, H);
        *out++ = query;
        for (int c = 0, m0 = 0; c < n_commits; m0 += commit_sizes[c], c++) {
            const int hc = orc_mmcs_log_max(commit_sizes[c], log_h + m0);
            size_t wsum = 0;
            for (int m = m0; m < m0 + commit_sizes[c]; m++) wsum += (size_t)width[m];
            /* reduced_index = query >> bits_reduced (pcs/mod.rs:7553-7554) */
            orc_mmcs_open(commit_sizes[c], log_h + m0, width + m0, (const uint64_t* const*)cw + m0, in_tree[c], query >> (H - hc), out,
                          out + wsum);
            out += wsum + 4 * (size_t)hc;
        }
        size_t idx = query;
        for (int r = 0; r < n; r++) {
            int h = H - r;
            st2(out, C[r][idx ^ 1]);
            out += 2;
            tree_path(ctree[r], (size_t)1 << (h - 1), h - 1, idx >> 1, out);
            out += 4 * (h - 1);
            idx >>= 1;
        }
    }
    for (int m = 0; m < n_mats; m++) { free(cw[m]); free(F[m]); free(E[m]); }
    for (int c = 0; c < n_commits; c++) free(in_tree[c]);
    free(log_h);
    for (int h = 0; h <= H; h++) free(B[h]);
    for (int r = 0; r <= n; r++) free(C[r]);
    for (int r = 0; r < n; r++) free(ctree[r]);
    free(cw); free(in_tree); free(F); free(E); free(S); free(B); free(C); free(ctree); free(chal); free(coeff); free(live_nv);
    return rc;
}

/* input commitment roots, as commit_traces publishes them: ONE root per commitment (cpu/mod.rs:559-584) */
void orc_basefold_commit_roots(int n_commits, const int* commit_sizes, const int* nv, const int* width, const uint64_t* const* traces,
                               int rate_log, const uint64_t* params, uint64_t* roots) {
    for (int c = 0, m0 = 0; c < n_commits; m0 += commit_sizes[c], c++) {
        const int k = commit_sizes[c];
        uint64_t** cw = calloc(k, sizeof(*cw));
        int* log_h = calloc(k, sizeof(int));
        for (int i = 0; i < k; i++) {
            const int m = m0 + i;
            size_t rows = (size_t)1 << nv[m], N = rows << rate_log;
            cw[i] = malloc(N * width[m] * 8);
            for (int col = 0; col < width[m]; col++) rs_encode_col(traces[m] + (size_t)col * rows, nv[m], rate_log, cw[i] + (size_t)col * N);
            log_h[i] = nv[m] + rate_log;
        }
        const int hc = orc_mmcs_log_max(k, log_h);
        uint64_t* t = malloc(4 * (((size_t)2 << hc) - 1) * 8);
        orc_mmcs_commit(k, log_h, width + m0, (const uint64_t* const*)cw, params, t);
        memcpy(roots + 4 * c, t + 4 * (((size_t)2 << hc) - 2), 32);
        for (int i = 0; i < k; i++) free(cw[i]);
        free(cw); free(log_h); free(t);
    }
}

/* returns 0 when the proof is accepted, a positive code naming the failed check otherwise */
int orc_basefold_verify(int n_commits, const int* commit_sizes, const int* nv, const int* width, const uint64_t* roots,
                        const uint64_t* const* points, const uint64_t* const* evals, int rate_log, int n_queries, int pow_bits,
                        const uint64_t* params, orc_transcript* tr, const uint64_t* proof) {
    int n = 0, total_cols = 0;
    const int n_mats = total_mats(n_commits, commit_sizes);
    if (!tr->append_base || !tr->sample_bits) return -1;
    for (int m = 0; m < n_mats; m++) {
        if (nv[m] > n) n = nv[m];
        total_cols += width[m];
    }
    const int H = n + rate_log;
    const uint64_t* msgs = proof;
    const uint64_t* commits = proof + 4 * (size_t)n;
    const uint64_t* finalm = proof + 8 * (size_t)n;
    const uint64_t* powp = finalm + 2 * (size_t)n_mats;
    const uint64_t* qbase = powp + 1;
    const size_t qwords = orc_basefold_query_words(n_commits, commit_sizes, nv, width, rate_log);
    int rc = 0;
    int* log_h = malloc(sizeof(int) * (size_t)n_mats);
    for (int m = 0; m < n_mats; m++) log_h[m] = nv[m] + rate_log;

    tr_label(tr, "batch coeffs");
    ext2 alpha = tr_sample(tr);
    ext2* coeff = malloc(sizeof(ext2) * total_cols);
    {
        ext2 c = e2_one();
        for (int i = 0; i < total_cols; i++) { coeff[i] = c; c = e2_mul(c, alpha); }
    }
    /* initial claim (pcs/mod.rs:1147-1183) */
    ext2 claim = e2_zero();
    {
        int ci = 0;
        for (int m = 0; m < n_mats; m++) {
            uint64_t scale = gl_pow(2, (uint64_t)(n - nv[m]));
            for (int c = 0; c < width[m]; c++, ci++) claim = e2_add(claim, e2_scale(e2_mul(coeff[ci], ld2(evals[m] + 2 * c)), scale));
        }
    }
    ext2* chal = malloc(sizeof(ext2) * (n ? n : 1));
    for (int r = 0; r < n; r++) {
        ext2 ev[2] = {ld2(msgs + 4 * r), ld2(msgs + 4 * r + 2)};
        tr_ext(tr, ev[0]);
        tr_ext(tr, ev[1]);
        tr_label(tr, "commit round");
        chal[r] = tr_sample(tr);
        ext2 p0 = e2_sub(claim, ev[0]);
        orc_extrapolate_uni_poly(p0.c, (const uint64_t*)ev, 2, chal[r].c, claim.c);
        tr_digest(tr, commits + 4 * r);
    }
    /* final claim (pcs/mod.rs:496-580): a point with nv coordinates meets the LAST nv fold challenges */
    ext2 expect = e2_zero(), total = e2_zero();
    for (int m = 0; m < n_mats; m++) {
        ext2 v = ld2(finalm + 2 * m), e;
        tr_ext(tr, v);
        orc_eq_eval(points[m], (const uint64_t*)(chal + (n - nv[m])), nv[m], e.c);
        expect = e2_add(expect, e2_mul(e, v));
        total = e2_add(total, v);
    }
    if (!e2_eq(expect, claim)) rc = 1;
    if (!rc && pow_bits > 0) {
        if (*powp >= GL_P || !orc_tr_check_witness(tr, pow_bits, *powp)) rc = 2; /* pcs/mod.rs:1248-1251 */
    }
    tr_label(tr, "query indices");
    ext2* reduced = malloc(sizeof(ext2) * (H + 1));
    char* has = malloc(H + 1);
    for (int q = 0; q < n_queries && !rc; q++) {
        const uint64_t* in = qbase + (size_t)q * qwords;
        size_t query = (size_t)orc_tr_sample_bits(tr, H);
        if (*in++ != query) { rc = 3; break; }
        memset(has, 0, H + 1);
        int ci = 0;
        for (int c = 0, m0 = 0; c < n_commits && !rc; m0 += commit_sizes[c], c++) {
            const int hc = orc_mmcs_log_max(commit_sizes[c], log_h + m0);
            size_t wsum = 0;
            for (int m = m0; m < m0 + commit_sizes[c]; m++) wsum += (size_t)width[m];
            /* one MMCS opening per commitment at reduced_index (pcs/mod.rs:7553-7566) */
            if (orc_mmcs_verify(commit_sizes[c], log_h + m0, width + m0, roots + 4 * c, query >> (H - hc), in, in + wsum, params)) rc = 4;
            /* reduce the opened values per height with the batch coefficients (pcs/mod.rs:7567-7612) */
            for (int m = m0; m < m0 + commit_sizes[c]; m++) {
                const int hm = log_h[m];
                if (!has[hm]) { reduced[hm] = e2_zero(); has[hm] = 1; }
                for (int k = 0; k < width[m]; k++, ci++) {
                    if (in[k] >= GL_P) rc = rc ? rc : 5;
                    reduced[hm] = e2_add(reduced[hm], e2_scale(coeff[ci], in[k]));
                }
                in += width[m];
            }
            in += 4 * (size_t)hc;
        }
        size_t idx = query;
        ext2 folded = e2_zero();
        for (int r = 0; r < n && !rc; r++) {
            int h = H - r;
            ext2 leafs[2];
            leafs[(idx & 1) ^ 1] = ld2(in);
            leafs[idx & 1] = has[h] ? e2_add(folded, reduced[h]) : folded;
            has[h] = 0;
            in += 2;
            uint64_t leaf[4], root[4];
            hash_pair(leafs[0], leafs[1], params, leaf);
            path_root(leaf, idx >> 1, in, h - 1, params, root);
            if (memcmp(root, commits + 4 * r, 32) != 0) rc = 6;
            in += 4 * (h - 1);
            folded = fold_pair(leafs[0], leafs[1], chal[r], folding_coeff(h, idx >> 1));
            idx >>= 1;
        }
        for (int h = 0; h <= H; h++) if (has[h] && h != rate_log) rc = rc ? rc : 7; /* unused reduced openings */
        /* a matrix may not have height rate_log (nv >= 1), so nothing joins after the last fold */
        if (!rc && !e2_eq(folded, total)) rc = 8; /* constant final codeword = sum of the message rows */
    }
    free(coeff); free(chal); free(reduced); free(has); free(log_h);
    return rc;
}
