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
 * Deviations forced by what is absent in-tree (all inside "unpinned" territory):
 *   - field/hash instances are this repo's Goldilocks ones (commit.c): Poseidon2 width 8, one tree per trace matrix
 *     (p3's mixed-height MMCS is an EXT crate);
 *   - digests are observed as two extension elements; query indices are the low bits of c0 of a sampled ext;
 *   - proof of work: seed = sample, witness = least w with permute(seed.c0, seed.c1, w, 0..)[0] = 0 mod 2^bits,
 *     then observed (p3's grinding challenger needs a transcript fork the C transcript interface does not have).
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

size_t orc_basefold_query_words(int n_mats, const int* nv, const int* width, int rate_log) {
    int n = 0;
    for (int m = 0; m < n_mats; m++) if (nv[m] > n) n = nv[m];
    size_t w = 1;
    for (int m = 0; m < n_mats; m++) w += (size_t)width[m] + 4 * (size_t)(nv[m] + rate_log);
    for (int r = 0; r < n; r++) w += 2 + 4 * (size_t)(n + rate_log - r - 1);
    return w;
}
/* [msgs 4n][commits 4n][final 2*n_mats][pow 1][n_queries * query_words] */
size_t orc_basefold_proof_words(int n_mats, const int* nv, const int* width, int rate_log, int n_queries) {
    int n = 0;
    for (int m = 0; m < n_mats; m++) if (nv[m] > n) n = nv[m];
    return 8 * (size_t)n + 2 * (size_t)n_mats + 1 + (size_t)n_queries * orc_basefold_query_words(n_mats, nv, width, rate_log);
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

static uint64_t pow_hash(ext2 seed, uint64_t w, const uint64_t* params) {
    uint64_t s[8] = {seed.c[0], seed.c[1], w, 0, 0, 0, 0, 0};
    orc_poseidon2_permute(s, params);
    return s[0];
}

int orc_basefold_open(int n_mats, const int* nv, const int* width, const uint64_t* const* traces, const uint64_t* const* points,
                      const uint64_t* const* evals, int rate_log, int n_queries, int pow_bits, const uint64_t* params,
                      orc_transcript* tr, uint64_t* proof) {
    int n = 0, total_cols = 0;
    if (n_mats < 1 || n_mats > 4096) return -1;
    for (int m = 0; m < n_mats; m++) {
        if (nv[m] < 1 || width[m] < 1) return -1;
        if (nv[m] > n) n = nv[m];
        total_cols += width[m];
    }
    const int H = n + rate_log;
    int rc = 0;
    /* input codewords + trees (what commit_traces produced) */
    uint64_t** cw = calloc(n_mats, sizeof(*cw));
    uint64_t** in_tree = calloc(n_mats, sizeof(*in_tree));
    for (int m = 0; m < n_mats; m++) {
        size_t rows = (size_t)1 << nv[m], N = rows << rate_log;
        cw[m] = malloc(N * width[m] * 8);
        for (int c = 0; c < width[m]; c++) rs_encode_col(traces[m] + (size_t)c * rows, nv[m], rate_log, cw[m] + (size_t)c * N);
        in_tree[m] = malloc(4 * (2 * N - 1) * 8);
        orc_merkle_commit(cw[m], nv[m] + rate_log, width[m], params, in_tree[m]);
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
    const size_t qwords = orc_basefold_query_words(n_mats, nv, width, rate_log);

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
    /* proof of work */
    *powp = 0;
    if (pow_bits > 0) {
        ext2 seed = tr_sample(tr);
        uint64_t w = 0, mask = ((uint64_t)1 << pow_bits) - 1;
        while (pow_hash(seed, w, params) & mask) w++;
        *powp = w;
        ext2 we = {{w, 0}};
        tr_ext(tr, we);
    }
    /* queries */
    tr_label(tr, "query indices");
    for (int q = 0; q < n_queries; q++) {
        uint64_t* out = qbase + (size_t)q * qwords;
        ext2 s = tr_sample(tr);
        size_t query = (size_t)(s.c[0] & (((uint64_t)1 << H) - 1));
        *out++ = query;
        for (int m = 0; m < n_mats; m++) {
            int hm = nv[m] + rate_log;
            size_t N = (size_t)1 << hm, idx = query >> (H - hm);
            for (int c = 0; c < width[m]; c++) *out++ = cw[m][(size_t)c * N + idx];
            tree_path(in_tree[m], N, hm, idx, out);
            out += 4 * hm;
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
    for (int m = 0; m < n_mats; m++) { free(cw[m]); free(in_tree[m]); free(F[m]); free(E[m]); }
    for (int h = 0; h <= H; h++) free(B[h]);
    for (int r = 0; r <= n; r++) free(C[r]);
    for (int r = 0; r < n; r++) free(ctree[r]);
    free(cw); free(in_tree); free(F); free(E); free(S); free(B); free(C); free(ctree); free(chal); free(coeff); free(live_nv);
    return rc;
}

/* input commitment roots, as commit_traces publishes them */
void orc_basefold_commit_roots(int n_mats, const int* nv, const int* width, const uint64_t* const* traces, int rate_log,
                               const uint64_t* params, uint64_t* roots) {
    for (int m = 0; m < n_mats; m++) {
        size_t rows = (size_t)1 << nv[m], N = rows << rate_log;
        uint64_t* cw = malloc(N * width[m] * 8);
        for (int c = 0; c < width[m]; c++) rs_encode_col(traces[m] + (size_t)c * rows, nv[m], rate_log, cw + (size_t)c * N);
        uint64_t* t = malloc(4 * (2 * N - 1) * 8);
        orc_merkle_commit(cw, nv[m] + rate_log, width[m], params, t);
        memcpy(roots + 4 * m, t + 4 * (2 * N - 2), 32);
        free(cw);
        free(t);
    }
}

/* returns 0 when the proof is accepted, a positive code naming the failed check otherwise */
int orc_basefold_verify(int n_mats, const int* nv, const int* width, const uint64_t* roots, const uint64_t* const* points,
                        const uint64_t* const* evals, int rate_log, int n_queries, int pow_bits, const uint64_t* params,
                        orc_transcript* tr, const uint64_t* proof) {
    int n = 0, total_cols = 0;
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
    const size_t qwords = orc_basefold_query_words(n_mats, nv, width, rate_log);
    int rc = 0;

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
        ext2 seed = tr_sample(tr);
        if (pow_hash(seed, *powp, params) & (((uint64_t)1 << pow_bits) - 1)) rc = 2;
        ext2 we = {{*powp, 0}};
        tr_ext(tr, we);
    }
    tr_label(tr, "query indices");
    ext2* reduced = malloc(sizeof(ext2) * (H + 1));
    char* has = malloc(H + 1);
    for (int q = 0; q < n_queries && !rc; q++) {
        const uint64_t* in = qbase + (size_t)q * qwords;
        ext2 s = tr_sample(tr);
        size_t query = (size_t)(s.c[0] & (((uint64_t)1 << H) - 1));
        if (*in++ != query) { rc = 3; break; }
        memset(has, 0, H + 1);
        int ci = 0;
        for (int m = 0; m < n_mats && !rc; m++) {
            int hm = nv[m] + rate_log;
            size_t idx = query >> (H - hm);
            /* leaf = sponge over the opened row (commit.c orc_merkle_commit) */
            uint64_t st[8] = {0};
            for (int c = 0; c < width[m]; c += 4) {
                for (int k = 0; k < 4 && c + k < width[m]; k++) st[k] = in[c + k];
                orc_poseidon2_permute(st, params);
            }
            uint64_t root[4];
            path_root(st, idx, in + width[m], hm, params, root);
            if (memcmp(root, roots + 4 * m, 32) != 0) rc = 4;
            if (!has[hm]) { reduced[hm] = e2_zero(); has[hm] = 1; }
            for (int c = 0; c < width[m]; c++, ci++) {
                if (in[c] >= GL_P) rc = 5;
                reduced[hm] = e2_add(reduced[hm], e2_scale(coeff[ci], in[c]));
            }
            in += width[m] + 4 * hm;
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
    free(coeff); free(chal); free(reduced); free(has);
    return rc;
}
