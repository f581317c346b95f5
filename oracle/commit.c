/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  Basefold commit-path primitives:
 * DFT over Goldilocks by definition (O(N^2), bit-reversed output), Poseidon2 (width 8) permutation,
 * overwrite-mode row sponge and Merkle tree.  PARITY UNPINNED: the reference's instances live in EXT
 * crates (mpcs / poseidon / p3-*) — this file restates the published p3 0.4.3 shapes:
 *   - two-adic generator 7^((p-1)/2^32) (p3-goldilocks `TWO_ADIC_GENERATOR`),
 *   - Poseidon2 external layer = M4 circulant-light [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] + column sums,
 *     internal layer = diag * x + sum(x), S-box x^7, 4 + 22 + 4 rounds,
 *   - PaddingFreeSponge<8,4,4> leaf hash, TruncatedPermutation 2-to-1 compression.
 */
#include "oracle.h"
#include "gl64.h"
#include <stdlib.h>
#include <string.h>

static unsigned bitrev(unsigned x, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

uint64_t orc_two_adic_generator(int bits) {
    uint64_t g = gl_pow(7, (GL_P - 1) >> 32); /* order 2^32 */
    for (int i = bits; i < 32; i++) g = gl_mul(g, g);
    return g;
}

/* out[bitrev(k)] = sum_j in[j] w^(jk), w = primitive 2^log_n-th root (inverse: w^-1 and 1/N, natural output
 * from bit-reversed input) */
void orc_dft_bitrev(const uint64_t* in, int log_n, int inverse, uint64_t* out) {
    size_t n = (size_t)1 << log_n;
    uint64_t w = orc_two_adic_generator(log_n);
    if (!inverse) {
        for (size_t k = 0; k < n; k++) {
            uint64_t wk = gl_pow(w, k), acc = 0, x = 1;
            for (size_t j = 0; j < n; j++) { acc = gl_add(acc, gl_mul(in[j], x)); x = gl_mul(x, wk); }
            out[bitrev((unsigned)k, log_n)] = acc;
        }
    } else {
        uint64_t wi = gl_inv(w), ninv = gl_inv((uint64_t)n % GL_P);
        for (size_t j = 0; j < n; j++) {
            uint64_t wj = gl_pow(wi, j), acc = 0, x = 1;
            for (size_t k = 0; k < n; k++) { acc = gl_add(acc, gl_mul(in[bitrev((unsigned)k, log_n)], x)); x = gl_mul(x, wj); }
            out[j] = gl_mul(acc, ninv);
        }
    }
}

/* params layout: ext_rc[8][8], int_rc[22], int_diag[8] (138 words) */
static uint64_t sbox7(uint64_t x) {
    uint64_t x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x3 = gl_mul(x2, x);
    return gl_mul(x4, x3);
}
static void ext_layer(uint64_t* s) {
    static const uint64_t M4[4][4] = {{2, 3, 1, 1}, {1, 2, 3, 1}, {1, 1, 2, 3}, {3, 1, 1, 2}};
    uint64_t t[8];
    for (int c = 0; c < 2; c++)
        for (int i = 0; i < 4; i++) {
            uint64_t acc = 0;
            for (int j = 0; j < 4; j++) acc = gl_add(acc, gl_mul(M4[i][j], s[4 * c + j]));
            t[4 * c + i] = acc;
        }
    for (int i = 0; i < 8; i++) s[i] = gl_add(t[i], gl_add(t[i & 3], t[(i & 3) + 4]));
}
void orc_poseidon2_permute(uint64_t* s, const uint64_t* params) {
    const uint64_t* ext_rc = params;
    const uint64_t* int_rc = params + 64;
    const uint64_t* diag = params + 64 + 22;
    ext_layer(s);
    for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 8; i++) s[i] = sbox7(gl_add(s[i], ext_rc[8 * r + i]));
        ext_layer(s);
    }
    for (int r = 0; r < 22; r++) {
        s[0] = sbox7(gl_add(s[0], int_rc[r]));
        uint64_t sum = 0;
        for (int i = 0; i < 8; i++) sum = gl_add(sum, s[i]);
        for (int i = 0; i < 8; i++) s[i] = gl_add(gl_mul(s[i], diag[i]), sum);
    }
    for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 8; i++) s[i] = sbox7(gl_add(s[i], ext_rc[8 * r + i]));
        ext_layer(s);
    }
}
void orc_poseidon2_default_params(uint64_t* params) {
    const uint64_t seed = 0x706f736569646f6eULL;
    for (int i = 0; i < 64 + 22; i++) params[i] = splitmix_gl(seed, (uint64_t)i);
    static const uint64_t diag[8] = {0xa98811a1fed4e3a5ULL, 0x1cc48b54f377e2a0ULL, 0xe40cd4f6c5609a26ULL, 0x11de79ebca97a4a3ULL,
                                     0x9177c73d8b7e929cULL, 0x2a6fe8085797e791ULL, 0x3de6e93329f8d5adULL, 0x3f7af9125da962feULL};
    memcpy(params + 86, diag, sizeof(diag));
}
/* column-major matrix (rows = 2^log_rows, `width` columns) -> all tree levels concatenated:
 * level 0 (leaves, 4 words each) ... root.  out must hold 4 * (2^(log_rows+1) - 1) words. */
void orc_merkle_commit(const uint64_t* m, int log_rows, int width, const uint64_t* params, uint64_t* out) {
    size_t rows = (size_t)1 << log_rows;
    for (size_t r = 0; r < rows; r++) {
        uint64_t s[8] = {0};
        for (int c = 0; c < width; c += 4) {
            for (int k = 0; k < 4 && c + k < width; k++) s[k] = m[(size_t)(c + k) * rows + r];
            orc_poseidon2_permute(s, params);
        }
        memcpy(out + 4 * r, s, 32);
    }
    uint64_t* child = out;
    size_t n = rows;
    while (n > 1) {
        uint64_t* parent = child + 4 * n;
        for (size_t i = 0; i < n / 2; i++) {
            uint64_t s[8];
            memcpy(s, child + 8 * i, 64);
            orc_poseidon2_permute(s, params);
            memcpy(parent + 4 * i, s, 32);
        }
        child = parent;
        n /= 2;
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Mixed-height Merkle commitment: ONE root for matrices of several power-of-two heights.  The reference commits all
 * trace matrices of a `commit_traces` call under one `PCS::Commitment` (ceno_zkvm/src/scheme/cpu/mod.rs:559-584,
 * PCS::batch_commit) and opens several `num_vars` under one commitment (ceno_recursion_v2/src/pcs/mod.rs:1123-1135,
 * 7547-7565: reduced_index = query >> bits_reduced; :7783-7802 one opening proof per commitment).  The tree itself is
 * p3-merkle-tree 0.4.3 (EXT) `MerkleTree::new` / `MerkleTreeMmcs::{open_batch, verify_batch}`, whose published
 * algorithm this restates:
 *   - matrices sorted by height, tallest first, STABLE (equal heights keep the caller's order);
 *   - first digest layer: row i of ALL tallest matrices, concatenated in that order, through the padding-free sponge;
 *   - each next layer has half the nodes: next[i] = compress(prev[2i], prev[2i+1]); when matrices of exactly that
 *     height exist ("inject"): next[i] = compress( compress(prev[2i], prev[2i+1]), sponge(row i of those matrices) );
 *   - opening at `index` (a row of the tallest height): matrix m shows row index >> (log_max - log_rows[m]); the proof
 *     is the sibling at every layer, digest_layers[l][(index >> l) ^ 1], l = 0 .. log_max - 1;
 *   - verify: sponge of the tallest rows, then per sibling compress in index order and, where shorter matrices join,
 *     compress with the sponge of their opened rows.
 * Heights here are powers of two (codewords), so p3's padding rules for odd layers never apply.
 * ------------------------------------------------------------------------------------------------------------------ */
static void sponge_rows(int n_mats, const int* log_rows, const int* width, const uint64_t* const* col_major, const int* order,
                        int log_h, size_t row, const uint64_t* params, uint64_t* digest4) {
    uint64_t s[8] = {0};
    int k = 0;
    for (int oi = 0; oi < n_mats; oi++) {
        int m = order[oi];
        if (log_rows[m] != log_h) continue;
        size_t rows = (size_t)1 << log_h;
        for (int c = 0; c < width[m]; c++) {
            s[k++] = col_major[m][(size_t)c * rows + row];
            if (k == 4) { orc_poseidon2_permute(s, params); k = 0; }
        }
    }
    if (k) orc_poseidon2_permute(s, params);
    memcpy(digest4, s, 32);
}
static void mmcs_order(int n_mats, const int* log_rows, int* order) {
    for (int i = 0; i < n_mats; i++) order[i] = i;
    for (int i = 1; i < n_mats; i++) { /* stable insertion sort, tallest first */
        int v = order[i], j = i;
        while (j > 0 && log_rows[order[j - 1]] < log_rows[v]) { order[j] = order[j - 1]; j--; }
        order[j] = v;
    }
}
int orc_mmcs_log_max(int n_mats, const int* log_rows) {
    int h = 0;
    for (int m = 0; m < n_mats; m++) if (log_rows[m] > h) h = log_rows[m];
    return h;
}
/* out_levels: layer 0 (2^log_max digests) ... root; 4 * (2^(log_max+1) - 1) words */
void orc_mmcs_commit(int n_mats, const int* log_rows, const int* width, const uint64_t* const* col_major, const uint64_t* params,
                     uint64_t* out_levels) {
    int* order = malloc(sizeof(int) * (size_t)n_mats);
    mmcs_order(n_mats, log_rows, order);
    const int H = orc_mmcs_log_max(n_mats, log_rows);
    size_t n = (size_t)1 << H;
    for (size_t r = 0; r < n; r++) sponge_rows(n_mats, log_rows, width, col_major, order, H, r, params, out_levels + 4 * r);
    uint64_t* child = out_levels;
    for (int h = H - 1; h >= 0; h--) {
        uint64_t* parent = child + 4 * n;
        n /= 2;
        int inject = 0;
        for (int m = 0; m < n_mats; m++) if (log_rows[m] == h) inject = 1;
        for (size_t i = 0; i < n; i++) {
            uint64_t s[8];
            memcpy(s, child + 8 * i, 64);
            orc_poseidon2_permute(s, params);
            if (inject) {
                sponge_rows(n_mats, log_rows, width, col_major, order, h, i, params, s + 4);
                orc_poseidon2_permute(s, params);
            }
            memcpy(parent + 4 * i, s, 32);
        }
        child = parent;
    }
    free(order);
}
/* rows_out: the opened row of every matrix in the CALLER's order (sum of widths words); path_out: 4 * log_max words */
void orc_mmcs_open(int n_mats, const int* log_rows, const int* width, const uint64_t* const* col_major, const uint64_t* levels,
                   size_t index, uint64_t* rows_out, uint64_t* path_out) {
    const int H = orc_mmcs_log_max(n_mats, log_rows);
    for (int m = 0; m < n_mats; m++) {
        size_t rows = (size_t)1 << log_rows[m], r = index >> (H - log_rows[m]);
        for (int c = 0; c < width[m]; c++) *rows_out++ = col_major[m][(size_t)c * rows + r];
    }
    const uint64_t* lv = levels;
    size_t n = (size_t)1 << H, idx = index;
    for (int l = 0; l < H; l++) {
        memcpy(path_out + 4 * l, lv + 4 * (idx ^ 1), 32);
        lv += 4 * n;
        n /= 2;
        idx >>= 1;
    }
}
/* 0 when the opened rows and the path lead to `root4` */
int orc_mmcs_verify(int n_mats, const int* log_rows, const int* width, const uint64_t* root4, size_t index, const uint64_t* rows,
                    const uint64_t* path, const uint64_t* params) {
    int* order = malloc(sizeof(int) * (size_t)n_mats);
    mmcs_order(n_mats, log_rows, order);
    const int H = orc_mmcs_log_max(n_mats, log_rows);
    /* one-row "matrices" so that sponge_rows can be reused: column c of matrix m = rows[off_m + c] */
    const uint64_t** one = malloc(sizeof(*one) * (size_t)n_mats);
    int* zero_log = malloc(sizeof(int) * (size_t)n_mats);
    {
        size_t off = 0;
        for (int m = 0; m < n_mats; m++) { one[m] = rows + off; off += (size_t)width[m]; zero_log[m] = 0; }
    }
    uint64_t cur[8];
    {   /* sponge over the tallest rows: select them by giving every other matrix a non-matching height */
        for (int m = 0; m < n_mats; m++) zero_log[m] = (log_rows[m] == H) ? 0 : -1;
        sponge_rows(n_mats, zero_log, width, one, order, 0, 0, params, cur);
    }
    size_t idx = index;
    for (int l = 0; l < H; l++) {
        uint64_t s[8];
        if (idx & 1) { memcpy(s, path + 4 * l, 32); memcpy(s + 4, cur, 32); }
        else { memcpy(s, cur, 32); memcpy(s + 4, path + 4 * l, 32); }
        orc_poseidon2_permute(s, params);
        idx >>= 1;
        const int h = H - 1 - l;
        int inject = 0;
        for (int m = 0; m < n_mats; m++) { zero_log[m] = (log_rows[m] == h) ? 0 : -1; inject |= (log_rows[m] == h); }
        if (inject) {
            sponge_rows(n_mats, zero_log, width, one, order, 0, 0, params, s + 4);
            orc_poseidon2_permute(s, params);
        }
        memcpy(cur, s, 32);
    }
    free(order); free(one); free(zero_log);
    return memcmp(cur, root4, 32) != 0;
}
