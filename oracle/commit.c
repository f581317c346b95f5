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
