// Poseidon2 permutation over Goldilocks, width 8 (rate 4 / capacity 4), S-box x^7,
// 8 external (4 + 4) and 22 internal rounds — the shape of p3-poseidon2 0.4.3 for Goldilocks that the
// reference's EXT `poseidon` / `transcript` / `mpcs` crates instantiate (reference Cargo.lock:4111-4375).
//
// PARITY UNPINNED (SURVEY.md §8c(i),(ii)): the round constants are not present anywhere under
// /root/reference.  The defaults below are PLACEHOLDERS derived from SplitMix64("poseidon2-goldilocks-8")
// — structurally valid, not the HorizenLabs constants — and can be replaced at run time through
// ceno_hip_poseidon2_set_constants() once goldens from the real BasicTranscript are available.
// The internal diagonal is the published MATRIX_DIAG_8_GOLDILOCKS.
#pragma once
#include "gl64.hpp"

namespace p2 {

constexpr int WIDTH = 8;
constexpr int RATE = 4;
constexpr int ROUNDS_F = 8;
constexpr int ROUNDS_P = 22;

struct Params {
    uint64_t ext_rc[ROUNDS_F][WIDTH];
    uint64_t int_rc[ROUNDS_P];
    uint64_t int_diag[WIDTH];
};

inline void default_params(Params& p) {
    const uint64_t seed = 0x706f736569646f6eULL;  // "poseidon"
    uint64_t i = 0;
    for (int r = 0; r < ROUNDS_F; r++)
        for (int k = 0; k < WIDTH; k++) p.ext_rc[r][k] = gl::splitmix_gl(seed, i++);
    for (int r = 0; r < ROUNDS_P; r++) p.int_rc[r] = gl::splitmix_gl(seed, i++);
    const uint64_t diag[WIDTH] = {0xa98811a1fed4e3a5ULL, 0x1cc48b54f377e2a0ULL, 0xe40cd4f6c5609a26ULL, 0x11de79ebca97a4a3ULL,
                                  0x9177c73d8b7e929cULL, 0x2a6fe8085797e791ULL, 0x3de6e93329f8d5adULL, 0x3f7af9125da962feULL};
    for (int k = 0; k < WIDTH; k++) p.int_diag[k] = diag[k];
}

GL_HD uint64_t sbox7(uint64_t x) {
    // the three inner products are only multiplied again: they skip canonicalisation (gl::mul_nc)
    uint64_t x2 = gl::mul_nc(x, x);
    uint64_t x3 = gl::mul_nc(x2, x);
    uint64_t x4 = gl::mul_nc(x2, x2);
    return gl::mul(x4, x3);
}

// [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]]
GL_HD void mat4(uint64_t* x) {
    using gl::add;
    uint64_t t01 = add(x[0], x[1]), t23 = add(x[2], x[3]);
    uint64_t t0123 = add(t01, t23);
    uint64_t t01123 = add(t0123, x[1]), t01233 = add(t0123, x[3]);
    uint64_t n3 = add(t01233, gl::dbl(x[0]));
    uint64_t n1 = add(t01123, gl::dbl(x[2]));
    uint64_t n0 = add(t01123, t01);
    uint64_t n2 = add(t01233, t23);
    x[0] = n0; x[1] = n1; x[2] = n2; x[3] = n3;
}

GL_HD void external_linear(uint64_t* s) {
    mat4(s);
    mat4(s + 4);
    uint64_t sums[4];
#pragma unroll
    for (int i = 0; i < 4; i++) sums[i] = gl::add(s[i], s[i + 4]);
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::add(s[i], sums[i & 3]);
}

GL_HD void internal_linear(uint64_t* s, const Params& p) {
    uint64_t sum = 0;
#pragma unroll
    for (int i = 0; i < WIDTH; i++) sum = gl::add(sum, s[i]);
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::mul_add(s[i], p.int_diag[i], sum);
}

// The permutation proper keeps its state NON-CANONICAL (any 64-bit representative of the residue) from the first
// linear layer to the last: every value is next either multiplied (mul_wide takes plain 64-bit integers) or summed,
// and the sums of the linear layers are formed as plain 96-bit integers (3 instructions per addition instead of the
// 8 of a modular one) and reduced once per output word.  One canonicalisation per word at the very end.
GL_HD uint64_t sbox7_nc(uint64_t x) {
    uint64_t x2 = gl::mul_ncm(x, x);
    uint64_t x3 = gl::mul_ncm(x2, x);
    uint64_t x4 = gl::mul_ncm(x2, x2);
    return gl::mul_ncm(x4, x3);
}
// rows [2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]: coefficients sum to 7, so every output is < 7 * 2^64
GL_HD void mat4_lazy(const uint64_t* x, gl::S96* n) {
    using gl::S96;
    const S96 t01 = gl::s96_sum(x[0], x[1]), t23 = gl::s96_sum(x[2], x[3]);
    const S96 t0123 = t01 + t23;
    const S96 t01123 = t0123 + x[1], t01233 = t0123 + x[3];
    n[3] = t01233 + x[0] + x[0];
    n[1] = t01123 + x[2] + x[2];
    n[0] = t01123 + t01;
    n[2] = t01233 + t23;
}
GL_HD void external_linear_nc(uint64_t* s) {
    gl::S96 n[WIDTH];
    mat4_lazy(s, n);
    mat4_lazy(s + 4, n + 4);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const gl::S96 sum = n[i] + n[i + 4];            // < 14 * 2^64
        s[i] = gl::s96_reduce_nc(n[i] + sum);           // < 21 * 2^64
        s[i + 4] = gl::s96_reduce_nc(n[i + 4] + sum);
    }
}
GL_HD void internal_linear_nc(uint64_t* s, const Params& p) {
    gl::S96 sum = gl::s96_sum(s[0], s[1]);
#pragma unroll
    for (int i = 2; i < WIDTH; i++) sum = sum + s[i];   // < 8 * 2^64
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::mul_add_s96_ncm(s[i], p.int_diag[i], sum);
}

GL_HD void permute(uint64_t* s, const Params& p) {
    external_linear_nc(s);
    for (int r = 0; r < ROUNDS_F / 2; r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7_nc(gl::add_nc(s[i], p.ext_rc[r][i]));
        external_linear_nc(s);
    }
    for (int r = 0; r < ROUNDS_P; r++) {
        s[0] = sbox7_nc(gl::add_nc(s[0], p.int_rc[r]));
        internal_linear_nc(s, p);
    }
    for (int r = ROUNDS_F / 2; r < ROUNDS_F; r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7_nc(gl::add_nc(s[i], p.ext_rc[r][i]));
        external_linear_nc(s);
    }
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}
// the same permutation with every intermediate canonical (the straightforward form; tests compare the two)
GL_HD void permute_canonical(uint64_t* s, const Params& p) {
    external_linear(s);
    for (int r = 0; r < ROUNDS_F / 2; r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add(s[i], p.ext_rc[r][i]));
        external_linear(s);
    }
    for (int r = 0; r < ROUNDS_P; r++) {
        s[0] = sbox7(gl::add(s[0], p.int_rc[r]));
        internal_linear(s, p);
    }
    for (int r = ROUNDS_F / 2; r < ROUNDS_F; r++) {
#pragma unroll
        for (int i = 0; i < WIDTH; i++) s[i] = sbox7(gl::add(s[i], p.ext_rc[r][i]));
        external_linear(s);
    }
}

#if defined(__HIPCC__)
// ---- one permutation spread over 8 adjacent lanes (lane g = threadIdx & 7 holds state word g) ----
// A tree level with few nodes is bound by the LATENCY of one permutation (a ~14k-instruction dependent chain on
// one lane, ~45 us); spreading the state over 8 lanes shortens the chain several times at ~2x the total work.
// Used for Merkle levels too small to fill the chip.  Linear layers via cross-lane permutes:
//   mat4 row i = sum(chunk) + x_i + 2 x_{(i+1)&3};  external: 2 t_i + t_{i^4};  internal: diag_g x_g + sum(all 8).
// DPP lane permutes (VALU, a few cycles) instead of ds_bpermute shuffles (~100 cycles each, dependent):
// quad_perm [1,0,3,2] = lane^1, [2,3,0,1] = lane^2, [1,2,3,0] = next lane of the quad,
// row_half_mirror (j -> 7-j) followed by quad_perm [3,2,1,0] (j -> j^3 inside the quad) = lane^4.
template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
template <int CTRL>
__device__ __forceinline__ gl::S96 dpp96(gl::S96 v) {
    gl::S96 r;
    r.w0 = (uint32_t)__builtin_amdgcn_update_dpp((int)v.w0, (int)v.w0, CTRL, 0xF, 0xF, false);
    r.w1 = (uint32_t)__builtin_amdgcn_update_dpp((int)v.w1, (int)v.w1, CTRL, 0xF, 0xF, false);
    r.w2 = (uint32_t)__builtin_amdgcn_update_dpp((int)v.w2, (int)v.w2, CTRL, 0xF, 0xF, false);
    return r;
}
// state words are any 64-bit representatives; the layer's sums are plain 96-bit integers (coefficients add up to 21)
__device__ __forceinline__ uint64_t lanes8_external(uint64_t x) {
    using gl::S96;
    const S96 s1 = gl::s96_sum(x, dpp64<0xB1>(x));
    const S96 s = s1 + dpp96<0x4E>(s1);
    const uint64_t nb = dpp64<0x39>(x);
    const S96 t = s + x + nb + nb;
    const S96 o = dpp96<0x1B>(dpp96<0x141>(t));
    return gl::s96_reduce_nc(t + t + o);
}
__device__ __forceinline__ uint64_t permute_lanes8(uint64_t x, const Params& p) {
    const int g = threadIdx.x & 7;
    x = lanes8_external(x);
    for (int r = 0; r < ROUNDS_F / 2; r++) {
        x = sbox7_nc(gl::add_nc(x, p.ext_rc[r][g]));
        x = lanes8_external(x);
    }
    const uint64_t dg = p.int_diag[g];
    for (int r = 0; r < ROUNDS_P; r++) {
        const uint64_t y = sbox7_nc(gl::add_nc(x, p.int_rc[r]));
        x = g == 0 ? y : x;
        gl::S96 sum = gl::s96_sum(x, dpp64<0xB1>(x));
        sum = sum + dpp96<0x4E>(sum);
        sum = sum + dpp96<0x141>(sum);  // every lane of a quad holds the quad sum: the mirror lane is in the other quad
        x = gl::mul_add_s96_ncm(x, dg, sum);
    }
    for (int r = ROUNDS_F / 2; r < ROUNDS_F; r++) {
        x = sbox7_nc(gl::add_nc(x, p.ext_rc[r][g]));
        x = lanes8_external(x);
    }
    return gl::canon(x);
}
#endif

}  // namespace p2
