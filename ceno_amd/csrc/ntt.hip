// Radix-2 NTT over Goldilocks and Reed–Solomon encoding of trace columns (Basefold commit path).
//
// Reference: the RS encoding inside EXT `mpcs::Basefold::batch_commit` (call sites
// ceno_zkvm/src/scheme/cpu/mod.rs:559-584, gpu/mod.rs:1642-1646); bit-reversed codeword order as in
// `encode_small_final_message` (ceno_recursion_v2/src/pcs/mod.rs:7739-7743).  PARITY UNPINNED: rate,
// evaluation domain and layout live in the EXT crate.  The two-adic generator is the published
// p3-goldilocks one: 7^((p-1)/2^32) = 1753635133440165772 (order 2^32).
//
// Forward = decimation in frequency (natural order in, bit-reversed out), in place, per column:
//   passes over HBM of `R` <= 5 stages each held in registers (up to 32 strided elements per lane, coalesced
//   along the contiguous index), then ONE pass that finishes the last <= 12 stages of every 4096-element
//   block in LDS (groups of 4 stages in registers between barriers).  log N = 21 -> 2 HBM passes (5 + 5 stages) + 1 LDS pass (vs 21 for stage-per-launch); an RS encoding
//   reads the un-extended column in its first pass, so the zero extension costs no traffic.
// Inverse = the mirrored decimation in time with inverse twiddles and the 1/N scale fused in the last pass.
// Twiddles w^i (i < N/2) are tabulated once per size in HBM (8 MB at N = 2^21) and stay L2-resident.
#include "common.hpp"

#include <map>

using namespace gl;

static constexpr int NT = 256;
static constexpr int LOCAL_LOG = 12;  // 4096 elements = 34 KB of LDS per block (with padding)
static constexpr uint64_t TWO_ADIC_GEN_2_32 = 1753635133440165772ULL;


__global__ void __launch_bounds__(NT) k_twiddles(uint64_t* tw, size_t half, uint64_t w) {
    size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < half; i += stride) tw[i] = gl::pow(w, i);
}

static int get_twiddles(ceno_hip_ctx* ctx, int log_n, bool inverse, hipStream_t st, const uint64_t** out) {
    std::lock_guard<std::mutex> g(ctx->tw_mu);
    auto& cache = ctx->twiddles;
    auto key = std::make_pair(log_n, inverse ? 1 : 0);
    auto it = cache.find(key);
    if (it != cache.end()) {
        *out = it->second;
        return 0;
    }
    uint64_t w = gl::pow(TWO_ADIC_GEN_2_32, (uint64_t)1 << (32 - log_n));  // primitive 2^log_n-th root
    if (inverse) w = gl::inv(w);
    size_t half = log_n ? (size_t)1 << (log_n - 1) : 1;
    void* p = nullptr;
    TRY(ctx_alloc(ctx, half * 8, &p));  // from the pool: counted by mem_info / booking, released by ceno_hip_destroy
    hipLaunchKernelGGL(k_twiddles, dim3(grid_for(half, NT, 2048)), dim3(NT), 0, st, (uint64_t*)p, half, w);
    HIP_TRY(ctx, hipGetLastError());
    // the table is shared by every stream of the context from now on: it must be complete before the cache shows it (a
    // second stream — the commit's helper, another lane — would otherwise transform with a half-written table).  Once per size.
    HIP_TRY(ctx, hipStreamSynchronize(st));
    cache[key] = (uint64_t*)p;
    *out = (uint64_t*)p;
    return 0;
}

// R register-resident stages starting at global stage `s` (forward DIF) — or ending at stage s (inverse DIT).
// Block of size M = N >> s splits into L = M >> R interleaved sub-sequences; a lane owns indices
// blk*M + k*L + l, k < 2^R.
// R register-resident radix-2 stages on the 2^R values a lane holds: value k sits at distance k << log_l inside a
// sub-block of 2^(R + log_l) elements whose low index is l; `s` is the global stage of the first (forward) / last
// (inverse) of the R stages.  Shared by the strided HBM passes and the LDS pass.
// Twiddles.  The butterfly of stage q at in-block position kk needs tw[((kk << log_l) + l) << (s + q)]; read as such, the
// loads of the later stages stride 2^(s+q) elements between adjacent lanes — one cache line per lane, 15 (R = 4) or 31
// (R = 5) such loads per item — and the passes were bound by the address coalescer, not by HBM or the ALU (PMC: 69 % of
// the wave cycles of the first RS-encode pass were issue stalls at 10 % VALU activity).  tw[] holds plain powers, so
//     tw[((kk << log_l) + l) << (s + q)] = B_q * Z[kk << q],   B_q = tw[l << s]^(2^q),   Z[j] = tw[j << (log_l + s)]:
// ONE per-lane load (B_0), R - 1 squarings, and 2^(R-1) constants that are uniform over the wave (scalar loads); a
// butterfly whose kk is not zero pays one more multiplication.
template <int R, bool INVERSE>
__device__ __forceinline__ void radix_stages(uint64_t (&v)[1 << R], int log_l, size_t l, int s, const uint64_t* __restrict__ tw) {
    constexpr int E = 1 << R;
    uint64_t B[R];
    B[0] = tw[l << s];
#pragma unroll
    for (int q = 1; q < R; q++) B[q] = mul(B[q - 1], B[q - 1]);
    uint64_t Z[E / 2];
#pragma unroll
    for (int j = 0; j < E / 2; j++) Z[j] = tw[(size_t)j << (log_l + s)];
    if (!INVERSE) {
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int hb = 1 << (R - q - 1);
#pragma unroll
            for (int kk = 0; kk < hb; kk++) {  // position inside the sub-block (lower half); shared by 2^q sub-blocks
                const uint64_t w = kk == 0 ? B[q] : mul(B[q], Z[kk << q]);
#pragma unroll
                for (int m = 0; m < (1 << q); m++) {
                    const int k = kk + m * 2 * hb;
                    const uint64_t a = v[k], b = v[k + hb];
                    v[k] = add(a, b);
                    v[k + hb] = mul(sub(a, b), w);
                }
            }
        }
    } else {
#pragma unroll
        for (int q = R - 1; q >= 0; q--) {
            const int hb = 1 << (R - q - 1);
#pragma unroll
            for (int kk = 0; kk < hb; kk++) {
                const uint64_t w = kk == 0 ? B[q] : mul(B[q], Z[kk << q]);
#pragma unroll
                for (int m = 0; m < (1 << q); m++) {
                    const int k = kk + m * 2 * hb;
                    const uint64_t a = v[k], b = mul(v[k + hb], w);
                    v[k] = add(a, b);
                    v[k + hb] = sub(a, b);
                }
            }
        }
    }
}

template <int R, bool INVERSE>
__global__ void __launch_bounds__(NT) k_ntt_strided(uint64_t* __restrict__ data, int log_n, int s, const uint64_t* __restrict__ tw,
                                                    uint64_t scale, const uint64_t* __restrict__ src, size_t src_len) {
    // src != NULL: out-of-place first pass of an RS encoding — the column is read from `src` (src_len elements per
    // column, zero beyond: the zero extension is never materialised) and the result is written to `data`
    constexpr int E = 1 << R;
    const size_t n = (size_t)1 << log_n;
    uint64_t* col = data + (size_t)blockIdx.y * n;
    const uint64_t* scol = src ? src + (size_t)blockIdx.y * src_len : nullptr;
    const int log_m = log_n - s;
    const int log_l = log_m - R;
    const size_t L = (size_t)1 << log_l;
    const size_t items = n >> R;
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t it = (size_t)blockIdx.x * NT + threadIdx.x; it < items; it += stride) {
        const size_t blk = it >> log_l, l = it & (L - 1);
        uint64_t* base = col + (blk << log_m) + l;
        uint64_t v[E];
        if (scol) {
            const size_t g0 = (blk << log_m) + l;
#pragma unroll
            for (int k = 0; k < E; k++) {
                const size_t gi = g0 + ((size_t)k << log_l);
                v[k] = gi < src_len ? scol[gi] : 0;
            }
        } else {
#pragma unroll
            for (int k = 0; k < E; k++) v[k] = base[(size_t)k << log_l];
        }
        radix_stages<R, INVERSE>(v, log_l, l, s, tw);
        if (INVERSE) {
            if (scale != 1) {
#pragma unroll
                for (int k = 0; k < E; k++) v[k] = mul(v[k], scale);
            }
        }
#pragma unroll
        for (int k = 0; k < E; k++) base[(size_t)k << log_l] = v[k];
    }
}

// last (forward) / first (inverse) `lb` stages of every contiguous 2^lb block, in LDS: groups of up to 4 stages run in
// registers (16 values per lane) between barriers — 3 LDS round trips for 12 stages instead of 12.  The LDS image is
// padded by one word every 16 so that the last group, whose lanes own 16 CONSECUTIVE elements (stride 128 B), does not
// pile every lane onto one bank.
__device__ __forceinline__ size_t lds_pad(size_t i) { return i + (i >> 4); }

template <int R, bool INVERSE>
__device__ __forceinline__ void local_group(uint64_t* sm, int lb, int sl, int s0, const uint64_t* __restrict__ tw) {
    constexpr int E = 1 << R;
    const int log_m = lb - sl, log_l = log_m - R;
    const size_t L = (size_t)1 << log_l, items = ((size_t)1 << lb) >> R;
    for (size_t it = threadIdx.x; it < items; it += NT) {
        const size_t blk = it >> log_l, l = it & (L - 1);
        const size_t b0 = (blk << log_m) + l;
        uint64_t v[E];
#pragma unroll
        for (int k = 0; k < E; k++) v[k] = sm[lds_pad(b0 + ((size_t)k << log_l))];
        radix_stages<R, INVERSE>(v, log_l, l, s0 + sl, tw);
#pragma unroll
        for (int k = 0; k < E; k++) sm[lds_pad(b0 + ((size_t)k << log_l))] = v[k];
    }
    __syncthreads();
}
template <bool INVERSE>
__device__ __forceinline__ void local_group_r(int R, uint64_t* sm, int lb, int sl, int s0, const uint64_t* __restrict__ tw) {
    switch (R) {
    case 1: local_group<1, INVERSE>(sm, lb, sl, s0, tw); break;
    case 2: local_group<2, INVERSE>(sm, lb, sl, s0, tw); break;
    case 3: local_group<3, INVERSE>(sm, lb, sl, s0, tw); break;
    default: local_group<4, INVERSE>(sm, lb, sl, s0, tw); break;
    }
}

template <bool INVERSE>
__global__ void __launch_bounds__(NT) k_ntt_local(uint64_t* __restrict__ data, int log_n, int lb, const uint64_t* __restrict__ tw, uint64_t scale) {
    __shared__ uint64_t sm[(1 << LOCAL_LOG) + (1 << (LOCAL_LOG - 4)) + 1];
    const size_t n = (size_t)1 << log_n;
    const size_t bsz = (size_t)1 << lb;
    const size_t n_blocks = n >> lb;
    uint64_t* col = data + (size_t)blockIdx.y * n;
    const int s0 = log_n - lb;  // first global stage handled here
    for (size_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        uint64_t* base = col + blk * bsz;
        for (size_t i = threadIdx.x; i < bsz; i += NT) sm[lds_pad(i)] = base[i];
        __syncthreads();
        if (!INVERSE) {
            for (int sl = 0; sl < lb;) {
                const int R = lb - sl >= 4 ? 4 : lb - sl;
                local_group_r<false>(R, sm, lb, sl, s0, tw);
                sl += R;
            }
        } else {  // mirrored: innermost stages first
            for (int hi = lb; hi > 0;) {
                const int R = hi >= 4 ? 4 : hi;
                hi -= R;
                local_group_r<true>(R, sm, lb, hi, s0, tw);
            }
        }
        for (size_t i = threadIdx.x; i < bsz; i += NT) base[i] = (INVERSE && scale != 1) ? mul(sm[lds_pad(i)], scale) : sm[lds_pad(i)];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(NT) k_pad_copy(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, size_t n_in, size_t n_out) {
    const uint64_t* src = in + (size_t)blockIdx.y * n_in;
    uint64_t* dst = out + (size_t)blockIdx.y * n_out;
    size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n_out; i += stride) dst[i] = i < n_in ? src[i] : 0;
}

template <bool INV>
static void launch_strided(int R, uint64_t* d, int log_n, int s, const uint64_t* tw, uint64_t scale, int n_cols, hipStream_t st,
                           const uint64_t* src = nullptr, size_t src_len = 0) {
    size_t items = ((size_t)1 << log_n) >> R;
    dim3 grid(grid_for(items, NT, 2048), (unsigned)n_cols);
    switch (R) {
    case 1: hipLaunchKernelGGL((k_ntt_strided<1, INV>), grid, dim3(NT), 0, st, d, log_n, s, tw, scale, src, src_len); break;
    case 2: hipLaunchKernelGGL((k_ntt_strided<2, INV>), grid, dim3(NT), 0, st, d, log_n, s, tw, scale, src, src_len); break;
    case 3: hipLaunchKernelGGL((k_ntt_strided<3, INV>), grid, dim3(NT), 0, st, d, log_n, s, tw, scale, src, src_len); break;
    case 4: hipLaunchKernelGGL((k_ntt_strided<4, INV>), grid, dim3(NT), 0, st, d, log_n, s, tw, scale, src, src_len); break;
    default: hipLaunchKernelGGL((k_ntt_strided<5, INV>), grid, dim3(NT), 0, st, d, log_n, s, tw, scale, src, src_len); break;
    }
}

// split `total` register-resident stages into the fewest passes of at most 5 stages, as evenly as possible
// (10 -> 5 + 5, not 4 + 4 + 2: every pass costs a full read + write of the data)
static int next_pass(int remaining) {
    const int passes = (remaining + 4) / 5;
    return (remaining + passes - 1) / passes;
}

static int ntt_impl(ceno_hip_ctx* ctx, uint64_t* d, int log_n, int n_cols, bool inverse, hipStream_t st, const uint64_t* src = nullptr,
                    size_t src_len = 0) {
    CHECK_ARG(ctx, d && log_n >= 0 && log_n <= 32 && n_cols >= 1 && n_cols <= 65535, "bad ntt arguments (log_n %d, cols %d)", log_n, n_cols);
    if (log_n == 0) return 0;
    const uint64_t* tw = nullptr;
    TRY(get_twiddles(ctx, log_n, inverse, st, &tw));
    const int lb = log_n < LOCAL_LOG ? log_n : LOCAL_LOG;
    const int n_strided = log_n - lb;  // stages done by strided passes
    const size_t n_blocks = ((size_t)1 << log_n) >> lb;
    dim3 lgrid((unsigned)(n_blocks < 1024 ? n_blocks : 1024), (unsigned)n_cols);
    if (!inverse) {
        int s = 0;
        if (src && n_strided == 0) {  // small transforms: no strided pass to fold the zero extension into
            hipLaunchKernelGGL(k_pad_copy, dim3(grid_for((size_t)1 << log_n, NT, 2048), (unsigned)n_cols), dim3(NT), 0, st, src, d, src_len,
                               (size_t)1 << log_n);
            src = nullptr;
        }
        while (s < n_strided) {
            const int R = next_pass(n_strided - s);
            launch_strided<false>(R, d, log_n, s, tw, 1, n_cols, st, s == 0 ? src : nullptr, src_len);
            s += R;
        }
        hipLaunchKernelGGL(k_ntt_local<false>, lgrid, dim3(NT), 0, st, d, log_n, lb, tw, (uint64_t)1);
    } else {
        const uint64_t n_inv = gl::inv(((uint64_t)1 << log_n) % gl::P);
        // mirrored order: local stages first, then strided groups from the innermost outwards
        hipLaunchKernelGGL(k_ntt_local<true>, lgrid, dim3(NT), 0, st, d, log_n, lb, tw, n_strided == 0 ? n_inv : (uint64_t)1);
        int s = n_strided;
        while (s > 0) {
            const int R = next_pass(s);
            s -= R;
            launch_strided<true>(R, d, log_n, s, tw, s == 0 ? n_inv : (uint64_t)1, n_cols, st);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}


extern "C" {

int ceno_hip_ntt_batch(ceno_hip_ctx* ctx, uint64_t* dev_cols, int log_n, int n_cols, int inverse, ceno_hip_stream s) {
    return ntt_impl(ctx, dev_cols, log_n, n_cols, inverse != 0, ctx_stream(ctx, s));
}

int ceno_hip_rs_encode(ceno_hip_ctx* ctx, const uint64_t* dev_cols, int log_n, int n_cols, int log_blowup, uint64_t* dev_codewords, ceno_hip_stream s) {
    CHECK_ARG(ctx, dev_cols && dev_codewords && log_blowup >= 0 && log_n >= 0 && log_n + log_blowup <= 32, "bad rs_encode arguments");
    CHECK_ARG(ctx, n_cols >= 1 && n_cols <= 65535, "bad column count");
    hipStream_t st = ctx_stream(ctx, s);
    // the zero extension is folded into the first pass (read `dev_cols`, write `dev_codewords`)
    return ntt_impl(ctx, dev_codewords, log_n + log_blowup, n_cols, false, st, dev_cols, (size_t)1 << log_n);
}

}  // extern "C"
