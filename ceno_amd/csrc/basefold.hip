// Basefold batch-open kernels for gfx950.
//
// Reference: `OpeningProver::open` -> `PCS::batch_open` (ceno_zkvm/src/scheme/hal.rs:284-294, cpu/mod.rs:1418-1457;
// the implementation is the EXT crate mpcs).  Protocol shape restated in-tree by the recursion verifier:
// ceno_recursion_v2/src/pcs/mod.rs:1111-1316 (transcript, degree-2 rounds), :7494-7720 (query phase),
// :7765-7781 (fold: lo=(a+b)/2, hi=(a-b) g^-bitrev(i)/2, lo + r(hi-lo)).  PARITY UNPINNED.
//
//   k_batch_cols    acc[i] (+)= sum_c coeff_c * col_c[i]      batch codewords / trace columns with the ext
//                   batch coefficients: reads every column once (8 B/row/col), 160-bit unreduced accumulators
//   k_fold_commit   one pass over the running codeword per commit round: hash every pair (the Merkle leaf of
//                   this round), fold it with the round challenge, add the codeword that joins at the next height
//   k_gather        query answers: rows / siblings / authentication paths gathered on device, one D2H copy
//   k_pow_grind     proof-of-work search, one Poseidon2 permutation per candidate
#include <algorithm>
#include <cstring>
#include <vector>

#include "merkle.hpp"

using namespace gl;

static constexpr int NT = 256;
static constexpr unsigned MAXB = 2048;
static constexpr uint64_t TWO_ADIC_GEN_2_32 = 1753635133440165772ULL;
static constexpr uint64_t INV2 = 0x7FFFFFFF80000001ULL;  // (p + 1) / 2

// ---- fold coefficients: T[j] = g_H^(-bitrev_{H-1}(j)) / 2.  The table of a smaller height is a prefix. ----

__global__ void __launch_bounds__(NT) k_fold_twiddles(uint64_t* t, size_t n, int bits, uint64_t g_inv) {
    size_t stride = (size_t)gridDim.x * NT;
    for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < n; j += stride) {
        const unsigned e = bits ? (__brev((unsigned)j) >> (32 - bits)) : 0u;
        t[j] = mul(gl::pow(g_inv, e), INV2);
    }
}

static int get_fold_twiddles(ceno_hip_ctx* ctx, int log_h, hipStream_t st, const uint64_t** out) {
    std::lock_guard<std::mutex> g(ctx->tw_mu);
    for (auto& kv : ctx->fold_twiddles)
        if (kv.first >= log_h) {  // any taller table serves as a prefix
            *out = kv.second;
            return 0;
        }
    const uint64_t gh = gl::pow(TWO_ADIC_GEN_2_32, (uint64_t)1 << (32 - log_h));
    const size_t n = (size_t)1 << (log_h - 1);
    void* p = nullptr;
    TRY(ctx_alloc(ctx, n * 8, &p));  // pool block: booked, released with the context
    hipLaunchKernelGGL(k_fold_twiddles, dim3(grid_for(n, NT, MAXB)), dim3(NT), 0, st, (uint64_t*)p, n, log_h - 1, gl::inv(gh));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));  // shared by every stream from now on: complete before the cache shows it (once per height)
    ctx->fold_twiddles[log_h] = (uint64_t*)p;
    *out = (uint64_t*)p;
    return 0;
}

// ---- batching ----
static constexpr int COEFF_CHUNK = 512;

__global__ void __launch_bounds__(NT) k_batch_cols(const uint64_t* __restrict__ cols, size_t len, int n_cols, const E2* __restrict__ coeffs,
                                                   E2* __restrict__ acc, int accumulate) {
    __shared__ E2 sc[COEFF_CHUNK];
    const size_t stride = (size_t)gridDim.x * NT;
    // all lanes walk the chunk loop together (the LDS refill is block-wide), rows beyond `len` just idle
    const size_t rounds = (len + stride - 1) / stride;
    for (size_t it = 0; it < rounds; it++) {
        const size_t i = it * stride + (size_t)blockIdx.x * NT + threadIdx.x;
        Acc5 a0{0, 0, 0, 0, 0}, a1{0, 0, 0, 0, 0};
        for (int c0 = 0; c0 < n_cols; c0 += COEFF_CHUNK) {
            const int nc = min(COEFF_CHUNK, n_cols - c0);
            __syncthreads();
            for (int k = threadIdx.x; k < nc; k += NT) sc[k] = coeffs[c0 + k];
            __syncthreads();
            if (i < len) {
                for (int k = 0; k < nc; k++) {
                    const uint64_t v = cols[(size_t)(c0 + k) * len + i];
                    acc5_add(a0, mul_wide(sc[k].c0, v));
                    acc5_add(a1, mul_wide(sc[k].c1, v));
                }
            }
        }
        if (i < len) {
            E2 r{acc5_reduce(a0), acc5_reduce(a1)};
            if (accumulate) r = r + acc[i];
            acc[i] = r;
        }
    }
}

// the same for MANY (columns, accumulator) jobs of different lengths in ONE launch (the opening batches ~60 trace matrices and ~12 codeword classes:
// one launch + one synchronisation each was 3 of its 8 ms): block b works on job blk[b].job as its blk[b].idx-th of jobs[job].n_blocks blocks
struct BatchJob {
    const uint64_t* cols;
    E2* acc;
    size_t len;
    uint32_t n_cols, coeff0, n_blocks, accumulate;
};
struct BatchBlk {
    uint32_t job, idx;
};
__global__ void __launch_bounds__(NT) k_batch_cols_multi(const BatchJob* __restrict__ jobs, const BatchBlk* __restrict__ blk, const E2* __restrict__ coeffs) {
    __shared__ E2 sc[COEFF_CHUNK];
    const BatchBlk B = blk[blockIdx.x];
    const BatchJob J = jobs[B.job];
    const uint64_t* __restrict__ cols = J.cols;
    const size_t len = J.len, stride = (size_t)J.n_blocks * NT;
    const size_t rounds = (len + stride - 1) / stride;
    for (size_t it = 0; it < rounds; it++) {
        const size_t i = it * stride + (size_t)B.idx * NT + threadIdx.x;
        Acc5 a0{0, 0, 0, 0, 0}, a1{0, 0, 0, 0, 0};
        for (uint32_t c0 = 0; c0 < J.n_cols; c0 += COEFF_CHUNK) {
            const int nc = (int)min((uint32_t)COEFF_CHUNK, J.n_cols - c0);
            __syncthreads();
            for (int k = threadIdx.x; k < nc; k += NT) sc[k] = coeffs[J.coeff0 + c0 + k];
            __syncthreads();
            if (i < len) {
                for (int k = 0; k < nc; k++) {
                    const uint64_t v = cols[(size_t)(c0 + k) * len + i];
                    acc5_add(a0, mul_wide(sc[k].c0, v));
                    acc5_add(a1, mul_wide(sc[k].c1, v));
                }
            }
        }
        if (i < len) {
            E2 r{acc5_reduce(a0), acc5_reduce(a1)};
            if (J.accumulate) r = r + J.acc[i];
            J.acc[i] = r;
        }
    }
}

// ---- commit-phase round ----
__global__ void __launch_bounds__(NT) k_fold_commit(const E2* __restrict__ cw, size_t n_pairs, E2 c, const E2* __restrict__ addend,
                                                    const uint64_t* __restrict__ tw, E2* __restrict__ out, uint64_t* __restrict__ digests,
                                                    const p2::Params* __restrict__ pp) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    const E2Pre cp = e2_pre(c);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < n_pairs; j += stride) {
        const E2 a = cw[2 * j], b = cw[2 * j + 1];
        uint64_t s[8] = {a.c0, a.c1, b.c0, b.c1, 0, 0, 0, 0};
        p2::permute(s, sp);
        *reinterpret_cast<ulonglong2*>(digests + 4 * j) = make_ulonglong2(s[0], s[1]);
        *reinterpret_cast<ulonglong2*>(digests + 4 * j + 2) = make_ulonglong2(s[2], s[3]);
        const E2 lo = e2_mul_base(a + b, INV2);
        const E2 hi = e2_mul_base(a - b, tw[j]);
        E2 v = lo + e2_mul_pre(cp, hi - lo);
        if (addend) v = v + addend[j];
        out[j] = v;
    }
}

// ---- one-round-ahead variant: fold the running codeword and hash the pairs of the RESULT ----
// The tree of the codeword a round commits to depends only on the PREVIOUS challenge, so it can be built while
// the current round's sumcheck message and transcript work are still in flight (host: two alternating streams).
__device__ __forceinline__ E2 fold_one(E2 a, E2 b, const E2Pre& cp, uint64_t tw) {
    const E2 lo = e2_mul_base(a + b, INV2);
    const E2 hi = e2_mul_base(a - b, tw);
    return lo + e2_mul_pre(cp, hi - lo);
}
// out[j] = fold(cw[2j], cw[2j+1]) (+ addend[j]); lane i produces out[2i], out[2i+1] and, when digests != NULL, their leaf
__global__ void __launch_bounds__(NT) k_fold_hash(const E2* __restrict__ cw, size_t n_out, E2 c, const E2* __restrict__ addend,
                                                  const uint64_t* __restrict__ tw, E2* __restrict__ out, uint64_t* __restrict__ digests,
                                                  const p2::Params* __restrict__ pp) {
    __shared__ p2::Params sp;
    if (digests) {
        for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
        __syncthreads();
    }
    const E2Pre cp = e2_pre(c);
    const size_t stride = (size_t)gridDim.x * NT;
    if (n_out == 1) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            E2 v = fold_one(cw[0], cw[1], cp, tw[0]);
            if (addend) v = v + addend[0];
            out[0] = v;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n_out / 2; i += stride) {
        E2 v0 = fold_one(cw[4 * i], cw[4 * i + 1], cp, tw[2 * i]);
        E2 v1 = fold_one(cw[4 * i + 2], cw[4 * i + 3], cp, tw[2 * i + 1]);
        if (addend) {
            v0 = v0 + addend[2 * i];
            v1 = v1 + addend[2 * i + 1];
        }
        out[2 * i] = v0;
        out[2 * i + 1] = v1;
        if (digests) {
            uint64_t s[8] = {v0.c0, v0.c1, v1.c0, v1.c1, 0, 0, 0, 0};
            p2::permute(s, sp);
            *reinterpret_cast<ulonglong2*>(digests + 4 * i) = make_ulonglong2(s[0], s[1]);
            *reinterpret_cast<ulonglong2*>(digests + 4 * i + 2) = make_ulonglong2(s[2], s[3]);
        }
    }
}
// leaf digests of the adjacent pairs of an ext codeword, 8 lanes per leaf (latency-bound sizes)
__global__ void __launch_bounds__(NT) k_hash_pairs8(const uint64_t* __restrict__ cw_words, size_t n_leaf, uint64_t* __restrict__ digests,
                                                    const p2::Params* __restrict__ pp) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    const size_t t = (size_t)blockIdx.x * NT + threadIdx.x, i = t >> 3;
    const int g = threadIdx.x & 7;
    const bool live = i < n_leaf;
    uint64_t x = (live && g < 4) ? cw_words[4 * i + g] : 0;  // [a.c0, a.c1, b.c0, b.c1, 0, 0, 0, 0]
    x = p2::permute_lanes8(x, sp);
    if (live && g < 4) digests[4 * i + g] = x;
}
// leaf digests, one lane per leaf (throughput-bound sizes)
__global__ void __launch_bounds__(NT) k_hash_pairs(const E2* __restrict__ cw, size_t n_leaf, uint64_t* __restrict__ digests,
                                                   const p2::Params* __restrict__ pp) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < n_leaf; j += stride) {
        const E2 a = cw[2 * j], b = cw[2 * j + 1];
        uint64_t s[8] = {a.c0, a.c1, b.c0, b.c1, 0, 0, 0, 0};
        p2::permute(s, sp);
        *reinterpret_cast<ulonglong2*>(digests + 4 * j) = make_ulonglong2(s[0], s[1]);
        *reinterpret_cast<ulonglong2*>(digests + 4 * j + 2) = make_ulonglong2(s[2], s[3]);
    }
}

// ---- gathers for the query phase ----
// out[(q * n_cols + c) * elem_words + e] = src[c * col_stride + idx[q] * elem_words + e]
__global__ void __launch_bounds__(NT) k_gather(const uint64_t* __restrict__ src, size_t col_stride, int n_cols, int elem_words,
                                               const uint64_t* __restrict__ idx, size_t n_q, int shift, int flip, uint64_t* __restrict__ out) {
    const size_t per_q = (size_t)n_cols * elem_words, total = n_q * per_q;
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t t = (size_t)blockIdx.x * NT + threadIdx.x; t < total; t += stride) {
        const size_t q = t / per_q, rem = t % per_q, c = rem / elem_words, e = rem % elem_words;
        const size_t i = (idx[q] >> shift) ^ (size_t)flip;
        out[t] = src[c * col_stride + i * elem_words + e];
    }
}
// authentication paths: out[(q * depth + l) * 4 + k] = levels[l][((idx[q] >> shift) >> l) ^ 1][k]
struct LevelPtrs {
    const uint64_t* p[63];  // by value in the kernel arguments: no pointer table to upload per tree
};
__global__ void __launch_bounds__(NT) k_gather_paths(LevelPtrs lv, int depth, const uint64_t* __restrict__ idx, size_t n_q, int shift,
                                                     uint64_t* __restrict__ out, size_t out_stride) {
    const size_t total = n_q * (size_t)depth * 4;
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t t = (size_t)blockIdx.x * NT + threadIdx.x; t < total; t += stride) {
        const size_t q = t / ((size_t)depth * 4), rem = t % ((size_t)depth * 4);
        const int l = (int)(rem / 4), k = (int)(rem % 4);
        const size_t node = ((idx[q] >> shift) >> l) ^ 1;
        out[q * out_stride + rem] = lv.p[l][4 * node + k];
    }
}
// one commit round of the query phase: round r answers query q with the sibling of its codeword entry (idx >> r) and the path of
// the leaf (idx >> (r + 1)) in that round's tree
struct QRound {
    static constexpr int MAXL = 40;
    const uint64_t* cw;      // the round's running codeword (ext elements)
    uint64_t sib_off;        // word offsets inside the output: [n_q x 2 sibling][n_q x 4 depth path]
    uint64_t path_off;
    int depth, shift;
    const uint64_t* levels[MAXL];
};
__global__ void __launch_bounds__(NT) k_query_rounds(const QRound* __restrict__ rounds, const uint64_t* __restrict__ idx, size_t n_q,
                                                     uint64_t* __restrict__ out) {
    const QRound& R = rounds[blockIdx.y];  // wave-uniform: read through the scalar cache
    const size_t per_q = 2 + 4 * (size_t)R.depth, total = n_q * per_q;
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t t = (size_t)blockIdx.x * NT + threadIdx.x; t < total; t += stride) {
        const size_t q = t / per_q;
        int rem = (int)(t % per_q);
        const uint64_t pos = idx[q] >> R.shift;
        if (rem < 2) {
            out[R.sib_off + 2 * q + rem] = R.cw[2 * (pos ^ 1) + rem];
        } else {
            rem -= 2;
            const int l = rem >> 2, k = rem & 3;
            const uint64_t node = ((pos >> 1) >> l) ^ 1;
            out[R.path_off + q * 4 * (size_t)R.depth + rem] = R.levels[l][4 * node + k];
        }
    }
}
int merkle_gather_paths(ceno_hip_ctx* ctx, ceno_hip_merkle* t, const uint64_t* dev_indices, size_t n, int shift, uint64_t* dev_out,
                        size_t out_stride_words, hipStream_t st) {
    CHECK_ARG(ctx, t->log_rows <= 62, "tree too tall");
    TRY(merkle_ensure_top(ctx, t));  // the host half of the tree, if nobody has asked for the root yet
    LevelPtrs lv{};
    for (int l = 0; l < t->log_rows; l++) lv.p[l] = t->levels[l];
    hipLaunchKernelGGL(k_gather_paths, dim3(grid_for(n * t->log_rows * 4, NT, MAXB)), dim3(NT), 0, st, lv, t->log_rows, dev_indices, n, shift, dev_out,
                       out_stride_words);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ---- proof of work (p3-challenger GrindingChallenger::grind / check_witness, ceno_recursion_v2/src/pcs/mod.rs:8125-8155):
// a clone of the duplex challenger observes the candidate and samples ONE base element; the candidate is a witness when the
// low `bits` bits of that sample are zero.  Whatever the number of pending inputs (< RATE) this is exactly one permutation:
// the pending inputs and then the candidate overwrite state[0 .. pos], permute, and the sample is the BACK of the fresh
// output buffer = state[RATE - 1].  The state arrives with the pending inputs already written; least witness >= base wins.
struct DuplexState {
    uint64_t s[8];
};
__global__ void __launch_bounds__(NT) k_pow_grind(DuplexState st0, int pos, uint64_t base, uint64_t count, uint64_t mask,
                                                  const p2::Params* __restrict__ pp, unsigned long long* __restrict__ best) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * NT;
    for (uint64_t t = (uint64_t)blockIdx.x * NT + threadIdx.x; t < count; t += stride) {
        const uint64_t w = base + t;
        uint64_t s[8];
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = (k == pos) ? w : st0.s[k];
        p2::permute(s, sp);
        if ((s[p2::RATE - 1] & mask) == 0) atomicMin(best, (unsigned long long)w);
    }
}

extern "C" {

int ceno_hip_batch_columns(ceno_hip_ctx* ctx, const uint64_t* dev_cols, size_t len, int n_cols, const uint64_t* coeffs_ext, uint64_t* dev_acc_ext,
                           int accumulate, ceno_hip_stream s) {
    CHECK_ARG(ctx, dev_cols && coeffs_ext && dev_acc_ext && len >= 1 && n_cols >= 1, "bad batch_columns arguments");
    for (int i = 0; i < 2 * n_cols; i++) CHECK_ARG(ctx, coeffs_ext[i] < gl::P, "batch coefficient word %d is not canonical", i);
    hipStream_t st = ctx_stream(ctx, s);
    void* d_coeff = nullptr;
    TRY(ctx_alloc(ctx, (size_t)n_cols * 16, &d_coeff));
    hipError_t e = hipMemcpyAsync(d_coeff, coeffs_ext, (size_t)n_cols * 16, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_batch_cols, dim3(grid_for(len, NT, MAXB)), dim3(NT), 0, st, dev_cols, len, n_cols, (const E2*)d_coeff, (E2*)dev_acc_ext,
                           accumulate);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the coefficient buffer goes back to the pool
    ctx_free(ctx, d_coeff);
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "batch_columns: %s", hipGetErrorString(e));
    return 0;
}

int ceno_hip_batch_columns_multi(ceno_hip_ctx* ctx, int n_jobs, const uint64_t* const* dev_cols, const size_t* lens, const int* n_cols,
                                 const uint64_t* coeffs_ext, uint64_t* const* dev_acc_ext, const int* accumulate, ceno_hip_stream s) {
    CHECK_ARG(ctx, n_jobs >= 1 && dev_cols && lens && n_cols && coeffs_ext && dev_acc_ext, "bad batch_columns_multi arguments");
    size_t total_cols = 0;
    for (int j = 0; j < n_jobs; j++) {
        CHECK_ARG(ctx, dev_cols[j] && dev_acc_ext[j] && lens[j] >= 1 && n_cols[j] >= 1, "batch_columns_multi: empty job %d", j);
        total_cols += (size_t)n_cols[j];
    }
    for (size_t i = 0; i < 2 * total_cols; i++) CHECK_ARG(ctx, coeffs_ext[i] < gl::P, "batch coefficient word %zu is not canonical", i);
    // jobs that ADD to an accumulator run after the job that wrote it: wave w = how many earlier jobs name the same accumulator
    std::vector<int> wave((size_t)n_jobs, 0);
    int n_waves = 1;
    for (int j = 0; j < n_jobs; j++) {
        for (int i = 0; i < j; i++)
            if (dev_acc_ext[i] == dev_acc_ext[j]) wave[(size_t)j]++;
        CHECK_ARG(ctx, (wave[(size_t)j] == 0) == !(accumulate && accumulate[j]), "batch_columns_multi: job %d: the first job of an accumulator writes it, later ones add", j);
        n_waves = std::max(n_waves, wave[(size_t)j] + 1);
    }
    std::vector<BatchJob> jobs((size_t)n_jobs);
    std::vector<std::vector<BatchBlk>> blks((size_t)n_waves);
    size_t c0 = 0;
    for (int j = 0; j < n_jobs; j++) {
        const uint32_t nb = (uint32_t)std::min<size_t>((lens[j] + NT - 1) / NT, MAXB);
        jobs[(size_t)j] = BatchJob{dev_cols[j], (E2*)dev_acc_ext[j], lens[j], (uint32_t)n_cols[j], (uint32_t)c0, nb, (uint32_t)(wave[(size_t)j] > 0)};
        for (uint32_t b = 0; b < nb; b++) blks[(size_t)wave[(size_t)j]].push_back(BatchBlk{(uint32_t)j, b});
        c0 += (size_t)n_cols[j];
    }
    size_t n_blk = 0;
    for (auto& w : blks) n_blk += w.size();
    // one upload: [jobs][blocks of every wave][coefficients]
    const size_t off_blk = (sizeof(BatchJob) * (size_t)n_jobs + 15) & ~(size_t)15, off_co = (off_blk + sizeof(BatchBlk) * n_blk + 15) & ~(size_t)15;
    const size_t bytes = off_co + total_cols * 16;
    std::vector<unsigned char> host(bytes);
    memcpy(host.data(), jobs.data(), sizeof(BatchJob) * (size_t)n_jobs);
    {
        size_t o = off_blk;
        for (auto& w : blks) {
            memcpy(host.data() + o, w.data(), sizeof(BatchBlk) * w.size());
            o += sizeof(BatchBlk) * w.size();
        }
    }
    memcpy(host.data() + off_co, coeffs_ext, total_cols * 16);
    hipStream_t st = ctx_stream(ctx, s);
    void* d = nullptr;
    TRY(ctx_alloc(ctx, bytes, &d));
    hipError_t e = hipMemcpyAsync(d, host.data(), bytes, hipMemcpyHostToDevice, st);  // (pageable source: the copy is staged before the call returns)
    size_t o = off_blk;
    for (auto& w : blks) {
        if (e == hipSuccess && !w.empty()) {
            hipLaunchKernelGGL(k_batch_cols_multi, dim3((unsigned)w.size()), dim3(NT), 0, st, (const BatchJob*)d, (const BatchBlk*)((const char*)d + o),
                               (const E2*)((const char*)d + off_co));
            e = hipGetLastError();
        }
        o += sizeof(BatchBlk) * w.size();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the job table goes back to the pool (ONE synchronisation for all the jobs)
    ctx_free(ctx, d);
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "batch_columns_multi: %s", hipGetErrorString(e));
    return 0;
}

// out[i] = sum_b in[b * len + i] (mod p), extension elements: the modular all-reduce of per-rank partial batchings (no collective library
// has a mod-p reduction; the blocks arrive by an all-gather)
__global__ void __launch_bounds__(NT) k_ext_sum_blocks(const E2* __restrict__ in, int n_blocks, size_t len, E2* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < len; i += stride) {
        E2 acc = in[i];
        for (int b = 1; b < n_blocks; b++) acc = acc + in[(size_t)b * len + i];
        out[i] = acc;
    }
}
int ceno_hip_ext_sum_blocks(ceno_hip_ctx* ctx, const uint64_t* dev_in_ext, int n_blocks, size_t len, uint64_t* dev_out_ext, ceno_hip_stream s) {
    CHECK_ARG(ctx, dev_in_ext && dev_out_ext && n_blocks >= 1 && len >= 1, "bad ext_sum_blocks arguments");
    hipStream_t st = ctx_stream(ctx, s);
    hipLaunchKernelGGL(k_ext_sum_blocks, dim3(grid_for(len, NT, MAXB)), dim3(NT), 0, st, (const E2*)dev_in_ext, n_blocks, len, (E2*)dev_out_ext);
    if (hipGetLastError() != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "ext_sum_blocks: launch failed");
    return 0;
}

int ceno_hip_basefold_fold_commit(ceno_hip_ctx* ctx, const uint64_t* dev_codeword_ext, int log_h, const uint64_t* challenge2,
                                  const uint64_t* dev_addend_ext, uint64_t* dev_out_ext, ceno_hip_stream s, ceno_hip_merkle** out_tree) {
    CHECK_ARG(ctx, dev_codeword_ext && challenge2 && dev_out_ext && out_tree && log_h >= 1 && log_h <= 32, "bad fold_commit arguments");
    CHECK_ARG(ctx, challenge2[0] < gl::P && challenge2[1] < gl::P, "challenge is not canonical");
    hipStream_t st = ctx_stream(ctx, s);
    const p2::Params* pp;
    TRY(get_params(ctx, &pp));
    const uint64_t* tw = nullptr;
    TRY(get_fold_twiddles(ctx, log_h, st, &tw));
    ceno_hip_merkle* t = nullptr;
    TRY(merkle_alloc(ctx, log_h - 1, &t));
    const size_t n_pairs = (size_t)1 << (log_h - 1);
    hipLaunchKernelGGL(k_fold_commit, dim3(grid_for(n_pairs, NT, MAXB)), dim3(NT), 0, st, (const E2*)dev_codeword_ext, n_pairs,
                       E2{challenge2[0], challenge2[1]}, (const E2*)dev_addend_ext, tw, (E2*)dev_out_ext, t->levels[0], pp);
    int rc = merkle_build_upper(ctx, t, st);
    if (rc) {
        merkle_release(ctx, t);
        return rc;
    }
    *out_tree = t;
    return 0;
}

static constexpr size_t LANES8_MAX_LEAVES = (size_t)1 << 14;

int ceno_hip_basefold_commit_codeword(ceno_hip_ctx* ctx, const uint64_t* dev_codeword_ext, int log_h, ceno_hip_stream s, ceno_hip_merkle** out_tree) {
    CHECK_ARG(ctx, dev_codeword_ext && out_tree && log_h >= 1 && log_h <= 32, "bad commit_codeword arguments");
    hipStream_t st = ctx_stream(ctx, s);
    const p2::Params* pp;
    TRY(get_params(ctx, &pp));
    ceno_hip_merkle* t = nullptr;
    TRY(merkle_alloc(ctx, log_h - 1, &t));
    const size_t n_leaf = (size_t)1 << (log_h - 1);
    if (n_leaf <= LANES8_MAX_LEAVES)
        hipLaunchKernelGGL(k_hash_pairs8, dim3((unsigned)((n_leaf * 8 + NT - 1) / NT)), dim3(NT), 0, st, dev_codeword_ext, n_leaf, t->levels[0], pp);
    else
        hipLaunchKernelGGL(k_hash_pairs, dim3(grid_for(n_leaf, NT, MAXB)), dim3(NT), 0, st, (const E2*)dev_codeword_ext, n_leaf, t->levels[0], pp);
    int rc = merkle_build_upper(ctx, t, st);
    if (rc) {
        merkle_release(ctx, t);
        return rc;
    }
    *out_tree = t;
    return 0;
}

int ceno_hip_basefold_fold(ceno_hip_ctx* ctx, const uint64_t* dev_codeword_ext, int log_h, const uint64_t* challenge2,
                           const uint64_t* dev_addend_ext, uint64_t* dev_out_ext, ceno_hip_stream s) {
    CHECK_ARG(ctx, dev_codeword_ext && challenge2 && dev_out_ext && log_h >= 1 && log_h <= 32, "bad basefold_fold arguments");
    CHECK_ARG(ctx, challenge2[0] < gl::P && challenge2[1] < gl::P, "challenge is not canonical");
    hipStream_t st = ctx_stream(ctx, s);
    const uint64_t* tw = nullptr;
    TRY(get_fold_twiddles(ctx, log_h, st, &tw));
    const size_t n_out = (size_t)1 << (log_h - 1);
    hipLaunchKernelGGL(k_fold_hash, dim3(grid_for(std::max<size_t>(n_out / 2, 1), NT, MAXB)), dim3(NT), 0, st, (const E2*)dev_codeword_ext, n_out,
                       E2{challenge2[0], challenge2[1]}, (const E2*)dev_addend_ext, tw, (E2*)dev_out_ext, (uint64_t*)nullptr,
                       (const p2::Params*)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int ceno_hip_gather(ceno_hip_ctx* ctx, const uint64_t* dev_src, size_t col_stride_words, int n_cols, int elem_words, const uint64_t* dev_indices,
                    size_t n, int shift, int flip_low_bit, uint64_t* dev_out, ceno_hip_stream s) {
    CHECK_ARG(ctx, dev_src && dev_indices && dev_out && n_cols >= 1 && elem_words >= 1 && shift >= 0 && shift < 64, "bad gather arguments");
    if (n == 0) return 0;
    hipStream_t st = ctx_stream(ctx, s);
    hipLaunchKernelGGL(k_gather, dim3(grid_for(n * n_cols * elem_words, NT, MAXB)), dim3(NT), 0, st, dev_src, col_stride_words, n_cols, elem_words,
                       dev_indices, n, shift, flip_low_bit ? 1 : 0, dev_out);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int ceno_hip_merkle_open_batch(ceno_hip_ctx* ctx, ceno_hip_merkle* t, const uint64_t* dev_indices, size_t n, int shift, uint64_t* dev_out,
                               ceno_hip_stream s) {
    CHECK_ARG(ctx, t && dev_indices && dev_out && shift >= 0 && shift < 64, "bad merkle_open_batch arguments");
    if (n == 0 || t->log_rows == 0) return 0;
    hipStream_t st = ctx_stream(ctx, s);
    return merkle_gather_paths(ctx, t, dev_indices, n, shift, dev_out, 4 * (size_t)t->log_rows, st);
}

// every commit round of the query phase in ONE launch (a gather of the sibling values and one of the Merkle paths per round were
// 2 x 20 dependent launches of ~2 us of work each: ~0.3 ms of launch gaps in an opening of 2^20 rows)
int ceno_hip_basefold_query_rounds(ceno_hip_ctx* ctx, const uint64_t* const* dev_codewords_ext, ceno_hip_merkle* const* trees, int n_rounds,
                                   const uint64_t* dev_indices, size_t n_q, uint64_t* dev_out, ceno_hip_stream s) {
    CHECK_ARG(ctx, dev_codewords_ext && trees && dev_indices && dev_out && n_rounds >= 0 && n_rounds < 64, "bad query_rounds arguments");
    if (n_rounds == 0 || n_q == 0) return 0;
    hipStream_t st = ctx_stream(ctx, s);
    void *h = nullptr, *d = nullptr;
    TRY(ctx_pinned_alloc(ctx, (size_t)n_rounds * sizeof(QRound), &h, &d));
    QRound* tab = (QRound*)h;
    size_t off = 0, widest = 0;
    for (int r = 0; r < n_rounds; r++) {
        ceno_hip_merkle* t = trees[r];
        if (!t || !dev_codewords_ext[r] || t->log_rows > QRound::MAXL) {
            ctx_pinned_free(ctx, h);
            return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "query_rounds: bad round %d", r);
        }
        int rc = merkle_ensure_top(ctx, t);  // the host half of the tree, if nobody has asked for the root yet
        if (rc) {
            ctx_pinned_free(ctx, h);
            return rc;
        }
        QRound& R = tab[r];
        R.cw = dev_codewords_ext[r];
        R.depth = t->log_rows;
        R.shift = r;
        R.sib_off = off;
        off += n_q * 2;
        R.path_off = off;
        off += n_q * 4 * (size_t)t->log_rows;
        for (int l = 0; l < t->log_rows; l++) R.levels[l] = t->levels[l];
        widest = std::max(widest, n_q * (2 + 4 * (size_t)t->log_rows));
    }
    hipLaunchKernelGGL(k_query_rounds, dim3(grid_for(widest, NT, 64), (unsigned)n_rounds), dim3(NT), 0, st, (const QRound*)d, dev_indices, n_q, dev_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // the table is read through its host mapping: it goes back to the pool once the kernel is done
    ctx_pinned_free(ctx, h);
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "query_rounds: %s", hipGetErrorString(e));
    return 0;
}

int ceno_hip_pow_grind_duplex(ceno_hip_ctx* ctx, const uint64_t* state16, int bits, uint64_t* out_witness, ceno_hip_stream s) {
    CHECK_ARG(ctx, state16 && out_witness && bits >= 0 && bits <= 40, "bad pow_grind arguments");
    const int pos = (int)state16[8];
    CHECK_ARG(ctx, state16[8] < (uint64_t)p2::RATE && state16[13] <= (uint64_t)p2::RATE, "bad duplex state (pending inputs %llu, outputs %llu)",
              (unsigned long long)state16[8], (unsigned long long)state16[13]);
    DuplexState st0;
    for (int k = 0; k < 8; k++) st0.s[k] = (k < pos) ? state16[9 + k] : state16[k];  // pending inputs overwrite the front of the state
    for (int k = 0; k < 8; k++) CHECK_ARG(ctx, st0.s[k] < gl::P, "duplex state word %d is not canonical", k);
    if (bits == 0) {
        *out_witness = 0;
        return 0;
    }
    hipStream_t st = ctx_stream(ctx, s);
    const p2::Params* pp;
    TRY(get_params(ctx, &pp));
    void* d = nullptr;
    TRY(ctx_alloc(ctx, 8, &d));
    const uint64_t mask = ((uint64_t)1 << bits) - 1;
    // expected 2^bits candidates: search windows of 4 * 2^bits until one holds a witness (the least one wins)
    const uint64_t window = (uint64_t)4 << (bits < 16 ? 16 : bits);
    int rc = 0;
    for (uint64_t base = 0;; base += window) {
        unsigned long long best = ~0ull;
        hipError_t e = hipMemcpyAsync(d, &best, 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_pow_grind, dim3(grid_for(window, NT, MAXB)), dim3(NT), 0, st, st0, pos, base, window, mask, pp,
                               (unsigned long long*)d);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&best, d, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            rc = ctx_fail(ctx, CENO_HIP_ERR_HIP, "pow_grind: %s", hipGetErrorString(e));
            break;
        }
        if (best != ~0ull) {
            *out_witness = best;
            break;
        }
        if (base > ((uint64_t)1 << 60)) {
            rc = ctx_fail(ctx, CENO_HIP_ERR_STATE, "pow_grind: no witness found");
            break;
        }
    }
    ctx_free(ctx, d);
    return rc;
}

}  // extern "C"
