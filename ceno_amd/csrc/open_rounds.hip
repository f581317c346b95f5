// The sumcheck of a Basefold batch opening's commit phase, every live matrix of a round in ONE launch.
//
// Reference: `PCS::batch_open` (ceno_zkvm/src/scheme/cpu/mod.rs:1418-1457 -> EXT crate mpcs); protocol shape restated in-tree by the recursion
// verifier, ceno_recursion_v2/src/pcs/mod.rs:1111-1316: the claim is sum_m sum_x eq(x, point_m) * F_m(x) with F_m the matrix's columns batched by
// the powers of the batch challenge, a matrix of fewer variables joining when the running codeword reaches its height (suffix alignment), messages of
// degree 2 sent as (p(1), p(2)).  PARITY UNPINNED.
//
// The generic prover runs this as one handle per height group (ceno_hip_sumcheck_begin / _round_dev): a shard's commitment has ~10 groups, so a
// round was ~8 slot-table blits + 8 launches of ~12 us queued by ONE host thread — ~135 us of a 265 us round (profiles/r06_open_rounds_before.txt).
// Here the tables of ALL matrices sit behind one job table written once at begin; a round is one launch over (live matrix x pair) items that folds
// with the previous challenge and accumulates the two message points, the last workgroup to arrive adds the partial sums and stores the message into
// pinned host memory the host watches (no D2H blit, no stream synchronisation).
//
//   k_open_round   HBM-bound at the top (round 1 reads 64 B + 64 B and writes 64 B per pair and matrix), a latency chain below 2^12 pairs
//   k_open_finish  the last fold: F_m at the challenges, one lane per matrix
#include <algorithm>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "reduce.hpp"

using namespace gl;

namespace {

constexpr int NT = 256;
constexpr unsigned MAXB = 1024;
constexpr uint64_t MSG_INVALID = ~0ull;  // >= p: never a field element (the host arms the message words with it)

struct OpenJob {
    const E2* eq_in;  // the tables this round reads: the matrix's own (joining round and the round after) or the previous fold's
    const E2* f_in;
    E2* eq_out;       // where this round's fold goes (NULL in the joining round: nothing to fold yet)
    E2* f_out;
};

__device__ __forceinline__ void st_agent(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ld_agent(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// item = (job, pair); all jobs of a round have 2^log_pairs pairs.  p(1) += eq(1) f(1), p(2) += (2 eq(1) - eq(0)) (2 f(1) - f(0))
__global__ void __launch_bounds__(NT) k_open_round(const OpenJob* __restrict__ jobs, int n_jobs, int log_pairs, E2 r, uint64_t* __restrict__ partials,
                                                   unsigned* __restrict__ counter, uint64_t* __restrict__ out_msg /* pinned host, 4 words */) {
    __shared__ E2 smem[(NT / 64) * 2];
    __shared__ int s_last;
    const E2Pre rp = e2_pre(r);
    const size_t items = (size_t)n_jobs << log_pairs, mask = ((size_t)1 << log_pairs) - 1, stride = (size_t)gridDim.x * NT;
    E2 acc[2] = {e2_zero(), e2_zero()};
    for (size_t it = (size_t)blockIdx.x * NT + threadIdx.x; it < items; it += stride) {
        const OpenJob J = jobs[it >> log_pairs];
        const size_t p = it & mask;
        E2 e0, e1, f0, f1;
        if (J.eq_out) {
            const E2 *qe = J.eq_in + 4 * p, *qf = J.f_in + 4 * p;
            const E2 a0 = qe[0], a1 = qe[1], a2 = qe[2], a3 = qe[3];
            const E2 b0 = qf[0], b1 = qf[1], b2 = qf[2], b3 = qf[3];
            e0 = e2_fma_pre(rp, a1 - a0, a0);
            e1 = e2_fma_pre(rp, a3 - a2, a2);
            f0 = e2_fma_pre(rp, b1 - b0, b0);
            f1 = e2_fma_pre(rp, b3 - b2, b2);
            J.eq_out[2 * p] = e0;
            J.eq_out[2 * p + 1] = e1;
            J.f_out[2 * p] = f0;
            J.f_out[2 * p + 1] = f1;
        } else {
            e0 = J.eq_in[2 * p];
            e1 = J.eq_in[2 * p + 1];
            f0 = J.f_in[2 * p];
            f1 = J.f_in[2 * p + 1];
        }
        acc[0] = acc[0] + e1 * f1;
        acc[1] = acc[1] + (e1 + e1 - e0) * (f1 + f1 - f0);
    }
    red::block_sum<2, NT>(acc, smem);
    auto publish = [&](const E2 (&t)[2]) {
        // two 16-byte write-through system-scope stores: each lands atomically per 8-byte word, the host takes the message once all four have changed
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const u4 w = {(unsigned)t[k].c0, (unsigned)(t[k].c0 >> 32), (unsigned)t[k].c1, (unsigned)(t[k].c1 >> 32)};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out_msg + 2 * k), "v"(w) : "memory");
        }
    };
    if (gridDim.x == 1) {
        if (threadIdx.x == 0) publish(acc);
        return;
    }
    // "last block done" (the recipe of sumcheck_dev.hpp's epilogue: per-XCD L2s are not coherent — write-through partial stores drained before the
    // agent-scope counter add, acquire fence + agent-scope loads in the last block, no release fence)
    if (threadIdx.x == 0) {
        uint64_t* row = partials + (size_t)blockIdx.x * 4;
        st_agent(row, acc[0].c0);
        st_agent(row + 1, acc[0].c1);
        st_agent(row + 2, acc[1].c0);
        st_agent(row + 3, acc[1].c1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (prev == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    E2 tot[2] = {e2_zero(), e2_zero()};
    for (unsigned b = threadIdx.x; b < gridDim.x; b += NT) {
        const uint64_t* row = partials + (size_t)b * 4;
        tot[0] = tot[0] + E2{ld_agent(row), ld_agent(row + 1)};
        tot[1] = tot[1] + E2{ld_agent(row + 2), ld_agent(row + 3)};
    }
    __syncthreads();  // smem is reused
    red::block_sum<2, NT>(tot, smem);
    if (threadIdx.x == 0) {
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        publish(tot);
    }
}

// F_m(r_0 .. r_{nv_m - 1}): the last fold of every matrix (a matrix of zero variables is its single entry)
__global__ void __launch_bounds__(NT) k_open_finish(const OpenJob* __restrict__ jobs, int n_mats, E2 r, uint64_t* __restrict__ out /* pinned host, 2 words per matrix */) {
    const int m = blockIdx.x * NT + threadIdx.x;
    if (m >= n_mats) return;
    const OpenJob J = jobs[m];
    E2 v = J.f_in[0];
    if (J.f_out) v = e2_fma_pre(e2_pre(r), J.f_in[1] - v, v);  // (f_out only marks "has a variable left to bind")
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 w = {(unsigned)v.c0, (unsigned)(v.c0 >> 32), (unsigned)v.c1, (unsigned)(v.c1 >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out + 2 * m), "v"(w) : "memory");
}

}  // namespace

struct ceno_hip_open_rounds {
    ceno_hip_ctx* ctx = nullptr;
    hipStream_t st = nullptr;
    int n_mats = 0, n = 0;       // n = the most variables of any matrix = the number of rounds
    int round = 0;               // the next round
    std::vector<int> nv;         // per matrix
    std::vector<int> order;      // matrices by falling number of variables: the live ones of a round are a prefix
    std::vector<int> first_job;  // [n + 2]: where round r's jobs start in the table (round n = the finish jobs, in the CALLER's order)
    std::vector<int> n_live;     // [n]
    void* arena = nullptr;       // fold buffers of every matrix + partial sums + [job table, counter] (one upload at begin)
    void* h_pin = nullptr;       // pinned: [job table][counter: 64 zero bytes][message 4 words][finals 2 words per matrix]
    void* d_pin = nullptr;
    size_t msg_off = 0, fin_off = 0;
    uint64_t* partials = nullptr;
    unsigned* counter = nullptr;
    const void* d_jobs = nullptr;     // OpenJob[]: the table in device memory (a kernel that walks it through the host mapping pays a PCIe fetch per round)
};

namespace {

// wait until `n_words` pinned words (armed with MSG_INVALID) have all been written by the device
int wait_words(ceno_hip_open_rounds* h, const uint64_t* words, size_t n_words, const char* what) {
    unsigned long long spins = 0;
    auto all_there = [&]() {
        size_t k = 0;
        while (k < n_words && __atomic_load_n(&words[k], __ATOMIC_ACQUIRE) != MSG_INVALID) k++;
        return k == n_words;
    };
    for (;;) {
        if (all_there()) return 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        __builtin_ia32_pause();
#endif
        if ((++spins & 0xFFFFF) == 0) {  // every ~1M polls make sure the stream is still alive (a faulted kernel never writes its message)
            hipError_t q = hipStreamQuery(h->st);
            if (q != hipSuccess && q != hipErrorNotReady) return ctx_fail(h->ctx, CENO_HIP_ERR_HIP, "open_rounds: the kernel failed: %s", hipGetErrorString(q));
            if (q == hipSuccess) {
                if (all_there()) return 0;
                return ctx_fail(h->ctx, CENO_HIP_ERR_HIP, "open_rounds: round %d of %d finished without publishing %s", h->round, h->n, what);
            }
        }
    }
}

}  // namespace

extern "C" {

int ceno_hip_open_rounds_begin(ceno_hip_ctx* ctx, int n_mats, const uint64_t* const* dev_eq_ext, const uint64_t* const* dev_f_ext, const int* num_vars,
                               ceno_hip_stream s, ceno_hip_open_rounds** out) {
    CHECK_ARG(ctx, n_mats >= 1 && n_mats <= (1 << 16) && dev_eq_ext && dev_f_ext && num_vars && out, "bad open_rounds_begin arguments");
    int n = 0;
    for (int m = 0; m < n_mats; m++) {
        CHECK_ARG(ctx, dev_eq_ext[m] && dev_f_ext[m] && num_vars[m] >= 0 && num_vars[m] <= 40, "open_rounds_begin: bad matrix %d", m);
        n = std::max(n, num_vars[m]);
    }
    hipStream_t st = ctx_stream(ctx, s);
    auto* h = new ceno_hip_open_rounds();
    h->ctx = ctx;
    h->st = st;
    h->n_mats = n_mats;
    h->n = n;
    h->nv.assign(num_vars, num_vars + n_mats);
    h->order.resize(n_mats);
    for (int m = 0; m < n_mats; m++) h->order[m] = m;
    std::stable_sort(h->order.begin(), h->order.end(), [&](int a, int b) { return num_vars[a] > num_vars[b]; });
    // fold buffers: a matrix of v variables joins in round n - v (read in place), is folded into A (2^(v-1) entries) in the next round, into
    // B (2^(v-2)) in the one after, then A, B, ... — two tables (eq, F) each
    std::vector<size_t> off_a(n_mats), off_b(n_mats);
    size_t words = 0;
    for (int m = 0; m < n_mats; m++) {
        const int v = num_vars[m];
        off_a[m] = words;
        words += v >= 2 ? (size_t)4 << (v - 1) : 0;  // eq + F, 2 words per entry
        off_b[m] = words;
        words += v >= 3 ? (size_t)4 << (v - 2) : 0;
    }
    const size_t part_off = words;
    words += (size_t)MAXB * 4;
    // (the job table goes behind the partial sums: sized below, once the rounds are laid out — at most n_mats jobs per round and the finish)
    size_t max_jobs = (size_t)n_mats;
    for (int m = 0; m < n_mats; m++) max_jobs += (size_t)num_vars[m];
    const size_t jobs_off = words;
    words += (max_jobs * sizeof(OpenJob) + 64) / 8 + 8;
    int rc = ctx_alloc(ctx, words * 8, &h->arena);
    if (rc) {
        delete h;
        return rc;
    }
    uint64_t* base = (uint64_t*)h->arena;
    h->partials = base + part_off;
    h->d_jobs = base + jobs_off;
    // the job table of every round, and of the finish
    std::vector<OpenJob> jobs;
    h->first_job.assign(n + 2, 0);
    h->n_live.assign(std::max(n, 1), 0);
    std::vector<OpenJob> last(n_mats);  // what the finish reads
    for (int m = 0; m < n_mats; m++) last[m] = OpenJob{(const E2*)dev_eq_ext[m], (const E2*)dev_f_ext[m], nullptr, num_vars[m] >= 1 ? (E2*)base : nullptr};
    for (int r = 0; r < n; r++) {
        h->first_job[r] = (int)jobs.size();
        for (int m : h->order) {
            const int v = num_vars[m], r0 = n - v;  // joins in round r0
            if (v == 0 || r0 > r) break;            // (sorted: nobody behind it is live either)
            OpenJob J{};
            const int k = r - r0;  // folds done before this round: k - 1 ... this round does fold number k (none when k = 0)
            if (k == 0) {
                J = OpenJob{(const E2*)dev_eq_ext[m], (const E2*)dev_f_ext[m], nullptr, nullptr};
            } else {
                // fold k writes 2^(v - k) entries per table into A (k odd) or B (k even); eq first, F behind it
                const size_t len = (size_t)1 << (v - k);
                E2* dst = (E2*)(base + ((k & 1) ? off_a[m] : off_b[m]));
                J.eq_in = last[m].eq_in;
                J.f_in = last[m].f_in;
                J.eq_out = dst;
                J.f_out = dst + len;
                last[m].eq_in = J.eq_out;
                last[m].f_in = J.f_out;
            }
            jobs.push_back(J);
            h->n_live[r]++;
        }
    }
    h->first_job[n] = (int)jobs.size();
    for (int m = 0; m < n_mats; m++) jobs.push_back(last[m]);
    h->first_job[n + 1] = (int)jobs.size();
    const size_t job_bytes = (jobs.size() * sizeof(OpenJob) + 63) & ~(size_t)63;
    h->counter = (unsigned*)((char*)h->d_jobs + job_bytes);
    h->msg_off = job_bytes + 64;
    h->fin_off = h->msg_off + 64;
    const size_t pin_bytes = h->fin_off + (size_t)n_mats * 16;
    rc = ctx_pinned_alloc(ctx, pin_bytes, &h->h_pin, &h->d_pin);
    if (rc) {
        ctx_free(ctx, h->arena);
        delete h;
        return rc;
    }
    memset(h->h_pin, 0, h->msg_off);
    memcpy(h->h_pin, jobs.data(), jobs.size() * sizeof(OpenJob));
    uint64_t* w = (uint64_t*)((char*)h->h_pin + h->msg_off);
    for (size_t k = 0; k < (pin_bytes - h->msg_off) / 8; k++) w[k] = MSG_INVALID;
    hipError_t e = hipMemcpyAsync((void*)h->d_jobs, h->h_pin, h->msg_off, hipMemcpyHostToDevice, st);  // jobs + the zeroed counter
    if (e != hipSuccess) {
        ctx_pinned_free(ctx, h->h_pin);
        ctx_free(ctx, h->arena);
        delete h;
        return ctx_fail(ctx, CENO_HIP_ERR_HIP, "open_rounds_begin: %s", hipGetErrorString(e));
    }
    *out = h;
    return 0;
}

int ceno_hip_open_rounds_round(ceno_hip_ctx* ctx, ceno_hip_open_rounds* h, const uint64_t* challenge_prev2, uint64_t* out_evals4) {
    CHECK_ARG(ctx, h && h->ctx == ctx && out_evals4, "bad open_rounds_round arguments");
    CHECK_ARG(ctx, h->round < h->n, "open_rounds_round: all %d rounds are done", h->n);
    CHECK_ARG(ctx, (h->round == 0) == (challenge_prev2 == nullptr), "open_rounds_round: round %d %s a challenge", h->round, h->round ? "needs" : "takes no");
    E2 r = e2_zero();
    if (challenge_prev2) {
        CHECK_ARG(ctx, challenge_prev2[0] < gl::P && challenge_prev2[1] < gl::P, "challenge is not canonical");
        r = E2{challenge_prev2[0], challenge_prev2[1]};
    }
    (void)ctx_stream(ctx, (ceno_hip_stream)h->st);
    const int rr = h->round, nj = h->n_live[rr], log_pairs = h->n - rr - 1;
    uint64_t* hw = (uint64_t*)((char*)h->h_pin + h->msg_off);
    const size_t items = (size_t)nj << log_pairs;
    hipLaunchKernelGGL(k_open_round, dim3(grid_for(items, NT, MAXB)), dim3(NT), 0, h->st, (const OpenJob*)h->d_jobs + h->first_job[rr], nj, log_pairs, r,
                       h->partials, h->counter, (uint64_t*)((char*)h->d_pin + h->msg_off));
    HIP_TRY(ctx, hipGetLastError());
    TRY(wait_words(h, hw, 4, "its message"));
    for (int k = 0; k < 4; k++) {
        out_evals4[k] = hw[k];
        __atomic_store_n(&hw[k], MSG_INVALID, __ATOMIC_RELAXED);  // armed for the next round (ordered before its launch: the doorbell)
    }
    h->round++;
    return 0;
}

int ceno_hip_open_rounds_finish(ceno_hip_ctx* ctx, ceno_hip_open_rounds* h, const uint64_t* challenge_last2, uint64_t* out_finals) {
    CHECK_ARG(ctx, h && h->ctx == ctx && out_finals, "bad open_rounds_finish arguments");
    CHECK_ARG(ctx, h->round == h->n, "open_rounds_finish after round %d of %d", h->round, h->n);
    CHECK_ARG(ctx, (h->n == 0) == (challenge_last2 == nullptr), "open_rounds_finish: %s challenge", h->n ? "needs the last" : "takes no");
    E2 r = e2_zero();
    if (challenge_last2) {
        CHECK_ARG(ctx, challenge_last2[0] < gl::P && challenge_last2[1] < gl::P, "challenge is not canonical");
        r = E2{challenge_last2[0], challenge_last2[1]};
    }
    (void)ctx_stream(ctx, (ceno_hip_stream)h->st);
    uint64_t* hw = (uint64_t*)((char*)h->h_pin + h->fin_off);
    hipLaunchKernelGGL(k_open_finish, dim3(grid_for((size_t)h->n_mats, NT, MAXB)), dim3(NT), 0, h->st, (const OpenJob*)h->d_jobs + h->first_job[h->n], h->n_mats, r,
                       (uint64_t*)((char*)h->d_pin + h->fin_off));
    HIP_TRY(ctx, hipGetLastError());
    TRY(wait_words(h, hw, 2 * (size_t)h->n_mats, "the final evaluations"));
    memcpy(out_finals, hw, (size_t)h->n_mats * 16);
    h->round++;
    return 0;
}

int ceno_hip_open_rounds_done(const ceno_hip_open_rounds* h) { return h ? h->round : 0; }

void ceno_hip_open_rounds_free(ceno_hip_ctx* ctx, ceno_hip_open_rounds* h) {
    if (!h) return;
    (void)hipStreamSynchronize(h->st);  // the job table and the message words are read / written through their host mapping
    (void)ctx_stream(ctx, (ceno_hip_stream)h->st);
    if (h->h_pin) ctx_pinned_free(ctx, h->h_pin);
    if (h->arena) ctx_free(ctx, h->arena);
    delete h;
}

}  // extern "C"
