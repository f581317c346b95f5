// Basefold batch open — host control flow of `OpeningProver::open` -> `PCS::batch_open`
// (ceno_zkvm/src/scheme/hal.rs:284-294, CPU scheme/cpu/mod.rs:1418-1457) over the device C ABI.
//
// The implementation the reference calls is in the EXT crate mpcs; the protocol is taken from the in-tree verifier
// replay ceno_recursion_v2/src/pcs/mod.rs:1111-1316 (transcript script, initial/final claims), :7494-7720 (queries),
// :7765-7781 (fold), :8125-8204 (proof of work = check_witness, sample_bits), :1252-1266 (query indices), :7547-7565 (one input
// opening per commitment at query >> bits_reduced).  PARITY UNPINNED (include/ceno_prover.h: constants, label packing);
// oracle/basefold.c is the CPU restatement of the same script plus the verifier, and the parity tests compare complete proofs
// word for word.
//
// Per opening group (one committed trace matrix with its point): F_m = sum_col coeff * column (ext table) and
// E_m = eq(point_m, .).  The degree-2 sumcheck of sum_m 2^(n - nv_m) <E_m, F_m> runs over n = max nv rounds; a
// matrix with fewer variables contributes a constant until round n - nv_m and is live afterwards (it meets the LAST
// nv_m challenges).  In lock step the running codeword (ext, bit-reversed order) is Merkle-committed pair-wise and
// folded once per round by ONE fused kernel, and the batched codeword of the next height is added as it joins.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <map>
#include <vector>

#include "../csrc/gl64.hpp"
#include "pcs_data.hpp"

using gl::E2;

int prover_set_error(int code, const char* msg);  // prover.cpp
#include "open_hook.hpp"
int basefold_open_hooked(ceno_hip_ctx* ctx, ceno_pcs_data* const* commits, int n_commits, const uint64_t* const* points, const uint64_t* const* evals,
                         int n_queries, int pow_bits, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_proof, const BasefoldOpenHook* hook);

namespace {

void tr_label(ceno_transcript* t, const char* s) { t->append_label(t->self, (const uint8_t*)s, strlen(s)); }
void tr_ext(ceno_transcript* t, E2 e) {
    uint64_t w[2] = {e.c0, e.c1};
    t->append_ext(t->self, w);
}
E2 tr_sample(ceno_transcript* t) {
    uint64_t o[2];
    t->sample_ext(t->self, o);
    return E2{o[0], o[1]};
}

int max_nv(ceno_pcs_data* const* commits, int n_commits) {
    int n = 0;
    for (int c = 0; c < n_commits; c++) n = std::max(n, commits[c]->max_log_rows());
    return n;
}
size_t query_words(ceno_pcs_data* const* commits, int n_commits, int n) {
    size_t w = 1;
    const int rate_log = commits[0]->log_blowup;
    for (int c = 0; c < n_commits; c++) w += ceno_hip_mmcs_opening_words(commits[c]->tree);
    for (int r = 0; r < n; r++) w += 2 + 4 * (size_t)(n + rate_log - r - 1);
    return w;
}
bool commits_ok(ceno_pcs_data* const* commits, int n_commits) {
    if (!commits || n_commits < 1) return false;
    for (int c = 0; c < n_commits; c++)
        if (!commits[c] || !commits[c]->tree || commits[c]->mats.empty() || commits[c]->log_blowup != commits[0]->log_blowup) return false;
    return true;
}

struct Group {  // matrices with the same number of variables share one sumcheck handle
    std::vector<int> mats;
    ceno_hip_sumcheck* sc = nullptr;
    bool started = false;
};

}  // namespace

// Pairs of tree streams (highest / lowest priority, see below) are kept between opens: creating and destroying two
// streams costs a few hundred microseconds per call, more than a commit round.  A pair is taken out of the pool for the
// duration of an open, so concurrent opens never share one.
namespace {
struct TreeStreams {
    hipStream_t s[2];
    int device;
};
std::mutex g_ts_mu;
std::vector<TreeStreams> g_ts_pool;
bool tree_streams_acquire(ceno_hip_ctx* ctx, TreeStreams* out) {
    // keyed on the CONTEXT's device (a lane worker thread starts with device 0 current), which is made current first
    if (ceno_hip_make_current(ctx) != 0) return false;
    const int dev = ceno_hip_device(ctx);
    {
        std::lock_guard<std::mutex> g(g_ts_mu);
        for (size_t i = 0; i < g_ts_pool.size(); i++)
            if (g_ts_pool[i].device == dev) {
                *out = g_ts_pool[i];
                g_ts_pool.erase(g_ts_pool.begin() + (long)i);
                for (int k = 0; k < 2; k++) (void)ceno_hip_stream_adopt(ctx, (ceno_hip_stream)out->s[k]);
                return true;
            }
    }
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    const int prio[2] = {greatest, least};
    out->device = dev;
    out->s[0] = out->s[1] = nullptr;
    for (int i = 0; i < 2; i++)
        if (hipStreamCreateWithPriority(&out->s[i], hipStreamNonBlocking, prio[i]) != hipSuccess) {
            if (out->s[0]) (void)hipStreamDestroy(out->s[0]);
            return false;
        }
    for (int k = 0; k < 2; k++) (void)ceno_hip_stream_adopt(ctx, (ceno_hip_stream)out->s[k]);  // pool blocks freed on them are ordered
    return true;
}
void tree_streams_release(const TreeStreams& ts) {
    std::lock_guard<std::mutex> g(g_ts_mu);
    g_ts_pool.push_back(ts);
}
}  // namespace

// the same pooled pair serves commit_traces as its helper streams (pcs_data.hpp): another stream object per context would
// shift the runtime's stream -> hardware-queue mapping and put the opening's two tree streams on one queue
bool ceno_aux_streams_acquire(ceno_hip_ctx* ctx, hipStream_t out[2], int* device) {
    TreeStreams ts;
    if (!tree_streams_acquire(ctx, &ts)) return false;
    out[0] = ts.s[0];
    out[1] = ts.s[1];
    *device = ts.device;
    return true;
}
void ceno_aux_streams_release(hipStream_t s[2], int device) { tree_streams_release(TreeStreams{{s[0], s[1]}, device}); }

extern "C" {

size_t ceno_prover_basefold_proof_words(ceno_pcs_data* const* commits, int n_commits, int n_queries) {
    if (!commits_ok(commits, n_commits) || n_queries < 0) return 0;
    const int n = max_nv(commits, n_commits);
    size_t n_mats = 0;
    for (int c = 0; c < n_commits; c++) n_mats += commits[c]->mats.size();
    return 8 * (size_t)n + 2 * n_mats + 1 + (size_t)n_queries * query_words(commits, n_commits, n);
}

int ceno_prover_basefold_open(ceno_hip_ctx* ctx, ceno_pcs_data* const* commits, int n_commits, const uint64_t* const* points,
                              const uint64_t* const* evals, int n_queries, int pow_bits, ceno_transcript* tr, ceno_hip_stream s,
                              uint64_t* out_proof) {
    return basefold_open_hooked(ctx, commits, n_commits, points, evals, n_queries, pow_bits, tr, s, out_proof, nullptr);
}
size_t ceno_prover_basefold_proof_words_commits(int n_commits, const int* n_mats, const int* total_width, const int* max_log_rows, int log_blowup,
                                                int n_queries) {
    if (n_commits < 1 || !n_mats || !total_width || !max_log_rows) return 0;
    int n = 0;
    size_t mats = 0, w = 1;
    for (int c = 0; c < n_commits; c++) {
        n = std::max(n, max_log_rows[c]);
        mats += (size_t)n_mats[c];
        w += (size_t)total_width[c] + 4 * (size_t)(max_log_rows[c] + log_blowup);
    }
    for (int r = 0; r < n; r++) w += 2 + 4 * (size_t)(n + log_blowup - r - 1);
    return 8 * (size_t)n + 2 * mats + 1 + (size_t)n_queries * w;
}
size_t ceno_prover_basefold_proof_words_meta(int n_mats, int total_width, int max_log_rows, int log_blowup, int n_queries) {
    const int n = max_log_rows;
    size_t w = 1 + (size_t)total_width + 4 * (size_t)(n + log_blowup);
    for (int r = 0; r < n; r++) w += 2 + 4 * (size_t)(n + log_blowup - r - 1);
    return 8 * (size_t)n + 2 * (size_t)n_mats + 1 + (size_t)n_queries * w;
}
}  // extern "C"

// hook == NULL: the commitments hold their traces, codewords and trees on this device.  hook != NULL (ceno_dist_basefold_open, dist_open.cpp):
// `commits` carry the SHAPES only (mats, classes, log_blowup; no tables, no tree) and the three places that touch the committed data go
// through the hook: the batched codeword of a height class, the batched trace polynomial of a matrix, the opening of the commitment at the
// query indices.  Everything else — the sumcheck, the folds, the round trees, proof of work, the round answers — runs here on whatever the
// hook produced, so a sharded opening ends with the words of the single-device one.
int basefold_open_hooked(ceno_hip_ctx* ctx, ceno_pcs_data* const* commits, int n_commits, const uint64_t* const* points,
                         const uint64_t* const* evals, int n_queries, int pow_bits, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_proof,
                         const BasefoldOpenHook* hook) {
    if (!ctx || !(hook ? (commits && n_commits >= 1) : commits_ok(commits, n_commits)) || !points || !evals || !tr || !out_proof || n_queries < 0 || pow_bits < 0 ||
        pow_bits > 40)
        return prover_set_error(CENO_HIP_ERR_INVALID, "bad basefold_open arguments");
    if (!s) return prover_set_error(CENO_HIP_ERR_INVALID, "basefold_open needs an explicit stream (ceno_hip_stream_create)");
    if (!tr->sample_bits || !tr->append_base)
        return prover_set_error(CENO_HIP_ERR_INVALID, "basefold_open: the transcript lacks the base-field operations (append_base / sample_bits)");
    (void)ceno_hip_stream_bind(ctx, s);  // this thread drives three streams: say which one every allocation is for (pool tags)
    hipStream_t st = (hipStream_t)s;
    const int rate_log = commits[0]->log_blowup, n = max_nv(commits, n_commits), H = n + rate_log;
    // the matrices of all commitments in opening order (batch coefficients, final message)
    struct Flat {
        ceno_pcs_data* d;
        int m;
        size_t coeff0;  // index of its first batch coefficient
    };
    std::vector<Flat> flat;
    size_t total_cols = 0;
    for (int c = 0; c < n_commits; c++)
        for (int m = 0; m < (int)commits[c]->mats.size(); m++) {
            flat.push_back({commits[c], m, total_cols});
            total_cols += commits[c]->mats[m].width;
        }
    const int n_mats = (int)flat.size();

    // everything allocated here is released by `cleanup`
    std::vector<ceno_hip_mle*> owned;
    std::vector<std::pair<ceno_hip_mle*, ceno_hip_stream>> owned_on;  // tables made (and used) on a tree stream
    std::vector<ceno_hip_merkle*> trees;
    std::map<int, Group> groups;  // key: nv
    void* d_scratch = nullptr;
    ceno_hip_stream sx[2] = {nullptr, nullptr};  // tree building runs one round ahead on two alternating streams
    hipEvent_t ev[2] = {nullptr, nullptr};
    int ts_device = 0;
    ceno_hip_open_rounds* all_rounds = nullptr;  // several height groups: the sumcheck of all matrices behind one handle (one launch per round)
    auto cleanup = [&]() {
        for (int i = 0; i < 2; i++) {
            if (sx[i]) (void)hipStreamSynchronize((hipStream_t)sx[i]);
            if (ev[i]) (void)hipEventDestroy(ev[i]);
        }
        // the running codewords go back to the pool from the stream they were made on: a block freed under another stream's tag is out of reach
        // of that stream's next request while the tag's stream has work queued — every opening then took its ~16 small codewords from the driver
        // (hipMalloc, ~8 us each) and left the last opening's in the cache for good (CENO_HIP_POOL_TRACE)
        for (auto& po : owned_on) {
            (void)ceno_hip_stream_bind(ctx, po.second);
            ceno_hip_mle_free(ctx, po.first);
        }
        owned_on.clear();
        (void)ceno_hip_stream_bind(ctx, s);
        if (sx[0]) tree_streams_release(TreeStreams{{(hipStream_t)sx[0], (hipStream_t)sx[1]}, ts_device});
        sx[0] = sx[1] = nullptr;
        for (auto& g : groups)
            if (g.second.sc) ceno_hip_sumcheck_free(ctx, g.second.sc);
        if (all_rounds) ceno_hip_open_rounds_free(ctx, all_rounds);
        all_rounds = nullptr;
        for (auto* t : trees)
            if (t) ceno_hip_merkle_free(ctx, t);
        for (auto* m : owned) ceno_hip_mle_free(ctx, m);
    };
    auto fail = [&](int rc, const char* what = nullptr) {
        std::string msg = what ? std::string("basefold_open: ") + what : std::string(ceno_hip_last_error(ctx));
        cleanup();
        return prover_set_error(rc, msg.c_str());
    };
    auto alloc_ext = [&](int nv, ceno_hip_mle** out) {
        int rc = ceno_hip_mle_alloc(ctx, nv, 1, out);
        if (!rc) owned.push_back(*out);
        return rc;
    };

    const bool dbg = getenv("CENO_HIP_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(st);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ceno_prover] basefold_open %-16s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // ---- batch coefficients (pcs/mod.rs:1130-1131) ----
    tr_label(tr, "batch coeffs");
    const E2 alpha = tr_sample(tr);
    std::vector<uint64_t> coeff(2 * total_cols);
    {
        E2 c = gl::e2_one();
        for (size_t i = 0; i < total_cols; i++) {
            coeff[2 * i] = c.c0;
            coeff[2 * i + 1] = c.c1;
            c = c * alpha;
        }
    }
    // ---- batched codeword per height, F_m / E_m per matrix, S_m = sum coeff * eval ----
    std::vector<ceno_hip_mle*> B(H + 1, nullptr), F(n_mats, nullptr), Eq(n_mats, nullptr);
    std::vector<E2> S(n_mats);
    {
        // codewords: one pass per height CLASS of every commitment (its matrices are stored back to back, so the class is one
        // wide column-major matrix); the class's coefficients are gathered from the flat order.
        // Single device: every class codeword and every F_m is a JOB of one batched launch (ceno_hip_batch_columns_multi) and the eq tables of all
        // opening points come from one batched build (ceno_hip_selector_build_batch, Whole = eq(., point)) — a shard of the reference's population
        // has ~60 matrices in ~12 height classes, and a launch + a synchronisation per matrix was 3 of the opening's 8 ms.  With a hook
        // (multi-rank opening) the pieces are gathered across ranks one by one as before.
        std::vector<const uint64_t*> j_cols;
        std::vector<size_t> j_len;
        std::vector<int> j_ncols, j_acc;
        std::vector<uint64_t> j_coeffs;
        std::vector<uint64_t*> j_dst;
        size_t m0 = 0;
        for (int c = 0; c < n_commits; c++) {
            ceno_pcs_data* d = commits[c];
            for (size_t k = 0; k < d->classes.size(); k++) {
                auto& K = d->classes[k];
                std::vector<uint64_t> cc(2 * K.width);
                for (size_t m = 0; m < d->mats.size(); m++) {
                    auto& M = d->mats[m];
                    if (M.cls != (int)k) continue;
                    memcpy(cc.data() + 2 * M.col0, coeff.data() + 2 * flat[m0 + m].coeff0, 16 * M.width);
                }
                const int h = K.log_rows + rate_log;
                int rc = 0, fresh = 0;
                if (!B[h]) {
                    rc = alloc_ext(h, &B[h]);
                    fresh = 1;
                }
                if (!rc && hook) rc = hook->batch_codeword(hook->self, c, (int)k, cc.data(), ceno_hip_mle_device_ptr(B[h]), h, fresh ? 0 : 1, s);
                else if (!rc) {
                    j_cols.push_back(ceno_hip_mle_device_ptr(K.codeword));
                    j_len.push_back((size_t)1 << h);
                    j_ncols.push_back((int)K.width);
                    j_acc.push_back(fresh ? 0 : 1);
                    j_dst.push_back(ceno_hip_mle_device_ptr(B[h]));
                    j_coeffs.insert(j_coeffs.end(), cc.begin(), cc.end());
                }
                if (rc) return fail(rc);
            }
            m0 += d->mats.size();
        }
        for (int m = 0; m < n_mats; m++) {
            auto& M = flat[m].d->mats[flat[m].m];
            const size_t ci = flat[m].coeff0;
            int rc = alloc_ext(M.log_rows, &F[m]);
            if (!rc && hook) {
                int ci_commit = 0;
                for (int c2 = 0; c2 < n_commits; c2++)
                    if (commits[c2] == flat[m].d) ci_commit = c2;
                rc = hook->batch_trace(hook->self, ci_commit, flat[m].m, coeff.data() + 2 * ci, ceno_hip_mle_device_ptr(F[m]), s);
            } else if (!rc) {
                j_cols.push_back(flat[m].d->trace_ptr(flat[m].m));
                j_len.push_back(M.rows);
                j_ncols.push_back((int)M.width);
                j_acc.push_back(0);
                j_dst.push_back(ceno_hip_mle_device_ptr(F[m]));
                j_coeffs.insert(j_coeffs.end(), coeff.begin() + 2 * ci, coeff.begin() + 2 * (ci + M.width));
            }
            if (!rc && hook) {
                rc = ceno_hip_eq_build(ctx, points[m], M.log_rows, nullptr, s, &Eq[m]);
                if (!rc) owned.push_back(Eq[m]);
            }
            if (rc) return fail(rc);
            E2 acc = gl::e2_zero();
            for (size_t c = 0; c < M.width; c++)
                acc = acc + E2{coeff[2 * (ci + c)], coeff[2 * (ci + c) + 1]} * E2{evals[m][2 * c], evals[m][2 * c + 1]};
            S[m] = acc;
            groups[M.log_rows].mats.push_back(m);
        }
        if (!hook) {
            (void)ceno_hip_stream_bind(ctx, s);
            int rc = ceno_hip_batch_columns_multi(ctx, (int)j_cols.size(), j_cols.data(), j_len.data(), j_ncols.data(), j_coeffs.data(), j_dst.data(),
                                                  j_acc.data(), s);
            if (rc) return fail(rc);
            std::vector<int> kinds((size_t)n_mats, CENO_HIP_SEL_WHOLE), nvs((size_t)n_mats);
            std::vector<size_t> zero((size_t)n_mats, 0);
            for (int m = 0; m < n_mats; m++) nvs[(size_t)m] = flat[m].d->mats[flat[m].m].log_rows;
            rc = ceno_hip_selector_build_batch(ctx, n_mats, kinds.data(), points, nvs.data(), zero.data(), zero.data(), s, Eq.data());
            if (rc) return fail(rc);
            for (int m = 0; m < n_mats; m++) owned.push_back(Eq[m]);
        }
    }
    lap("batching");
    uint64_t* msgs = out_proof;
    uint64_t* round_roots = out_proof + 4 * (size_t)n;
    uint64_t* finalm = out_proof + 8 * (size_t)n;
    uint64_t* powp = finalm + 2 * (size_t)n_mats;
    uint64_t* qbase = powp + 1;

    // ---- commit phase ----
    // Round r commits to the running codeword C[r], which depends only on challenge r-1: its tree is built as soon as
    // C[r] exists — on the stream that folded it — while round r's sumcheck message and the transcript work run on `s`.
    // Two alternating streams let the (latency-bound) trees of consecutive rounds overlap.
    std::vector<ceno_hip_mle*> C(n + 1, nullptr);
    C[0] = B[H];
    std::vector<uint64_t> ch(2 * (size_t)std::max(n, 1));
    trees.assign(n, nullptr);
    {
        // HIP spreads streams over a few hardware queues round-robin, and two streams that land on the same queue run
        // their kernels back to back (seen in the kernel trace: both tree streams on queue 4).  Streams of different
        // PRIORITY get queues of their own, so the two tree streams take the highest and the lowest level and leave the
        // caller's (normal) stream alone.
        TreeStreams ts;
        if (!tree_streams_acquire(ctx, &ts)) return fail(CENO_HIP_ERR_HIP, "hipStreamCreateWithPriority failed");
        ts_device = ts.device;
        for (int i = 0; i < 2; i++) {
            sx[i] = (ceno_hip_stream)ts.s[i];
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return fail(CENO_HIP_ERR_HIP, "hipEventCreate failed");
        }
    }
    if (n > 0) {
        int rc = ceno_hip_basefold_commit_codeword(ctx, ceno_hip_mle_device_ptr(C[0]), H, sx[0], &trees[0]);
        if (rc) return fail(rc);
    }
    auto t_round = std::chrono::steady_clock::now();
    // Several height groups (a shard's commitment: ~10): every live group's handle has tables of the SAME current size in round r (suffix
    // alignment: n - r variables left), and their round kernels do not depend on each other — while the tables are large the rounds of all live
    // groups are queued back to back with their messages left on the device (ceno_hip_sumcheck_round_dev) and fetched by ONE copy + wait, instead of
    // a launch and a wait per group; from 2^8 entries down the handles finish on the host as before (no device trip at all).
    // CENO_BASEFOLD_GROUP_ASYNC=0: the old one-by-one rounds (A/B)
    static const bool group_async_on = !(getenv("CENO_BASEFOLD_GROUP_ASYNC") && atoi(getenv("CENO_BASEFOLD_GROUP_ASYNC")) == 0);
    // ... and since round 6 in ONE launch per round for all of them (ceno_hip_open_rounds: the ~8 live groups of a shard's round were 8 slot-table
    // blits + 8 launches, ~135 us of a 265 us round, queued by this one thread).  CENO_BASEFOLD_OPEN_ROUNDS=0: the handles per group (A/B; read
    // per call).  With a hook too (the tables are whole on this device either way, and every launch ends without waiting for the host).
    const bool open_rounds_on = !(getenv("CENO_BASEFOLD_OPEN_ROUNDS") && atoi(getenv("CENO_BASEFOLD_OPEN_ROUNDS")) == 0) && groups.size() > 1 && n > 0;
    const bool group_async = group_async_on && !hook && groups.size() > 1 && !open_rounds_on;
    ceno_hip_mle* round_buf = nullptr;
    std::vector<uint64_t> round_host;
    if (group_async) {
        (void)ceno_hip_stream_bind(ctx, s);
        int nvb = 1;
        while (((size_t)1 << nvb) < 2 * groups.size() + 2) nvb++;
        if (int rc = alloc_ext(nvb, &round_buf)) return fail(rc);
        round_host.resize(4 * groups.size());
    }
    if (open_rounds_on) {
        std::vector<const uint64_t*> pe((size_t)n_mats), pf((size_t)n_mats);
        std::vector<int> nvs((size_t)n_mats);
        for (int m = 0; m < n_mats; m++) {
            pe[(size_t)m] = ceno_hip_mle_device_ptr(Eq[m]);
            pf[(size_t)m] = ceno_hip_mle_device_ptr(F[m]);
            nvs[(size_t)m] = flat[m].d->mats[flat[m].m].log_rows;
        }
        (void)ceno_hip_stream_bind(ctx, s);
        if (int rc = ceno_hip_open_rounds_begin(ctx, n_mats, pe.data(), pf.data(), nvs.data(), s, &all_rounds)) return fail(rc);
    }
    for (int r = 0; r < n; r++) {
        const int h = H - r;
        E2 p1 = gl::e2_zero(), p2 = gl::e2_zero();
        const bool async_round = group_async && (n - r) > 8;
        int n_async = 0;
        if (all_rounds) {
            uint64_t ev[4];
            if (int rc = ceno_hip_open_rounds_round(ctx, all_rounds, r == 0 ? nullptr : &ch[2 * (r - 1)], ev)) return fail(rc);
            p1 = E2{ev[0], ev[1]};
            p2 = E2{ev[2], ev[3]};
        }
        for (auto& kv : groups) {
            Group& g = kv.second;
            const int s_m = n - kv.first;
            if (s_m > r) {  // joins later: constant in this variable
                const uint64_t scale = gl::pow(2, (uint64_t)(s_m - r - 1));
                for (int m : g.mats) {
                    E2 c = gl::e2_mul_base(S[m], scale);
                    p1 = p1 + c;
                    p2 = p2 + c;
                }
                continue;
            }
            if (all_rounds) continue;  // live: in the one launch above
            int rc = 0;
            if (!g.started) {  // becomes live now: terms E_m * F_m over kv.first variables
                std::vector<ceno_hip_mle*> mles;
                std::vector<uint64_t> tc;
                std::vector<uint32_t> toff{0}, tidx;
                for (int m : g.mats) {
                    tidx.push_back((uint32_t)mles.size());
                    mles.push_back(Eq[m]);
                    tidx.push_back((uint32_t)mles.size());
                    mles.push_back(F[m]);
                    toff.push_back((uint32_t)tidx.size());
                    tc.push_back(1);
                    tc.push_back(0);
                }
                ceno_hip_sumcheck_plan plan{};
                plan.num_mles = (int)mles.size();
                plan.num_terms = (int)g.mats.size();
                plan.term_coeffs = tc.data();
                plan.term_offsets = toff.data();
                plan.term_mle_idx = tidx.data();
                plan.max_num_vars = kv.first;
                plan.max_degree = 2;
                rc = ceno_hip_sumcheck_begin(ctx, mles.data(), &plan, s, &g.sc);
                // rounds queued ahead of their challenges (persistent small-round kernels, host-finished tail): between two rounds
                // this loop only waits for a tree root, and a round kernel that waits a little longer for its challenge costs nothing.
                // Only when ONE height group exists: the queued kernels of a pipelined sumcheck wait for this thread, and a second
                // group's kernels behind them on the same stream would never produce the message this thread waits for.
                // CENO_BASEFOLD_PIPELINE=0: one launch + synchronisation per round (A/B measurements)
                static const bool pipe = !(getenv("CENO_BASEFOLD_PIPELINE") && atoi(getenv("CENO_BASEFOLD_PIPELINE")) == 0);
                if (!rc && pipe && groups.size() == 1 && !(hook && hook->serial_rounds)) (void)ceno_hip_sumcheck_set_pipelined(ctx, g.sc, 1);
                g.started = true;
            }
            uint64_t ev[4];
            if (!rc && async_round) {
                rc = ceno_hip_sumcheck_round_dev(ctx, g.sc, (r - s_m) == 0 ? nullptr : &ch[2 * (r - 1)], ceno_hip_mle_device_ptr(round_buf) + 4 * n_async);
                n_async++;
                if (rc) return fail(rc);
                continue;
            }
            if (!rc) rc = ceno_hip_sumcheck_round(ctx, g.sc, (r - s_m) == 0 ? nullptr : &ch[2 * (r - 1)], ev);
            if (rc) return fail(rc);
            p1 = p1 + E2{ev[0], ev[1]};
            p2 = p2 + E2{ev[2], ev[3]};
        }
        if (n_async) {
            if (hipMemcpyAsync(round_host.data(), ceno_hip_mle_device_ptr(round_buf), (size_t)32 * n_async, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess)
                return fail(CENO_HIP_ERR_HIP, "download of the round messages failed");
            for (int a = 0; a < n_async; a++) {
                p1 = p1 + E2{round_host[4 * a], round_host[4 * a + 1]};
                p2 = p2 + E2{round_host[4 * a + 2], round_host[4 * a + 3]};
            }
        }
        msgs[4 * r] = p1.c0; msgs[4 * r + 1] = p1.c1; msgs[4 * r + 2] = p2.c0; msgs[4 * r + 3] = p2.c1;
        tr_ext(tr, p1);
        tr_ext(tr, p2);
        tr_label(tr, "commit round");
        const E2 c = tr_sample(tr);
        ch[2 * r] = c.c0;
        ch[2 * r + 1] = c.c1;
        // fold with c_r (the codeword of the next height joins) and start the NEXT round's tree right away
        ceno_hip_stream fs = sx[(r + 1) & 1];  // C[r] was produced on sx[r & 1] (event ev[r & 1]); C[0] by the batching on `s` (synchronised)
        (void)ceno_hip_stream_bind(ctx, fs);  // C[r + 1] is produced and consumed on `fs`
        int rc = ceno_hip_mle_alloc(ctx, h - 1, 1, &C[r + 1]);
        if (!rc) owned_on.push_back({C[r + 1], fs});
        if (!rc && r > 0 && hipStreamWaitEvent((hipStream_t)fs, ev[r & 1], 0) != hipSuccess) rc = CENO_HIP_ERR_HIP;
        if (!rc) rc = ceno_hip_basefold_fold(ctx, ceno_hip_mle_device_ptr(C[r]), h, &ch[2 * r], B[h - 1] ? ceno_hip_mle_device_ptr(B[h - 1]) : nullptr,
                                             ceno_hip_mle_device_ptr(C[r + 1]), fs);
        if (!rc && hipEventRecord(ev[(r + 1) & 1], (hipStream_t)fs) != hipSuccess) rc = CENO_HIP_ERR_HIP;
        if (!rc && r + 1 < n) rc = ceno_hip_basefold_commit_codeword(ctx, ceno_hip_mle_device_ptr(C[r + 1]), h - 1, fs, &trees[r + 1]);
        // the root of THIS round's tree (built one round ago on the other stream) is observed after the challenge
        if (!rc) rc = ceno_hip_merkle_root(ctx, trees[r], round_roots + 4 * r, sx[r & 1]);
        if (rc) return fail(rc);
        tr->append_ext(tr->self, round_roots + 4 * r);
        tr->append_ext(tr->self, round_roots + 4 * r + 2);
        if (dbg) {
            auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[ceno_prover] basefold_open round %2d (height %2d): %7.1f us since the previous round\n", r, h,
                    std::chrono::duration<double, std::micro>(now - t_round).count());
            t_round = now;
        }
    }
    lap("commit phase");
    // ---- final message: F_m at the challenges, one row per opening point ----
    E2 total = gl::e2_zero();
    if (all_rounds) {
        if (int rc = ceno_hip_open_rounds_finish(ctx, all_rounds, &ch[2 * (n - 1)], finalm)) return fail(rc);
    }
    for (auto& kv : groups) {
        if (all_rounds) break;
        Group& g = kv.second;
        std::vector<uint64_t> fin(4 * g.mats.size());
        int rc = ceno_hip_sumcheck_finish(ctx, g.sc, n ? &ch[2 * (n - 1)] : nullptr, fin.data());
        if (rc) return fail(rc);
        for (size_t i = 0; i < g.mats.size(); i++) {
            finalm[2 * g.mats[i]] = fin[4 * i + 2];
            finalm[2 * g.mats[i] + 1] = fin[4 * i + 3];
        }
    }
    for (int m = 0; m < n_mats; m++) {
        const E2 v{finalm[2 * m], finalm[2 * m + 1]};
        total = total + v;
        tr_ext(tr, v);
    }
    {   // the fully folded codeword must be the constant codeword of the message
        std::vector<uint64_t> last(2 * ((size_t)1 << rate_log));
        int rc = ceno_hip_mle_download(ctx, C[n], last.data(), n > 0 ? sx[n & 1] : s);  // the stream of the last fold
        if (rc) return fail(rc);
        for (size_t i = 0; i < ((size_t)1 << rate_log); i++)
            if (last[2 * i] != total.c0 || last[2 * i + 1] != total.c1) {
                cleanup();
                return prover_set_error(CENO_HIP_ERR_STATE, "basefold_open: folded codeword is not the encoding of the final message");
            }
    }
    lap("final message");
    // ---- proof of work: p3 grinding — a witness that check_witness accepts (pcs/mod.rs:1248-1251, 8125-8155) ----
    *powp = 0;
    if (pow_bits > 0) {
        int rc = ceno_prover_transcript_grind(ctx, tr, pow_bits, s, powp);
        if (rc) {
            cleanup();
            return rc;
        }
    }
    lap("proof of work");
    // ---- queries ----
    tr_label(tr, "query indices");
    if (n_queries == 0) {
        cleanup();
        return 0;
    }
    size_t qw = 1;
    for (int c = 0; c < n_commits; c++) qw += hook ? hook->opening_words(hook->self, c) : ceno_hip_mmcs_opening_words(commits[c]->tree);
    for (int r = 0; r < n; r++) qw += 2 + 4 * (size_t)(n + rate_log - r - 1);
    const size_t Q = (size_t)n_queries;
    std::vector<uint64_t> qidx(Q);
    for (size_t q = 0; q < Q; q++) qidx[q] = ceno_transcript_sample_bits(tr, H);  // ONE base sample per query (pcs/mod.rs:1252-1266)
    // device scratch: [indices Q][piece-major answers]; host buffer mirrors the answers
    const size_t ans_words = Q * (qw - 1);
    {   // scratch from the library's pool (a base-field table of enough words): hipMalloc / hipFree cost ~0.1 ms per open
        int snv = 0;
        while (((size_t)1 << snv) < Q + ans_words) snv++;
        ceno_hip_mle* scratch = nullptr;
        (void)ceno_hip_stream_bind(ctx, s);
        if (ceno_hip_mle_alloc(ctx, snv, 0, &scratch) != 0) {
            cleanup();
            return prover_set_error(CENO_HIP_ERR_OOM, "basefold_open: query scratch allocation failed");
        }
        owned.push_back(scratch);
        d_scratch = ceno_hip_mle_device_ptr(scratch);
    }
    uint64_t* d_idx = (uint64_t*)d_scratch;
    uint64_t* d_ans = d_idx + Q;
    if (hipMemcpyAsync(d_idx, qidx.data(), Q * 8, hipMemcpyHostToDevice, st) != hipSuccess) return fail(CENO_HIP_ERR_HIP, "upload of the query indices failed");
    struct Piece { size_t off, per_q; };
    std::vector<Piece> pieces;
    size_t off = 0;
    int rc = 0;
    for (int c = 0; c < n_commits && !rc; c++) {  // one MMCS opening per commitment at reduced_index = query >> bits_reduced
        ceno_pcs_data* d = commits[c];
        const int hc = d->max_log_rows() + rate_log;
        const size_t per_q = hook ? hook->opening_words(hook->self, c) : ceno_hip_mmcs_opening_words(d->tree);
        if (hook) rc = hook->mmcs_open(hook->self, c, qidx.data(), d_idx, Q, H - hc, d_ans + off, per_q, s);
        else rc = ceno_hip_mmcs_open_batch(ctx, d->tree, d_idx, Q, H - hc, d_ans + off, per_q, s);
        pieces.push_back({off, per_q});
        off += Q * per_q;
    }
    if (!rc && n > 0) {  // every commit round in one launch: [sibling 2][path 4 (h - 1)] per round
        std::vector<const uint64_t*> cws(n);
        for (int r = 0; r < n; r++) cws[r] = ceno_hip_mle_device_ptr(C[r]);
        rc = ceno_hip_basefold_query_rounds(ctx, cws.data(), trees.data(), n, d_idx, Q, d_ans + off, s);
        for (int r = 0; r < n; r++) {
            const int h = H - r;
            pieces.push_back({off, 2});
            off += Q * 2;
            pieces.push_back({off, 4 * (size_t)(h - 1)});
            off += Q * 4 * (size_t)(h - 1);
        }
    }
    if (rc) return fail(rc);
    std::vector<uint64_t> ans(ans_words);
    if (hipMemcpyAsync(ans.data(), d_ans, ans_words * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(CENO_HIP_ERR_HIP, "download of the query answers failed");
    lap("query gathers");
    for (size_t q = 0; q < Q; q++) {
        uint64_t* out = qbase + q * qw;
        *out++ = qidx[q];
        for (auto& p : pieces) {
            memcpy(out, ans.data() + p.off + q * p.per_q, p.per_q * 8);
            out += p.per_q;
        }
    }
    cleanup();
    return 0;
}
