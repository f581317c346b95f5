// Basefold opening of a commitment whose matrices were committed ACROSS ranks (ceno_dist_commit_traces_mmcs): the proof equals
// ceno_prover_basefold_open's on the single-device commitment of the same matrices, word for word, on every rank.
//
// Reference flow (single device): OpeningProver::open -> PCS::batch_open, ceno_zkvm/src/scheme/cpu/mod.rs:1418-1457; protocol restated in
// ceno_recursion_v2/src/pcs/mod.rs:1111-1316,7494-7781.  The reference has no distribution (docs/src/optimizations.md:3-5).
//
// What is sharded and what is not.  Config #3's opening is 3.1 ms of which ~2.7 ms are ~21 DEPENDENT commit rounds (fold, tree, root, transcript)
// on a running codeword that is 1/width of the committed data: latency, which more GPUs do not shorten.  What scales with the data is the
// batching — one pass over every codeword and trace column (555 MB for 2^20 x 22) — and the commitment's own rows and paths at the queries.  So:
//   * batched codeword: every rank batches ITS ROWS of all columns (the row shard the commit left it with), the shards are all-gathered
//     (bulk transport) into the full batched codeword on every rank;
//   * batched trace polynomial F = sum_c coeff_c col_c: every rank batches ITS COLUMNS (the column shard it committed), the partial sums are
//     all-gathered and added mod p (ceno_hip_ext_sum_blocks; no collective library reduces mod p);
//   * sumcheck, folds, round trees, final message, proof of work, round answers: replicated, the single-device code (basefold.cpp) on the
//     gathered tables — same transcript on every rank;
//   * the commitment's opening at a query: the rank that owns the row answers from its rows and its sub-tree (ceno_hip_mmcs_open_batch on
//     local indices), every rank appends the top log2(world) levels from the replicated top tree; the answers travel by the small-message
//     transport.
// Matrices of any heights (the traces of a shard's chips: one batched codeword per height class, the sub-trees are mixed-height trees over the
// row shards); a matrix whose codeword has fewer rows than there are ranks is held whole by every rank and lives in the replicated top tree
// (ceno_dist_commit_traces_mmcs): its class is batched locally, its rows at a query come from the top tree's opening.  The tallest codeword of a
// commitment needs at least `world` rows.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ceno_prover.h"
#include "open_hook.hpp"
#include "pcs_data.hpp"

int prover_set_error(int code, const char* msg);  // prover.cpp
int basefold_open_hooked(ceno_hip_ctx* ctx, ceno_pcs_data* const* commits, int n_commits, const uint64_t* const* points, const uint64_t* const* evals,
                         int n_queries, int pow_bits, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_proof, const BasefoldOpenHook* hook);
int dist_comm_world(const ceno_dist_comm* c);  // dist.cpp
int dist_comm_rank(const ceno_dist_comm* c);
bool dist_comm_ranks_share_device(ceno_dist_comm* c, hipStream_t st);  // dist.cpp: from the ranks' host names + PCI bus ids, gathered once
bool dist_comm_has_rccl(const ceno_dist_comm* c);  // one rank per GPU by construction (RCCL does not place two ranks on a device)
int dist_allgather_words(ceno_dist_comm* c, const uint64_t* mine, size_t n_words, uint64_t* out, hipStream_t st);
int dist_allgather_device(ceno_dist_comm* c, const uint64_t* send_dev, size_t n_words, uint64_t* recv_dev, hipStream_t st);

namespace {

struct DistCommit {                            // one commitment made across the ranks
    int n_mats = 0, log_rows = 0;              // log_rows: of its TALLEST trace
    std::vector<int> log_rows_of;              // per matrix
    std::vector<std::vector<int>> class_mats;  // height classes, tallest first; the matrices of a class in the caller's order (commit.cpp)
    const int* widths = nullptr;               // [m * W + g]
    const uint64_t* const* local_trace_cols = nullptr;  // [m]: this rank's columns of matrix m, column-major, 2^log_rows rows
    const uint64_t* const* local_cw_rows = nullptr;     // [m]: ALL columns of matrix m x (R / W) rows, column-major (ceno_dist_commit_traces_mmcs out_rows_dev)
    ceno_hip_merkle* subtree = nullptr;
    ceno_hip_merkle* top = nullptr;
    std::vector<size_t> width_of;              // per matrix: all ranks' columns
    std::vector<char> is_short;                // per matrix: fewer codeword rows than ranks — every rank holds ALL its rows, it joins the tree in the replicated top
    size_t total_width = 0, short_width = 0;
};
struct DistOpen {
    ceno_hip_ctx* ctx;
    ceno_dist_comm* comm;
    int W, rank, k;
    int log_blowup;
    std::vector<DistCommit> commits;
};

int fail_ctx(ceno_hip_ctx* ctx, int rc) { return prover_set_error(rc, ceno_hip_last_error(ctx)); }

struct DevBuf {
    ceno_hip_ctx* ctx;
    ceno_hip_mle* m = nullptr;
    ~DevBuf() {
        if (m) ceno_hip_mle_free(ctx, m);
    }
    int alloc_words(size_t words) {  // a base-field table of at least `words` words from the library's pool
        int nv = 0;
        while (((size_t)1 << nv) < words) nv++;
        return ceno_hip_mle_alloc(ctx, nv, 0, &m);
    }
    uint64_t* ptr() const { return ceno_hip_mle_device_ptr(m); }
};

int hook_batch_codeword(void* self, int commit, int cls, const uint64_t* coeffs, uint64_t* dev_B, int log_h, int accumulate, ceno_hip_stream s) {
    DistOpen& D = *static_cast<DistOpen*>(self);
    if (commit < 0 || commit >= (int)D.commits.size()) return prover_set_error(CENO_HIP_ERR_STATE, "dist_basefold_open: commitment index");
    const DistCommit& K = D.commits[(size_t)commit];
    if (cls < 0 || cls >= (int)K.class_mats.size()) return prover_set_error(CENO_HIP_ERR_STATE, "dist_basefold_open: height class index");
    const size_t R = (size_t)1 << log_h;
    const bool shorter = log_h < D.k;  // fewer rows than ranks: nothing is sharded, every rank batches the whole codeword itself
    const size_t Rl = shorter ? R : R / (size_t)D.W;
    DevBuf loc{D.ctx}, both{D.ctx};
    if (int rc = loc.alloc_words(2 * Rl)) return fail_ctx(D.ctx, rc);
    // this rank's rows of every matrix of the class: the class's columns are its matrices' columns back to back (coeffs in that order)
    size_t c0 = 0;
    bool first = true;
    for (int m : K.class_mats[(size_t)cls]) {
        int rc = ceno_hip_batch_columns(D.ctx, K.local_cw_rows[m], Rl, (int)K.width_of[(size_t)m], coeffs + 2 * c0, loc.ptr(), first ? 0 : 1, s);
        if (rc) return fail_ctx(D.ctx, rc);
        c0 += K.width_of[(size_t)m];
        first = false;
    }
    if (shorter) {
        if (!accumulate) {
            if (hipMemcpyAsync(dev_B, loc.ptr(), R * 16, hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess) return prover_set_error(CENO_HIP_ERR_HIP, "dist_basefold_open: copy failed");
        } else {
            if (int rc = both.alloc_words(4 * R)) return fail_ctx(D.ctx, rc);
            if (hipMemcpyAsync(both.ptr(), dev_B, R * 16, hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess ||
                hipMemcpyAsync(both.ptr() + 2 * R, loc.ptr(), R * 16, hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess)
                return prover_set_error(CENO_HIP_ERR_HIP, "dist_basefold_open: copy failed");
            if (int rc = ceno_hip_ext_sum_blocks(D.ctx, both.ptr(), 2, R, dev_B, s)) return fail_ctx(D.ctx, rc);
        }
    } else if (!accumulate) {
        if (int rc = dist_allgather_device(D.comm, loc.ptr(), 2 * Rl, dev_B, (hipStream_t)s)) return prover_set_error(rc, ceno_dist_last_error());
    } else {
        // a second commitment with a class of this height (witness + fixed traces of one size): B += the gathered codeword — the running B and the
        // gathered block side by side, added mod p
        if (int rc = both.alloc_words(4 * R)) return fail_ctx(D.ctx, rc);
        if (hipMemcpyAsync(both.ptr(), dev_B, R * 16, hipMemcpyDeviceToDevice, (hipStream_t)s) != hipSuccess)
            return prover_set_error(CENO_HIP_ERR_HIP, "dist_basefold_open: copy failed");
        if (int rc = dist_allgather_device(D.comm, loc.ptr(), 2 * Rl, both.ptr() + 2 * R, (hipStream_t)s)) return prover_set_error(rc, ceno_dist_last_error());
        if (int rc = ceno_hip_ext_sum_blocks(D.ctx, both.ptr(), 2, R, dev_B, s)) return fail_ctx(D.ctx, rc);
    }
    if (hipStreamSynchronize((hipStream_t)s) != hipSuccess) return prover_set_error(CENO_HIP_ERR_HIP, "dist_basefold_open: sync failed");
    return 0;
}

int hook_batch_trace(void* self, int commit, int mat, const uint64_t* coeffs, uint64_t* dev_F, ceno_hip_stream s) {
    DistOpen& D = *static_cast<DistOpen*>(self);
    const DistCommit& K = D.commits[(size_t)commit];
    const size_t rows = (size_t)1 << K.log_rows_of[(size_t)mat];
    DevBuf part{D.ctx}, all{D.ctx};
    // the ranks' partial sums are gathered and added in CHUNKS of at most 2^20 rows: the gather scratch stays at W x 16 MB whatever the height of the
    // matrix (whole: 2 x rows x W extension words rounded up to a power of two — ~2 GB at 2^24 rows and 8 ranks, which no booking estimate knew of)
    const size_t chunk = std::min<size_t>(rows, (size_t)1 << 20);
    if (int rc = part.alloc_words(2 * rows)) return fail_ctx(D.ctx, rc);
    if (int rc = all.alloc_words(2 * chunk * (size_t)D.W)) return fail_ctx(D.ctx, rc);
    size_t col0 = 0;
    for (int g = 0; g < D.rank; g++) col0 += (size_t)K.widths[(size_t)mat * D.W + g];
    const int mine = K.widths[(size_t)mat * D.W + D.rank];
    if (mine > 0) {
        int rc = ceno_hip_batch_columns(D.ctx, K.local_trace_cols[mat], rows, mine, coeffs + 2 * col0, part.ptr(), 0, s);
        if (rc) return fail_ctx(D.ctx, rc);
    } else if (hipMemsetAsync(part.ptr(), 0, rows * 16, (hipStream_t)s) != hipSuccess) {
        return prover_set_error(CENO_HIP_ERR_HIP, "dist_basefold_open: memset failed");
    }
    for (size_t off = 0; off < rows; off += chunk) {
        if (int rc = dist_allgather_device(D.comm, part.ptr() + 2 * off, 2 * chunk, all.ptr(), (hipStream_t)s)) return prover_set_error(rc, ceno_dist_last_error());
        if (int rc = ceno_hip_ext_sum_blocks(D.ctx, all.ptr(), D.W, chunk, dev_F + 2 * off, s)) return fail_ctx(D.ctx, rc);
    }
    if (hipStreamSynchronize((hipStream_t)s) != hipSuccess) return prover_set_error(CENO_HIP_ERR_HIP, "dist_basefold_open: sync failed");
    return 0;
}

size_t hook_opening_words(void* self, int commit) {
    DistOpen& D = *static_cast<DistOpen*>(self);
    const DistCommit& K = D.commits[(size_t)commit];
    return K.total_width + 4 * (size_t)(K.log_rows + D.log_blowup);
}

int hook_mmcs_open(void* self, int commit, const uint64_t* idx, const uint64_t*, size_t n, int shift, uint64_t* dev_out, size_t per_q, ceno_hip_stream s) {
    DistOpen& D = *static_cast<DistOpen*>(self);
    const DistCommit& K = D.commits[(size_t)commit];
    hipStream_t st = (hipStream_t)s;
    const int H = K.log_rows + D.log_blowup, hl = H - D.k;  // levels of a rank's sub-tree
    // the sub-trees hold the matrices with at least W codeword rows, the replicated top tree the shorter ones (their rows, then its log2 W levels)
    const size_t tall_width = K.total_width - K.short_width;
    const size_t per_loc = tall_width + 4 * (size_t)hl, per_top = K.short_width + 4 * (size_t)D.k;
    if (per_q != per_loc + per_top) return prover_set_error(CENO_HIP_ERR_STATE, "dist_basefold_open: opening size mismatch");
    std::vector<uint64_t> loc_idx(n), own(n);
    for (size_t q = 0; q < n; q++) {
        const uint64_t row = idx[q] >> shift;
        own[q] = row >> hl;
        loc_idx[q] = row & (((uint64_t)1 << hl) - 1);
    }
    DevBuf buf{D.ctx};
    if (int rc = buf.alloc_words(2 * n + n * per_loc + n * per_top)) return fail_ctx(D.ctx, rc);
    uint64_t *d_loc = buf.ptr(), *d_own = d_loc + n, *d_ans = d_own + n, *d_top = d_ans + n * per_loc;
    if (hipMemcpyAsync(d_loc, loc_idx.data(), n * 8, hipMemcpyHostToDevice, st) != hipSuccess || hipMemcpyAsync(d_own, own.data(), n * 8, hipMemcpyHostToDevice, st) != hipSuccess)
        return prover_set_error(CENO_HIP_ERR_HIP, "dist_basefold_open: index upload failed");
    // every rank opens its own sub-tree at the local index of every query (only the owner's answer is used) ...
    int rc = ceno_hip_mmcs_open_batch(D.ctx, K.subtree, d_loc, n, 0, d_ans, per_loc, s);
    // ... and the replicated top tree at the owner's leaf: the rows of the short matrices + the top log2 W levels
    if (!rc && K.short_width > 0) rc = ceno_hip_mmcs_open_batch(D.ctx, K.top, d_own, n, 0, d_top, per_top, s);
    else if (!rc && D.k > 0) rc = ceno_hip_merkle_open_batch(D.ctx, K.top, d_own, n, 0, d_top, s);
    if (rc) return fail_ctx(D.ctx, rc);
    std::vector<uint64_t> mine(n * per_loc), top(n * per_top), all((size_t)D.W * n * per_loc), out(n * per_q);
    if (hipMemcpyAsync(mine.data(), d_ans, mine.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
        (per_top && hipMemcpyAsync(top.data(), d_top, top.size() * 8, hipMemcpyDeviceToHost, st) != hipSuccess) || hipStreamSynchronize(st) != hipSuccess)
        return prover_set_error(CENO_HIP_ERR_HIP, "dist_basefold_open: answer download failed");
    if (int rc2 = dist_allgather_words(D.comm, mine.data(), mine.size(), all.data(), st)) return prover_set_error(rc2, ceno_dist_last_error());
    for (size_t q = 0; q < n; q++) {
        // [row of every matrix in the caller's order][path of the sub-tree][path of the top]
        const uint64_t* tall = all.data() + (size_t)own[q] * n * per_loc + q * per_loc;
        const uint64_t* shrt = top.data() + q * per_top;
        uint64_t* o = out.data() + q * per_q;
        for (int m = 0; m < K.n_mats; m++) {
            const size_t w = K.width_of[(size_t)m];
            const uint64_t*& src = K.is_short[(size_t)m] ? shrt : tall;
            memcpy(o, src, w * 8);
            o += w;
            src += w;
        }
        memcpy(o, tall, 4 * (size_t)hl * 8);
        o += 4 * (size_t)hl;
        memcpy(o, shrt, 4 * (size_t)D.k * 8);
    }
    if (hipMemcpyAsync(dev_out, out.data(), out.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return prover_set_error(CENO_HIP_ERR_HIP, "dist_basefold_open: answer upload failed");
    return 0;
}

}  // namespace

extern "C" {

int ceno_dist_basefold_open_commits(ceno_hip_ctx* ctx, ceno_dist_comm* comm, int n_commits, const ceno_dist_commit_view* views, int log_blowup,
                                    const uint64_t* const* points, const uint64_t* const* evals, int n_queries, int pow_bits, ceno_transcript* tr,
                                    ceno_hip_stream s, uint64_t* out_proof) {
    if (!ctx || !comm || n_commits < 1 || !views || !points || !evals || !tr || !out_proof) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_basefold_open: bad arguments");
    DistOpen D;
    D.ctx = ctx;
    D.comm = comm;
    D.W = dist_comm_world(comm);
    D.rank = dist_comm_rank(comm);
    D.k = 0;
    while ((1 << D.k) < D.W) D.k++;
    if ((1 << D.k) != D.W) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_basefold_open: the number of ranks must be a power of two");
    D.log_blowup = log_blowup;
    D.commits.resize((size_t)n_commits);
    // the SHAPE of every commitment for the single-device code (no tables, no tree): height classes tallest first, the matrices of a class in
    // the caller's order — what commit_traces builds (commit.cpp)
    std::vector<ceno_pcs_data> shapes((size_t)n_commits);
    std::vector<ceno_pcs_data*> shape_ptrs;
    for (int c = 0; c < n_commits; c++) {
        const ceno_dist_commit_view& V = views[c];
        if (V.n_mats < 1 || !V.log_rows || !V.widths || !V.local_trace_cols || !V.local_cw_rows || !V.subtree)
            return prover_set_error(CENO_HIP_ERR_INVALID, "dist_basefold_open: bad commitment view");
        if (D.W > 1 && !V.top) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_basefold_open: the replicated top tree is missing");
        DistCommit& K = D.commits[(size_t)c];
        K.n_mats = V.n_mats;
        for (int m = 0; m < V.n_mats; m++) {
            // (a trace has at least two rows: next_pow2_instance_padding, ceno_zkvm/src/scheme/hal.rs:127-128)
            if (V.log_rows[m] < 1 || V.log_rows[m] > 40) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_basefold_open: bad matrix height");
            K.log_rows = std::max(K.log_rows, V.log_rows[m]);
        }
        if (K.log_rows + log_blowup < D.k)
            return prover_set_error(CENO_HIP_ERR_UNSUPPORTED, "dist_basefold_open: a commitment whose tallest codeword has fewer rows than there are ranks");
        K.log_rows_of.assign(V.log_rows, V.log_rows + V.n_mats);
        K.widths = V.widths;
        K.local_trace_cols = V.local_trace_cols;
        K.local_cw_rows = V.local_cw_rows;
        K.subtree = V.subtree;
        K.top = V.top;
        ceno_pcs_data& shape = shapes[(size_t)c];
        shape.log_blowup = log_blowup;
        std::vector<int> heights(V.log_rows, V.log_rows + V.n_mats);
        std::sort(heights.begin(), heights.end(), [](int a, int b) { return a > b; });
        heights.erase(std::unique(heights.begin(), heights.end()), heights.end());
        for (int h : heights) {
            ceno_pcs_data::Class C;
            C.log_rows = h;
            shape.classes.push_back(C);
            K.class_mats.emplace_back();
        }
        for (int m = 0; m < V.n_mats; m++) {
            size_t w = 0;
            for (int g = 0; g < D.W; g++) w += (size_t)V.widths[(size_t)m * D.W + g];
            if (w < 1) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_basefold_open: a matrix without columns");
            const int cls = (int)(std::find(heights.begin(), heights.end(), V.log_rows[m]) - heights.begin());
            ceno_pcs_data::Class& C = shape.classes[(size_t)cls];
            ceno_pcs_data::Mat M;
            M.rows = (size_t)1 << V.log_rows[m];
            M.width = w;
            M.log_rows = V.log_rows[m];
            M.cls = cls;
            M.col0 = C.width;
            C.width += w;
            shape.mats.push_back(M);
            K.class_mats[(size_t)cls].push_back(m);
            K.width_of.push_back(w);
            K.total_width += w;
            K.is_short.push_back(V.log_rows[m] + log_blowup < D.k ? 1 : 0);
            if (K.is_short.back()) K.short_width += w;
        }
        shape_ptrs.push_back(&shape);
    }
    // (CENO_DIST_OPEN_PIPELINE=1 / 0 overrides the choice: A/B and the stress run that shows what the serial rounds are for, tools/dev/open_ranks_stress.sh)
    const char* pe = getenv("CENO_DIST_OPEN_PIPELINE");
    // pipelined rounds only when no two ranks share a device — asked of the placement (host name + PCI bus id of every rank), not of the transports:
    // a communicator that has RCCL AND a shared segment, or a one-device RCCL set-up, must not get the pipelined path by accident
    const bool shared = dist_comm_ranks_share_device(comm, (hipStream_t)s);   // (collective: asked on every rank, whatever the override says)
    const bool serial = pe ? atoi(pe) == 0 : shared;
    BasefoldOpenHook hook{&D, serial, hook_batch_codeword, hook_batch_trace, hook_opening_words, hook_mmcs_open};
    return basefold_open_hooked(ctx, shape_ptrs.data(), n_commits, points, evals, n_queries, pow_bits, tr, s, out_proof, &hook);
}

int ceno_dist_basefold_open_mmcs(ceno_hip_ctx* ctx, ceno_dist_comm* comm, int n_mats, const int* log_rows, const int* widths, int log_blowup,
                                 const uint64_t* const* local_trace_cols, const uint64_t* const* local_cw_rows, ceno_hip_merkle* subtree,
                                 ceno_hip_merkle* top, const uint64_t* const* points, const uint64_t* const* evals, int n_queries, int pow_bits,
                                 ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_proof) {
    ceno_dist_commit_view V{n_mats, log_rows, widths, local_trace_cols, local_cw_rows, subtree, top};
    return ceno_dist_basefold_open_commits(ctx, comm, 1, &V, log_blowup, points, evals, n_queries, pow_bits, tr, s, out_proof);
}

int ceno_dist_basefold_open(ceno_hip_ctx* ctx, ceno_dist_comm* comm, int n_mats, int log_rows, const int* widths, int log_blowup,
                            const uint64_t* const* local_trace_cols, const uint64_t* const* local_cw_rows, ceno_hip_merkle* subtree,
                            ceno_hip_merkle* top, const uint64_t* const* points, const uint64_t* const* evals, int n_queries, int pow_bits,
                            ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_proof) {
    if (n_mats < 1) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_basefold_open: bad arguments");
    std::vector<int> lr((size_t)n_mats, log_rows);
    return ceno_dist_basefold_open_mmcs(ctx, comm, n_mats, lr.data(), widths, log_blowup, local_trace_cols, local_cw_rows, subtree, top, points, evals, n_queries,
                                        pow_bits, tr, s, out_proof);
}

}  // extern "C"
