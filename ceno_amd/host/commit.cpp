// commit_traces — host control flow of `TraceCommitter::commit_traces` (ceno_zkvm/src/scheme/cpu/mod.rs:559-584,
// GPU arm scheme/gpu/mod.rs:1519-1660) over the device C ABI.  See include/ceno_prover.h.
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/ceno_prover.h"

int prover_set_error(int code, const char* msg);  // prover.cpp

#include "pcs_data.hpp"

// (a negative count that crossed the ABI as size_t must not reach the padding loop)
static constexpr size_t MAX_ROWS = (size_t)1 << 40, MAX_WIDTH = (size_t)1 << 20;

static int ceil_log2_sz(size_t x) {
    int l = 0;
    while (((size_t)1 << l) < x) l++;
    return l;
}

extern "C" {

void ceno_pcs_data_free(ceno_hip_ctx* ctx, ceno_pcs_data* d) {
    if (!d) return;
    if (d->tree) ceno_hip_merkle_free(ctx, d->tree);
    for (auto& c : d->classes) {
        if (c.codeword) ceno_hip_mle_free(ctx, c.codeword);
        if (c.trace) ceno_hip_mle_free(ctx, c.trace);
    }
    delete d;
}

static int commit_impl(ceno_hip_ctx* ctx, const uint64_t* const* host_row_major, const size_t* num_instances, const size_t* widths,
                       int n_matrices, int log_blowup, ceno_hip_stream s, ceno_pcs_data** out, bool on_device);

int ceno_prover_commit_traces(ceno_hip_ctx* ctx, const uint64_t* const* host_row_major, const size_t* num_instances, const size_t* widths,
                              int n_matrices, int log_blowup, ceno_hip_stream s, ceno_pcs_data** out) {
    return commit_impl(ctx, host_row_major, num_instances, widths, n_matrices, log_blowup, s, out, false);
}
int ceno_prover_commit_traces_dev(ceno_hip_ctx* ctx, const uint64_t* const* dev_row_major, const size_t* num_instances, const size_t* widths,
                                  int n_matrices, int log_blowup, ceno_hip_stream s, ceno_pcs_data** out) {
    return commit_impl(ctx, dev_row_major, num_instances, widths, n_matrices, log_blowup, s, out, true);
}

// the height classes of a commitment and their storage (trace + codeword per class): what commit_traces computes before it touches a
// matrix, and ALL that ceno_prover_commit_reserve does
static int commit_layout(ceno_hip_ctx* ctx, const size_t* num_instances, const size_t* widths, int n_matrices, int log_blowup, ceno_pcs_data* d) {
    d->log_blowup = log_blowup;
    d->mats.resize(n_matrices);
    // height classes, tallest first; inside a class the matrices keep the caller's order (what the tree's stable sort does)
    std::map<int, size_t, std::greater<int>> class_width;
    for (int i = 0; i < n_matrices; i++) {
        auto& M = d->mats[i];
        // next_pow2_instance_padding: at least 2 rows (ceno_zkvm/src/scheme/hal.rs:127-128)
        size_t rows = 2;
        while (rows < num_instances[i]) rows <<= 1;
        M.rows = rows;
        M.width = widths[i];
        M.log_rows = ceil_log2_sz(rows);
        M.col0 = class_width[M.log_rows];
        class_width[M.log_rows] += M.width;
    }
    std::map<int, int> class_of;
    for (auto& kv : class_width) {
        ceno_pcs_data::Class c;
        c.log_rows = kv.first;
        c.width = kv.second;
        const size_t words = ((size_t)1 << c.log_rows) * c.width;
        int rc = ceno_hip_mle_alloc(ctx, ceil_log2_sz(words), 0, &c.trace);
        if (!rc) rc = ceno_hip_mle_alloc(ctx, ceil_log2_sz(words << log_blowup), 0, &c.codeword);
        class_of[kv.first] = (int)d->classes.size();
        d->classes.push_back(c);
        if (rc) return rc;
    }
    for (int i = 0; i < n_matrices; i++) d->mats[i].cls = class_of[d->mats[i].log_rows];
    return 0;
}

// per CLASS one Reed-Solomon encoding of all its columns, then ONE launch that hashes the rows of every class and one tree
static int commit_encode_and_hash(ceno_hip_ctx* ctx, ceno_pcs_data* d, ceno_hip_stream s) {
    const int n_matrices = (int)d->mats.size();
    for (auto& c : d->classes) {
        int rc = ceno_hip_rs_encode(ctx, ceno_hip_mle_device_ptr(c.trace), c.log_rows, (int)c.width, d->log_blowup, ceno_hip_mle_device_ptr(c.codeword), s);
        if (rc) return rc;
    }
    std::vector<const uint64_t*> ptrs(n_matrices);
    std::vector<int> lr(n_matrices), w(n_matrices);
    for (int i = 0; i < n_matrices; i++) {
        ptrs[i] = d->codeword_ptr(i);
        lr[i] = d->mats[i].log_rows + d->log_blowup;
        w[i] = (int)d->mats[i].width;
    }
    return ceno_hip_mmcs_commit(ctx, ptrs.data(), lr.data(), w.data(), n_matrices, s, &d->tree);
}

static int commit_impl(ceno_hip_ctx* ctx, const uint64_t* const* host_row_major, const size_t* num_instances, const size_t* widths,
                       int n_matrices, int log_blowup, ceno_hip_stream s, ceno_pcs_data** out, bool on_device) {
    if (!ctx || !host_row_major || !num_instances || !widths || !out || n_matrices < 1 || log_blowup < 0)
        return prover_set_error(CENO_HIP_ERR_INVALID, "bad commit_traces arguments");
    if (!s) return prover_set_error(CENO_HIP_ERR_INVALID, "commit_traces needs an explicit stream (ceno_hip_stream_create)");
    for (int i = 0; i < n_matrices; i++)
        if (!host_row_major[i] || widths[i] < 1 || num_instances[i] < 1) return prover_set_error(CENO_HIP_ERR_INVALID, "commit_traces: empty matrix");
    for (int i = 0; i < n_matrices; i++)
        if (num_instances[i] > MAX_ROWS || widths[i] > MAX_WIDTH) return prover_set_error(CENO_HIP_ERR_INVALID, "commit_traces: matrix larger than 2^40 rows / 2^20 columns");
    auto* d = new ceno_pcs_data();
    // Everything is queued on the caller's stream, in order behind whatever produced a device-resident input, and nothing
    // waits until the end: per matrix pad / copy + transpose into its height class, per CLASS one Reed-Solomon encoding of all
    // its columns, then ONE launch that hashes the rows of every class and one tree (ceno_hip_mmcs_commit).
    (void)ceno_hip_stream_bind(ctx, s);
    std::vector<ceno_hip_mle*> stagings;
    auto drain = [&]() {
        int rc = ceno_hip_stream_sync(ctx, s);
        for (auto* m : stagings) ceno_hip_mle_free(ctx, m);
        stagings.clear();
        return rc;
    };
    auto bail = [&](int rc, const char* msg) {
        (void)drain();
        ceno_pcs_data_free(ctx, d);
        return prover_set_error(rc, msg);
    };
    if (int rc = commit_layout(ctx, num_instances, widths, n_matrices, log_blowup, d)) return bail(rc, ceno_hip_last_error(ctx));
    hipStream_t st = (hipStream_t)s;
    for (int i = 0; i < n_matrices; i++) {
        auto& M = d->mats[i];
        const size_t words = M.rows * M.width;
        // a device-resident matrix that already has all `rows` rows is transposed straight out of the caller's buffer
        const bool direct = on_device && num_instances[i] == M.rows;
        ceno_hip_mle* staging = nullptr;
        if (!direct) {
            int rc = ceno_hip_mle_alloc(ctx, ceil_log2_sz(words), 0, &staging);
            if (rc) return bail(rc, ceno_hip_last_error(ctx));
            stagings.push_back(staging);
        }
        const uint64_t* d_stage = direct ? host_row_major[i] : ceno_hip_mle_device_ptr(staging);
        hipError_t e = hipSuccess;
        if (!direct) {
            uint64_t* dst = ceno_hip_mle_device_ptr(staging);
            const size_t used = num_instances[i] * M.width;
            if (used < words) e = hipMemsetAsync(dst + used, 0, (words - used) * 8, st);  // rows beyond num_instances are zero (InstancePaddingStrategy::Default)
            if (e == hipSuccess) e = hipMemcpyAsync(dst, host_row_major[i], used * 8, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
        }
        if (e != hipSuccess) return bail(CENO_HIP_ERR_HIP, hipGetErrorString(e));
        int rc = ceno_hip_transpose(ctx, d_stage, M.rows, M.width, const_cast<uint64_t*>(d->trace_ptr(i)), s);
        if (rc) return bail(rc, ceno_hip_last_error(ctx));
    }
    if (int rc = commit_encode_and_hash(ctx, d, s)) return bail(rc, ceno_hip_last_error(ctx));
    if (int rc = drain()) {
        ceno_pcs_data_free(ctx, d);
        return prover_set_error(rc, ceno_hip_last_error(ctx));
    }
    *out = d;
    return 0;
}

// ---- commit_traces for traces that are PRODUCED on the device, in two steps (include/ceno_prover.h) ----
int ceno_prover_commit_reserve(ceno_hip_ctx* ctx, const size_t* num_instances, const size_t* widths, int n_matrices, int log_blowup,
                               ceno_hip_stream s, ceno_pcs_data** out) {
    if (!ctx || !num_instances || !widths || !out || n_matrices < 1 || log_blowup < 0)
        return prover_set_error(CENO_HIP_ERR_INVALID, "bad commit_reserve arguments");
    if (!s) return prover_set_error(CENO_HIP_ERR_INVALID, "commit_reserve needs an explicit stream (ceno_hip_stream_create)");
    for (int i = 0; i < n_matrices; i++)
        if (widths[i] < 1 || num_instances[i] < 1 || num_instances[i] > MAX_ROWS || widths[i] > MAX_WIDTH)
            return prover_set_error(CENO_HIP_ERR_INVALID, "commit_reserve: empty matrix, or larger than 2^40 rows / 2^20 columns");
    auto* d = new ceno_pcs_data();
    (void)ceno_hip_stream_bind(ctx, s);
    if (int rc = commit_layout(ctx, num_instances, widths, n_matrices, log_blowup, d)) {
        std::string msg = ceno_hip_last_error(ctx);
        ceno_pcs_data_free(ctx, d);
        return prover_set_error(rc, msg.c_str());
    }
    *out = d;
    return 0;
}

uint64_t* ceno_pcs_data_trace_ptr(ceno_pcs_data* d, int matrix) {
    if (!d || matrix < 0 || matrix >= (int)d->mats.size()) return nullptr;
    return const_cast<uint64_t*>(d->trace_ptr(matrix));
}

size_t ceno_pcs_data_rows(const ceno_pcs_data* d, int matrix) {
    if (!d || matrix < 0 || matrix >= (int)d->mats.size()) return 0;
    return d->mats[matrix].rows;
}

int ceno_prover_commit_finish(ceno_hip_ctx* ctx, ceno_pcs_data* d, ceno_hip_stream s) {
    if (!ctx || !d || d->mats.empty()) return prover_set_error(CENO_HIP_ERR_INVALID, "bad commit_finish arguments");
    if (!s) return prover_set_error(CENO_HIP_ERR_INVALID, "commit_finish needs an explicit stream (ceno_hip_stream_create)");
    if (d->tree) return prover_set_error(CENO_HIP_ERR_STATE, "commit_finish: this commitment is already finished");
    (void)ceno_hip_stream_bind(ctx, s);
    int rc = commit_encode_and_hash(ctx, d, s);
    if (!rc) rc = ceno_hip_stream_sync(ctx, s);
    return rc ? prover_set_error(rc, ceno_hip_last_error(ctx)) : 0;
}

int ceno_pcs_data_num_matrices(const ceno_pcs_data* d) { return d ? (int)d->mats.size() : -1; }
int ceno_pcs_data_num_vars(const ceno_pcs_data* d, int matrix) {
    if (!d || matrix < 0 || matrix >= (int)d->mats.size()) return -1;
    return d->mats[matrix].log_rows;
}
int ceno_pcs_data_width(const ceno_pcs_data* d, int matrix) {
    if (!d || matrix < 0 || matrix >= (int)d->mats.size()) return -1;
    return (int)d->mats[matrix].width;
}

int ceno_pcs_data_root(ceno_hip_ctx* ctx, ceno_pcs_data* d, uint64_t* root4, ceno_hip_stream s) {
    if (!d || !d->tree || !root4) return prover_set_error(CENO_HIP_ERR_INVALID, "bad pcs_data_root arguments");
    int rc = ceno_hip_merkle_root(ctx, d->tree, root4, s);
    return rc ? prover_set_error(rc, ceno_hip_last_error(ctx)) : 0;
}

int ceno_pcs_data_witness_mle(ceno_hip_ctx* ctx, ceno_pcs_data* d, int matrix, size_t col, ceno_hip_mle** out) {
    if (!d || matrix < 0 || matrix >= (int)d->mats.size() || col >= d->mats[matrix].width) return prover_set_error(CENO_HIP_ERR_INVALID, "column out of range");
    auto& M = d->mats[matrix];
    int rc = ceno_hip_mle_wrap(ctx, const_cast<uint64_t*>(d->trace_ptr(matrix)) + col * M.rows, M.log_rows, 0, out);
    return rc ? prover_set_error(rc, ceno_hip_last_error(ctx)) : 0;
}

size_t ceno_pcs_data_opening_words(const ceno_pcs_data* d) { return d && d->tree ? ceno_hip_mmcs_opening_words(d->tree) : 0; }

int ceno_pcs_data_open(ceno_hip_ctx* ctx, ceno_pcs_data* d, size_t index, uint64_t* out, ceno_hip_stream s) {
    if (!d || !d->tree || !out) return prover_set_error(CENO_HIP_ERR_INVALID, "bad pcs_data_open arguments");
    const int H = d->max_log_rows() + d->log_blowup;
    if (index >= ((size_t)1 << H)) return prover_set_error(CENO_HIP_ERR_INVALID, "row index out of range");
    const size_t words = ceno_hip_mmcs_opening_words(d->tree);
    ceno_hip_mle* scratch = nullptr;
    (void)ceno_hip_stream_bind(ctx, s);
    int rc = ceno_hip_mle_alloc(ctx, ceil_log2_sz(words + 1), 0, &scratch);
    if (rc) return prover_set_error(rc, ceno_hip_last_error(ctx));
    uint64_t* dv = ceno_hip_mle_device_ptr(scratch);
    hipStream_t st = (hipStream_t)s;
    const uint64_t idx = index;
    hipError_t e = hipMemcpyAsync(dv, &idx, 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        rc = ceno_hip_mmcs_open_batch(ctx, d->tree, dv, 1, 0, dv + 1, words, s);
        if (!rc) e = hipMemcpyAsync(out, dv + 1, words * 8, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    ceno_hip_mle_free(ctx, scratch);
    if (rc) return prover_set_error(rc, ceno_hip_last_error(ctx));
    if (e != hipSuccess) return prover_set_error(CENO_HIP_ERR_HIP, hipGetErrorString(e));
    return 0;
}

}  // extern "C"
