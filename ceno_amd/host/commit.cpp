// commit_traces — host control flow of `TraceCommitter::commit_traces` (ceno_zkvm/src/scheme/cpu/mod.rs:559-584,
// GPU arm scheme/gpu/mod.rs:1519-1660) over the device C ABI.  See include/ceno_prover.h.
#include <hip/hip_runtime_api.h>

#include <cstring>
#include <vector>

#include "../../include/ceno_prover.h"

int prover_set_error(int code, const char* msg);  // prover.cpp

#include "pcs_data.hpp"

static int ceil_log2_sz(size_t x) {
    int l = 0;
    while (((size_t)1 << l) < x) l++;
    return l;
}

extern "C" {

void ceno_pcs_data_free(ceno_hip_ctx* ctx, ceno_pcs_data* d) {
    if (!d) return;
    for (auto& m : d->mats) {
        if (m.tree) ceno_hip_merkle_free(ctx, m.tree);
        if (m.codeword) ceno_hip_mle_free(ctx, m.codeword);
        if (m.trace) ceno_hip_mle_free(ctx, m.trace);
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

static int commit_impl(ceno_hip_ctx* ctx, const uint64_t* const* host_row_major, const size_t* num_instances, const size_t* widths,
                       int n_matrices, int log_blowup, ceno_hip_stream s, ceno_pcs_data** out, bool on_device) {
    if (!ctx || !host_row_major || !num_instances || !widths || !out || n_matrices < 1 || log_blowup < 0)
        return prover_set_error(CENO_HIP_ERR_INVALID, "bad commit_traces arguments");
    if (!s) return prover_set_error(CENO_HIP_ERR_INVALID, "commit_traces needs an explicit stream (ceno_hip_stream_create)");
    auto* d = new ceno_pcs_data();
    d->log_blowup = log_blowup;
    d->mats.resize(n_matrices);
    // Nothing waits between matrices: every matrix is queued (pad / copy, transpose, encode, leaf hash, tree) as soon as the
    // previous one has been, alternately on the caller's stream and on a helper stream, so that the latency-bound tree tops of
    // one matrix run under the leaf hashing of the next (eight traces of a 2^20-cycle shard: 8.0 -> measured below).  ONE
    // wait per stream at the end: the caller's matrices and the staging buffers are only borrowed until then.
    ceno_hip_stream helper = nullptr;
    hipStream_t aux[2] = {nullptr, nullptr};
    int aux_dev = -1;
    if (n_matrices > 1 && !(getenv("CENO_COMMIT_ONE_STREAM") && atoi(getenv("CENO_COMMIT_ONE_STREAM")) != 0) &&
        ceno_aux_streams_acquire(ctx, aux, &aux_dev))
        helper = (ceno_hip_stream)aux[1];
    std::vector<ceno_hip_mle*> stagings;
    auto drain = [&]() {
        int rc = ceno_hip_stream_sync(ctx, s);
        if (helper) {
            const int rc2 = ceno_hip_stream_sync(ctx, helper);
            if (!rc) rc = rc2;
        }
        for (auto* m : stagings) ceno_hip_mle_free(ctx, m);
        stagings.clear();
        if (aux[0]) ceno_aux_streams_release(aux, aux_dev);
        aux[0] = aux[1] = nullptr;
        helper = nullptr;
        (void)ceno_hip_stream_bind(ctx, s);  // the thread is back on the caller's stream
        return rc;
    };
    for (int i = 0; i < n_matrices; i++) {
        auto& M = d->mats[i];
        ceno_hip_stream si = (helper && (i & 1)) ? helper : s;
        // next_pow2_instance_padding: at least 2 rows (ceno_zkvm/src/scheme/hal.rs:127-128)
        size_t rows = 2;
        while (rows < num_instances[i]) rows <<= 1;
        M.rows = rows;
        M.width = widths[i];
        M.log_rows = ceil_log2_sz(rows);
        const size_t words = rows * M.width, cw_words = words << log_blowup;
        // a device-resident matrix that already has all `rows` rows is transposed straight out of the caller's buffer
        const bool direct = on_device && num_instances[i] == rows;
        ceno_hip_mle* staging = nullptr;
        int rc = ceno_hip_stream_bind(ctx, si);  // the blocks allocated next are used on `si`
        if (!rc && !direct) rc = ceno_hip_mle_alloc(ctx, ceil_log2_sz(words), 0, &staging);
        if (staging) stagings.push_back(staging);
        if (!rc) rc = ceno_hip_mle_alloc(ctx, ceil_log2_sz(words), 0, &M.trace);
        if (!rc) rc = ceno_hip_mle_alloc(ctx, ceil_log2_sz(cw_words), 0, &M.codeword);
        if (rc) {
            (void)drain();
            ceno_pcs_data_free(ctx, d);
            return prover_set_error(rc, ceno_hip_last_error(ctx));
        }
        const uint64_t* d_stage = direct ? host_row_major[i] : ceno_hip_mle_device_ptr(staging);
        hipStream_t st = (hipStream_t)si;
        hipError_t e = hipSuccess;
        if (!direct) {
            uint64_t* dst = ceno_hip_mle_device_ptr(staging);
            const size_t used = num_instances[i] * M.width;
            if (used < words) e = hipMemsetAsync(dst + used, 0, (words - used) * 8, st);  // rows beyond num_instances are zero (InstancePaddingStrategy::Default)
            if (e == hipSuccess) e = hipMemcpyAsync(dst, host_row_major[i], used * 8, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st);
        }
        if (e != hipSuccess) {
            (void)drain();
            ceno_pcs_data_free(ctx, d);
            return prover_set_error(CENO_HIP_ERR_HIP, hipGetErrorString(e));
        }
        rc = ceno_hip_transpose(ctx, d_stage, rows, M.width, ceno_hip_mle_device_ptr(M.trace), si);
        if (!rc) rc = ceno_hip_rs_encode(ctx, ceno_hip_mle_device_ptr(M.trace), M.log_rows, (int)M.width, log_blowup, ceno_hip_mle_device_ptr(M.codeword), si);
        if (!rc) rc = ceno_hip_merkle_commit(ctx, ceno_hip_mle_device_ptr(M.codeword), M.log_rows + log_blowup, (int)M.width, si, &M.tree);
        if (rc) {
            (void)drain();
            ceno_pcs_data_free(ctx, d);
            return prover_set_error(rc, ceno_hip_last_error(ctx));
        }
    }
    if (int rc = drain()) {
        ceno_pcs_data_free(ctx, d);
        return prover_set_error(rc, ceno_hip_last_error(ctx));
    }
    *out = d;
    return 0;
}

int ceno_pcs_data_num_vars(const ceno_pcs_data* d, int matrix) {
    if (!d || matrix < 0 || matrix >= (int)d->mats.size()) return -1;
    return d->mats[matrix].log_rows;
}

int ceno_pcs_data_root(ceno_hip_ctx* ctx, ceno_pcs_data* d, int matrix, uint64_t* root4, ceno_hip_stream s) {
    if (!d || matrix < 0 || matrix >= (int)d->mats.size()) return prover_set_error(CENO_HIP_ERR_INVALID, "matrix out of range");
    int rc = ceno_hip_merkle_root(ctx, d->mats[matrix].tree, root4, s);
    return rc ? prover_set_error(rc, ceno_hip_last_error(ctx)) : 0;
}

int ceno_pcs_data_witness_mle(ceno_hip_ctx* ctx, ceno_pcs_data* d, int matrix, size_t col, ceno_hip_mle** out) {
    if (!d || matrix < 0 || matrix >= (int)d->mats.size() || col >= d->mats[matrix].width) return prover_set_error(CENO_HIP_ERR_INVALID, "column out of range");
    auto& M = d->mats[matrix];
    int rc = ceno_hip_mle_wrap(ctx, ceno_hip_mle_device_ptr(M.trace) + col * M.rows, M.log_rows, 0, out);
    return rc ? prover_set_error(rc, ceno_hip_last_error(ctx)) : 0;
}

int ceno_pcs_data_open_row(ceno_hip_ctx* ctx, ceno_pcs_data* d, int matrix, size_t index, uint64_t* row_out, uint64_t* path_out, ceno_hip_stream s) {
    if (!d || matrix < 0 || matrix >= (int)d->mats.size()) return prover_set_error(CENO_HIP_ERR_INVALID, "matrix out of range");
    auto& M = d->mats[matrix];
    const size_t cw_rows = M.rows << d->log_blowup;
    if (index >= cw_rows) return prover_set_error(CENO_HIP_ERR_INVALID, "row index out of range");
    hipStream_t st = (hipStream_t)s;
    const uint64_t* cw = ceno_hip_mle_device_ptr(M.codeword);
    // one strided 2D copy: `width` words, source pitch = one column
    hipError_t e = hipMemcpy2DAsync(row_out, 8, cw + index, cw_rows * 8, 8, M.width, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return prover_set_error(CENO_HIP_ERR_HIP, hipGetErrorString(e));
    int rc = ceno_hip_merkle_open(ctx, M.tree, index, path_out, s);  // synchronises
    return rc ? prover_set_error(rc, ceno_hip_last_error(ctx)) : 0;
}

}  // extern "C"
