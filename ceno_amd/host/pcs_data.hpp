// PcsData of the trace commitment, shared by commit.cpp (commit_traces) and basefold.cpp (batch open).
#pragma once
#include <hip/hip_runtime_api.h>

#include <vector>

#include "../../include/ceno_prover.h"

// One commitment = ONE mixed-height Merkle tree over the codewords of all its matrices (PCS::batch_commit,
// ceno_zkvm/src/scheme/cpu/mod.rs:559-584).  Matrices of equal height form a CLASS whose traces / codewords are stored back to
// back in one column-major allocation (equal row counts: concatenating matrices = concatenating columns), so a class is
// encoded, hashed and batched as ONE wide matrix.
struct ceno_pcs_data {
    struct Class {
        int log_rows = 0;    // of the trace (codeword: + log_blowup)
        size_t width = 0;    // columns of all its matrices
        ceno_hip_mle* trace = nullptr;     // column-major, raw buffer (base words)
        ceno_hip_mle* codeword = nullptr;  // column-major codewords
    };
    struct Mat {
        size_t rows = 0, width = 0;  // padded rows
        int log_rows = 0;
        int cls = 0;                 // index into classes
        size_t col0 = 0;             // first column inside the class
    };
    std::vector<Mat> mats;           // caller's order (the order of the BTreeMap of traces)
    std::vector<Class> classes;      // tallest first
    ceno_hip_merkle* tree = nullptr;
    int log_blowup = 0;

    const uint64_t* trace_ptr(int m) const { return ceno_hip_mle_device_ptr(classes[mats[m].cls].trace) + mats[m].col0 * mats[m].rows; }
    const uint64_t* codeword_ptr(int m) const { return ceno_hip_mle_device_ptr(classes[mats[m].cls].codeword) + mats[m].col0 * (mats[m].rows << log_blowup); }
    int max_log_rows() const { return classes.empty() ? 0 : classes[0].log_rows; }
};

// pooled pair of auxiliary streams per device (highest / lowest priority; basefold.cpp): acquire, use, release
bool ceno_aux_streams_acquire(ceno_hip_ctx* ctx, hipStream_t out[2], int* device);
void ceno_aux_streams_release(hipStream_t s[2], int device);
