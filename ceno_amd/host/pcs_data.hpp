// PcsData of the trace commitment, shared by commit.cpp (commit_traces) and basefold.cpp (batch open).
#pragma once
#include <hip/hip_runtime_api.h>

#include <vector>

#include "../../include/ceno_prover.h"

struct ceno_pcs_data {
    struct Mat {
        size_t rows = 0, width = 0;  // padded rows
        int log_rows = 0;
        ceno_hip_mle* trace = nullptr;     // column-major trace, raw buffer (base words)
        ceno_hip_mle* codeword = nullptr;  // column-major codewords
        ceno_hip_merkle* tree = nullptr;
    };
    std::vector<Mat> mats;
    int log_blowup = 0;
};

// pooled pair of auxiliary streams per device (highest / lowest priority; basefold.cpp): acquire, use, release
bool ceno_aux_streams_acquire(ceno_hip_ctx* ctx, hipStream_t out[2], int* device);
void ceno_aux_streams_release(hipStream_t s[2], int device);
