// The dense fused kernels (sumcheck_dense.hip): one product term of K <= 4 tables, the fold of round i fused with the evaluation of round i + 1.
// The host driver (sumcheck.hip) hands over the tables of one launch; everything else about a round (grid, epilogue, the challenge) is its own.
#pragma once
#include "sumcheck_dev.hpp"

struct DenseTables {
    const uint64_t* in[4];
    uint64_t* out[4];
};
// mode 0: accumulate only, ext input    1: accumulate only, base input    2: fold + accumulate, ext input    3: fold + accumulate, base input
void launch_dense_tables(ceno_hip_ctx* ctx, int K, int mode, const DenseTables& t, size_t pairs, gl::E2 r, const Epilogue& ep, unsigned grid, hipStream_t st);
