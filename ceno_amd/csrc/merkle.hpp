// Merkle tree handle and helpers shared by poseidon2.hip (trace commitment) and basefold.hip (commit phase).
#pragma once
#include "common.hpp"
#include "poseidon2.cuh"

struct PoseidonParams {
    p2::Params p;
};

struct ceno_hip_merkle {
    int log_rows = 0;
    std::vector<uint64_t*> levels;   // levels[0] = 2^log_rows leaf digests (4 words each) ... levels[log_rows] = root
    uint64_t** all_ptrs = nullptr;   // device array of all level pointers (batched path gathers)
};

int get_params(ceno_hip_ctx* ctx, const p2::Params** out);
int merkle_alloc(ceno_hip_ctx* ctx, int log_rows, ceno_hip_merkle** out);
int merkle_build_upper(ceno_hip_ctx* ctx, ceno_hip_merkle* t, hipStream_t st);  // levels 1.. from the leaf digests in levels[0]
void merkle_release(ceno_hip_ctx* ctx, ceno_hip_merkle* t);
