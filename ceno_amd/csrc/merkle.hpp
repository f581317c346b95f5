// Merkle tree handle and helpers shared by poseidon2.hip (trace commitment) and basefold.hip (commit phase).
#pragma once
#include "common.hpp"
#include "poseidon2.hpp"

struct PoseidonParams {
    p2::Params p;
};

struct ceno_hip_merkle {
    int log_rows = 0;
    std::vector<uint64_t*> levels;   // levels[0] = 2^log_rows leaf digests (4 words each) ... levels[log_rows] = root
    uint64_t* h_root = nullptr;      // pinned host copy of the root, written by the kernel that computes it (no D2H blit)
    uint64_t* d_root_view = nullptr; // device view of h_root
    bool root_on_host = false;       // the tree-top kernel has been told to write h_root
};

int get_params(ceno_hip_ctx* ctx, const p2::Params** out);
int merkle_alloc(ceno_hip_ctx* ctx, int log_rows, ceno_hip_merkle** out);
int merkle_build_upper(ceno_hip_ctx* ctx, ceno_hip_merkle* t, hipStream_t st);  // levels 1.. from the leaf digests in levels[0]
void merkle_release(ceno_hip_ctx* ctx, ceno_hip_merkle* t);
