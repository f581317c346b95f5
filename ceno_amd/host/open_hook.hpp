// Hook of the Basefold opening for a commitment whose data is sharded across ranks (basefold.cpp <-> dist_open.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/ceno_prover.h"

struct BasefoldOpenHook {
    void* self;
    // the ranks may share ONE device (virtual ranks of an in-process group, processes over the shared segment): the opening's sumcheck then runs
    // round by round instead of pipelined — queued round kernels that wait for their host occupy hardware queues the other ranks' streams share,
    // and the ranks wait for each other in the gathers (a multi-rank opening on 8 virtual ranks once lost three rounds to the 60 s give-up)
    bool serial_rounds;
    // B (+)= sum_c coeffs[c] * codeword column c of height class `cls` of commitment `commit`: the FULL batched codeword of 2^log_h extension
    // elements on this device (coeffs: 2 words per column of the class, in the class's column order)
    int (*batch_codeword)(void* self, int commit, int cls, const uint64_t* coeffs, uint64_t* dev_B_ext, int log_h, int accumulate, ceno_hip_stream s);
    // F = sum_c coeffs[c] * trace column c of matrix `mat`: the FULL batched polynomial (2^log_rows extension elements) on this device
    int (*batch_trace)(void* self, int commit, int mat, const uint64_t* coeffs, uint64_t* dev_F_ext, ceno_hip_stream s);
    // words of one opening of the commitment: sum of widths + 4 * log2(rows of the tallest codeword)
    size_t (*opening_words)(void* self, int commit);
    // MerkleTreeMmcs::open_batch at the indices (host copy idx, device copy dev_idx) >> shift: per query [rows of every matrix][path], stride per_q words
    int (*mmcs_open)(void* self, int commit, const uint64_t* idx, const uint64_t* dev_idx, size_t n, int shift, uint64_t* dev_out, size_t per_q, ceno_hip_stream s);
};
