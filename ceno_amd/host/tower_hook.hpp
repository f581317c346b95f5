// Hook of the tower prover for towers whose large layers are sharded across ranks (prover.cpp <-> dist_gkr.cpp).
#pragma once
#include <cstdint>

#include "../../include/ceno_prover.h"

struct TowerDistHook {
    int r_rep;              // rounds 1 .. r_rep run on the towers passed to the prover (replicated on every rank)
    const int* nv_global;   // [n_prod + n_logup]: number of variables of the GLOBAL towers
    // round `round` > r_rep: the whole layer sumcheck (prologue, `round` messages of 3 ext, `round` challenges, final evaluations in the order
    // [eq, (a, b) per active product tower, (p1, p2, q1, q2) per active LogUp tower]); alpha: n_prod + 2 n_logup ext of this round
    int (*layer)(void* self, int round, const uint64_t* out_rt, const uint64_t* alpha, ceno_transcript* tr, uint64_t* msgs, uint64_t* chal, uint64_t* fin);
    void* self;
};

// The rotation argument over ROW-SHARDED witness columns (prover.cpp prover_prove_rotation_sharded <-> dist_gkr.cpp): rank `rank` of `world` = 2^k
// holds the rows whose index bits [q, q + k) equal its number; `allgather` collects n_words words of every rank, rank-major.
struct RotationShard {
    int world, rank, k, q;
    int (*allgather)(void* self, const uint64_t* mine, size_t n_words, uint64_t* all);
    void* self;
};
int prover_prove_rotation_sharded(ceno_hip_ctx* ctx, ceno_hip_mle* const* wit_local, const int* source_idx, const int* target_idx, int n_pairs,
                                  int cyclic_subgroup_size, int cyclic_group_log2, const uint64_t* rt, int n, ceno_transcript* tr, ceno_hip_stream s,
                                  uint64_t* out_msgs, uint64_t* out_evals, uint64_t* out_origin, uint64_t* out_left, uint64_t* out_right,
                                  const RotationShard* sh);

// prove_batched_main_constraints over row-sharded tables in the same layout (main_constraints.cpp; sh == NULL: whole tables)
int prover_main_constraints_sharded(ceno_hip_ctx* ctx, const ceno_main_job* jobs, int n_jobs, const uint64_t* gc4, ceno_transcript* tr, ceno_hip_stream s,
                                    uint64_t* out_claimed_sum, uint64_t* out_msgs, uint64_t* out_global_rt, uint64_t* out_evals, int* out_num_vars,
                                    int* out_degree, const RotationShard* sh);
