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
