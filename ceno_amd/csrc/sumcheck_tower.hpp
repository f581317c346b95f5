// Large rounds of a tower layer's sumcheck (sumcheck_tower.hip):  sum_x eq(x, rt) * G(x),
//   G = sum_i alpha_i a_i b_i + sum_k [ an_k (p1_k q2_k + p2_k q1_k) + ad_k q1_k q2_k ]      (scheme/cpu/mod.rs:417-494)
// One fused pass per round (fold every table with the previous challenge, write it, evaluate) and ONE evaluation point fewer than
// the message has: the eq table bound so far factors as  EQ_i[2y + b] = P_i * E_i[y] * eq(b, rt_i),  so the round polynomial is
//   p_i(X) = eq(X, rt_i) * q_i(X),   q_i(X) = sum_y (EQ_i[2y] + EQ_i[2y+1]) * G(X, y)   of degree 2,
// and the kernel produces q_i(1) and q_i's leading coefficient (round 0: q_0(0) too); the host derives q_i(0) from the running claim
// p_{i-1}(r_{i-1}) = (1 - rt_i) q_i(0) + rt_i q_i(1) and publishes p_i(1..3) — the same field elements the generic rounds produce.
#pragma once
#include "sumcheck_dev.hpp"

constexpr int TOWER_FAST_MAX_PROD = 3, TOWER_FAST_MAX_LOGUP = 2;
struct TowerCoef {
    gl::E2 prod[TOWER_FAST_MAX_PROD];        // alpha_i
    gl::E2 logup[TOWER_FAST_MAX_LOGUP][2];   // (an_k, ad_k)
};
bool tower_fast_shape(int n_prod, int n_logup);
// slots: eq, then (a, b) per product tower, then (p1, p2, q1, q2) per logup tower — all extension tables; round 0 reads slot.in as
// pairs, later rounds fold slot.in (four entries per pair) into slot.out.  Message words: q(1), leading coefficient and, in mode 0, q(0).
// mode 0: round 0 without a claim (three values), 1: round 0 under a claim the caller knows (two values), 2: a later round (fold, two values)
void launch_tower_round(ceno_hip_ctx* ctx, int n_prod, int n_logup, int mode, const MleSlot* slots, const TowerCoef& coef, size_t pairs, const Epilogue& ep,
                        unsigned grid, hipStream_t st);
