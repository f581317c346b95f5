// The tower prover as a RESUMABLE object (prover.cpp): CpuTowerProver::create_proof (ceno_zkvm/src/scheme/cpu/mod.rs:346-554) is a loop over the
// layers of the towers; the cohort driver (cohort.cpp) runs the first layers of MANY chips chip by chip, then the middle layers of all of them
// in lock-step in one launch per layer, then the rest chip by chip again — on the same state, the same transcript, the same output buffers.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/ceno_prover.h"
#include "../csrc/gl64.hpp"
#include "tower_hook.hpp"

struct HostTowerTop {
    std::vector<uint64_t> words;  // ceno_hip_tower_download_top layout
    int n_limbs = 0, n_layers = 0;
    const gl::E2* limb(int layer, int b) const {
        return reinterpret_cast<const gl::E2*>(words.data()) + (size_t)n_limbs * (((size_t)1 << layer) - 1) + ((size_t)b << layer);
    }
};

struct TowerProveState {
    ceno_hip_ctx* ctx = nullptr;
    ceno_hip_tower* const* prod = nullptr;
    int n_prod = 0;
    ceno_hip_tower* const* logup = nullptr;
    int n_logup = 0;
    ceno_transcript* tr = nullptr;
    ceno_hip_stream s = nullptr;
    ceno_tower_proof* out = nullptr;
    const TowerDistHook* hook = nullptr;
    int max_nv = 0, R = 0, n_alpha = 0, host_layers = 0;
    int round = 1;                        // the next layer to prove (1 .. R)
    std::vector<uint64_t> alpha, out_rt;  // this layer's alpha powers (n_prod + 2 n_logup ext) and point (round ext)
    size_t msg_off = 0;                   // words of out->msgs written so far
    gl::E2 claim{0, 0};                   // the sum the next layer's sumcheck proves (known from the layer before)
    bool have_claim = false;
    std::vector<HostTowerTop> top_prod, top_logup;
    int nv_of(const ceno_hip_tower* t) const;  // the GLOBAL height of a tower (the hook's when the large layers are sharded)
    bool done() const { return round > R; }
};
// alpha powers, "product_sum", the first point, the towers' small layers fetched; the state then stands before layer 1
int tower_state_init(TowerProveState& st, ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup, int n_logup,
                     ceno_transcript* tr, ceno_hip_stream s, ceno_tower_proof* out, const TowerDistHook* hook);
// layer st.round by the route the per-chip prover takes (host / sharded hook / device sumcheck), epilogue included
int tower_state_step(TowerProveState& st);
// what follows a layer's sumcheck whoever ran it (`chal`: round ext, `fin`: [eq, (a, b) per active product tower, (p1, p2, q1, q2) per active LogUp
// tower]; its messages are already at out->msgs + msg_off): the evaluations into the proof and the transcript, "merge", the next point and alpha
// powers, the next claim; st.round advances
int tower_state_layer_epilogue(TowerProveState& st, const uint64_t* chal, const uint64_t* fin);
void tower_state_finish(TowerProveState& st);  // the final point into the proof

// transcript steps of IOPProverState::prove as the layer sumchecks take them (prover.cpp)
void prover_tr_usize(ceno_transcript* t, uint64_t v);
gl::E2 prover_tr_round(ceno_transcript* t, const uint64_t* msg6);  // three message points, "Internal round", the challenge
