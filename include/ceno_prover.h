/*
 * ceno_prover.h — C++ host layer (libceno_prover.so) that mirrors the reference's operator
 * interface for the hot path ON TOP of the device C ABI (ceno_hip.h).  The reference host code is
 * Rust; this image has no Rust toolchain, so the same control flow is written in C++ and exported
 * with a C ABI so that tests and bench.py can drive it (INTEGRATION.md shows the Rust shim that
 * replaces this layer when a toolchain is available).
 *
 *   ceno_prover_sumcheck_prove      <->  IOPProverState::prove            (EXT sumcheck; call sites
 *                                        gkr_iop/src/gkr/layer/cpu/mod.rs:217-227, scheme/cpu/mod.rs:490-493)
 *   ceno_prover_tower_create_proof  <->  CpuTowerProver::create_proof     (scheme/cpu/mod.rs:346-554)
 *   ceno_prover_prove_tower_relation<->  TowerProver::prove_tower_relation(scheme/cpu/mod.rs:765-797)
 *   ceno_transcript                 <->  transcript::Transcript<E>        (EXT; script restated in
 *                                        ceno_recursion_v2/src/tower/mod.rs:1541-1646)
 */
#ifndef CENO_PROVER_H
#define CENO_PROVER_H

#include "ceno_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Transcript as seen by the prover loops: append a byte label (`append_message`), append an
 * extension element (`append_field_element_ext`), squeeze an extension challenge. */
typedef struct ceno_transcript {
    void (*append_label)(void* self, const uint8_t* bytes, size_t n);
    void (*append_ext)(void* self, const uint64_t* e2);
    void (*sample_ext)(void* self, uint64_t* out2);
    void* self;
    void (*destroy)(void* self);
} ceno_transcript;

/* deterministic data-dependent stand-in (SplitMix64 chaining) — identical to the oracle's stub so
 * that parity tests compare complete proofs; NOT the reference's Poseidon2 transcript. */
ceno_transcript* ceno_transcript_stub_new(uint64_t seed);
/* Poseidon2-Goldilocks duplex challenger (width 8, rate 4) run on the host.  PARITY UNPINNED:
 * constants / label encoding of the reference's BasicTranscript live in EXT crates. */
ceno_transcript* ceno_transcript_poseidon2_new(const uint8_t* label, size_t n);
void ceno_transcript_free(ceno_transcript* t);
/* convenience for bindings that cannot call through the function-pointer table */
void ceno_transcript_append_label(ceno_transcript* t, const uint8_t* bytes, size_t n);
void ceno_transcript_append_ext(ceno_transcript* t, const uint64_t* e2);
void ceno_transcript_sample_ext(ceno_transcript* t, uint64_t* out2);

/* IOPProverState::prove: appends n and d (usize le-bytes), then per round the d evaluations and
 * the label "Internal round", samples the challenge.  out_msgs: n*d ext, out_challenges: n ext,
 * out_final_evals: num_mles ext. */
int ceno_prover_sumcheck_prove(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan,
                               ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_msgs, uint64_t* out_challenges,
                               uint64_t* out_final_evals);
/* same loop over an already begun sumcheck handle (consumes rounds 0..n-1 and finishes) */
int ceno_prover_sumcheck_run(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int num_vars, int degree, int num_mles,
                             ceno_transcript* tr, uint64_t* out_msgs, uint64_t* out_challenges, uint64_t* out_final_evals);

/* TowerProofs (ceno_zkvm/src/structs.rs:87-101) flattened:
 *   msgs: for tower round r = 1..R (R = max_num_vars - 1): r rounds x 3 ext, concatenated
 *   prod_evals[spec][r-1][2], logup_evals[spec][r-1][4] (zero where the spec is inactive)
 *   point: final rt, max_num_vars ext */
typedef struct ceno_tower_proof {
    int num_rounds;
    uint64_t* msgs;
    uint64_t* prod_evals;
    uint64_t* logup_evals;
    uint64_t* point;
} ceno_tower_proof;
size_t ceno_tower_msgs_words(int max_num_vars);

int ceno_prover_tower_create_proof(ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup,
                                   int n_logup, ceno_transcript* tr, ceno_hip_stream s, ceno_tower_proof* out);

/* prove_tower_relation: append every out-eval (r, w, lk order) to the transcript, then create_proof.
 * out_evals: (n_prod*2 + n_logup*4) ext in that order. */
int ceno_prover_prove_tower_relation(ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup,
                                     int n_logup, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_evals,
                                     ceno_tower_proof* out);

const char* ceno_prover_last_error(void);

/* ---- hypercube-sharded sumcheck over the GPUs of one node (ceno_amd/host/dist.cpp) ----
 * One process per GPU; the 128-byte RCCL unique id is created on rank 0 and distributed by the launcher
 * (torch.distributed broadcast).  See DESIGN.md section 6. */
typedef struct ceno_dist_comm ceno_dist_comm;
int ceno_dist_unique_id(uint8_t* out128);
int ceno_dist_comm_init(int world, int rank, const uint8_t* id128, ceno_dist_comm** out);
void ceno_dist_comm_destroy(ceno_dist_comm* c);
int ceno_dist_sumcheck_prove(ceno_hip_ctx* ctx, ceno_dist_comm* c, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan_local,
                             int n_total, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_msgs, uint64_t* out_challenges,
                             uint64_t* out_final_evals);
const char* ceno_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
