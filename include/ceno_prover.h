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
 * extension element (`append_field_element_ext`), squeeze an extension challenge; and, for the PCS, the base-field side
 * of the challenger (p3-challenger `CanObserve<F>` / `CanSample<F>`; in-tree use ceno_recursion_v2/src/pcs/mod.rs:8125-8204):
 * observe one base element, sample bits of ONE base element, clone the challenger (grinding checks candidates on clones), and
 * export the duplex state so that the proof-of-work search can run on the device. */
typedef struct ceno_transcript {
    void (*append_label)(void* self, const uint8_t* bytes, size_t n);
    void (*append_ext)(void* self, const uint64_t* e2);
    void (*sample_ext)(void* self, uint64_t* out2);
    void* self;
    void (*destroy)(void* self);
    void (*append_base)(void* self, uint64_t v);   /* `append_field_element` (instance counts, circuit ids: prover.rs:682-689) */
    /* `CanSampleBits::sample_bits(bits)`: the low `bits` bits of the canonical value of ONE base-field sample (p3 DuplexChallenger;
     * pcs/mod.rs:8164-8204; a supertrait of the reference's `Transcript`, ceno_recursion_v2/src/tower/tower.rs:89-93).  bits >= 64
     * returns the whole sample (the library's own challengers only) */
    uint64_t (*sample_bits)(void* self, int bits);
    void* (*fork)(void* self);                     /* `challenger.clone()`: a new `self` with the same state, released by fork_free */
    void (*fork_free)(void* forked);
    /* Poseidon2 duplex state as 16 words: [sponge state 8][n pending inputs][pending inputs 4, zero padded][n outputs left]
     * [0][0]; the output buffer is state[0 .. n_out) and samples pop from its back (p3-challenger DuplexChallenger: the pub
     * fields sponge_state / input_buffer / output_buffer).  Returns CENO_TRANSCRIPT_DUPLEX8 (1), or 0 when the transcript
     * is of another kind (the proof-of-work search then runs through fork / append_base / sample_bits on the host).
     * May be NULL (same meaning as returning 0). */
    int (*export_state)(void* self, uint64_t* out16);
    int (*import_state)(void* self, const uint64_t* in16);   /* 0 = ok; may be NULL */
    /* the transcript's OWN `GrindingChallenger::grind(bits)` (a supertrait of the reference's `Transcript`, tower.rs:95-101): finds a
     * witness, observes it, samples the bits; returns the witness.  May be NULL.  Used when the state cannot be exported to the
     * device search (a Rust transcript whose challenger is private). */
    uint64_t (*grind)(void* self, int bits);
} ceno_transcript;
#define CENO_TRANSCRIPT_DUPLEX8 1

/* deterministic data-dependent stand-in (SplitMix64 chaining) — identical to the oracle's stub so
 * that parity tests compare complete proofs; NOT the reference's Poseidon2 transcript. */
ceno_transcript* ceno_transcript_stub_new(uint64_t seed);
/* Poseidon2-Goldilocks duplex challenger (width 8, rate 4) run on the host.  PARITY UNPINNED:
 * constants / label encoding of the reference's BasicTranscript live in EXT crates. */
ceno_transcript* ceno_transcript_poseidon2_new(const uint8_t* label, size_t n);
/* parameter table of the host challenger (process-wide; same layout as ceno_hip_poseidon2_set_constants, NULL = placeholder part).
 * NON-INTEROPERABLE UNTIL PINNED: with the built-in placeholder constants the challenges are not the reference's; first use
 * warns on stderr, and with CENO_HIP_REQUIRE_PINNED_POSEIDON2=1 ceno_transcript_poseidon2_new returns NULL until a complete
 * table has been supplied. */
int ceno_transcript_poseidon2_set_constants(const uint64_t* external_rc /* 8 x 8 */, const uint64_t* internal_rc /* 22 */,
                                            const uint64_t* internal_diag /* 8 */);
int ceno_transcript_poseidon2_is_pinned(void);
void ceno_transcript_free(ceno_transcript* t);
/* convenience for bindings that cannot call through the function-pointer table */
void ceno_transcript_append_label(ceno_transcript* t, const uint8_t* bytes, size_t n);
void ceno_transcript_append_ext(ceno_transcript* t, const uint64_t* e2);
void ceno_transcript_append_base(ceno_transcript* t, uint64_t v);
void ceno_transcript_sample_ext(ceno_transcript* t, uint64_t* out2);
/* sample_bits (pcs/mod.rs:8164-8204): the low `bits` bits of the canonical value of ONE base sample */
uint64_t ceno_transcript_sample_bits(ceno_transcript* t, int bits);
/* check_witness (pcs/mod.rs:8125-8155): observe `witness`, then sample_bits(bits) == 0.  Advances the transcript.  1 = accepted */
int ceno_transcript_check_witness(ceno_transcript* t, int bits, uint64_t witness);
/* a complete copy (table + state) of a transcript that implements fork; release with ceno_transcript_free.  NULL when it cannot fork */
ceno_transcript* ceno_transcript_clone(const ceno_transcript* t);
int ceno_transcript_export_state(ceno_transcript* t, uint64_t* out16);          /* CENO_TRANSCRIPT_DUPLEX8 or 0 */
int ceno_transcript_import_state(ceno_transcript* t, const uint64_t* in16);     /* 0 = ok */
/* `GrindingChallenger::grind` (p3-challenger; the prover side of check_witness): a witness w for which a clone of the
 * transcript passes check_witness(bits, w) — the LEAST one, found on the device, when the transcript exports a Poseidon2
 * duplex state (one permutation per candidate: ceno_hip_pow_grind_duplex); else the transcript's own `grind` when the table
 * has one; else the least one through fork / append_base / sample_bits on the host.  Then check_witness on `t` itself.
 * p3 accepts any valid witness (`find_any`). */
int ceno_prover_transcript_grind(ceno_hip_ctx* ctx, ceno_transcript* t, int bits, ceno_hip_stream s, uint64_t* out_witness);

/* IOPProverState::prove: appends n and d (usize le-bytes), then per round the d evaluations and
 * the label "Internal round", samples the challenge.  out_msgs: n*d ext, out_challenges: n ext,
 * out_final_evals: num_mles ext. */
int ceno_prover_sumcheck_prove(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan,
                               ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_msgs, uint64_t* out_challenges,
                               uint64_t* out_final_evals);
/* the same with selector declarations handed through to ceno_hip_sumcheck_begin_eq (include/ceno_hip.h): table eq_mle_idx[k] is
 * eq(., eq_points[k]) on the rows [eq_lo[k], eq_hi[k]); the messages are the same words, the rounds of the declared chips cheaper */
int ceno_prover_sumcheck_prove_eq(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan, int num_eq, const int* eq_mle_idx,
                                  const uint64_t* const* eq_points, const size_t* eq_lo, const size_t* eq_hi, ceno_transcript* tr, ceno_hip_stream s,
                                  uint64_t* out_msgs, uint64_t* out_challenges, uint64_t* out_final_evals);
/* VirtualPolynomialsBuilder (EXT multilinear_extensions; call sites gkr_iop/src/gkr/layer/cpu/mod.rs:80-88,213-226,
 * ceno_zkvm/src/scheme/cpu/mod.rs:98,135-137,413-418,1255-1334): register MLEs — `lift` returns the expression id,
 * the same id for the same handle; take_ownership != 0 is the reference's `Either::Right` (owned, freed with the builder),
 * 0 is `Either::Left` (borrowed) — add monomial terms `Term{scalar, product}`, then prove
 * (`to_virtual_polys_with_monomial_terms` + `IOPProverState::prove`).  max_degree <= 0 means the largest product. */
typedef struct ceno_vp_builder ceno_vp_builder;
ceno_vp_builder* ceno_vp_builder_new(int max_num_vars);
void ceno_vp_builder_free(ceno_hip_ctx* ctx, ceno_vp_builder* b);
int ceno_vp_builder_lift(ceno_vp_builder* b, ceno_hip_mle* mle, int take_ownership);                 /* >= 0: id, < 0: error */
int ceno_vp_builder_add_term(ceno_vp_builder* b, const uint64_t* scalar2, const int* product, int n);  /* >= 0: term id */
int ceno_vp_builder_num_mles(const ceno_vp_builder* b);
int ceno_vp_builder_max_degree(const ceno_vp_builder* b);
int ceno_vp_builder_prove(ceno_hip_ctx* ctx, ceno_vp_builder* b, int max_degree, ceno_transcript* tr, ceno_hip_stream s,
                          uint64_t* out_msgs, uint64_t* out_challenges, uint64_t* out_final_evals);
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

/* ---- per-chip proof flow (a9 + create_chip_proof) ----
 * build_tower_witness (ceno_zkvm/src/scheme/cpu/mod.rs:608-757; GPU arm scheme/gpu/mod.rs:2136-2407): slice the record MLEs
 * [reads | writes | lookup numerators | lookup denominators] (numerators exist for table circuits only: num_lk_tables =
 * cs.lk_table_expressions.len(), then the denominators are the table expressions too; otherwise num_lk = cs.lk_expressions.len()
 * denominators and all-one numerators), interleave every group over ALL 2^(log2_num_instances + rotation_vars) rows with
 * default ONE (read / write sets) or challenges[0] (lookup limbs), and build the product / LogUp towers of
 * group_num_vars = row_vars + ceil_log2(next_pow2(group size)) layers.  Out-evals = layer 0 of every tower. */
typedef struct ceno_tower_witness {
    ceno_hip_tower* prod[2];       /* read set, write set (an absent set is skipped) */
    int n_prod;
    ceno_hip_tower* logup[1];
    int n_logup;
    int has_r, has_w, has_lk;
    uint64_t r_out_evals[4];       /* 2 ext */
    uint64_t w_out_evals[4];       /* 2 ext */
    uint64_t lk_out_evals[8];      /* 4 ext: p1, p2, q1, q2 */
} ceno_tower_witness;
int ceno_prover_build_tower_witness(ceno_hip_ctx* ctx, ceno_hip_mle* const* records, int num_reads, int num_writes, int num_lk_tables,
                                    int num_lk, int log2_num_instances, int rotation_vars, const uint64_t* challenges4, ceno_hip_stream s,
                                    ceno_tower_witness* out);
void ceno_tower_witness_free(ceno_hip_ctx* ctx, ceno_tower_witness* w);

/* ZKVMProver::create_chip_proof (ceno_zkvm/src/scheme/prover.rs:717-833; harness benches/riscv_add.rs:86-141):
 * build_main_witness at the tower stage (record MLEs by wit_infer over witness ++ fixed ++ structural, scheme/utils.rs:667-771),
 * prove_tower_relation (out-evals into the transcript, tower proof), rt_main = the last num_var_with_rotation coordinates of
 * the tower point, prove_rotation for chips with a rotation argument.  The main constraints are NOT proved here: the caller
 * collects one ceno_main_job per chip with selector points = rt_main and calls ceno_prover_prove_batched_main_constraints
 * (prover.rs:818-826, 577-586).  ECC quark (shard-RAM chips) is out of scope. */
typedef struct ceno_chip_task {
    int circuit_idx;
    size_t num_instances;          /* input.num_instances */
    int log2_num_instances;        /* of the padded instance count */
    int rotation_vars;             /* 0 without rotation */
    int n_witin, n_fixed, n_structural;
    ceno_hip_mle* const* mles;     /* witness ++ fixed ++ structural; every table has log2_num_instances + rotation_vars variables */
    /* record expressions in monomial form, coefficients already evaluated at the two global challenges
     * (wit_infer_by_monomial_expr, gkr_iop/src/cpu/mod.rs:119-176); outputs in the order reads, writes, lk numerators, lk denominators */
    int num_reads, num_writes, num_lk_tables, num_lk;
    int n_record_terms;
    const uint64_t* record_coeffs;            /* 2 words per term */
    const uint32_t* record_term_offsets;      /* n_record_terms + 1 */
    const uint32_t* record_term_mle_idx;
    const uint32_t* record_out_term_offsets;  /* n_records + 1 */
    /* rotation argument (n_rotation_pairs = 0: none); indices into `mles` */
    int n_rotation_pairs;
    const int* rotation_source_idx;
    const int* rotation_target_idx;
    int cyclic_subgroup_size;
    int cyclic_group_log2;
} ceno_chip_task;
/* ZKVMChipProof (ceno_zkvm/src/scheme.rs:59-76) flattened; buffers are allocated by create_chip_proof, released by
 * ceno_chip_proof_free.  tower: layout of ceno_tower_proof for max_num_vars = tower_num_vars. */
typedef struct ceno_chip_proof {
    size_t num_instances;
    int n_r_out, n_w_out, n_lk_out;           /* 0 or 2, 0 or 2, 0 or 4 */
    uint64_t r_out_evals[4], w_out_evals[4], lk_out_evals[8];
    int tower_num_vars, n_prod, n_logup;
    ceno_tower_proof tower;
    int num_var_with_rotation;
    uint64_t* rt_main;                        /* num_var_with_rotation ext: MainConstraintJob.rt_tower */
    int n_rotation_pairs;
    uint64_t* rotation_msgs;                  /* num_var_with_rotation x 2 ext */
    uint64_t* rotation_evals;                 /* 3 x n_rotation_pairs ext: [left, right, target] per pair */
    uint64_t* rotation_points;                /* origin | left | right, num_var_with_rotation ext each */
} ceno_chip_proof;
int ceno_prover_create_chip_proof(ceno_hip_ctx* ctx, const ceno_chip_task* task, const uint64_t* challenges4, ceno_transcript* tr,
                                  ceno_hip_stream s, ceno_chip_proof* out);
void ceno_chip_proof_free(ceno_chip_proof* p);
/* device bytes a chip proof allocates on top of its borrowed tables: the booking estimate ceno_prover_create_chip_proofs uses (the
 * reference's estimator, ceno_zkvm/src/scheme/gpu/memory.rs:54-145); held between 1x and 2x the pool's high-water mark by the GPU tests */
size_t ceno_prover_chip_proof_estimate_bytes(const ceno_chip_task* task);

/* prove_rotation (gkr_iop/src/gkr/layer/cpu/mod.rs:249-389; GPU arm layer/gpu/mod.rs:305-462): for pairs
 * (source_j, target_j) of base-field witness tables prove  0 = sum_b sel(b) sum_j alpha^j (rotated(source_j)(b) - target_j(b))
 * with sel = eq(., rt) on the cyclic subgroup.  Transcript: get_challenge_pows(n_pairs), the degree-2 sumcheck,
 * then the 3*n_pairs evaluations [left, right, target] are appended.
 * out_msgs: n*2 ext; out_evals: 3*n_pairs ext; out_origin/left/right: n ext each. */
int ceno_prover_prove_rotation(ceno_hip_ctx* ctx, ceno_hip_mle* const* wit, const int* source_idx, const int* target_idx, int n_pairs,
                               int cyclic_subgroup_size, int cyclic_group_log2, const uint64_t* rt, int num_vars, ceno_transcript* tr,
                               ceno_hip_stream s, uint64_t* out_msgs, uint64_t* out_evals, uint64_t* out_origin, uint64_t* out_left,
                               uint64_t* out_right);

/* ---- batched main-constraint sumcheck (a12/a13) ----
 * BatchedMainConstraintProver::prove_batched_main_constraints (ceno_zkvm/src/scheme/hal.rs:37-52, CPU body
 * scheme/cpu/mod.rs:1052-1390): ONE sumcheck over the first GKR layer of every chip.  Per chip the caller
 * passes its MLEs (witness ++ fixed ++ structural), the selector groups (built here as eq tables at the
 * given out-points and substituted for their structural witness id, first occurrence wins), and the layer's
 * monomial terms.  The scalar of a term is a polynomial in the "main sumcheck challenges"
 *   [global_challenges[0], global_challenges[1], alpha_pows[alpha_start .. alpha_start + n_exprs)]
 * encoded as monomials: scalar_t = sum_m mono_coeffs[m] * prod_k challenges[mono_chal_idx[k]].
 * A job list of length 1 is the per-layer ZerocheckLayerProver::prove (gkr_iop/src/gkr/layer/cpu/mod.rs:102-238). */
typedef struct ceno_main_job {
    int circuit_idx;
    int num_vars;                          /* log2_num_instances + rotation_vars */
    int n_witin, n_fixed, n_structural;
    ceno_hip_mle* const* mles;             /* n_witin + n_fixed + n_structural handles (a replaced structural slot may be NULL) */
    int n_selectors;
    const int* sel_kind;                   /* ceno_hip_selector_kind */
    const size_t* sel_offset;
    const size_t* sel_num_instances;
    const int* sel_structural_id;          /* structural witness id the selector stands for */
    const uint32_t* const* sel_sparse_indices;
    const int* sel_n_sparse;
    const int* sel_sparse_num_vars;
    const uint64_t* const* sel_points;     /* num_vars ext each */
    int n_exprs;                           /* number of alpha powers this chip consumes (layer.exprs.len()) */
    int max_degree;                        /* first_layer.max_expr_degree + 1 */
    int n_terms;
    const uint32_t* term_offsets;          /* n_terms + 1 */
    const uint32_t* term_mle_idx;          /* chip-local MLE ids */
    const uint32_t* scalar_offsets;        /* n_terms + 1 -> monomial range */
    const uint64_t* mono_coeffs;           /* 2 words per monomial */
    const uint32_t* mono_chal_offsets;     /* n_monomials + 1 */
    const uint32_t* mono_chal_idx;         /* ids into the chip's main-sumcheck challenge list */
    /* public-instance values the scalar expressions may reference (eval_by_expr_with_instance(.., &chip.pi, ..),
     * scheme/cpu/mod.rs:1241-1247,1297-1304): atoms with id >= 2 + n_exprs select pi[id - 2 - n_exprs] */
    int n_pi;
    const uint64_t* pi;                    /* n_pi ext (base-field instances embedded) */
} ceno_main_job;

/* outputs: claimed_sum (1 ext), msgs (max_num_vars * max_degree ext), global_rt (max_num_vars ext),
 * evals (sum of per-chip MLE counts, ext; chip order, witness ++ fixed ++ structural).
 * *out_num_vars / *out_degree receive max_num_variables / max_degree of the batched sumcheck. */
int ceno_prover_prove_batched_main_constraints(ceno_hip_ctx* ctx, const ceno_main_job* jobs, int n_jobs, const uint64_t* global_challenges4,
                                               ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_claimed_sum, uint64_t* out_msgs,
                                               uint64_t* out_global_rt, uint64_t* out_evals, int* out_num_vars, int* out_degree);

/* ---- multi-layer GKR circuit (a13) ----
 * GKRCircuit::prove (gkr_iop/src/gkr.rs:72-115): the running claims (`PointAndEval` per evaluation slot) start from the
 * circuit's out-evaluations; every layer, output side first, (1) reads its claims: per out-evaluation group the claimed values
 * and the group's point (Layer::extract_claim_and_point, gkr/layer.rs:289-313), (2) is proved by its layer prover
 * (gkr/layer.rs:198-243): ZEROCHECK = ZerocheckLayerProver::prove (gkr/layer/cpu/mod.rs:102-238: alpha powers, selector eq
 * tables at the group points, one sumcheck, final evaluations appended), LINEAR = LinearLayerProver::prove
 * (gkr/layer/cpu/mod.rs:47-66: no sumcheck, the witness evaluations at the out point), SUMCHECK = SumcheckLayerProver::prove
 * (gkr/layer/cpu/mod.rs:72-96: plain sumcheck of the layer expression), (3) writes its final evaluations back into the
 * claims at `in_eval_pos` with the layer's point (Layer::update_claims, gkr/layer.rs:315-322).
 * The keccak-style harness proves the rotation argument FIRST and feeds its left / right / target evaluations in as
 * out-evaluations (ceno_zkvm/src/precompiles/lookup_keccakf.rs:1338-1397); ceno_prover_prove_rotation is that step. */
typedef enum ceno_layer_type { CENO_LAYER_ZEROCHECK = 0, CENO_LAYER_LINEAR = 1, CENO_LAYER_SUMCHECK = 2 } ceno_layer_type;
/* EvalExpression (gkr_iop/src/evaluation.rs:17-93) with its coefficient expressions already evaluated at the challenges:
 * ZERO (claim 0 at the point of slot 0), SINGLE (claims[idx]), LINEAR (claims[idx] * c0 + c1).  Partition is not supported. */
typedef enum ceno_eval_kind { CENO_EVAL_ZERO = 0, CENO_EVAL_SINGLE = 1, CENO_EVAL_LINEAR = 2 } ceno_eval_kind;
typedef struct ceno_eval_expr {
    int kind;
    int idx;
    uint64_t c0[2], c1[2];
} ceno_eval_expr;
typedef struct ceno_gkr_layer {
    int type;                              /* ceno_layer_type */
    int num_vars;                          /* variables of the layer's tables */
    int n_witin, n_fixed, n_structural;
    ceno_hip_mle* const* mles;             /* witin ++ fixed ++ structural (a selector's structural slot may be NULL) */
    /* out-evaluation groups (layer.out_sel_and_eval_exprs): selector + the claims it gates */
    int n_groups;
    const int* group_sel_kind;             /* ceno_hip_selector_kind, or -1 = SelectorType::None */
    const int* group_sel_structural_id;
    const size_t* group_sel_offset;
    const size_t* group_sel_num_instances;
    const uint32_t* const* group_sel_sparse_indices;
    const int* group_sel_n_sparse;
    const int* group_sel_sparse_num_vars;
    const uint32_t* group_expr_offsets;    /* n_groups + 1 -> out_exprs */
    const ceno_eval_expr* out_exprs;
    /* main sumcheck expression in monomial form, scalars as in ceno_main_job (challenge list: the two global challenges,
     * then alpha powers [ZEROCHECK only, n_exprs of them], then pub_io) */
    int n_exprs;
    int max_degree;
    int n_terms;
    const uint32_t* term_offsets;
    const uint32_t* term_mle_idx;
    const uint32_t* scalar_offsets;
    const uint64_t* mono_coeffs;
    const uint32_t* mono_chal_offsets;
    const uint32_t* mono_chal_idx;
    /* claim slots that receive this layer's final evaluations, in MLE order (layer.in_eval_expr) */
    int n_in_evals;
    const int* in_eval_pos;
} ceno_gkr_layer;
/* claims: n_evaluations slots; claim_points[i] has claim_point_len[i] ext elements (0 = slot not set yet), claim_evals 2 words
 * per slot.  Outputs per layer l: msgs at out_msgs[l] (num_vars * max_degree ext; nothing for LINEAR), the evaluations of all
 * the layer's MLEs at out_evals[l], the layer's point at out_points[l] (num_vars ext); out_claim_* receive the final claims
 * (points padded to max_num_vars ext per slot, lengths in out_claim_point_len). */
int ceno_prover_gkr_prove(ceno_hip_ctx* ctx, const ceno_gkr_layer* layers, int n_layers, int max_num_vars, int n_evaluations,
                          const uint64_t* const* claim_points, const int* claim_point_len, const uint64_t* claim_evals,
                          const uint64_t* pub_io, int n_pub_io, const uint64_t* challenges4, ceno_transcript* tr, ceno_hip_stream s,
                          uint64_t* const* out_msgs, uint64_t* const* out_evals, uint64_t* const* out_points,
                          uint64_t* out_claim_points, int* out_claim_point_len, uint64_t* out_claim_evals);

/* ---- trace commitment (a14) ----
 * TraceCommitter::commit_traces (ceno_zkvm/src/scheme/hal.rs:137-156, CPU scheme/cpu/mod.rs:559-584, GPU
 * scheme/gpu/mod.rs:1519-1660): per row-major trace matrix — pad the rows to next_pow2_instance_padding
 * (>= 2, zero rows), move it to the device, transpose to column-major, RS-encode every column (blow-up
 * 2^log_blowup), hash the codeword rows and build the Merkle tree.  The witness MLEs handed to the sumchecks
 * are borrowed views of the column-major trace (nothing is copied or re-uploaded).
 * ONE commitment for all the matrices of the call (PCS::batch_commit -> one PCS::Commitment): a mixed-height Merkle tree
 * (ceno_hip_mmcs_commit = p3 MerkleTreeMmcs).  PARITY UNPINNED (EXT mpcs / p3): placeholder Poseidon2 constants, rate /
 * layout assumptions — see DESIGN.md section 5. */
typedef struct ceno_pcs_data ceno_pcs_data;
int ceno_prover_commit_traces(ceno_hip_ctx* ctx, const uint64_t* const* host_row_major, const size_t* num_instances, const size_t* widths,
                              int n_matrices, int log_blowup, ceno_hip_stream s, ceno_pcs_data** out);
/* same with the row-major matrices already in device memory (the reference's device-backed RowMajorMatrix: witness generated
 * on the GPU, `has_device_backing()` / `device_backing_layout()`, scheme/gpu/mod.rs:1556-1582): no PCIe transfer.  All the work
 * is queued on `s`, behind whatever the caller queued there to produce the matrices. */
int ceno_prover_commit_traces_dev(ceno_hip_ctx* ctx, const uint64_t* const* dev_row_major, const size_t* num_instances, const size_t* widths,
                                  int n_matrices, int log_blowup, ceno_hip_stream s, ceno_pcs_data** out);
/* commit_traces for traces PRODUCED on the device (on-device witness generation: the reference keeps the GPU witgen's output as the
 * device backing of the RowMajorMatrix, column-major, and hands it to batch_commit without a copy —
 * `should_materialize_witness_on_gpu`, `normalize_traces_to_device_col_major`, scheme/gpu/mod.rs:1565-1592), in two steps so that the
 * producer writes INSIDE the commitment's own storage: `reserve` lays the matrices out in their height classes and allocates trace +
 * codeword storage; ceno_pcs_data_trace_ptr(d, m) is matrix m's destination — COLUMN-major, widths[m] columns of
 * ceno_pcs_data_rows(d, m) = next_pow2_instance_padding(num_instances[m]) words, the layout the ceno_hip_witgen_* kernels write
 * (rows >= num_instances must be written as zero: they do); `finish` RS-encodes every class and builds the tree, queued on `s`
 * behind the producers (which must have been queued on `s` or be complete).  Between the two calls the handle supports
 * ceno_pcs_data_num_vars / _width / _rows / _trace_ptr / _witness_mle only. */
int ceno_prover_commit_reserve(ceno_hip_ctx* ctx, const size_t* num_instances, const size_t* widths, int n_matrices, int log_blowup,
                               ceno_hip_stream s, ceno_pcs_data** out);
uint64_t* ceno_pcs_data_trace_ptr(ceno_pcs_data* d, int matrix);
size_t ceno_pcs_data_rows(const ceno_pcs_data* d, int matrix);
int ceno_prover_commit_finish(ceno_hip_ctx* ctx, ceno_pcs_data* d, ceno_hip_stream s);
int ceno_pcs_data_num_matrices(const ceno_pcs_data* d);
int ceno_pcs_data_num_vars(const ceno_pcs_data* d, int matrix);
int ceno_pcs_data_width(const ceno_pcs_data* d, int matrix);
/* PCS::get_pure_commitment: the root of the commitment (4 words) */
int ceno_pcs_data_root(ceno_hip_ctx* ctx, ceno_pcs_data* d, uint64_t* root4, ceno_hip_stream s);
/* borrowed base-field MLE view of column `col` of matrix `matrix` (valid while `d` lives; free the handle with ceno_hip_mle_free) */
int ceno_pcs_data_witness_mle(ceno_hip_ctx* ctx, ceno_pcs_data* d, int matrix, size_t col, ceno_hip_mle** out);
/* MerkleTreeMmcs::open_batch at `index` (a row of the tallest codeword): out = [row index >> (H - h_m) of every matrix's codeword, in
 * order (sum of widths words)][path 4 x H words], H = max num_vars + log_blowup; ceno_pcs_data_opening_words words in all */
size_t ceno_pcs_data_opening_words(const ceno_pcs_data* d);
int ceno_pcs_data_open(ceno_hip_ctx* ctx, ceno_pcs_data* d, size_t index, uint64_t* out, ceno_hip_stream s);
void ceno_pcs_data_free(ceno_hip_ctx* ctx, ceno_pcs_data* d);

/* ---- Basefold batch open (a15) ----
 * OpeningProver::open -> PCS::batch_open (ceno_zkvm/src/scheme/hal.rs:284-294, CPU scheme/cpu/mod.rs:1418-1457);
 * protocol as replayed by the in-tree verifier ceno_recursion_v2/src/pcs/mod.rs:1111-1316,7494-7781
 * (basecode_msg_size_log = 0).  `commits` are the `rounds` of batch_open (witness commitment, then the fixed one when present);
 * every committed matrix is opened at its own point (`points[m]`: num_vars ext) with the claimed column evaluations
 * `evals[m]` (width ext), m running over the matrices of commits[0], then commits[1], ...  Proof of work = p3 grinding
 * (check_witness, pcs/mod.rs:8125-8155), query indices = sample_bits(max num_vars + log_blowup) of one base sample each
 * (pcs/mod.rs:1252-1266), one input opening per commitment at query >> bits_reduced (pcs/mod.rs:7547-7565).
 * PARITY UNPINNED (EXT mpcs): Poseidon2 constants, label packing — DESIGN.md section 5.
 * Flat proof (ceno_prover_basefold_proof_words words), n = max num_vars over all commitments, H = n + log_blowup:
 *   [sumcheck messages n x (p(1), p(2)) ext][commit-round roots n x 4][final message: one ext per matrix][pow witness]
 *   then per query: [index] per commitment c [opened codeword rows of its matrices: sum of widths][MMCS path 4 x H_c]
 *                   per round r [sibling ext][path 4 x (H - r - 1)]                                         */
size_t ceno_prover_basefold_proof_words(ceno_pcs_data* const* commits, int n_commits, int n_queries);
int ceno_prover_basefold_open(ceno_hip_ctx* ctx, ceno_pcs_data* const* commits, int n_commits, const uint64_t* const* points,
                              const uint64_t* const* evals, int n_queries, int pow_bits, ceno_transcript* tr, ceno_hip_stream s,
                              uint64_t* out_proof);

/* ---- concurrent chip proving on lanes (scheduler.rs:231-336, memory booking :342-347,:622-652) ----
 * One worker thread per lane, each with the context's stream of that lane (ceno_hip_lane_stream: created once, never per run); tasks are taken largest-estimate
 * first, booked against the pool before they start (greedy back-fill with smaller tasks when the largest does not fit)
 * and unbooked when done.  `fn` runs the C ABI calls of one chip proof on the given stream and returns 0 or an error.
 * out_status / out_lane (n_tasks each, may be NULL) receive every task's return code and the lane that ran it.
 * More than four lanes run as four (the device dispatches four queues concurrently; CENO_HIP_MAX_LANES overrides). */
typedef int (*ceno_lane_task_fn)(void* arg, int lane, ceno_hip_stream stream);
typedef struct ceno_lane_task {
    ceno_lane_task_fn fn;
    void* arg;
    size_t estimated_bytes;
} ceno_lane_task;
int ceno_prover_lanes_run(ceno_hip_ctx* ctx, int n_lanes, const ceno_lane_task* tasks, int n_tasks, int* out_status, int* out_lane);
/* lanes a run of `n_lanes` really uses: the device dispatches four queues concurrently, more lanes are run as four (CENO_HIP_MAX_LANES
 * overrides).  Runs on one context take turns (the lane streams are the context's). */
int ceno_prover_lanes_effective(int n_lanes);
/* The chip-proof phase of create_proof on the scheduler (prover.rs:556-570, scheduler.rs:231-336): task i = one
 * ceno_prover_create_chip_proof with its OWN transcript (the caller forks the parent transcript per chip and merges one sample of
 * each fork back afterwards, prover.rs:567-570), run on whichever of the context's lanes picks it, largest estimate first, booked
 * against the pool.  out_proofs[i] are released with ceno_chip_proof_free; out_status (may be NULL) receives every task's code. */
int ceno_prover_create_chip_proofs(ceno_hip_ctx* ctx, const ceno_chip_task* tasks, int n_tasks, const uint64_t* challenges4,
                                   ceno_transcript* const* transcripts, int n_lanes, ceno_chip_proof* out_proofs, int* out_status);
/* ZKVMProver::run_chip_proofs (prover.rs:618-710) with the scheduler's forking (scheduler.rs:231-336): task i's transcript is a clone of
 * `fork_parent` bound to the two global challenges and then to bind_words[bind_offsets[i] .. bind_offsets[i + 1]) as base-field elements (task id,
 * circuit index, instance counts: prover.rs:646-654, the verifier's order); after the proofs ONE extension sample of every fork is returned in
 * out_samples (n_tasks x 2 words) for the caller to merge into the main transcript (prover.rs:567-570). */
int ceno_prover_run_chip_proofs(ceno_hip_ctx* ctx, const ceno_chip_task* tasks, int n_tasks, const uint64_t* challenges4, const ceno_transcript* fork_parent,
                                const uint64_t* bind_words, const uint32_t* bind_offsets, int n_lanes, ceno_chip_proof* out_proofs,
                                uint64_t* out_samples, int* out_status);

const char* ceno_prover_last_error(void);

/* ---- hypercube-sharded sumcheck over the GPUs of one node (ceno_amd/host/dist.cpp) ----
 * One process per GPU; the 128-byte RCCL unique id is created on rank 0 and distributed by the launcher
 * (torch.distributed broadcast).  See DESIGN.md section 6. */
typedef struct ceno_dist_comm ceno_dist_comm;
int ceno_dist_unique_id(uint8_t* out128);
int ceno_dist_comm_init(int world, int rank, const uint8_t* id128, ceno_dist_comm** out);
void ceno_dist_comm_destroy(ceno_dist_comm* c);
/* what RCCL itself reports for the communicator (ncclCommCount, ncclCommUserRank): > 0 = ranks of the RCCL communicator, 0 = no RCCL
 * part (shared-memory / in-process exchange only), < 0 = error.  bench.py prints it so that a multi-GPU line says which transport ran. */
int ceno_dist_comm_rccl_ranks(ceno_dist_comm* c, int* out_user_rank);
int ceno_dist_sumcheck_prove(ceno_hip_ctx* ctx, ceno_dist_comm* c, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan_local,
                             int n_total, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_msgs, uint64_t* out_challenges,
                             uint64_t* out_final_evals);
/* Host shared-memory exchange for the ranks of ONE node: the d partial evaluations of a round are already in host
 * memory (the transcript is there), so they are exchanged through a POSIX shared segment (~1-2 us) instead of a device
 * collective; with it attached ceno_dist_sumcheck_prove runs every rank's shard as an ordinary pipelined sumcheck.
 * *c == NULL creates a communicator without RCCL.  Rank 0 creates the segment (create != 0) before the other ranks
 * attach, and may unlink the name once all have (the mappings stay valid). */
int ceno_dist_comm_attach_shm(ceno_dist_comm** c, int world, int rank, const char* name, int create);
int ceno_dist_shm_unlink(const char* name);
/* What THIS rank put on the wire through the communicator since the last reset: out4 = [message exchanges (per-round partial sums, gathered
 * evaluations / tables, digests: host words over the small-message transport), bytes this rank sent in them, bulk exchanges (device buffers: the
 * codeword re-shard, gathered blocks), bytes this rank sent in them].  The per-rank critical path of DESIGN.md section 6 is read from these. */
int ceno_dist_comm_stats(ceno_dist_comm* c, uint64_t* out4, int reset);
/* Raise the segment's abort word: every rank that waits for a peer in a shared-memory exchange returns an error at once (instead of after
 * CENO_DIST_SHM_TIMEOUT_S, default 60 s of wall clock).  What a rank calls when ITS part of a collective entry failed before it published —
 * the library does it itself on the errors it raises inside an exchange and on a time-out.  The communicator is unusable afterwards. */
int ceno_dist_comm_abort(ceno_dist_comm* c);
int ceno_dist_shm_selftest(ceno_dist_comm* c, int iters);
/* Mixed-size (front-loaded) batched sumcheck across ranks (BASELINE config #4): every size class is
 * split along the top bits of ITS OWN hypercube or replicated; needs the shared-memory exchange.  Term MLE ids are
 * class-local.  out_final_evals: concatenated per class (num_mles ext each), identical on every rank. */
typedef struct ceno_dist_class {
    int num_vars;                  /* GLOBAL number of variables of the class */
    int sharded;                   /* 1: `mles` are slice `rank` (num_vars - log2(world) variables); 0: whole tables on every rank */
    int num_mles;
    ceno_hip_mle* const* mles;
    int num_terms;
    const uint64_t* term_coeffs;   /* 2 * num_terms words */
    const uint32_t* term_offsets;  /* num_terms + 1 */
    const uint32_t* term_mle_idx;
} ceno_dist_class;
int ceno_dist_batched_sumcheck_prove(ceno_hip_ctx* ctx, ceno_dist_comm* c, const ceno_dist_class* classes, int n_classes, int n_total,
                                     int max_degree, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_msgs,
                                     uint64_t* out_challenges, uint64_t* out_final_evals);
/* Trace commitment across ranks (SURVEY section 8e "Commit path"; single-device reference flow: commit_traces,
 * ceno_zkvm/src/scheme/gpu/mod.rs:927-1020).  Rank g holds widths[g] columns of the padded trace, column-major on the
 * device (column stride 2^log_rows); `widths` has `world` entries and is identical on every rank.  Every rank RS-encodes
 * its columns, ONE all-to-all of unequal blocks (RCCL grouped send/recv over the point-to-point xGMI links; device copies
 * inside a local group) re-shards the codeword by rows, rank g hashes the sub-tree over rows [g R/world, (g+1) R/world),
 * R = 2^(log_rows+log_blowup), the sub-tree roots are gathered (shared segment when attached, else RCCL) and the top
 * log2(world) levels are hashed on every rank.  out_rows_dev: (sum widths) x R/world words, column-major = this rank's
 * codeword rows (caller-allocated, kept for the openings); *out_subtree: Merkle tree over them (ceno_hip_merkle_free);
 * out_subtree_roots: world x 4 words or NULL; out_root: 4 words = the single-device commitment root, bit for bit. */
int ceno_dist_commit_traces(ceno_hip_ctx* ctx, ceno_dist_comm* c, const uint64_t* local_cols_dev, const int* widths, int log_rows,
                            int log_blowup, ceno_hip_stream s, uint64_t* out_rows_dev, ceno_hip_merkle** out_subtree,
                            uint64_t* out_subtree_roots, uint64_t* out_root);
/* The same for SEVERAL matrices of several heights under ONE root = the multi-rank form of ceno_prover_commit_traces: the result
 * equals the single-device mixed-height commitment (ceno_hip_mmcs_commit) bit for bit.  Rank g holds widths[m * world + g] columns
 * of matrix m (2^log_rows[m] rows; local_cols_dev[m] column-major).  Per matrix one all-to-all re-shards the codeword by rows —
 * out_rows_dev[m]: (all columns of matrix m) x (R_m / world) words, column-major, or the WHOLE codeword when it has fewer rows than
 * there are ranks; rank g's sub-tree (*out_subtree) is the mixed-height tree over its shards = the sub-tree under node g of the
 * global tree's level with `world` nodes; the top log2(world) levels (*out_top, may be NULL; NULL is returned for world 1 without
 * short matrices) are built on every rank over the gathered sub-tree roots, the matrices shorter than `world` joining there. */
int ceno_dist_commit_traces_mmcs(ceno_hip_ctx* ctx, ceno_dist_comm* c, int n_mats, const int* log_rows, const int* widths,
                                 const uint64_t* const* local_cols_dev, int log_blowup, ceno_hip_stream s, uint64_t* const* out_rows_dev,
                                 ceno_hip_merkle** out_subtree, ceno_hip_merkle** out_top, uint64_t* out_subtree_roots, uint64_t* out_root);
/* In-process group of `world` virtual ranks (threads of one process sharing a device, one stream each): the transport the
 * single-GPU tests run the multi-rank commit path on, and a single-process deployment's.  Create the group once, one
 * communicator per rank (ceno_dist_comm_destroy), destroy the group last. */
typedef struct ceno_dist_local_group ceno_dist_local_group;
ceno_dist_local_group* ceno_dist_local_group_create(int world);
void ceno_dist_local_group_destroy(ceno_dist_local_group* g);
int ceno_dist_comm_init_local(ceno_dist_local_group* group, int rank, ceno_dist_comm** out);
/* The GKR half of a chip proof over ROW-SHARDED witness columns (ceno_amd/host/dist_gkr.cpp; SURVEY section 8(e): a3, a5-a10): record
 * inference, tower witness and tower proof, bit for bit the proof ceno_prover_create_chip_proof produces from the whole columns
 * (single-device flow: ZKVMProver::create_chip_proof, ceno_zkvm/src/scheme/prover.rs:717-833; tower prover scheme/cpu/mod.rs:346-554;
 * the reference has no distribution, docs/src/optimizations.md:3-5).
 * Row layout: with k = log2(world) and q = row_block_log (<= 0: ceno_dist_chip_block_log(), 10), rank g holds the rows whose index bits
 * [q, q + k) equal g — blocks of 2^q rows dealt round-robin — in increasing order; `task->mles` are those local tables and
 * task->log2_num_instances = log2_num_instances_global - k their height (task->num_instances stays the GLOBAL count; record plans as in
 * ceno_chip_task).  A layout by the MIDDLE bits keeps both the product layers (which pair the top bit) and the LSB-first layer sumchecks
 * local; the top of every tower (<= 2^(q + log2 records + k) entries per limb) is gathered and proved replicated.  Needs
 * log2_num_instances_global (+ rotation_vars) >= q + k + 1.  A keccak-style chip (task->rotation_vars, rotation pairs) shards its ROWS the same way
 * (task->log2_num_instances + rotation_vars = the local height; q >= cyclic_group_log2): rotations are local, the rotation sumcheck runs q local
 * rounds and its gathered tail replicated.  Every rank passes a transcript
 * in the same state and ends with the same proof; works over any transport of the communicator (in-process group, shared segment, RCCL). */
int ceno_dist_chip_block_log(void);
int ceno_dist_create_chip_proof(ceno_hip_ctx* ctx, ceno_dist_comm* c, const ceno_chip_task* task, int log2_num_instances_global, int row_block_log,
                                const uint64_t* challenges4, ceno_transcript* tr, ceno_hip_stream s, ceno_chip_proof* out);
/* prove_batched_main_constraints over the same block layout (ceno_amd/host/main_constraints.cpp prover_main_constraints_sharded): the tables of every
 * job hold THIS rank's rows (num_vars - log2 world variables; job.num_vars, the selectors' offsets / counts and points stay GLOBAL; Whole and
 * Prefix selectors only).  A chip with num_vars < q + log2 world + 1 is too small to be sharded: every rank passes its WHOLE tables and it is
 * proved replicated beside the sharded ones (any selector kind; its part of a round's message is added once).  q local rounds (the partial
 * evaluations of a round summed over the ranks), the gathered tail replicated; outputs as ceno_prover_prove_batched_main_constraints on the whole
 * tables, word for word, on every rank. */
int ceno_dist_prove_batched_main_constraints(ceno_hip_ctx* ctx, ceno_dist_comm* c, const ceno_main_job* jobs_local, int n_jobs, int row_block_log,
                                             const uint64_t* global_challenges4, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_claimed_sum,
                                             uint64_t* out_msgs, uint64_t* out_global_rt, uint64_t* out_evals, int* out_num_vars, int* out_degree);
/* Basefold opening of a commitment made ACROSS ranks by ceno_dist_commit_traces_mmcs (ceno_amd/host/dist_open.cpp): on every rank the
 * proof ceno_prover_basefold_open produces for the single-device commitment of the same matrices, word for word (layout and size:
 * ceno_prover_basefold_proof_words_meta).  The bandwidth-bound part is sharded — the batched codeword from every rank's ROW shard
 * (local_cw_rows = the commit's out_rows_dev), the batched trace polynomial from every rank's COLUMN shard (local_trace_cols = the commit's
 * input), the commitment's rows and sub-tree paths at the queries from the rank that owns them — and gathered; the commit phase (~21
 * dependent rounds on a codeword 1 / width the size of the data: latency) runs replicated.  ONE commitment; ceno_dist_basefold_open takes
 * matrices of one height 2^log_rows, ceno_dist_basefold_open_mmcs matrices of any heights log_rows[m] (a shard's traces; one batched codeword
 * per height class; a matrix whose codeword has fewer rows than ranks is opened from the replicated top tree; only the TALLEST codeword of the
 * commitment needs at least `world` rows); widths[m * world + g] as in the commit; points / evals per matrix as in
 * ceno_prover_basefold_open (evals: all `sum_g widths` columns, rank-major).  The bulk all-gathers use RCCL, the in-process group's device copies, or — between
 * processes without RCCL — the shared segment (host-staged: a correctness path).  ceno_dist_basefold_open_commits: several such commitments in one opening. */
/* ... and of SEVERAL commitments in one opening (OpeningProver::open takes the witness and the fixed commitment, scheme/hal.rs:284-294): one view
 * per commitment, points / evals over the matrices of all of them in order; a height that two commitments share is one batched codeword */
typedef struct ceno_dist_commit_view {
    int n_mats;
    const int* log_rows;                       /* [n_mats] */
    const int* widths;                         /* [n_mats * world] */
    const uint64_t* const* local_trace_cols;   /* [n_mats]: this rank's columns (the commit's input) */
    const uint64_t* const* local_cw_rows;      /* [n_mats]: this rank's codeword rows of all columns (the commit's out_rows_dev) */
    ceno_hip_merkle* subtree;
    ceno_hip_merkle* top;
} ceno_dist_commit_view;
int ceno_dist_basefold_open_commits(ceno_hip_ctx* ctx, ceno_dist_comm* c, int n_commits, const ceno_dist_commit_view* commits, int log_blowup,
                                    const uint64_t* const* points, const uint64_t* const* evals, int n_queries, int pow_bits, ceno_transcript* tr,
                                    ceno_hip_stream s, uint64_t* out_proof);
/* words of that proof from the shapes alone: n_mats / total_width / max_log_rows per commitment */
size_t ceno_prover_basefold_proof_words_commits(int n_commits, const int* n_mats, const int* total_width, const int* max_log_rows, int log_blowup,
                                                int n_queries);
int ceno_dist_basefold_open_mmcs(ceno_hip_ctx* ctx, ceno_dist_comm* c, int n_mats, const int* log_rows, const int* widths, int log_blowup,
                                 const uint64_t* const* local_trace_cols, const uint64_t* const* local_cw_rows, ceno_hip_merkle* subtree,
                                 ceno_hip_merkle* top, const uint64_t* const* points, const uint64_t* const* evals, int n_queries, int pow_bits,
                                 ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_proof);
int ceno_dist_basefold_open(ceno_hip_ctx* ctx, ceno_dist_comm* c, int n_mats, int log_rows, const int* widths, int log_blowup,
                            const uint64_t* const* local_trace_cols, const uint64_t* const* local_cw_rows, ceno_hip_merkle* subtree,
                            ceno_hip_merkle* top, const uint64_t* const* points, const uint64_t* const* evals, int n_queries, int pow_bits,
                            ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_proof);
/* words of a Basefold opening proof from the shapes alone (n_mats matrices, total_width columns, tallest trace 2^max_log_rows): what
 * ceno_prover_basefold_proof_words returns for one commitment of such matrices */
size_t ceno_prover_basefold_proof_words_meta(int n_mats, int total_width, int max_log_rows, int log_blowup, int n_queries);
const char* ceno_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
