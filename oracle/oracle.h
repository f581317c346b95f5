/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see gl64.h).  CPU restatement, in plain C, of the
 * reference's GKR/sumcheck hot path.  PARITY STATUS:
 *   - tower-witness functions are pinned against the reference's literal known-answer
 *     tests (ceno_zkvm/src/scheme/utils.rs:934-1194) — tests/golden/tower_witness.json;
 *   - eq / MLE-evaluate / selector / succinct evaluators are pinned against the
 *     reference's identity tests (gkr_iop/src/utils.rs:332-441, selector.rs:396-435);
 *   - sumcheck round messages, final evaluations and tower proofs are pinned by the
 *     restated verifiers (scheme/verifier.rs:1282-1353,1372-1709) — values are unique
 *     field elements given identical inputs and challenges;
 *   - PARITY UNPINNED: Fiat–Shamir bytes (Poseidon2-Goldilocks constants, label
 *     encoding), Basefold commitment root, W of the extension (gl64.h) — SURVEY.md §8c.
 *
 * Field elements cross this API as canonical little-endian uint64 words; an extension
 * element is two consecutive words [c0, c1].
 */
#ifndef CENO_ORACLE_H
#define CENO_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- transcript abstraction (reference: EXT `transcript::Transcript`, call sites
 * ceno_zkvm/src/scheme/cpu/mod.rs:375-381,534; script restated in
 * ceno_recursion_v2/src/tower/mod.rs:1541-1646) ---- */
typedef struct orc_transcript {
    void (*append_label)(void* self, const uint8_t* bytes, size_t n);
    void (*append_ext)(void* self, const uint64_t* e2);
    void (*sample_ext)(void* self, uint64_t* out2);
    void* self;
    void (*reserved)(void* self);     /* keeps the table layout-compatible with the host library's ceno_transcript (its `destroy`), so that
                                       * tests can drive the oracle's verifiers with the product's transcript objects */
    /* base-field side of the challenger, needed by the PCS (p3-challenger 0.4.3 `CanObserve<F>` / `CanSample<F>` /
     * `CanSampleBits` / `GrindingChallenger`; in-tree use: ceno_recursion_v2/src/pcs/mod.rs:8125-8204).  NULL in
     * transcripts that only serve sumchecks (the PCS entry points then fail). */
    void (*append_base)(void* self, uint64_t v);
    /* `CanSampleBits::sample_bits` (p3-challenger 0.4.3 DuplexChallenger; in-tree ceno_recursion_v2/src/pcs/mod.rs:8164-8204 and the
     * supertraits of `Transcript`, ceno_recursion_v2/src/tower/tower.rs:85-101): the low `bits` bits of the canonical value of ONE
     * base-field sample */
    uint64_t (*sample_bits)(void* self, int bits);
    void* (*fork)(void* self);        /* heap copy of the challenger state (`self.clone()`) */
    void (*fork_free)(void* forked);
} orc_transcript;

uint64_t orc_tr_sample_bits(orc_transcript* t, int bits);
/* check_witness (pcs/mod.rs:8125-8155): observe the witness, then sample_bits(bits) == 0.  Advances the transcript. */
int orc_tr_check_witness(orc_transcript* t, int bits, uint64_t witness);
/* grind (p3-challenger `GrindingChallenger::grind`): a witness that check_witness accepts on a CLONE of the challenger, then
 * check_witness on the challenger itself.  p3 takes any such witness (`find_any`); this restatement takes the least. */
uint64_t orc_tr_grind(orc_transcript* t, int bits);

/* deterministic, data-dependent stand-in for BasicTranscript (NOT Poseidon2) */
typedef struct orc_stub_state { uint64_t s; } orc_stub_state;
void orc_stub_init(orc_stub_state* st, uint64_t seed);
void orc_stub_bind(orc_transcript* t, orc_stub_state* st);
void orc_stub_append_label(orc_stub_state* st, const uint8_t* bytes, size_t n);
void orc_stub_append_ext(orc_stub_state* st, const uint64_t* e2);
void orc_stub_append_base(orc_stub_state* st, uint64_t v);
void orc_stub_sample_ext(orc_stub_state* st, uint64_t* out2);
uint64_t orc_stub_sample_bits(orc_stub_state* st, int bits);

/* Poseidon2 duplex challenger over Goldilocks, width 8 / rate 4 (transcript.c): the shape of p3-challenger 0.4.3
 * `DuplexChallenger<F, Perm, 8, 4>` that the reference's EXT `transcript::BasicTranscript` wraps.  PARITY UNPINNED
 * (round constants, label packing): SURVEY.md section 8c(i). */
typedef struct orc_duplex_state {
    uint64_t state[8];
    uint64_t in[4];
    int n_in;
    int n_out;             /* the output buffer is state[0 .. n_out) (samples pop from the back) */
    uint64_t params[138];
} orc_duplex_state;
void orc_duplex_init(orc_duplex_state* st, const uint64_t* params138, const uint8_t* label, size_t n);
void orc_duplex_bind(orc_transcript* t, orc_duplex_state* st);
void orc_duplex_observe(orc_duplex_state* st, uint64_t v);
uint64_t orc_duplex_sample(orc_duplex_state* st);

/* ---- field helpers exported for Python cross-checks ---- */
uint64_t orc_gl_mul(uint64_t a, uint64_t b);
uint64_t orc_gl_mul_div(uint64_t a, uint64_t b); /* by division: cross-check */
uint64_t orc_gl_inv(uint64_t a);
void orc_e2_mul(const uint64_t* a, const uint64_t* b, uint64_t* out);
void orc_e2_inv(const uint64_t* a, uint64_t* out);
void orc_fill_splitmix(uint64_t* out, size_t n_words, uint64_t seed, uint64_t word_offset);

/* ---- MLE primitives (a3, EXT multilinear_extensions; LSB-first: gkr_iop/src/utils.rs:215-232) ---- */
void orc_build_eq_x_r_vec(const uint64_t* point, int n, uint64_t* out /* 2*2^n words */);
void orc_eq_eval(const uint64_t* a, const uint64_t* b, int n, uint64_t* out2);
void orc_mle_evaluate(const uint64_t* evals, int is_ext, int num_vars, const uint64_t* point, uint64_t* out2);
/* out[j] = f[2j] + r (f[2j+1] - f[2j]); out has 2^(num_vars-1) ext elements */
void orc_mle_fix_variable(const uint64_t* evals, int is_ext, int num_vars, const uint64_t* r2, uint64_t* out);
void orc_extrapolate_uni_poly(const uint64_t* p0, const uint64_t* evals_1_to_d, int d, const uint64_t* x, uint64_t* out2);

/* ---- succinct evaluators (gkr_iop/src/utils.rs:166-307) ---- */
void orc_eq_eval_less_or_equal_than(uint64_t max_idx, const uint64_t* a, int na, const uint64_t* b, int nb, uint64_t* out2);
void orc_eval_wellform_address_vec(uint64_t offset, uint64_t scaled, const uint64_t* r, int n, int descending, uint64_t* out2);
void orc_eval_stacked_wellform_address_vec(const uint64_t* r, int n, uint64_t* out2);
void orc_eval_stacked_constant_vec(const uint64_t* r, int n, uint64_t* out2);

/* ---- selectors (a4, gkr_iop/src/selector.rs:131-363) ---- */
enum { ORC_SEL_WHOLE = 0, ORC_SEL_PREFIX = 1, ORC_SEL_ORDERED_SPARSE = 2, ORC_SEL_QUARK_LT = 3 };
int orc_selector_compute(int kind, const uint64_t* out_point, int num_vars, size_t offset, size_t num_instances,
                         const uint32_t* sparse_indices, int n_sparse, int sparse_num_vars, uint64_t* out);
int orc_selector_evaluate(int kind, const uint64_t* out_point, const uint64_t* in_point, int num_vars, size_t offset,
                          size_t num_instances, const uint32_t* sparse_indices, int n_sparse, int sparse_num_vars,
                          uint64_t* out2);

/* ---- generic sumcheck (a1; message format SURVEY.md §3.4) ---- */
typedef struct orc_mle {
    const uint64_t* data; /* 2^num_vars base words or 2*2^num_vars ext words */
    int is_ext;
    int num_vars;
} orc_mle;

/* Run the n-round prover for  sum_x sum_t c_t prod_{j in S_t} f_j(x).
 * Challenges come from `tr` exactly as IOPProverState::prove drives a transcript
 * (append n, d as le-bytes labels; per round append d evals, label "Internal round",
 * sample) — ceno_recursion_v2/src/main/mod.rs:3503-3529.
 * out_msgs: n*d ext; out_challenges: n ext; out_final_evals: num_mles ext (pure
 * evaluations f_j(r_0..r_{nv_j-1}), no front-load tail factor). Returns 0 on success. */
int orc_sumcheck_prove(const orc_mle* mles, int num_mles, const uint64_t* term_coeffs, const uint32_t* term_offsets,
                       const uint32_t* term_mle_idx, int num_terms, int max_num_vars, int max_degree,
                       orc_transcript* tr, uint64_t* out_msgs, uint64_t* out_challenges, uint64_t* out_final_evals);

/* IOPVerifierState::verify semantics (scheme/verifier.rs:1286-1295; main/mod.rs:3508-3529):
 * returns the expected evaluation and the point; transcript driven identically. */
int orc_sumcheck_verify(const uint64_t* claimed_sum, const uint64_t* msgs, int num_vars, int degree, orc_transcript* tr,
                        uint64_t* out_point, uint64_t* out_expected2);

/* expected evaluation of the batched polynomial at the end (front-load rule,
 * scheme/verifier.rs:180-238 restated in ceno_recursion_v2/src/main/mod.rs:3414-3447) */
void orc_sumcheck_expected_from_evals(const int* mle_num_vars, int num_mles, const uint64_t* term_coeffs,
                                      const uint32_t* term_offsets, const uint32_t* term_mle_idx, int num_terms,
                                      int max_num_vars, const uint64_t* point, const uint64_t* final_evals,
                                      uint64_t* out2);

/* recover_sumcheck_claim_from_final (scheme/cpu/mod.rs:1393-1413) */
void orc_recover_claim_from_final(const uint64_t* final_claim, const uint64_t* msgs, const uint64_t* challenges, int n,
                                  int d, uint64_t* out2);

/* multi-threaded (OpenMP) fused fold+accumulate for a single product of `k` ext MLEs
 * with externally supplied per-round challenges: the cpu_baseline leg of bench.py.
 * tables are consumed (folded in place). */
int orc_sumcheck_dense_mt(uint64_t** tables, int k, int num_vars, const uint64_t* challenges, int threads,
                          uint64_t* out_msgs, uint64_t* out_final_evals);
/* the same, eight pairs per AVX-512 vector (oracle/dense_avx512.c): identical outputs; -2 when the CPU lacks AVX-512 F + DQ */
int orc_sumcheck_dense_mt_avx512(uint64_t** tables, int k, int num_vars, const uint64_t* challenges, int threads,
                                 uint64_t* out_msgs, uint64_t* out_final_evals);
int orc_have_avx512(void);

/* ---- element-wise witness inference (a5, EXT wit_infer_by_monomial_expr;
 * gkr_iop/src/cpu/mod.rs:119-176) ---- */
int orc_wit_infer(const orc_mle* mles, int num_mles, const uint64_t* term_coeffs, const uint32_t* term_offsets,
                  const uint32_t* term_mle_idx, int num_terms, int num_vars, uint64_t* out /* 2*2^num_vars */);

/* ---- tower witness (a6-a8, ceno_zkvm/src/scheme/utils.rs:402-659) ---- */
size_t orc_interleave_out_len(int num_mles, size_t num_instances, int num_limbs);
int orc_interleaving_mles_to_mles(const orc_mle* mles, int num_mles, size_t num_instances, int num_limbs,
                                  const uint64_t* default2, uint64_t** out_limbs /* num_limbs buffers */);
/* layers: for layer l in 0..num_vars-1 two buffers of 2^l ext each; layers[2*l+s] */
int orc_infer_tower_product_witness(int num_vars, const uint64_t* last0, const uint64_t* last1, uint64_t** layers);
/* layers[4*l + {p1,p2,q1,q2}], l in 0..num_vars (num_vars+1 layers); p may be NULL */
int orc_infer_tower_logup_witness(int num_vars, const uint64_t* p0, const uint64_t* p1, const uint64_t* q0,
                                  const uint64_t* q1, uint64_t** layers);

/* ---- tower prover/verifier (a10; scheme/cpu/mod.rs:346-554, scheme/verifier.rs:1372-1709) ---- */
typedef struct orc_tower_spec {
    int num_vars;            /* number of witness layers (layer l has limbs of 2^l ext) */
    uint64_t** layers;       /* prod: 2 per layer; logup: 4 per layer */
} orc_tower_spec;

typedef struct orc_tower_proof {
    int num_rounds;          /* max_num_vars - 1 */
    uint64_t* msgs;          /* per round r (1-based layer r): r*3 ext, concatenated */
    uint64_t* prod_evals;    /* [spec][round][2] ext; zero when inactive */
    uint64_t* logup_evals;   /* [spec][round][4] ext */
    uint64_t* point;         /* final rt: max_num_vars ext */
} orc_tower_proof;

size_t orc_tower_msgs_words(int max_num_vars);
int orc_tower_prove(const orc_tower_spec* prod, int n_prod, const orc_tower_spec* logup, int n_logup,
                    orc_transcript* tr, orc_tower_proof* out);
/* returns 0 if the proof verifies; out_point (max_nv ext), and the final claims */
int orc_tower_verify(const uint64_t* prod_out_evals /* n_prod*2 ext */, const uint64_t* logup_out_evals /* n_logup*4 */,
                     const int* num_variables, int n_prod, int n_logup, const orc_tower_proof* proof,
                     orc_transcript* tr, uint64_t* out_point, uint64_t* out_prod_claims /* n_prod ext */,
                     uint64_t* out_logup_p_claims, uint64_t* out_logup_q_claims);

/* ---- rotation argument (a11; gkr_iop/src/utils.rs:19-102, gkr/booleanhypercube.rs, layer/cpu/mod.rs:249-389) ---- */
int orc_cyclic_table(int log2, uint32_t* out);
int orc_rotation_next_base_mle(const uint64_t* in, int num_vars, int log2, uint64_t* out);
int orc_rotation_selector(const uint64_t* eq, int num_vars, int subgroup_size, int log2, uint64_t* out);
int orc_rotation_points(const uint64_t* point, int n, int log2, uint64_t* left, uint64_t* right);
int orc_prove_rotation(const orc_mle* wit, int n_wit, const int* src, const int* tgt, int n_pairs, int subgroup_size, int log2,
                       const uint64_t* rt, int n, orc_transcript* tr, uint64_t* out_msgs, uint64_t* out_evals, uint64_t* out_origin,
                       uint64_t* out_left, uint64_t* out_right);

/* ---- Basefold commit path (a14) — PARITY UNPINNED, see commit.c ---- */
uint64_t orc_two_adic_generator(int bits);
void orc_dft_bitrev(const uint64_t* in, int log_n, int inverse, uint64_t* out);
void orc_poseidon2_permute(uint64_t* state8, const uint64_t* params138);
void orc_poseidon2_default_params(uint64_t* params138);
void orc_merkle_commit(const uint64_t* col_major, int log_rows, int width, const uint64_t* params138, uint64_t* out_levels);


/* mixed-height Merkle commitment (p3 MerkleTreeMmcs), commit.c */
int orc_mmcs_log_max(int n_mats, const int* log_rows);
void orc_mmcs_commit(int n_mats, const int* log_rows, const int* width, const uint64_t* const* col_major, const uint64_t* params138,
                     uint64_t* out_levels /* 4 * (2^(log_max+1) - 1) */);
void orc_mmcs_open(int n_mats, const int* log_rows, const int* width, const uint64_t* const* col_major, const uint64_t* levels,
                   size_t index, uint64_t* rows_out /* sum of widths */, uint64_t* path_out /* 4 * log_max */);
int orc_mmcs_verify(int n_mats, const int* log_rows, const int* width, const uint64_t* root4, size_t index, const uint64_t* rows,
                    const uint64_t* path, const uint64_t* params138);

/* ---- Basefold batch open + verifier (a15), see basefold.c ----
 * `n_commits` commitments ("rounds" of PCS::batch_open: witness, fixed), commitment c holds commit_sizes[c] consecutive matrices
 * of the flat arrays nv / width / traces / points / evals.
 * proof layout (words): [sumcheck msgs 4n][commit roots 4n][final message 2*n_mats][pow witness 1] then per query
 *   [index 1] per commitment [opened rows of its matrices: sum width_m][MMCS path 4*(max nv_m + rate_log)]
 *   per round r [sibling 2][path 4*(n+rate_log-r-1)] */
void orc_fft_bitrev(uint64_t* a, int log_n);
size_t orc_basefold_query_words(int n_commits, const int* commit_sizes, const int* nv, const int* width, int rate_log);
size_t orc_basefold_proof_words(int n_commits, const int* commit_sizes, const int* nv, const int* width, int rate_log, int n_queries);
int orc_basefold_open(int n_commits, const int* commit_sizes, const int* nv, const int* width,
                      const uint64_t* const* traces /* column-major base */, const uint64_t* const* points,
                      const uint64_t* const* evals /* width ext per matrix */, int rate_log, int n_queries, int pow_bits,
                      const uint64_t* params138, orc_transcript* tr, uint64_t* proof);
void orc_basefold_commit_roots(int n_commits, const int* commit_sizes, const int* nv, const int* width, const uint64_t* const* traces,
                               int rate_log, const uint64_t* params138, uint64_t* roots /* 4 per commitment */);
int orc_basefold_verify(int n_commits, const int* commit_sizes, const int* nv, const int* width, const uint64_t* roots,
                        const uint64_t* const* points, const uint64_t* const* evals, int rate_log, int n_queries, int pow_bits,
                        const uint64_t* params138, orc_transcript* tr, const uint64_t* proof);

/* ---- witness assignment of the ADD / SUB chips (witgen.c; reference arith.rs:101-142 and the files cited there) ---- */
size_t orc_step_record_bytes(void);
void orc_step_record_r(void* out, uint64_t cycle, uint32_t pc, uint8_t kind, uint8_t rs1, uint8_t rs2, uint8_t rd, uint32_t rs1_val,
                       uint32_t rs2_val, uint32_t rd_before, uint32_t rd_after, uint64_t prev_cycle);
int orc_witgen_arith(const uint32_t* cols, int is_sub, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                     uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* AND / OR / XOR (logic_circuit.rs:66-160): cols[29] in LogicRColumnMap order; lk_logic = the 2^16 counters of the op's table */
int orc_witgen_logic_r(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                       uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_logic);
/* ADDI (arith_imm_circuit_v2.rs:85-117): cols[19] in AddiColumnMap order; orc_step_record_i = StepRecord::new_i_instruction */
void orc_step_record_i(void* out, uint64_t cycle, uint32_t pc, uint8_t kind, uint8_t rs1, uint8_t rd, int32_t imm, uint32_t rs1_val,
                       uint32_t rd_before, uint32_t rd_after, uint64_t prev_cycle);
int orc_witgen_addi(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                    uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* ANDI / ORI / XORI (logic_imm_circuit_v2.rs:105-130,195-224): cols[25] in LogicIColumnMap order */
int orc_witgen_logic_i(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                       uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_logic);
/* LUI (riscv/lui.rs:100-120): cols[17] in LuiColumnMap order */
int orc_witgen_lui(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* SLT / SLTU (slt_circuit_v2.rs:86-119, signed_limbs.rs:150-236): cols[27] in SltColumnMap order */
int orc_witgen_slt(const uint32_t* cols, int is_signed, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                   uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* SLTI / SLTIU (slti_circuit_v2.rs:104-140): cols[23] in SltiColumnMap order */
int orc_witgen_slti(const uint32_t* cols, int is_signed, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                    uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* branches (branch_circuit_v2.rs:143-209): is_eq = 0 -> cols[23] BranchCmpColumnMap order, flag = is_signed; is_eq = 1 -> cols[20] BranchEqColumnMap, flag = is_beq */
void orc_step_record_b(void* out, uint64_t cycle, uint32_t pc, uint32_t pc_after, uint8_t kind, uint8_t rs1, uint8_t rs2, int32_t imm, uint32_t rs1_val,
                       uint32_t rs2_val, uint64_t prev_cycle);
int orc_witgen_branch(const uint32_t* cols, int is_eq, int flag, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                      uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* LW / SW (load_v2.rs:197-255, store_v2.rs:138-177): cols[24] in LwColumnMap / SwColumnMap order */
void orc_step_record_mem(void* out, int is_store, uint64_t cycle, uint32_t pc, uint8_t kind, uint8_t rs1, uint8_t rs2_or_rd, int32_t imm, uint32_t rs1_val,
                         uint32_t rs2_val, uint32_t rd_before, uint32_t rd_after, uint32_t mem_byte_addr, uint32_t mem_before, uint32_t mem_after,
                         uint64_t prev_cycle, uint64_t mem_prev_cycle);
int orc_witgen_mem(const uint32_t* cols, int is_store, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* JALR (jalr_v2.rs:146-190): cols[23] in JalrColumnMap order */
int orc_witgen_jalr(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                    uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* shifts (shift_circuit_v2.rs:242-293,359-396,485-521): cols[48] in ShiftRColumnMap order / cols[41] in ShiftIColumnMap order; kind 0 left, 1 logical right, 2 arithmetic right */
int orc_witgen_shift(const uint32_t* cols, int is_imm, int kind, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                     uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_double_u8,
                     uint32_t* lk_xor);
/* LH / LHU / LB / LBU (load_v2.rs:197-255): cols[30] in LoadSubColumnMap order, 0xFFFFFFFF for the Option fields the variant lacks */
int orc_witgen_load_sub(const uint32_t* cols, int load_width, int is_signed, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                        uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* MUL / MULH / MULHU / MULHSU (mulh_circuit_v2.rs:234-333,427-487): cols[27] in MulColumnMap order, 0xFFFFFFFF in the Option fields for MUL */
int orc_witgen_mul(const uint32_t* cols, int kind, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* DIV / DIVU / REM / REMU (div_circuit_v2.rs:391-536): cols[40] in DivColumnMap order */
int orc_witgen_div(const uint32_t* cols, int kind, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch);
/* JAL (jal_v2.rs:99-127) and AUIPC (auipc.rs:149-187): cols[14] / cols[22] in JalColumnMap / AuipcColumnMap order; double_u8 key a << 8 | b */
void orc_step_record_j(void* out, uint64_t cycle, uint32_t pc, uint32_t pc_after, uint8_t kind, uint8_t rd, int32_t imm, uint32_t rd_before,
                       uint32_t rd_after, uint64_t prev_cycle);
int orc_witgen_jal(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_double_u8, uint32_t* lk_xor);
int orc_witgen_auipc(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                     uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_double_u8, uint32_t* lk_xor);

#ifdef __cplusplus
}
#endif
#endif
