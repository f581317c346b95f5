/*
 * ceno_hip.h — C ABI of libceno_hip.so, the MI355X (gfx950) device back end for Ceno's
 * GKR / sumcheck prover core.
 *
 * This is the drop-in boundary: the entry points below are what a Rust `HipBackend/HipProver`
 * implementing `gkr_iop::hal::ProverBackend` + `ceno_zkvm::scheme::hal::ProverDevice`
 * (reference gkr_iop/src/hal.rs:11-68, ceno_zkvm/src/scheme/hal.rs:19-35) binds over FFI, exactly
 * where the reference's GPU arm calls the external `ceno_gpu` HAL (catalogue: SURVEY.md §2.2).
 * INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *  - Field elements are canonical little-endian uint64 (< p = 2^64-2^32+1).  An extension element
 *    (GoldilocksExt2 = F_p[X]/(X^2-7)) is two consecutive words [c0, c1].
 *  - Multilinear polynomials are dense evaluation tables; index bit k <-> variable k (LSB first,
 *    reference gkr_iop/src/utils.rs:215-232).
 *  - Every function returns 0 on success or a negative ceno_hip_status; the message is available
 *    from ceno_hip_last_error().  Nothing throws or aborts across the boundary
 *    (reference: Result<_, HalError>, ceno_zkvm/src/scheme/gpu/mod.rs:348-352).
 *  - Every call that touches the device takes an explicit stream (reference: thread-bound streams,
 *    gkr_iop/src/gpu/mod.rs:87-154).  `NULL` selects the context's default stream.  The library is
 *    re-entrant across streams; host buffers passed in are only borrowed for the call.
 *  - Handles are opaque; device memory is owned by the library pool unless wrapped from outside.
 *  - Memory: freeing a handle does not wait for the device.  A freed block remembers the stream the freeing thread used last
 *    (one lane = one thread = one stream); the pool hands the block to a different stream only after that stream has drained
 *    (it never waits: it takes another block or allocates), so a lane may free tables whose kernels are still queued.  A caller's own
 *    HIP stream passed as ceno_hip_stream is registered with that bookkeeping the first time a thread uses it (ceno_hip_stream_adopt
 *    does it explicitly); handles own the stream they were begun on and free their blocks with it.  Blocks go back to the DRIVER
 *    (hipFree waits for every kernel on the device) only while no pipelined sumcheck is alive on any lane — from
 *    ceno_hip_mem_trim, from the pool_bytes cap, or when the cache is several times the memory in use (largest idle blocks first) —
 *    and no pipelined sumcheck starts while such a trim is under way: call ceno_hip_mem_trim between phases.
 *  - The latency-bound ends of the work are finished by the calling HOST thread: the last rounds of a pipelined sumcheck
 *    (CENO_HIP_HOST_TAIL) and the top levels of every Merkle tree (CENO_HIP_HOST_TOP).  Results are bit-identical for every split.
 *  - The Fiat–Shamir transcript stays with the caller: sumcheck is exposed round by round
 *    (the reference passes `&mut BasicTranscript` into the HAL, gkr_iop/src/gkr/layer/gpu/mod.rs:252-270;
 *    a C ABI cannot take a Rust generic, so control is inverted).
 */
#ifndef CENO_HIP_H
#define CENO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ceno_hip_ctx ceno_hip_ctx;
typedef struct ceno_hip_mle ceno_hip_mle;             /* device MLE; cf. MultilinearExtensionGpu, gkr_iop/src/gpu/mod.rs:157-370 */
typedef struct ceno_hip_sumcheck ceno_hip_sumcheck;   /* in-flight sumcheck; cf. prove_generic_sumcheck_gpu */
typedef struct ceno_hip_tower ceno_hip_tower;         /* built tower witness; cf. GpuProverSpec */
typedef struct ceno_hip_merkle ceno_hip_merkle;       /* committed Merkle tree; cf. basefold PcsData */
typedef void* ceno_hip_stream;                        /* hipStream_t */

typedef enum ceno_hip_status {
    CENO_HIP_OK = 0,
    CENO_HIP_ERR_INVALID = -1,     /* bad argument / plan */
    CENO_HIP_ERR_HIP = -2,         /* HIP runtime error */
    CENO_HIP_ERR_OOM = -3,         /* pool capacity exceeded / allocation failed */
    CENO_HIP_ERR_STATE = -4,       /* call out of order (e.g. round after finish) */
    CENO_HIP_ERR_UNSUPPORTED = -5
} ceno_hip_status;

/* ------------------------------------------------------------------------------------------------
 * context, streams, memory  (reference: CUDA_HAL lazy global + pool, gkr_iop/src/gpu/mod.rs:53-66;
 * get_cuda_mem_info / mem_pool booking, ceno_zkvm/src/scheme/gpu/mod.rs:226-267)
 * ---------------------------------------------------------------------------------------------- */
int ceno_hip_init(int device, size_t pool_bytes /* 0 = unlimited */, ceno_hip_ctx** out);
void ceno_hip_destroy(ceno_hip_ctx* ctx);
const char* ceno_hip_last_error(ceno_hip_ctx* ctx);   /* ctx may be NULL: last init error */
const char* ceno_hip_version(void);
/* make the context's device current for the calling thread (ensure_context, gkr_iop/src/gpu/mod.rs:91-92).  Every entry
 * point that takes a stream does this itself; worker threads that call the HIP runtime directly call it once. */
int ceno_hip_make_current(ceno_hip_ctx* ctx);
int ceno_hip_device(const ceno_hip_ctx* ctx);
int ceno_hip_stream_create(ceno_hip_ctx* ctx, ceno_hip_stream* out);
/* stream for proving lane `lane` (concurrent chip proving, ceno_zkvm/src/scheme/scheduler.rs:73-85): consecutive lanes
 * get different stream priorities so that they land on different hardware queues and really overlap */
int ceno_hip_stream_create_lane(ceno_hip_ctx* ctx, int lane, ceno_hip_stream* out);
/* the CONTEXT's stream of lane `lane` (0..63): created on first use (priorities rotate as in ceno_hip_stream_create_lane), reused by
 * every later call, destroyed with the context — creating a HIP stream costs ~4 ms and destroying one ~2 ms, so the chip scheduler
 * (ceno_prover_lanes_run; reference: the thread-bound streams of scheduler.rs:73-85,231-336) never does either per run.  Do NOT pass
 * it to ceno_hip_stream_destroy. */
int ceno_hip_lane_stream(ceno_hip_ctx* ctx, int lane, ceno_hip_stream* out);
int ceno_hip_stream_destroy(ceno_hip_ctx* ctx, ceno_hip_stream s);
/* registers a HIP stream the caller created itself with the pool's cross-stream ordering (see "Memory" above); the stream must
 * stay alive until ceno_hip_stream_destroy (which also forgets it) or the context is destroyed */
int ceno_hip_stream_adopt(ceno_hip_ctx* ctx, ceno_hip_stream s);
/* the calling thread's following allocations and frees belong to work on stream `s` (NULL: the default stream).  Every entry
 * point that takes a stream does this implicitly; a thread that drives SEVERAL streams calls it before allocating for one. */
int ceno_hip_stream_bind(ceno_hip_ctx* ctx, ceno_hip_stream s);
int ceno_hip_stream_sync(ceno_hip_ctx* ctx, ceno_hip_stream s);
/* free/total = device memory; pool_used = bytes held by live handles; pool_cached = bytes parked in the pool */
int ceno_hip_mem_info(ceno_hip_ctx* ctx, size_t* free_bytes, size_t* total_bytes, size_t* pool_used, size_t* pool_cached);
/* bookkeeping that must return to zero when no sumcheck handle is alive (tests): live pipelined sumchecks (the pool's soft-cap
 * trim waits for zero) and the residency budget booked by persistent mid-round kernels (units of 1/64 compute unit) */
int ceno_hip_debug_state(ceno_hip_ctx* ctx, int* pipelined_live, int* mid_units_in_flight);
/* CENO_HIP_HOST_TIMING=1: print where the library's HOST threads spent their time since the last dump (calls, total, mean and longest per labelled
 * scope; process-wide) to stderr and start counting afresh — ceno_hip_destroy does the same.  Nothing without the switch.  `what` (may be NULL)
 * heads the lines.  A measurement aid: call it between the repetitions of a benchmark to see the steady state apart from the first run. */
void ceno_hip_host_timing_dump(const char* what);
/* release cached blocks (trim_mem_pool, e2e.rs:3331-3334); waits for the device; a no-op while pipelined sumchecks are alive on any lane */
int ceno_hip_mem_trim(ceno_hip_ctx* ctx);
/* high-water mark of the bytes handed out by the pool since the last reset (reset != 0: restart the mark at the current usage).  What a
 * scheduler's booking estimates are checked against (the reference asserts its estimator against real usage,
 * ceno_zkvm/src/scheme/gpu/memory.rs:54-145). */
size_t ceno_hip_mem_peak(ceno_hip_ctx* ctx, int reset);
/* booking of estimated task footprints by a chip scheduler (mem_pool try_book_capacity / unbook_capacity /
 * get_booked_total, ceno_zkvm/src/scheme/scheduler.rs:342-347,390,622-652): refused (CENO_HIP_ERR_OOM, nothing is
 * allocated) when live allocations + bookings + bytes would exceed pool_bytes (or the device memory when unlimited) */
int ceno_hip_mem_book(ceno_hip_ctx* ctx, size_t bytes);
int ceno_hip_mem_unbook(ceno_hip_ctx* ctx, size_t bytes);
size_t ceno_hip_mem_booked(ceno_hip_ctx* ctx);
/* high-water mark of the booked total since the last reset: what the scheduler promised at its busiest moment, to hold against
 * ceno_hip_mem_peak (what the booked tasks really took) */
size_t ceno_hip_mem_booked_peak(ceno_hip_ctx* ctx, int reset);

/* ------------------------------------------------------------------------------------------------
 * MLE handles  (alloc_elems_on_device / alloc_ext_elems_from_host / Buffer::to_cpu_vec)
 * ---------------------------------------------------------------------------------------------- */
/* (takes its block for the stream the calling thread is bound to: the last call that took a stream, or ceno_hip_stream_bind) */
int ceno_hip_mle_alloc(ceno_hip_ctx* ctx, int num_vars, int is_ext, ceno_hip_mle** out);
int ceno_hip_mle_upload(ceno_hip_ctx* ctx, const uint64_t* host, int num_vars, int is_ext, ceno_hip_stream s, ceno_hip_mle** out);
/* borrow device memory owned by the caller (e.g. a torch tensor); never freed by the library */
int ceno_hip_mle_wrap(ceno_hip_ctx* ctx, uint64_t* device_ptr, int num_vars, int is_ext, ceno_hip_mle** out);
/* borrowed view of the contiguous chunk [chunk*2^sub_vars, (chunk+1)*2^sub_vars) of a parent (as_view_chunk, gkr_iop/src/gpu/mod.rs:244-253) */
int ceno_hip_mle_view_chunk(ceno_hip_ctx* ctx, ceno_hip_mle* parent, int sub_vars, size_t chunk, ceno_hip_mle** out);
/* new MLE with one variable less holding the even (odd = 0) or odd (odd = 1) entries of `m`
 * (filter_mle_even_odd_batch, ceno_zkvm/src/scheme/gpu/util.rs:186-266) */
int ceno_hip_mle_filter_even_odd(ceno_hip_ctx* ctx, const ceno_hip_mle* m, int odd, ceno_hip_stream s, ceno_hip_mle** out);
int ceno_hip_mle_download(ceno_hip_ctx* ctx, const ceno_hip_mle* m, uint64_t* host, ceno_hip_stream s);  /* synchronises s */
int ceno_hip_mle_free(ceno_hip_ctx* ctx, ceno_hip_mle* m);
int ceno_hip_mle_num_vars(const ceno_hip_mle* m);
int ceno_hip_mle_is_ext(const ceno_hip_mle* m);
uint64_t* ceno_hip_mle_device_ptr(const ceno_hip_mle* m);
/* word i of the table = SplitMix64(seed, word_offset + i) reduced mod p (BASELINE.md synthetic inputs) */
int ceno_hip_mle_fill_splitmix(ceno_hip_ctx* ctx, ceno_hip_mle* m, uint64_t seed, uint64_t word_offset, ceno_hip_stream s);
/* every word zero, queued on `s` (a fresh table is NOT zero: pool blocks are recycled).  MultilinearExtension of zeros /
 * RowMajorMatrix::new(.., InstancePaddingStrategy::Default) (witness crate); also how a caller clears lookup counters kept in a table's words */
int ceno_hip_mle_fill_zero(ceno_hip_ctx* ctx, ceno_hip_mle* m, ceno_hip_stream s);

/* MultilinearExtension::evaluate(point) — out2 = f(point), point has num_vars ext elements */
int ceno_hip_mle_evaluate(ceno_hip_ctx* ctx, const ceno_hip_mle* m, const uint64_t* point, uint64_t* out2, ceno_hip_stream s);
/* fix_variables (low variables first): out has num_vars - n_fix variables, always ext */
int ceno_hip_mle_fix_variables(ceno_hip_ctx* ctx, const ceno_hip_mle* m, const uint64_t* point, int n_fix, ceno_hip_stream s, ceno_hip_mle** out);
/* MultilinearExtension::evaluate for n base-field tables in one pass: out[2 j ..] = cols[j](point[0 .. num_vars_j)), every table at the
 * prefix of ONE point of point_len >= num_vars_j ext elements (the final evaluations of a batched sumcheck whose smaller tables bind the
 * first variables, ceno_zkvm/src/scheme/cpu/mod.rs:1338-1361).  `out` is host memory; returns when it is filled. */
int ceno_hip_mle_evaluate_prefix_batch(ceno_hip_ctx* ctx, int n, ceno_hip_mle* const* cols, const uint64_t* point, int point_len, ceno_hip_stream s,
                                       uint64_t* out);
/* linear combinations of base-field columns with ext coefficients, n_groups at once (group g = cols[group_offsets[g] .. group_offsets[g+1]),
 * one table size per group): out0[g][x] = sum_j coeffs[2j] cols[j][x],  out1[g][x] = sum_j coeffs[2j+1] cols[j][x]  (mod p), i.e.
 * sum_j c_j col_j = out0 + X out1 as two BASE-field tables.  The monomial form of a chip's record RLCs (selector x column for most
 * columns, gkr_iop/src/gkr/layer/zerocheck_layer.rs:118-140) enters prove_batched_main_constraints as two tables per selector this way. */
int ceno_hip_lincomb_base_batch(ceno_hip_ctx* ctx, int n_groups, const uint32_t* group_offsets, ceno_hip_mle* const* cols, const uint64_t* coeffs,
                                ceno_hip_stream s, ceno_hip_mle** out0, ceno_hip_mle** out1);

/* ------------------------------------------------------------------------------------------------
 * eq tables and selectors  (build_mle_as_ceno / ordered_sparse_selector_gpu,
 * gkr_iop/src/gkr/layer/gpu/utils.rs:121-190; semantics gkr_iop/src/selector.rs:131-245)
 * ---------------------------------------------------------------------------------------------- */
typedef enum ceno_hip_selector_kind {
    CENO_HIP_SEL_WHOLE = 0,
    CENO_HIP_SEL_PREFIX = 1,          /* rows [offset, offset+num_instances) keep eq, others 0 */
    CENO_HIP_SEL_ORDERED_SPARSE = 2,  /* within each 2^sparse_num_vars chunk keep `sparse_indices`; chunks >= num_instances are 0 */
    CENO_HIP_SEL_QUARK_LT = 3         /* QuarkBinaryTreeLessThan */
} ceno_hip_selector_kind;

/* out[x] = scalar * eq(x, point); scalar2 may be NULL (= 1).  The scalar carries eq over the
 * high (sharded-away) variables when the hypercube is split across GPUs. */
int ceno_hip_eq_build(ceno_hip_ctx* ctx, const uint64_t* point, int num_vars, const uint64_t* scalar2, ceno_hip_stream s, ceno_hip_mle** out);
int ceno_hip_selector_build(ceno_hip_ctx* ctx, int kind, const uint64_t* point, int num_vars, size_t offset, size_t num_instances,
                            const uint32_t* sparse_indices, int n_sparse, int sparse_num_vars, ceno_hip_stream s, ceno_hip_mle** out);

/* all Whole / Prefix selectors of a batch (the chips of prove_batched_main_constraints, ceno_zkvm/src/scheme/cpu/mod.rs:1200-1234) in two
 * launches; table k = ceno_hip_selector_build(kinds[k], points[k], num_vars[k], offsets[k], num_instances[k]).  Other kinds: one at a time. */
int ceno_hip_selector_build_batch(ceno_hip_ctx* ctx, int n, const int* kinds, const uint64_t* const* points, const int* num_vars,
                                  const size_t* offsets, const size_t* num_instances, ceno_hip_stream s, ceno_hip_mle** outs);

/* ------------------------------------------------------------------------------------------------
 * rotation argument of the keccak-style chips  (rotation_next_base_mle_gpu / rotation_selector_gpu,
 * gkr_iop/src/gkr/layer/gpu/utils.rs:231-336; semantics gkr_iop/src/utils.rs:19-76,
 * cyclic tables gkr_iop/src/gkr/booleanhypercube.rs:10-113)
 * ---------------------------------------------------------------------------------------------- */
/* out[c*G + x^i] = in[c*G + x^(i+1)] within every chunk of G = 2^cyclic_group_log2 (5 or 6) entries, position 0 fixed;
 * `in` is a base-field table, `out` a new base-field table */
int ceno_hip_rotation_next_base_mle(ceno_hip_ctx* ctx, const ceno_hip_mle* in, int cyclic_group_log2, ceno_hip_stream s, ceno_hip_mle** out);
/* eq(x, point) kept on the first `cyclic_subgroup_size` elements x^i of every chunk, zero elsewhere */
int ceno_hip_rotation_selector_build(ceno_hip_ctx* ctx, const uint64_t* point, int num_vars, int cyclic_subgroup_size,
                                     int cyclic_group_log2, ceno_hip_stream s, ceno_hip_mle** out);

/* ------------------------------------------------------------------------------------------------
 * element-wise witness inference  (wit_infer_by_monomial_expr, gkr_iop/src/gpu/mod.rs:599-609)
 *   outs[o][x] = sum_{t in terms of o} coeff_t * prod_{j in S_t} mles[j][x]
 * terms are CSR: output o owns terms [out_term_offsets[o], out_term_offsets[o+1]); term t owns
 * factors term_mle_idx[term_offsets[t] .. term_offsets[t+1]).  Outputs are ext tables.
 * ---------------------------------------------------------------------------------------------- */
int ceno_hip_wit_infer(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, int num_mles, const uint64_t* term_coeffs,
                       const uint32_t* term_offsets, const uint32_t* term_mle_idx, int num_terms,
                       const uint32_t* out_term_offsets, int num_outs, int num_vars, ceno_hip_stream s, ceno_hip_mle** outs);
/* MANY plans in one launch and one upload (the records of all chips of a shard: ceno_prover_create_chip_proofs); plan i as the arguments
 * of ceno_hip_wit_infer, its outputs into plan i's outs[0 .. num_outs).  On failure no output is left allocated. */
typedef struct ceno_hip_wit_plan {
    ceno_hip_mle* const* mles;
    int num_mles;
    const uint64_t* term_coeffs;
    const uint32_t* term_offsets;
    const uint32_t* term_mle_idx;
    int num_terms;
    const uint32_t* out_term_offsets;
    int num_outs, num_vars;
    ceno_hip_mle** outs;
} ceno_hip_wit_plan;
int ceno_hip_wit_infer_many(ceno_hip_ctx* ctx, const ceno_hip_wit_plan* plans, int n, ceno_hip_stream s);

/* ------------------------------------------------------------------------------------------------
 * generic sumcheck  (prove_generic_sumcheck_gpu / _v2, gkr_iop/src/gkr/layer/gpu/mod.rs:259-271,
 * ceno_zkvm/src/scheme/gpu/mod.rs:891-902,2968-2982)
 *
 * Claim: sum_{x in {0,1}^n} sum_t c_t prod_{j in S_t} f_j(x), n = max_num_vars, degree d = max_degree.
 * Round i binds variable i; the round message is exactly d extension elements p(1..d).
 * An MLE with n' < n variables is front-loaded: f(x_0..x_{n'-1}) * x_{n'} ... x_{n-1}
 * (scheme/verifier.rs:233-237); all factors of one term must have the same num_vars.
 * Optional common-factor plan (CommonTermPlan, layer/gpu/utils.rs:69-119): group g multiplies the
 * sum of its terms by prod of common_mle_idx[common_offsets[g]..common_offsets[g+1]); terms listed in
 * a group are given by group_term_idx[group_term_offsets[g]..]; terms not in any group stand alone.
 *
 * Inputs are never modified (the first fold writes into library-owned half-size buffers).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ceno_hip_sumcheck_plan {
    int num_mles;
    int num_terms;
    const uint64_t* term_coeffs;        /* 2 * num_terms words */
    const uint32_t* term_offsets;       /* num_terms + 1 */
    const uint32_t* term_mle_idx;       /* term_offsets[num_terms] */
    int num_groups;                     /* 0 = no common-factor plan */
    const uint32_t* group_term_offsets; /* num_groups + 1 */
    const uint32_t* group_term_idx;
    const uint32_t* common_offsets;     /* num_groups + 1 */
    const uint32_t* common_mle_idx;
    int max_num_vars;
    int max_degree;
} ceno_hip_sumcheck_plan;

int ceno_hip_sumcheck_begin(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan,
                            ceno_hip_stream s, ceno_hip_sumcheck** out);
/* ceno_hip_sumcheck_begin for a plan whose common factors are selector tables of a known form — the main-constraint sumchecks
 * (ceno_zkvm/src/scheme/cpu/mod.rs:1052-1390: every group is selector x sum of column products, the selectors are
 * SelectorType::Whole / Prefix, gkr_iop/src/selector.rs:131-245).  Table eq_mle_idx[k] of the plan IS eq(., eq_points[k]) on the rows
 * [eq_lo[k], eq_hi[k]) and 0 elsewhere (what ceno_hip_selector_build produces for WHOLE: [0, 2^n) and PREFIX: [offset, offset +
 * num_instances)); eq_points[k] holds as many extension elements as the table has variables.  The caller vouches for the declaration.
 * Where every group of a chip has one such common factor and all of them share the point, the rounds evaluate the chip's quotient by
 * eq(X, rt_i) at one point fewer and the library completes each message from the chip's running claim: the same words as without the
 * declaration (an all-zero declaration list is ceno_hip_sumcheck_begin).  Such a handle produces host messages only
 * (ceno_hip_sumcheck_round_dev fails).  CENO_HIP_GEN_EQF=0 ignores the declarations. */
int ceno_hip_sumcheck_begin_eq(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan, int num_eq, const int* eq_mle_idx,
                               const uint64_t* const* eq_points, const size_t* eq_lo, const size_t* eq_hi, ceno_hip_stream s, ceno_hip_sumcheck** out);
/* how many components (chips) of the plan run in the eq-factored form (0: none, the declarations did not apply) */
int ceno_hip_sumcheck_eq_components(const ceno_hip_sumcheck* sc);
/* eq-factored round launches this context has issued so far (a statistic for tests and A/B runs) */
uint64_t ceno_hip_stat_eq_launches(const ceno_hip_ctx* ctx);
/* With CENO_HIP_PLAN_REPORT=1 in the environment: a JSON array describing the round kernels the size classes of the multi-class sumcheck
 * built LAST on this context were given — per class its variables, tables, terms, "path" ("eq-factored" | "generic" | "two-kernel" |
 * "dense"), components, column blocks, tables staged twice, pairs per tile.  "" otherwise.  Valid until the next begin on the context. */
const char* ceno_hip_plan_report(const ceno_hip_ctx* ctx);
/* Produce the message of the next round.  `challenge2` is the challenge of the PREVIOUS round
 * (NULL for round 0): the tables are folded with it and the new message accumulated in one pass.
 * out_evals receives max_degree ext elements (host memory). Synchronises the stream. */
int ceno_hip_sumcheck_round(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, const uint64_t* challenge2, uint64_t* out_evals);
/* same, but the (partial) message is left in device memory `dev_out_evals` (max_degree ext) and the
 * call does not synchronise — for hypercube shards whose partials are combined by a collective. */
int ceno_hip_sumcheck_round_dev(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, const uint64_t* challenge2, uint64_t* dev_out_evals);
/* bind the last variable; final_evals receives num_mles ext elements: f_j(r_0..r_{nv_j - 1})
 * (get_mle_flatten_final_evaluations, gkr_iop/src/gkr/layer/cpu/mod.rs:229-230) */
int ceno_hip_sumcheck_finish(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, const uint64_t* last_challenge2, uint64_t* final_evals);
int ceno_hip_sumcheck_rounds_done(const ceno_hip_sumcheck* sc);
/* device bytes a sumcheck allocates on top of its inputs (estimate_sumcheck_memory, ceno_zkvm/src/scheme/gpu/memory.rs:413-433):
 * what a scheduler books (ceno_hip_mem_book) before starting the task.  mle_num_vars == NULL: all MLEs have max_num_vars. */
size_t ceno_hip_sumcheck_estimate_memory(int max_num_vars, int max_degree, const int* mle_num_vars, int num_mles, int num_terms);
/* Device pointer, element kind and current number of variables of table `mle_index` as the next round will
 * read it (i.e. not yet folded with the challenge that call is going to receive).  Valid until the next
 * round/finish call; used by the sharded driver to gather small shards onto every rank. */
int ceno_hip_sumcheck_table(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int mle_index, uint64_t** device_ptr, int* is_ext, int* num_vars);
/* The same table copied to HOST memory as extension elements (cap_ext = capacity of out_host in elements), wherever it lives: on the device,
 * or in the host's copy once the host has taken the last rounds of a round-by-round sumcheck over (then ceno_hip_sumcheck_table fails).
 * Synchronises the handle's stream.  Used by the row-sharded tower prover to gather the folded shards (ceno_amd/host/dist_gkr.cpp). */
int ceno_hip_sumcheck_table_host(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int mle_index, uint64_t* out_host, size_t cap_ext, int* num_vars);
/* n tables at once: every copy queued, ONE wait (the sharded main constraints fetch hundreds of folded tables before their gathered tail) */
int ceno_hip_sumcheck_tables_host(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int n, const int* mle_indices, uint64_t* const* outs_host,
                                  const size_t* caps_ext, int* num_vars);
/* Opt in (before round 0) to pipelined rounds: all round kernels are enqueued at round 0 and pick their
 * challenges up from a pinned-memory mailbox, which removes the launch latency from every round.  The
 * caller promises to call ceno_hip_sumcheck_round back to back (a queued kernel gives up after CENO_HIP_PIPE_TIMEOUT_S (60 s) without
 * its challenge and the next call then fails with CENO_HIP_ERR_HIP).  Only taken for plans the library can
 * pipeline (one dense product over tables of max_num_vars variables); otherwise a no-op. */
int ceno_hip_sumcheck_set_pipelined(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int on);
int ceno_hip_sumcheck_free(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc);

/* ------------------------------------------------------------------------------------------------
 * tower witness  (build_prod_tower_from_virtual_ext_batch / build_logup_tower_from_virtual_ext_batch,
 * GpuProverSpec::get_output_evals — ceno_zkvm/src/scheme/gpu/mod.rs:2365-2402,379-410; CPU semantics
 * ceno_zkvm/src/scheme/utils.rs:402-659)
 * ---------------------------------------------------------------------------------------------- */
/* interleave `k` record MLEs into 2 limbs (default-padded) and build all product layers.
 * Resulting tower has num_vars = row_vars' + ceil_log2(next_pow2(k)) layers, layer l = 2 limbs of 2^l. */
int ceno_hip_tower_build_prod(ceno_hip_ctx* ctx, ceno_hip_mle* const* records, int k, size_t num_instances,
                              const uint64_t* default2, ceno_hip_stream s, ceno_hip_tower** out);
/* LogUp tower over denominators q (records) and optional numerators p (NULL = all ones):
 * layer l = 4 limbs p1,p2,q1,q2 of 2^l */
int ceno_hip_tower_build_logup(ceno_hip_ctx* ctx, ceno_hip_mle* const* p_records, ceno_hip_mle* const* q_records, int k,
                               size_t num_instances, const uint64_t* default2, ceno_hip_stream s, ceno_hip_tower** out);
/* build directly from already interleaved last-layer limbs (infer_tower_product_witness / _logup_witness) */
int ceno_hip_tower_from_last_layer(ceno_hip_ctx* ctx, ceno_hip_mle* const* limbs, int n_limbs /* 2 = prod; 4 = logup p1,p2,q1,q2; */,
                                   ceno_hip_stream s, ceno_hip_tower** out);
/* MANY towers in level-synchronous launches (one interleave launch, one launch per layer size, one for the contiguous tops) instead of ~11
 * dependent launches per tower: the towers of all chips of a shard (ceno_prover_create_chip_proofs).  Tower i: logup == 0 — a product tower
 * over records[0 .. k) as ceno_hip_tower_build_prod; logup != 0 — a LogUp tower with denominators records[0 .. k) and numerators[0 .. k)
 * (NULL: all one) as ceno_hip_tower_build_logup.  Same layers, bit for bit.  On failure nothing is built. */
typedef struct ceno_hip_tower_spec {
    ceno_hip_mle* const* records;
    ceno_hip_mle* const* numerators;
    int k, logup;
    size_t num_instances;
    uint64_t default2[2];
} ceno_hip_tower_spec;
int ceno_hip_tower_build_many(ceno_hip_ctx* ctx, const ceno_hip_tower_spec* specs, int n, ceno_hip_stream s, ceno_hip_tower** out /* n */);
/* The same towers straight from the record EXPRESSIONS (the reference's build_prod_tower_from_virtual_ext_batch /
 * build_logup_tower_from_virtual_ext_batch, ceno_zkvm/src/scheme/gpu/mod.rs:2365-2402): tower i reads records first_record .. + k (and, a LogUp
 * tower with numerators, first_numerator .. + k) of plans[plan] — what ceno_hip_wit_infer would write into record tables of 2^num_vars rows
 * (one row per slot of the padded trace: num_instances = 2^num_vars) — without the tables ever existing (plans[..].outs is not used).
 * CENO_HIP_ERR_UNSUPPORTED when a plan is too large for the kernel's LDS stage: the caller materialises the records then. */
typedef struct ceno_hip_virtual_tower_spec {
    int plan, first_record, k;
    int first_numerator; /* < 0: none (a product tower, or a LogUp tower whose numerators are all one) */
    int logup;
    uint64_t default2[2];
} ceno_hip_virtual_tower_spec;
int ceno_hip_tower_build_many_virtual(ceno_hip_ctx* ctx, const ceno_hip_wit_plan* plans, int n_plans, const ceno_hip_virtual_tower_spec* specs, int n,
                                      ceno_hip_stream s, ceno_hip_tower** out /* n */);
int ceno_hip_tower_num_vars(const ceno_hip_tower* t);   /* number of layers */
int ceno_hip_tower_num_limbs(const ceno_hip_tower* t);  /* 2 or 4 */
/* borrowed handle of limb `limb` of layer `layer` (valid while the tower lives) */
int ceno_hip_tower_layer(ceno_hip_ctx* ctx, ceno_hip_tower* t, int layer, int limb, ceno_hip_mle** out);
/* the same table as a bare device pointer (2^layer extension elements; NULL when out of range) */
const uint64_t* ceno_hip_tower_layer_ptr(const ceno_hip_tower* t, int layer, int limb);
int ceno_hip_tower_out_evals(ceno_hip_ctx* ctx, ceno_hip_tower* t, uint64_t* out /* n_limbs ext */, ceno_hip_stream s);
/* The top layers of a tower (layer l has n_limbs * 2^l elements) are stored back to back so that a host-side prover of the small
 * layers needs ONE copy: layers 0 .. n_layers-1 into host_out, layer l limb b at element offset n_limbs * (2^l - 1) + b * 2^l
 * (n_limbs * (2^n_layers - 1) extension elements in all; n_layers <= ceno_hip_tower_top_layers).  Synchronises. */
int ceno_hip_tower_download_top(ceno_hip_ctx* ctx, ceno_hip_tower* t, int n_layers, uint64_t* host_out, ceno_hip_stream s);
/* the same copy for SEVERAL towers with one synchronisation (the towers of a chip): layers 0 .. min(n_layers, top layers) - 1 of each go to
 * a host-side cache inside the handle, from which ceno_hip_tower_out_evals and ceno_hip_tower_download_top are then served */
int ceno_hip_tower_prefetch_tops(ceno_hip_ctx* ctx, ceno_hip_tower* const* towers, int n_towers, int n_layers, ceno_hip_stream s);
int ceno_hip_tower_top_layers(const ceno_hip_tower* t);  /* how many layers are contiguous (min(num_vars, 11)) */
int ceno_hip_tower_free(ceno_hip_ctx* ctx, ceno_hip_tower* t);

/* One tower layer sumcheck (the body of CpuTowerProver::create_proof's round loop,
 * ceno_zkvm/src/scheme/cpu/mod.rs:409-494):  sum_x eq(x, out_rt) * [ sum_i alpha_i a_i(x) b_i(x)
 *   + sum_k ( alpha_n,k (p1 q2 + p2 q1) + alpha_d,k q1 q2 ) ]  over layer `layer` of every tower that has it.
 * alpha_pows has n_prod + 2*n_logup ext elements in the reference's order.  Returns a sumcheck handle
 * whose MLE order is [eq, prod0.a, prod0.b, ..., logup0.p1, p2, q1, q2, ...] restricted to active towers. */
int ceno_hip_tower_layer_sumcheck_begin(ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup, int n_logup,
                                        int layer, const uint64_t* out_rt /* `layer` ext */, const uint64_t* alpha_pows,
                                        ceno_hip_stream s, ceno_hip_sumcheck** out);
/* How many leading rounds of this handle run on the fused tower-layer kernel when it is driven pipelined (one pass per round, the eq
 * factor taken out of the evaluation points; the messages are the same field elements): 0 for every other handle.  Diagnostics and
 * tests only; CENO_HIP_TOWER_FAST=0 turns the kernel off, CENO_HIP_TOWER_FAST_MIN_LOG moves the hand-over to the generic rounds. */
int ceno_hip_sumcheck_fused_eq_rounds(const ceno_hip_sumcheck* sc);
/* Optional, before round 0: the sum this sumcheck proves (one extension element), when the caller has it — the tower prover does: the
 * claim of layer l + 1 is the alpha-combination of layer l's evaluations merged at r_merge (scheme/cpu/mod.rs:497-541, as
 * TowerVerify recomputes it, scheme/verifier.rs:1587-1680).  The fused tower rounds then evaluate two points in round 0 instead of three;
 * handles without them ignore it.  A WRONG claim yields wrong messages for those rounds — the reference's prover never needs the sum,
 * so nothing checks it here. */
int ceno_hip_sumcheck_set_claim(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, const uint64_t* claim2);

/* ------------------------------------------------------------------------------------------------
 * Basefold commit path  (cuda_hal.basefold.batch_commit, ceno_zkvm/src/scheme/gpu/mod.rs:1642-1646;
 * protocol shape restated in ceno_recursion_v2/src/pcs/mod.rs:1111-1316,7720-8002).  PARITY UNPINNED:
 * Poseidon2-Goldilocks round constants and the RS/leaf layout live in EXT crates.
 * ---------------------------------------------------------------------------------------------- */
/* in-place radix-2 NTT of `n_cols` columns of length 2^log_n stored column-major (col stride 2^log_n):
 * natural-order input -> bit-reversed-order output (inverse: bit-reversed in -> natural out, scaled by 1/N) */
int ceno_hip_ntt_batch(ceno_hip_ctx* ctx, uint64_t* dev_cols, int log_n, int n_cols, int inverse, ceno_hip_stream s);
/* Reed–Solomon encode: each column of 2^log_n evaluations-as-coefficients is zero-extended by 2^log_blowup and transformed. */
int ceno_hip_rs_encode(ceno_hip_ctx* ctx, const uint64_t* dev_cols, int log_n, int n_cols, int log_blowup, uint64_t* dev_codewords, ceno_hip_stream s);
/* row-major (rows x width) -> column-major (matrix_transpose, ceno_zkvm/src/scheme/gpu/mod.rs:84,963-968) */
int ceno_hip_transpose(ceno_hip_ctx* ctx, const uint64_t* dev_row_major, size_t rows, size_t width, uint64_t* dev_col_major, ceno_hip_stream s);
/* Poseidon2 (width 8, rate 4, x^7) parameter table; NULL restores the built-in placeholder constants.
 * NON-INTEROPERABLE UNTIL PINNED: the built-in round constants are placeholders (the reference's live in an EXT crate), so
 * every root / proof-of-work / opening computed before a complete table (all three arrays) has been supplied is
 * self-consistent only.  First use of the placeholders prints a warning on stderr; with
 * CENO_HIP_REQUIRE_PINNED_POSEIDON2=1 in the environment the commit / open entry points fail with CENO_HIP_ERR_STATE instead.
 * tools/goldens/ dumps the table (and known-answer vectors) from the reference where cargo is available. */
int ceno_hip_poseidon2_set_constants(ceno_hip_ctx* ctx, const uint64_t* external_rc /* 8 rounds x 8 */, const uint64_t* internal_rc /* 22 */,
                                     const uint64_t* internal_diag /* 8 */);
int ceno_hip_poseidon2_is_pinned(const ceno_hip_ctx* ctx);   /* 1 once a complete table has been supplied */
int ceno_hip_poseidon2_permute(ceno_hip_ctx* ctx, uint64_t* dev_states /* n x 8 */, size_t n, ceno_hip_stream s);
/* leaves = sponge hash of each row of a column-major matrix (rows = 2^log_rows); tree = 2-to-1 compression */
int ceno_hip_merkle_commit(ceno_hip_ctx* ctx, const uint64_t* dev_col_major, int log_rows, int width, ceno_hip_stream s, ceno_hip_merkle** out);
/* ONE commitment over matrices of several power-of-two heights — what PCS::batch_commit returns for all the traces of a
 * commit_traces call (ceno_zkvm/src/scheme/cpu/mod.rs:559-584; several num_vars opened under one commitment:
 * ceno_recursion_v2/src/pcs/mod.rs:1123-1135,7547-7565).  The tree is p3's MerkleTreeMmcs (EXT p3-merkle-tree 0.4.3): matrices
 * sorted tallest first (stable); leaf i = sponge over row i of ALL tallest matrices concatenated; every next level has half
 * the nodes, node i = compress(left, right), and where matrices of exactly that height exist
 * node i = compress(compress(left, right), sponge(row i of those matrices)).  log_rows[m] = log2 of matrix m's height,
 * dev_col_major[m] its columns (stride 2^log_rows[m]); the matrices are BORROWED for as long as the tree is opened.
 * The tree has max(log_rows) levels below the root; ceno_hip_merkle_root / _open / _open_batch / _free apply. */
int ceno_hip_mmcs_commit(ceno_hip_ctx* ctx, const uint64_t* const* dev_col_major, const int* log_rows, const int* widths, int n_mats,
                         ceno_hip_stream s, ceno_hip_merkle** out);
/* The same tree over a GIVEN bottom layer of 2^log_leaves digests (4 words each, device memory) instead of hashed rows: every
 * matrix (n_mats may be 0) is shorter than that layer and joins at its level above.  This is the top of a commitment whose rows are
 * sharded over the GPUs of a node: the layer = the ranks' sub-tree roots, the matrices = the classes with fewer rows than ranks
 * (ceno_dist_commit_traces_mmcs). */
int ceno_hip_mmcs_commit_over(ceno_hip_ctx* ctx, const uint64_t* dev_leaf_digests, int log_leaves, const uint64_t* const* dev_col_major,
                              const int* log_rows, const int* widths, int n_mats, ceno_hip_stream s, ceno_hip_merkle** out);
/* MerkleTreeMmcs::open_batch for many indices at once: for query q, index i_q = dev_indices[q] >> shift (a row of the tallest
 * height), dev_out[q * out_stride_words ..] = [row (i_q >> (log_max - log_rows[m])) of every matrix m in the caller's order]
 * [authentication path: 4 x log_max words, bottom-up].  ceno_hip_mmcs_opening_words = sum of widths + 4 log_max. */
size_t ceno_hip_mmcs_opening_words(const ceno_hip_merkle* t);
int ceno_hip_mmcs_open_batch(ceno_hip_ctx* ctx, ceno_hip_merkle* t, const uint64_t* dev_indices, size_t n, int shift, uint64_t* dev_out,
                             size_t out_stride_words, ceno_hip_stream s);
int ceno_hip_merkle_root(ceno_hip_ctx* ctx, ceno_hip_merkle* t, uint64_t* root4, ceno_hip_stream s);
/* authentication path of leaf `index`: log_rows sibling digests (4 words each), bottom-up */
int ceno_hip_merkle_open(ceno_hip_ctx* ctx, ceno_hip_merkle* t, size_t index, uint64_t* path /* log_rows*4 */, ceno_hip_stream s);
int ceno_hip_merkle_free(ceno_hip_ctx* ctx, ceno_hip_merkle* t);

/* ------------------------------------------------------------------------------------------------
 * A COHORT of tower-layer sumchecks in one launch (round 6; ceno_amd/csrc/tower_cohort.hip): the layer sumcheck of
 * CpuTowerProver::create_proof (ceno_zkvm/src/scheme/cpu/mod.rs:417-494;  sum_x eq(x, rt) [ sum_i alpha_i a_i b_i + sum_k (an_k (p1 q2 + p2 q1) +
 * ad_k q1 q2) ], degree 3, LSB first) for MANY independent chains side by side — one workgroup per job, each waiting for its own host.
 * The reference proves its chips' towers one `tower.create_proof` call at a time (scheme/gpu/mod.rs:343-353) and overlaps them with
 * scheduler lanes; on this hardware four queues run at a time, so the chains are put into ONE launch instead.
 * A job: n <= ceno_hip_tower_cohort_max_vars() (13) variables; tables = 2 n_prod + 4 n_logup device pointers, each 2^n extension elements
 * ([a, b] per product tower, [p1, p2, q1, q2] per LogUp tower); rt: n ext; alpha_*: one ext per tower.  A layer of more than 2^13 entries
 * is split by its TOP index bits into smaller jobs forming a group (below) whose messages the device adds, scaled by eq over the high variables.
 * Protocol per job and round i = 0 .. n - 1: try_message(job, i) until it returns 1 (out6 = p(1), p(2), p(3)), send_challenge(job, i, r_i)
 * (to the mailbox's owner alone when jobs share one);
 * after the last challenge try_final(job) yields the 1 + 2 n_prod + 4 n_logup evaluations [eq, a, b, .., p1, p2, q1, q2, ..] at the point.
 * begin launches and returns; end waits for the launch and frees.  abort releases every waiting workgroup (then call end).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ceno_hip_cohort ceno_hip_cohort;
typedef struct ceno_hip_cohort_job {
    const uint64_t* const* tables;
    int n_prod, n_logup, n;
    const uint64_t* rt;
    const uint64_t *alpha_prod, *alpha_num, *alpha_den;
    int share_mailbox_of; /* 0: the job stands alone.  j + 1: it belongs to the GROUP led by job j (consecutive jobs of one size, j first, j names
                           * itself): the sub-cubes of one layer.  A group takes its challenges from ONE mailbox (send_challenge to j alone)
                           * and publishes ONE message per round — the sum of its jobs' messages, each times its `scale` — through
                           * try_message(j): the device adds them up, so that a round costs one write to the host however fine the cut.
                           * Final evaluations stay per job. */
    const uint64_t* scale; /* one ext (NULL: one): eq of the sub-cube's index at the layer's high coordinates */
    const uint64_t* claim; /* one ext, a job that stands alone or leads a group: the sum its (group's) sumcheck proves — the device reports two
                            * values per round (the eq factor is taken out of the tables and of the evaluation points), the third comes from
                            * the running claim.  CENO_HIP_ERR_UNSUPPORTED from set_job / begin when a coordinate of rt is 1 (the claim does not
                            * determine the round polynomial then): prove that layer another way */
} ceno_hip_cohort_job;
int ceno_hip_tower_cohort_max_vars(void);
/* workgroups (= jobs) the device holds at once; a launch of more would leave jobs undispatched behind jobs that wait for their host.
 * 0 on a device without host-writable device memory (large BAR): no cohorts there */
int ceno_hip_tower_cohort_capacity(ceno_hip_ctx* ctx);
int ceno_hip_tower_cohort_begin(ceno_hip_ctx* ctx, const ceno_hip_cohort_job* jobs, int n_jobs, ceno_hip_stream s, ceno_hip_cohort** out);
/* begin in three steps, for callers with several host threads (a launch of ~1000 jobs: the job records are most of begin's time):
 * open — the jobs' shapes and groups (allocations, layout); set_job — one job's record, callable from any thread, distinct jobs
 * concurrently (the shape must be the one given to open); launch.  end releases a cohort whether it was launched or not. */
typedef struct ceno_hip_cohort_shape {
    int n_prod, n_logup, n, share_mailbox_of;
} ceno_hip_cohort_shape;
int ceno_hip_tower_cohort_open(ceno_hip_ctx* ctx, const ceno_hip_cohort_shape* shapes, int n_jobs, ceno_hip_stream s, ceno_hip_cohort** out);
int ceno_hip_tower_cohort_set_job(ceno_hip_cohort* c, int job, const ceno_hip_cohort_job* record);
int ceno_hip_tower_cohort_launch(ceno_hip_ctx* ctx, ceno_hip_cohort* c);
int ceno_hip_tower_cohort_try_message(ceno_hip_cohort* c, int job, int round, uint64_t* out6);
int ceno_hip_tower_cohort_send_challenge(ceno_hip_cohort* c, int job, int round, const uint64_t* chal2);
/* diagnostics, valid once try_message(job, round) returned 1: out2 = the device's 100 MHz clock when the job saw the challenge that opened
 * this round (round 0: when it started) and when it sent the round's message */
int ceno_hip_tower_cohort_round_times(ceno_hip_cohort* c, int job, int round, uint64_t* out2);
int ceno_hip_tower_cohort_try_final(ceno_hip_cohort* c, int job, uint64_t* out_evals);
int ceno_hip_tower_cohort_abort(ceno_hip_cohort* c);
int ceno_hip_tower_cohort_end(ceno_hip_ctx* ctx, ceno_hip_cohort* c);

/* ------------------------------------------------------------------------------------------------
 * Basefold batch open  (OpeningProver::open -> PCS::batch_open, ceno_zkvm/src/scheme/hal.rs:284-294,
 * cpu/mod.rs:1418-1457; protocol restated in ceno_recursion_v2/src/pcs/mod.rs:1111-1316,7494-7781).
 * PARITY UNPINNED like the commit path.  The round loop lives in the host layer (ceno_prover_basefold_open).
 * ---------------------------------------------------------------------------------------------- */
/* acc[i] (+)= sum_c coeff_c * col_c[i]: `n_cols` base columns of length `len` (column-major, stride len), ext
 * coefficients from the host, ext accumulator of length `len` on the device.  Synchronises the stream. */
int ceno_hip_batch_columns(ceno_hip_ctx* ctx, const uint64_t* dev_cols, size_t len, int n_cols, const uint64_t* coeffs_ext,
                           uint64_t* dev_acc_ext, int accumulate, ceno_hip_stream s);
/* n_jobs such batchings of different lengths in ONE launch (per wave of accumulation) and one synchronisation: job j reads n_cols[j] columns of
 * length lens[j] at dev_cols[j] with the next n_cols[j] coefficients of coeffs_ext (jobs' coefficients back to back, 2 words each) into
 * dev_acc_ext[j].  The first job that names an accumulator writes it (accumulate[j] = 0), later ones add to it (accumulate[j] = 1) and run in a
 * launch of their own after it.  What the opening's batching phase is: the batched codeword per height class and F_m per matrix
 * (ceno_recursion_v2/src/pcs/mod.rs:1130-1180).  Synchronises the stream once. */
int ceno_hip_batch_columns_multi(ceno_hip_ctx* ctx, int n_jobs, const uint64_t* const* dev_cols, const size_t* lens, const int* n_cols,
                                 const uint64_t* coeffs_ext, uint64_t* const* dev_acc_ext, const int* accumulate, ceno_hip_stream s);
/* out[i] = sum over b < n_blocks of in[b * len + i] (mod p), extension elements: the modular all-reduce of per-rank partial batchings
 * after an all-gather (the multi-rank opening, ceno_dist_basefold_open).  Asynchronous on `s`. */
int ceno_hip_ext_sum_blocks(ceno_hip_ctx* ctx, const uint64_t* dev_in_ext, int n_blocks, size_t len, uint64_t* dev_out_ext, ceno_hip_stream s);
/* One commit-phase round over the running ext codeword of length 2^log_h (bit-reversed order, pairs adjacent):
 * tree over the 2^(log_h-1) pair leaves (returned), and out[j] = fold(cw[2j], cw[2j+1]; challenge) (+ addend[j]). */
int ceno_hip_basefold_fold_commit(ceno_hip_ctx* ctx, const uint64_t* dev_codeword_ext, int log_h, const uint64_t* challenge2,
                                  const uint64_t* dev_addend_ext /* may be NULL */, uint64_t* dev_out_ext, ceno_hip_stream s,
                                  ceno_hip_merkle** out_tree);
/* One-round-ahead form of the same round: the tree a round commits to depends only on the PREVIOUS challenge, so the
 * host folds (ceno_hip_basefold_fold: out[j] = fold(cw[2j], cw[2j+1]; challenge) (+ addend[j])) and at once commits the
 * result (ceno_hip_basefold_commit_codeword: tree over its pair leaves) on a second stream while the next round's
 * sumcheck message is computed. */
int ceno_hip_basefold_fold(ceno_hip_ctx* ctx, const uint64_t* dev_codeword_ext, int log_h, const uint64_t* challenge2,
                           const uint64_t* dev_addend_ext /* may be NULL */, uint64_t* dev_out_ext, ceno_hip_stream s);
int ceno_hip_basefold_commit_codeword(ceno_hip_ctx* ctx, const uint64_t* dev_codeword_ext, int log_h, ceno_hip_stream s,
                                      ceno_hip_merkle** out_tree);
/* dev_out[(q*n_cols + c)*elem_words + e] = dev_src[c*col_stride_words + i_q*elem_words + e], i_q = (idx[q] >> shift) ^ flip */
int ceno_hip_gather(ceno_hip_ctx* ctx, const uint64_t* dev_src, size_t col_stride_words, int n_cols, int elem_words,
                    const uint64_t* dev_indices, size_t n, int shift, int flip_low_bit, uint64_t* dev_out, ceno_hip_stream s);
/* authentication paths of leaves (idx[q] >> shift): dev_out[q][level][4], log_rows levels, bottom-up */
int ceno_hip_merkle_open_batch(ceno_hip_ctx* ctx, ceno_hip_merkle* t, const uint64_t* dev_indices, size_t n, int shift,
                               uint64_t* dev_out, ceno_hip_stream s);
/* the commit-round part of ALL queries in one launch: for round r = 0 .. n_rounds-1 (codeword r, tree r) and query q the sibling
 * value of codeword entry (idx_q >> r) and the authentication path of leaf (idx_q >> (r + 1)); dev_out = per round
 * [n_q x 2 words sibling][n_q x 4 * depth_r words path, bottom-up], rounds back to back (the layout of n_rounds pairs of
 * ceno_hip_gather + ceno_hip_merkle_open_batch).  Query phase of ceno_recursion_v2/src/pcs/mod.rs:7611-7690.  Synchronises. */
int ceno_hip_basefold_query_rounds(ceno_hip_ctx* ctx, const uint64_t* const* dev_codewords_ext, ceno_hip_merkle* const* trees, int n_rounds,
                                   const uint64_t* dev_indices, size_t n_q, uint64_t* dev_out, ceno_hip_stream s);
/* The commit phase's sumcheck of a batch opening over matrices of MIXED heights, every live matrix of a round in one launch: the claim is
 * sum_m sum_x Eq_m(x) * F_m(x) (degree 2; Eq_m = eq(., point_m), F_m = the matrix's columns batched with the batch coefficients — both extension
 * tables of 2^num_vars[m] entries in device memory, borrowed until _free and left untouched); with n = the most variables of any matrix, matrix m
 * joins in round n - num_vars[m] (suffix alignment, ceno_recursion_v2/src/pcs/mod.rs:1111-1316).  Replaces one ceno_hip_sumcheck handle per
 * height group (what PCS::batch_open's prover does per matrix through the EXT crate mpcs; ceno_zkvm/src/scheme/cpu/mod.rs:1418-1457).
 *   _round:  round r = 0 .. n-1 in order; challenge_prev2 = round r-1's challenge (NULL for round 0).  out_evals4 = (p(1), p(2)) of the LIVE
 *            matrices' part of the round polynomial — a matrix that joins in a later round j is a constant in this variable (its claimed sum
 *            times 2^(j - r - 1) at both points): the caller adds those.  Returns when the message is in host memory (the stream is not drained).
 *   _finish: after round n-1, with its challenge: out_finals[2m .. 2m+1] = F_m(r_{n - nv_m} .. r_{n-1}), the caller's matrix order.
 * Device memory on top of the inputs: 1.5 x the inputs' bytes (two fold buffers per table) + 40 KB. */
typedef struct ceno_hip_open_rounds ceno_hip_open_rounds;
int ceno_hip_open_rounds_begin(ceno_hip_ctx* ctx, int n_mats, const uint64_t* const* dev_eq_ext, const uint64_t* const* dev_f_ext, const int* num_vars,
                               ceno_hip_stream s, ceno_hip_open_rounds** out);
int ceno_hip_open_rounds_round(ceno_hip_ctx* ctx, ceno_hip_open_rounds* h, const uint64_t* challenge_prev2, uint64_t* out_evals4);
int ceno_hip_open_rounds_finish(ceno_hip_ctx* ctx, ceno_hip_open_rounds* h, const uint64_t* challenge_last2, uint64_t* out_finals);
int ceno_hip_open_rounds_done(const ceno_hip_open_rounds* h); /* rounds produced so far (n + 1 after _finish) */
void ceno_hip_open_rounds_free(ceno_hip_ctx* ctx, ceno_hip_open_rounds* h);
/* Proof-of-work search of p3's grinding challenger (GrindingChallenger::grind; the verifier side is check_witness,
 * ceno_recursion_v2/src/pcs/mod.rs:8125-8155): the least w for which a CLONE of the Poseidon2 duplex challenger (width 8, rate 4)
 * that observes w samples a base element whose low `bits` bits are zero.  state16 = the challenger state as exported by
 * ceno_transcript.export_state (include/ceno_prover.h): [sponge state 8][n pending inputs][pending inputs 4][n outputs left][0][0].
 * One permutation per candidate.  Synchronises. */
int ceno_hip_pow_grind_duplex(ceno_hip_ctx* ctx, const uint64_t* state16, int bits, uint64_t* out_witness, ceno_hip_stream s);

/* ------------------------------------------------------------------------------------------------
 * on-device witness generation: the R-type chips ADD / SUB, then ADDI and AND / OR / XOR
 *   reference: hal.witgen.witgen_add / witgen_sub as called from ceno_zkvm/src/instructions/gpu/dispatch.rs:509-571
 *   (column maps: instructions/gpu/chips/add.rs:15-47, chips/sub.rs:15-46; CPU assignment being reproduced:
 *   instructions/riscv/arith.rs:101-142, r_insn.rs:67-86, insn_base.rs:61-77,112-145,223-257,337-400,
 *   gkr_iop/src/gadgets/is_lt.rs:243-287).
 * Input: the shard's StepRecord array as the emulator lays it out (#[repr(C)], 136 bytes per step,
 * ceno_emul/src/tracer.rs:33-60) already on the device, and the indices of the steps that belong to this chip.
 * Output: the witness matrix COLUMN-major, `num_cols` columns of `rows_padded` base-field words (rows >= n are zero =
 * InstancePaddingStrategy::Default), and the lookup multiplicities this chip contributes:
 *   dev_lk_dynamic[(1 << bits) + v]  (LookupTable::Dynamic, gkr_iop/src/utils/lk_multiplicity.rs:181-198; 2^19 counters: bits <= DYNAMIC_RANGE_MAX_BITS = 18, scheme/constants.rs:11)
 *   dev_lk_fetch[(pc - fetch_base_pc) / 4]  (LookupTable::Instruction, dispatch.rs:438-443)
 * Counters are ADDED to (atomics), so one pair of tables serves all chips of a shard; either pointer may be NULL.
 * With lookup tables the call synchronises the stream (it counts into per-XCD scratch copies and merges them).
 * The shard RAM records of the reference's kernels (F-3) are not produced here.
 * ---------------------------------------------------------------------------------------------- */
#define CENO_HIP_STEP_RECORD_BYTES 136
#define CENO_HIP_LK_DYNAMIC_SLOTS (1u << 19)
/* same fields, same order as ceno_gpu's AddColumnMap (chips/add.rs:29-46): every entry is a column id < num_cols */
typedef struct ceno_hip_add_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_limbs[2], rs2_limbs[2], rd_carries[2];
    uint32_t num_cols;
} ceno_hip_add_column_map;
/* SubColumnMap (chips/sub.rs:28-45): SUB proves rs1 = rs2 + rd, the carries are those of rs2 + rd */
typedef struct ceno_hip_sub_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs2_limbs[2], rd_limbs[2], carries[2];
    uint32_t num_cols;
} ceno_hip_sub_column_map;
int ceno_hip_witgen_add(ceno_hip_ctx* ctx, const ceno_hip_add_column_map* map, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                        uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                        uint32_t* dev_lk_fetch, ceno_hip_stream s);
int ceno_hip_witgen_sub(ceno_hip_ctx* ctx, const ceno_hip_sub_column_map* map, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                        uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                        uint32_t* dev_lk_fetch, ceno_hip_stream s);

/* I-type ADDI: hal.witgen.witgen_addi (dispatch.rs GpuWitgenKind::Addi; column map chips/addi.rs:12-43; CPU assignment
 * arith_imm/arith_imm_circuit_v2.rs:85-117 + i_insn.rs:66-82): rs1 + sign_extend(imm) = rd with the witnessed carries; `imm` = the low
 * 16 bits of the instruction's immediate (StepRecord.insn.imm), `imm_sign` = 1 when it is negative.  18 mapped columns. */
typedef struct ceno_hip_addi_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_limbs[2], imm, imm_sign, rd_carries[2];
    uint32_t num_cols;
} ceno_hip_addi_column_map;
int ceno_hip_witgen_addi(ceno_hip_ctx* ctx, const ceno_hip_addi_column_map* map, const void* dev_step_records, size_t num_records,
                         const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                         uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                         uint32_t* dev_lk_fetch, ceno_hip_stream s);
/* JAL: hal.witgen.witgen_jal (GpuWitgenKind::Jal; column map chips/jal.rs:10-32; CPU assignment riscv/jump/jal_v2.rs:99-127 + j_insn.rs:58-73):
 * state with next_pc, rd = pc + 4 as four bytes.  dev_lk_double_u8: the 2^16 counters of LookupTable::DoubleU8, key a << 8 | b
 * (lk_multiplicity.rs:200-203); dev_lk_xor: those of LookupTable::Xor, key a | b << 8 (the top byte of rd against 0xC0).  13 mapped columns.
 * AUIPC: hal.witgen.witgen_auipc (GpuWitgenKind::Auipc; chips/auipc.rs:10-45; riscv/auipc.rs:149-187): the I-instruction base, rd as four
 * bytes, bytes 1, 2 of pc, the three bytes of insn.imm as u32 >> 8, the top byte of pc against 0xC0.  21 mapped columns. */
typedef struct ceno_hip_jal_column_map {
    uint32_t pc, next_pc, ts;
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rd_bytes[4];
    uint32_t num_cols;
} ceno_hip_jal_column_map;
typedef struct ceno_hip_auipc_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rd_bytes[4], pc_limbs[2], imm_limbs[3];
    uint32_t num_cols;
} ceno_hip_auipc_column_map;
int ceno_hip_witgen_jal(ceno_hip_ctx* ctx, const ceno_hip_jal_column_map* map, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, uint32_t* dev_lk_double_u8,
                        uint32_t* dev_lk_xor, ceno_hip_stream s);
int ceno_hip_witgen_auipc(ceno_hip_ctx* ctx, const ceno_hip_auipc_column_map* map, const void* dev_step_records, size_t num_records,
                          const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                          uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch,
                          uint32_t* dev_lk_double_u8, uint32_t* dev_lk_xor, ceno_hip_stream s);
/* SLT / SLTU: hal.witgen.witgen_slt (GpuWitgenKind::Slt(is_signed): 1 = SLT, 0 = SLTU; column map chips/slt.rs:12-56; CPU assignment
 * riscv/slt/slt_circuit_v2.rs:86-119 + gadgets/signed_limbs.rs:150-236).  a_msb_f / b_msb_f are FIELD elements: a top limb that is
 * negative in a signed comparison is stored as p - (2^16 - limb) (Goldilocks p).  26 mapped columns, the comparison's ten first. */
typedef struct ceno_hip_slt_column_map {
    uint32_t rs1_limbs[2], rs2_limbs[2], cmp_lt, a_msb_f, b_msb_f, diff_marker[2], diff_val;
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t num_cols;
} ceno_hip_slt_column_map;
int ceno_hip_witgen_slt(ceno_hip_ctx* ctx, const ceno_hip_slt_column_map* map, int is_signed, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);
/* SLTI / SLTIU: hal.witgen.witgen_slti (GpuWitgenKind::Slti(is_signed); chips/slti.rs:10-52; riscv/slti/slti_circuit_v2.rs:104-140): rs1 against
 * the sign-extended immediate, same comparison gadget.  22 mapped columns. */
typedef struct ceno_hip_slti_column_map {
    uint32_t rs1_limbs[2], imm, imm_sign, cmp_lt, a_msb_f, b_msb_f, diff_marker[2], diff_val;
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t num_cols;
} ceno_hip_slti_column_map;
int ceno_hip_witgen_slti(ceno_hip_ctx* ctx, const ceno_hip_slti_column_map* map, int is_signed, const void* dev_step_records, size_t num_records,
                         const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                         uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);
/* Branches: hal.witgen.witgen_branch_cmp (GpuWitgenKind::BranchCmp(is_signed): BLT / BGE = 1, BLTU / BGEU = 0; chips/branch_cmp.rs:12-55) and
 * witgen_branch_eq (GpuWitgenKind::BranchEq(is_beq): BEQ = 1, BNE = 0; chips/branch_eq.rs:12-44); CPU assignment
 * riscv/branch/branch_circuit_v2.rs:143-209 + b_insn.rs:92-116.  `imm` is the branch offset as a field element (negative: p - |imm|);
 * diff_inv_marker holds the field inverse of (rs1_limb - rs2_limb) at the first differing limb.  22 / 19 mapped columns. */
typedef struct ceno_hip_branch_cmp_column_map {
    uint32_t rs1_limbs[2], rs2_limbs[2], cmp_lt, a_msb_f, b_msb_f, diff_marker[2], diff_val;
    uint32_t pc, next_pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t imm;
    uint32_t num_cols;
} ceno_hip_branch_cmp_column_map;
typedef struct ceno_hip_branch_eq_column_map {
    uint32_t rs1_limbs[2], rs2_limbs[2], branch_taken, diff_inv_marker[2];
    uint32_t pc, next_pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t imm;
    uint32_t num_cols;
} ceno_hip_branch_eq_column_map;
int ceno_hip_witgen_branch_cmp(ceno_hip_ctx* ctx, const ceno_hip_branch_cmp_column_map* map, int is_signed, const void* dev_step_records, size_t num_records,
                               const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                               uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);
int ceno_hip_witgen_branch_eq(ceno_hip_ctx* ctx, const ceno_hip_branch_eq_column_map* map, int is_beq, const void* dev_step_records, size_t num_records,
                              const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                              uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);
/* Shifts: hal.witgen.witgen_shift_r / witgen_shift_i (GpuWitgenKind::ShiftR(kind) / ShiftI(kind), kind 0 = SLL / SLLI, 1 = SRL / SRLI, 2 = SRA / SRAI;
 * chips/shift_r.rs:12-60, chips/shift_i.rs:9-54; CPU assignment riscv/shift/shift_circuit_v2.rs:359-396,485-521 with the ShiftBase gadget :242-293).
 * The result is StepRecord.rd.value.after; its byte pairs count into the double-byte table, SRA's sign lookup into the XOR table.  47 / 40 mapped columns. */
typedef struct ceno_hip_shift_r_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_bytes[4], rs2_bytes[4], rd_bytes[4];
    uint32_t bit_shift_marker[8], limb_shift_marker[4], bit_multiplier_left, bit_multiplier_right, b_sign, bit_shift_carry[4];
    uint32_t num_cols;
} ceno_hip_shift_r_column_map;
typedef struct ceno_hip_shift_i_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_bytes[4], rd_bytes[4], imm;
    uint32_t bit_shift_marker[8], limb_shift_marker[4], bit_multiplier_left, bit_multiplier_right, b_sign, bit_shift_carry[4];
    uint32_t num_cols;
} ceno_hip_shift_i_column_map;
int ceno_hip_witgen_shift_r(ceno_hip_ctx* ctx, const ceno_hip_shift_r_column_map* map, int kind, const void* dev_step_records, size_t num_records,
                            const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                            uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch,
                            uint32_t* dev_lk_double_u8, uint32_t* dev_lk_xor, ceno_hip_stream s);
int ceno_hip_witgen_shift_i(ceno_hip_ctx* ctx, const ceno_hip_shift_i_column_map* map, int kind, const void* dev_step_records, size_t num_records,
                            const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                            uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch,
                            uint32_t* dev_lk_double_u8, uint32_t* dev_lk_xor, ceno_hip_stream s);

/* JALR: hal.witgen.witgen_jalr (GpuWitgenKind::Jalr; chips/jalr.rs:12-50; CPU assignment riscv/jump/jalr_v2.rs:146-190 + i_insn.rs:66-82; the
 * jump target's MemAddr insn_base.rs:880-905 with max_bits = PC_BITS and both low bits witnessed).  22 mapped columns. */
typedef struct ceno_hip_jalr_column_map {
    uint32_t pc, next_pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_limbs[2], imm, imm_sign, jump_pc_addr[2], jump_pc_addr_bit[2], rd_high;
    uint32_t num_cols;
} ceno_hip_jalr_column_map;
int ceno_hip_witgen_jalr(ceno_hip_ctx* ctx, const ceno_hip_jalr_column_map* map, const void* dev_step_records, size_t num_records,
                         const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                         uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);

/* LW / SW: hal.witgen.witgen_lw / witgen_sw (GpuWitgenKind::Lw / Sw; chips/lw.rs:12-54, chips/sw.rs:12-50; CPU assignment
 * riscv/memory/load_v2.rs:197-255 + im_insn.rs:71-90 and store_v2.rs:138-177 + s_insn.rs:77-96; memory access insn_base.rs:517-545,650-680,
 * address checks :880-905).  The memory operand is StepRecord.memory_op; the shard RAM records of the access are not produced here.
 * 23 mapped columns each. */
typedef struct ceno_hip_lw_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], imm, imm_sign, mem_addr_limbs[2], mem_read_limbs[2];
    uint32_t num_cols;
} ceno_hip_lw_column_map;
typedef struct ceno_hip_sw_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], rs2_limbs[2], imm, imm_sign, prev_mem_val[2], mem_addr[2];
    uint32_t num_cols;
} ceno_hip_sw_column_map;
/* LH / LHU / LB / LBU: hal.witgen.witgen_load_sub (GpuWitgenKind::LoadSub { load_width: 16 | 8, is_signed }; chips/load_sub.rs:12-92; CPU assignment
 * riscv/memory/load_v2.rs:197-255, SignedExtendConfig gadgets/signed_ext.rs:92-103).  The reference's map holds Option columns: a halfword load has no
 * addr_bit_0 / target_byte / dummy_byte, an unsigned load no msb — such fields carry CENO_HIP_NO_COLUMN here and are checked.  25 .. 29 mapped columns. */
#define CENO_HIP_NO_COLUMN 0xffffffffu
typedef struct ceno_hip_load_sub_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], imm, imm_sign, mem_addr[2], mem_read[2];
    uint32_t addr_bit_1, target_limb, addr_bit_0, target_byte, dummy_byte, msb;
    uint32_t num_cols;
} ceno_hip_load_sub_column_map;
int ceno_hip_witgen_load_sub(ceno_hip_ctx* ctx, const ceno_hip_load_sub_column_map* map, int load_width, int is_signed, const void* dev_step_records,
                             size_t num_records, const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                             uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch,
                             ceno_hip_stream s);

/* MUL / MULH / MULHU / MULHSU: hal.witgen.witgen_mul (GpuWitgenKind::Mul(mul_kind), mul_kind 0 = MUL, 1 = MULH, 2 = MULHU, 3 = MULHSU; chips/mul.rs:11-57;
 * CPU assignment riscv/mulh/mulh_circuit_v2.rs:234-333 with run_mulh :427-487).  rd_high[2], rs1_ext, rs2_ext are Option columns of the reference's map:
 * CENO_HIP_NO_COLUMN for MUL.  22 / 26 mapped columns. */
typedef struct ceno_hip_mul_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_limbs[2], rs2_limbs[2], rd_low[2], rd_high[2], rs1_ext, rs2_ext;
    uint32_t num_cols;
} ceno_hip_mul_column_map;
int ceno_hip_witgen_mul(ceno_hip_ctx* ctx, const ceno_hip_mul_column_map* map, int mul_kind, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);

/* DIV / DIVU / REM / REMU: hal.witgen.witgen_div (GpuWitgenKind::Div(div_kind), div_kind 0 = DIV, 1 = DIVU, 2 = REM, 3 = REMU; chips/div.rs:11-85; CPU
 * assignment riscv/div/div_circuit_v2.rs:391-536 with run_divrem :628-697, run_mul_carries :711-752, run_sltu_diff_idx :699-709).  divisor_sum_inv,
 * remainder_sum_inv and remainder_inv[2] are FIELD elements (inverses; 0 for a zero sum), written as canonical Goldilocks values.  39 mapped columns. */
typedef struct ceno_hip_div_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t dividend[2], divisor[2], quotient[2], remainder[2];
    uint32_t dividend_sign, divisor_sign, quotient_sign, remainder_zero, divisor_zero;
    uint32_t divisor_sum_inv, remainder_sum_inv, remainder_inv[2], sign_xor, remainder_prime[2], lt_marker[2], lt_diff;
    uint32_t num_cols;
} ceno_hip_div_column_map;
int ceno_hip_witgen_div(ceno_hip_ctx* ctx, const ceno_hip_div_column_map* map, int div_kind, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);

/* SH / SB: hal.witgen.witgen_sh / witgen_sb (GpuWitgenKind::Sh / Sb; chips/sh.rs:12-59, chips/sb.rs:10-82; StoreConfig<E, 1> / <E, 0>,
 * store_v2.rs:100-177, MemWordUtil riscv/memory/gadget.rs:134-185): SW's columns, the free address bits, and for SB the byte columns of the
 * addressed limb.  24 / 29 mapped columns. */
typedef struct ceno_hip_sh_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], rs2_limbs[2], imm, imm_sign, prev_mem_val[2], mem_addr[2];
    uint32_t mem_addr_bit_1;
    uint32_t num_cols;
} ceno_hip_sh_column_map;
typedef struct ceno_hip_sb_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], rs2_limbs[2], imm, imm_sign, prev_mem_val[2], mem_addr[2];
    uint32_t mem_addr_bit_0, mem_addr_bit_1, prev_limb_bytes[2], rs2_limb_byte, expected_limb;
    uint32_t num_cols;
} ceno_hip_sb_column_map;
int ceno_hip_witgen_sh(ceno_hip_ctx* ctx, const ceno_hip_sh_column_map* map, const void* dev_step_records, size_t num_records,
                       const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                       uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);
int ceno_hip_witgen_sb(ceno_hip_ctx* ctx, const ceno_hip_sb_column_map* map, const void* dev_step_records, size_t num_records,
                       const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                       uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);
int ceno_hip_witgen_lw(ceno_hip_ctx* ctx, const ceno_hip_lw_column_map* map, const void* dev_step_records, size_t num_records,
                       const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                       uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);
int ceno_hip_witgen_sw(ceno_hip_ctx* ctx, const ceno_hip_sw_column_map* map, const void* dev_step_records, size_t num_records,
                       const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                       uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);
/* LUI: hal.witgen.witgen_lui (GpuWitgenKind::Lui; column map chips/lui.rs:10-42; CPU assignment riscv/lui.rs:100-120): the I-instruction
 * base, bytes 1..3 of rd (each counted as a byte of the dynamic table), imm = insn.imm as u32 >> 12.  16 mapped columns. */
typedef struct ceno_hip_lui_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rd_bytes[3], imm;
    uint32_t num_cols;
} ceno_hip_lui_column_map;
int ceno_hip_witgen_lui(ceno_hip_ctx* ctx, const ceno_hip_lui_column_map* map, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s);
/* I-type logic chips ANDI / ORI / XORI: hal.witgen.witgen_logic_i (dispatch.rs:612-650; column map chips/logic_i.rs:10-42; CPU assignment
 * logic_imm/logic_imm_circuit_v2.rs:105-130,195-224).  logic_kind = 0 ANDI, 1 ORI, 2 XORI (GpuWitgenKind::LogicI); dev_lk_logic as for
 * the R-type chips, key rs1_byte | imm_byte << 8 with the immediate's high half = its sign spread over 16 bits.  24 mapped columns. */
typedef struct ceno_hip_logic_i_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_bytes[4], rd_bytes[4], imm_lo_bytes[2], imm_hi_bytes[2];
    uint32_t num_cols;
} ceno_hip_logic_i_column_map;
int ceno_hip_witgen_logic_i(ceno_hip_ctx* ctx, const ceno_hip_logic_i_column_map* map, int logic_kind, const void* dev_step_records,
                            size_t num_records, const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                            uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                            uint32_t* dev_lk_fetch, uint32_t* dev_lk_logic, ceno_hip_stream s);
/* R-type logic chips AND / OR / XOR: hal.witgen.witgen_logic_r (dispatch.rs:574-611; column map chips/logic_r.rs:12-43; CPU
 * assignment logic_circuit.rs:66-160: the R-instruction base as above, the three registers as four bytes each).  logic_kind =
 * 0 AND, 1 OR, 2 XOR as in GpuWitgenKind::LogicR.  dev_lk_logic (may be NULL): the 2^16 counters of THAT operation's table
 * (LookupTable::And / Or / Xor), key rs1_byte | rs2_byte << 8 (OpsTable::pack, gkr_iop/src/tables/mod.rs:29-31), four per instance. */
typedef struct ceno_hip_logic_r_column_map {
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_bytes[4], rs2_bytes[4], rd_bytes[4];
    uint32_t num_cols;
} ceno_hip_logic_r_column_map;
int ceno_hip_witgen_logic_r(ceno_hip_ctx* ctx, const ceno_hip_logic_r_column_map* map, int logic_kind, const void* dev_step_records,
                            size_t num_records, const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                            uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                            uint32_t* dev_lk_fetch, uint32_t* dev_lk_logic, ceno_hip_stream s);

/* A SHARD's witness generation as one session: the ~45 opcode chips of a shard count into the same few lookup tables (the reference
 * accumulates them in one LkMultiplicity per shard, gkr_iop/src/utils/lk_multiplicity.rs:181-198, and its GPU path fetches the device
 * counters back chip by chip, instructions/gpu/dispatch.rs:234-236 `gpu_lk_d2h`).  Between begin and end the ceno_hip_witgen_* calls on
 * `s` that name a registered table neither clear, merge nor wait: the per-XCD copies of the registered tables are cleared once at
 * begin, every chip's kernel is queued behind the previous one, and `end` adds the totals to the registered tables with one merge per
 * table and synchronises.  dev_tables[t]: slots[t] counters, e.g. the dynamic-range table (CENO_HIP_LK_DYNAMIC_SLOTS), the fetch table
 * (fetch_num_slots), the double-u8 and the AND / OR / XOR tables (2^16 each).  A table used inside the session must be registered.
 * One open session per context.  Chips inside a session may run on ANY stream of the context: a stream's first chip waits for the session's set-up, and
 * ceno_hip_witgen_session_end (on the session's stream) waits for every stream the chips ran on before it merges. */
int ceno_hip_witgen_session_begin(ceno_hip_ctx* ctx, uint32_t* const* dev_tables, const size_t* slots, int n_tables, ceno_hip_stream s);
int ceno_hip_witgen_session_end(ceno_hip_ctx* ctx, ceno_hip_stream s);
/* The `mlt` witness column of a table circuit straight from the device counters (no trip through the host): dev_column[i] = dev_counters[i]
 * for i < n, zero up to rows_padded (TableConfig::assign_instances: ceno_zkvm/src/tables/ops/ops_impl.rs:85-100,
 * tables/range/range_impl.rs:60-96 set `mlt` from the shard's multiplicity map; InstancePaddingStrategy::Default beyond the table).
 * dev_column is typically ceno_pcs_data_trace_ptr of the table's one-column witness matrix (include/ceno_prover.h). */
int ceno_hip_lk_to_mlt_column(ceno_hip_ctx* ctx, const uint32_t* dev_counters, size_t n, uint64_t* dev_column, size_t rows_padded,
                              ceno_hip_stream s);

/* ------------------------------------------------------------------------------------------------
 * diagnostics used by bench.py (HIP-event timing of the dominant kernel on the launch stream)
 * ---------------------------------------------------------------------------------------------- */
/* accumulated device time (ms) and launch count of the fused sumcheck round kernel since the last reset */
int ceno_hip_prof_reset(ceno_hip_ctx* ctx);
/* on = 1: every round is launched when it is asked for (no pipelining), events bracket the bare kernels.  on = 2: pipelined sumchecks STAY
 * pipelined — what a timed region runs — and the events bracket their large dense rounds as queued (a queued round ends when its finishing
 * workgroup has the next challenge, so the sum includes those waits and can never exceed the wall time of the same region). */
int ceno_hip_prof_enable(ceno_hip_ctx* ctx, int on);
int ceno_hip_prof_get(ceno_hip_ctx* ctx, double* kernel_ms, uint64_t* launches, double* algorithmic_bytes);

/* ------------------------------------------------------------------------------------------------
 * self-test hook of the device field arithmetic (test infrastructure; no reference counterpart): runs the reduction
 * (which = 0: n records of five 32-bit limbs w0..w3, c -> canon(reduce128), reduce_limbs), the base-field operations
 * (which = 1: n pairs of arbitrary 64-bit words -> mul, mul_nc, add, sub, add_nc, mul_add) or the extension multiply and its
 * unreduced accumulator (which = 2: n pairs of ext -> a*b, a*b + b*a + a*a) on host-supplied inputs.
 * ------------------------------------------------------------------------------------------------ */
int ceno_hip_selftest_field(ceno_hip_ctx* ctx, int which, const void* host_in, size_t n, uint64_t* host_out);

#ifdef __cplusplus
}
#endif
#endif
