// Per-chip proof flow of the zkVM prover, written against the device C ABI.
//
//   ceno_prover_build_tower_witness   CpuProver::build_tower_witness        ceno_zkvm/src/scheme/cpu/mod.rs:608-757
//                                     (GPU arm: build_tower_witness_gpu      ceno_zkvm/src/scheme/gpu/mod.rs:2136-2407)
//   ceno_prover_create_chip_proof     ZKVMProver::create_chip_proof          ceno_zkvm/src/scheme/prover.rs:717-833
//                                     = build_main_witness (tower stage)     ceno_zkvm/src/scheme/utils.rs:667-771
//                                       -> prove_tower_relation              ceno_zkvm/src/scheme/cpu/mod.rs:765-797
//                                       -> prove_rotation (keccak-style)     gkr_iop/src/gkr/layer/cpu/mod.rs:249-389
//                                     harness shape: ceno_zkvm/benches/riscv_add.rs:86-141
// The ECC quark step (prover.rs:766, shard-RAM chips only) is out of scope: BabyBear-only in the reference.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/ceno_prover.h"

int prover_set_error(int code, const char* msg);
#include "chip_run.hpp"
int prover_tower_host_layers();  // prover.cpp: tower layers the host proves (CENO_TOWER_HOST_LAYERS)  // prover.cpp

namespace {

int ceil_log2(size_t x) {
    int l = 0;
    while (((size_t)1 << l) < x) l++;
    return l;
}
size_t next_pow2(size_t x) { return (size_t)1 << ceil_log2(x ? x : 1); }

int fail_ctx(ceno_hip_ctx* ctx, int rc) { return prover_set_error(rc, ceno_hip_last_error(ctx)); }

}  // namespace

extern "C" {

void ceno_tower_witness_free(ceno_hip_ctx* ctx, ceno_tower_witness* w) {
    if (!w) return;
    for (int i = 0; i < 2; i++) {
        if (w->prod[i]) ceno_hip_tower_free(ctx, w->prod[i]);
        w->prod[i] = nullptr;
    }
    if (w->logup[0]) ceno_hip_tower_free(ctx, w->logup[0]);
    w->logup[0] = nullptr;
    w->n_prod = w->n_logup = 0;
}

int ceno_prover_build_tower_witness(ceno_hip_ctx* ctx, ceno_hip_mle* const* records, int num_reads, int num_writes, int num_lk_tables, int num_lk,
                                    int log2_num_instances, int rotation_vars, const uint64_t* challenges4, ceno_hip_stream s,
                                    ceno_tower_witness* out) {
    if (!ctx || !out || !challenges4 || num_reads < 0 || num_writes < 0 || num_lk_tables < 0 || num_lk < 0)
        return prover_set_error(CENO_HIP_ERR_INVALID, "build_tower_witness: bad arguments");
    memset(out, 0, sizeof(*out));
    // record slicing (cpu/mod.rs:626-644): [reads | writes | lk numerators (table circuits only) | lk denominators]
    const int n_lk_num = num_lk_tables;
    const int n_lk_den = num_lk_tables > 0 ? num_lk_tables : num_lk;
    const int total = num_reads + num_writes + n_lk_num + n_lk_den;
    if (total > 0 && !records) return prover_set_error(CENO_HIP_ERR_INVALID, "build_tower_witness: records is NULL");
    const int active_row_vars = log2_num_instances + rotation_vars;          // cpu/mod.rs:645
    const size_t active_rows = (size_t)1 << active_row_vars;                 // cpu/mod.rs:646
    for (int i = 0; i < total; i++)
        if (!records[i] || ceno_hip_mle_num_vars(records[i]) != active_row_vars)
            return prover_set_error(CENO_HIP_ERR_INVALID, "build_tower_witness: every record needs log2_num_instances + rotation_vars variables");
    auto group_num_vars = [&](int num_ops) { return active_row_vars + ceil_log2(next_pow2((size_t)num_ops)); };  // cpu/mod.rs:647-648
    ceno_hip_mle* const* r_set = records;
    ceno_hip_mle* const* w_set = records + num_reads;
    ceno_hip_mle* const* lk_n = records + num_reads + num_writes;
    ceno_hip_mle* const* lk_d = lk_n + n_lk_num;
    const uint64_t one[2] = {1, 0};
    const uint64_t* alpha = challenges4;  // challenges[0]: default of the lookup limbs (cpu/mod.rs:658-661)
    int rc = 0;
    // interleaving_mles_to_mles(.., active_rows, NUM_FANIN, default) + infer_tower_product_witness / _logup_witness
    // (cpu/mod.rs:652-677): one device call per tower
    if (num_reads > 0) {
        rc = ceno_hip_tower_build_prod(ctx, r_set, num_reads, active_rows, one, s, &out->prod[out->n_prod]);
        if (!rc && ceno_hip_tower_num_vars(out->prod[out->n_prod]) != group_num_vars(num_reads)) rc = CENO_HIP_ERR_STATE;
        if (!rc) { out->has_r = 1; out->n_prod++; }
    }
    if (!rc && num_writes > 0) {
        rc = ceno_hip_tower_build_prod(ctx, w_set, num_writes, active_rows, one, s, &out->prod[out->n_prod]);
        if (!rc && ceno_hip_tower_num_vars(out->prod[out->n_prod]) != group_num_vars(num_writes)) rc = CENO_HIP_ERR_STATE;
        if (!rc) { out->has_w = 1; out->n_prod++; }
    }
    if (!rc && n_lk_den > 0) {
        rc = ceno_hip_tower_build_logup(ctx, n_lk_num > 0 ? lk_n : nullptr, lk_d, n_lk_den, active_rows, alpha, s, &out->logup[0]);
        if (!rc && ceno_hip_tower_num_vars(out->logup[0]) != group_num_vars(n_lk_den)) rc = CENO_HIP_ERR_STATE;
        if (!rc) { out->has_lk = 1; out->n_logup = 1; }
    }
    if (!rc) {
        // the three builds are queued back to back; ONE copy + wait brings the top layers of all towers to the host — the out-evaluations
        // (layer 0) now, the layers the tower prover proves on the host later (ceno_hip_tower_prefetch_tops)
        ceno_hip_tower* all[3];
        int n_all = 0;
        for (int i = 0; i < out->n_prod; i++) all[n_all++] = out->prod[i];
        if (out->n_logup) all[n_all++] = out->logup[0];
        rc = ceno_hip_tower_prefetch_tops(ctx, all, n_all, prover_tower_host_layers() + 1, s);
        int k = 0;
        if (!rc && out->has_r) rc = ceno_hip_tower_out_evals(ctx, out->prod[k++], out->r_out_evals, s);
        if (!rc && out->has_w) rc = ceno_hip_tower_out_evals(ctx, out->prod[k++], out->w_out_evals, s);
        if (!rc && out->has_lk) rc = ceno_hip_tower_out_evals(ctx, out->logup[0], out->lk_out_evals, s);
    }
    if (rc) {
        std::string msg = rc == CENO_HIP_ERR_STATE ? "build_tower_witness: tower height differs from group_num_vars" : ceno_hip_last_error(ctx);
        ceno_tower_witness_free(ctx, out);
        return prover_set_error(rc, msg.c_str());
    }
    return 0;
}

void ceno_chip_proof_free(ceno_chip_proof* p) {
    if (!p) return;
    free(p->tower.msgs);
    free(p->tower.prod_evals);
    free(p->tower.logup_evals);
    free(p->tower.point);
    free(p->rt_main);
    free(p->rotation_msgs);
    free(p->rotation_evals);
    free(p->rotation_points);
    memset(p, 0, sizeof(*p));
}

int ceno_prover_create_chip_proof(ceno_hip_ctx* ctx, const ceno_chip_task* task, const uint64_t* challenges4, ceno_transcript* tr,
                                  ceno_hip_stream s, ceno_chip_proof* out) {
    ChipProofRun run;
    if (int rc = chip_run_begin(run, ctx, task, challenges4, tr, s, out)) return rc;
    return chip_run_finish(run, s);
}

}  // extern "C"

// ---- ZKVMProver::create_chip_proof in two halves (chip_run.hpp) ----------------------------------------------------------------------------
int chip_run_records_plan(ChipProofRun& run, ceno_hip_ctx* ctx, const ceno_chip_task* task, const uint64_t* challenges4, ceno_transcript* tr,
                          ceno_chip_proof* out, ceno_hip_wit_plan* plan) {
    if (!ctx || !task || !challenges4 || !tr || !out) return prover_set_error(CENO_HIP_ERR_INVALID, "create_chip_proof: NULL argument");
    memset(out, 0, sizeof(*out));
    run.ctx = ctx;
    run.task = task;
    run.tr = tr;
    run.out = out;
    run.challenges4 = challenges4;
    run.live = false;
    const int n_mles = task->n_witin + task->n_fixed + task->n_structural;
    const int num_var_with_rotation = task->log2_num_instances + task->rotation_vars;   // prover.rs:728-729
    run.num_var_with_rotation = num_var_with_rotation;
    const int n_lk_num = task->num_lk_tables, n_lk_den = task->num_lk_tables > 0 ? task->num_lk_tables : task->num_lk;
    const int n_records = task->num_reads + task->num_writes + n_lk_num + n_lk_den;
    if (n_records < 1) return prover_set_error(CENO_HIP_ERR_INVALID, "create_chip_proof: a circuit needs at least one read / write / lookup");  // utils.rs:701-710
    if (n_mles < 1 || !task->mles || !task->record_out_term_offsets || !task->record_term_offsets)
        return prover_set_error(CENO_HIP_ERR_INVALID, "create_chip_proof: bad task");
    for (int j = 0; j < n_mles; j++)
        if (task->mles[j] && ceno_hip_mle_num_vars(task->mles[j]) != num_var_with_rotation)                      // utils.rs:713-723
            return prover_set_error(CENO_HIP_ERR_INVALID, "create_chip_proof: witness sizes differ from log2_num_instances + rotation_vars");
    out->num_instances = task->num_instances;
    // ---- build_main_witness, tower stage (prover.rs:735-745): only the tower-facing records are materialised ----
    run.records.assign((size_t)n_records, nullptr);
    // structural witnesses may be absent at this stage ("they are `eq`, and will be filled later", utils.rs:690-695): the
    // record expressions never read them, so the inference runs on the tables that exist
    std::vector<ceno_hip_mle*>& present = run.present;
    std::vector<uint32_t>& ridx = run.ridx;
    present.clear();
    ridx.clear();
    std::vector<uint32_t> remap(n_mles, UINT32_MAX);
    for (int j = 0; j < n_mles; j++)
        if (task->mles[j]) {
            remap[j] = (uint32_t)present.size();
            present.push_back(task->mles[j]);
        }
    const uint32_t n_factors = task->record_term_offsets[task->n_record_terms];
    ridx.reserve(n_factors);
    for (uint32_t k = 0; k < n_factors; k++) {
        const uint32_t j = task->record_term_mle_idx[k];
        if ((int)j >= n_mles || remap[j] == UINT32_MAX) return prover_set_error(CENO_HIP_ERR_INVALID, "create_chip_proof: a record expression reads an absent table");
        ridx.push_back(remap[j]);
    }
    *plan = ceno_hip_wit_plan{present.data(), (int)present.size(), task->record_coeffs, task->record_term_offsets, ridx.data(), task->n_record_terms,
                              task->record_out_term_offsets, n_records, num_var_with_rotation, run.records.data()};
    return 0;
}

int chip_run_records(ChipProofRun& run, ceno_hip_ctx* ctx, const ceno_chip_task* task, const uint64_t* challenges4, ceno_transcript* tr,
                     ceno_hip_stream s, ceno_chip_proof* out) {
    ceno_hip_wit_plan p{};
    if (int rc = chip_run_records_plan(run, ctx, task, challenges4, tr, out, &p)) return rc;
    const int rc = ceno_hip_wit_infer(ctx, p.mles, p.num_mles, p.term_coeffs, p.term_offsets, p.term_mle_idx, p.num_terms, p.out_term_offsets, p.num_outs, p.num_vars,
                                      s, p.outs);
    if (rc) return fail_ctx(ctx, rc);
    return 0;
}

void chip_run_free_records(ChipProofRun& run) {  // prover.rs:756 drop(records): the towers own their interleaved copies
    for (auto*& m : run.records) {
        if (m) ceno_hip_mle_free(run.ctx, m);
        m = nullptr;
    }
    run.records.clear();
}

// the record slicing of build_tower_witness (cpu/mod.rs:626-677) as tower specs: [reads], [writes], [lookups]
int chip_run_tower_specs(ChipProofRun& run, ceno_hip_tower_spec* specs3) {
    const ceno_chip_task* t = run.task;
    const int n_lk_num = t->num_lk_tables, n_lk_den = t->num_lk_tables > 0 ? t->num_lk_tables : t->num_lk;
    ceno_hip_mle* const* rec = run.records.data();
    const size_t active_rows = (size_t)1 << (t->log2_num_instances + t->rotation_vars);
    int n = 0;
    auto put = [&](ceno_hip_mle* const* r, ceno_hip_mle* const* num, int k, int logup, uint64_t d0, uint64_t d1) {
        specs3[n++] = ceno_hip_tower_spec{r, num, k, logup, active_rows, {d0, d1}};
    };
    if (t->num_reads > 0) put(rec, nullptr, t->num_reads, 0, 1, 0);
    if (t->num_writes > 0) put(rec + t->num_reads, nullptr, t->num_writes, 0, 1, 0);
    if (n_lk_den > 0) {
        ceno_hip_mle* const* lk_n = rec + t->num_reads + t->num_writes;
        put(lk_n + n_lk_num, n_lk_num > 0 ? lk_n : nullptr, n_lk_den, 1, run.challenges4[0], run.challenges4[1]);  // challenges[0]: cpu/mod.rs:658-661
    }
    return n;
}

// the same towers named by their records' places in the chip's record plan (ceno_hip_tower_build_many_virtual: no record tables)
int chip_run_virtual_tower_specs(ChipProofRun& run, int plan_index, ceno_hip_virtual_tower_spec* specs3) {
    const ceno_chip_task* t = run.task;
    const int n_lk_num = t->num_lk_tables, n_lk_den = t->num_lk_tables > 0 ? t->num_lk_tables : t->num_lk;
    int n = 0;
    if (t->num_reads > 0) specs3[n++] = ceno_hip_virtual_tower_spec{plan_index, 0, t->num_reads, -1, 0, {1, 0}};
    if (t->num_writes > 0) specs3[n++] = ceno_hip_virtual_tower_spec{plan_index, t->num_reads, t->num_writes, -1, 0, {1, 0}};
    if (n_lk_den > 0) {
        const int lk0 = t->num_reads + t->num_writes;
        specs3[n++] = ceno_hip_virtual_tower_spec{plan_index, lk0 + n_lk_num, n_lk_den, n_lk_num > 0 ? lk0 : -1, 1, {run.challenges4[0], run.challenges4[1]}};  // cpu/mod.rs:658-661
    }
    return n;
}

int chip_run_adopt_towers(ChipProofRun& run, ceno_hip_tower* const* towers, int n) {
    const ceno_chip_task* t = run.task;
    ceno_tower_witness& tw = run.tw;
    memset(&tw, 0, sizeof(tw));
    const int n_lk_den = t->num_lk_tables > 0 ? t->num_lk_tables : t->num_lk;
    const int active_row_vars = t->log2_num_instances + t->rotation_vars;
    auto group_num_vars = [&](int num_ops) { return active_row_vars + ceil_log2(next_pow2((size_t)num_ops)); };  // cpu/mod.rs:647-648
    int k = 0;
    bool ok = true;
    if (t->num_reads > 0) {
        tw.prod[tw.n_prod++] = towers[k];
        tw.has_r = 1;
        ok = ok && ceno_hip_tower_num_vars(towers[k++]) == group_num_vars(t->num_reads);
    }
    if (t->num_writes > 0) {
        tw.prod[tw.n_prod++] = towers[k];
        tw.has_w = 1;
        ok = ok && ceno_hip_tower_num_vars(towers[k++]) == group_num_vars(t->num_writes);
    }
    if (n_lk_den > 0) {
        tw.logup[0] = towers[k];
        tw.n_logup = 1;
        tw.has_lk = 1;
        ok = ok && ceno_hip_tower_num_vars(towers[k++]) == group_num_vars(n_lk_den);
    }
    run.live = true;
    if (k != n || !ok) {
        chip_run_abandon(run);
        return prover_set_error(CENO_HIP_ERR_STATE, "build_tower_witness: tower height differs from group_num_vars");
    }
    return 0;
}

// the towers stand (their tops prefetched): out-evaluations, the proof's buffers, the transcript up to the first layer
int chip_run_after_towers(ChipProofRun& run, ceno_hip_stream s) {
    ceno_hip_ctx* ctx = run.ctx;
    ceno_chip_proof* out = run.out;
    ceno_transcript* tr = run.tr;
    ceno_tower_witness& tw = run.tw;
    int rc = 0, k = 0;
    if (tw.has_r) rc = ceno_hip_tower_out_evals(ctx, tw.prod[k++], tw.r_out_evals, s);
    if (!rc && tw.has_w) rc = ceno_hip_tower_out_evals(ctx, tw.prod[k++], tw.w_out_evals, s);
    if (!rc && tw.has_lk) rc = ceno_hip_tower_out_evals(ctx, tw.logup[0], tw.lk_out_evals, s);
    if (rc) {
        chip_run_abandon(run);
        return fail_ctx(ctx, rc);
    }
    int max_nv = 0;
    for (int i = 0; i < tw.n_prod; i++) max_nv = std::max(max_nv, ceno_hip_tower_num_vars(tw.prod[i]));
    for (int i = 0; i < tw.n_logup; i++) max_nv = std::max(max_nv, ceno_hip_tower_num_vars(tw.logup[i]));
    const int R = max_nv - 1;
    out->tower_num_vars = max_nv;
    out->n_prod = tw.n_prod;
    out->n_logup = tw.n_logup;
    out->tower.msgs = (uint64_t*)calloc(std::max<size_t>(1, ceno_tower_msgs_words(max_nv)), 8);
    out->tower.prod_evals = (uint64_t*)calloc((size_t)std::max(1, tw.n_prod) * std::max(1, R) * 4, 8);
    out->tower.logup_evals = (uint64_t*)calloc((size_t)std::max(1, tw.n_logup) * std::max(1, R) * 8, 8);
    out->tower.point = (uint64_t*)calloc((size_t)2 * (max_nv + 1), 8);
    out->rt_main = (uint64_t*)calloc((size_t)2 * std::max(1, run.num_var_with_rotation), 8);
    if (!out->tower.msgs || !out->tower.prod_evals || !out->tower.logup_evals || !out->tower.point || !out->rt_main) {
        chip_run_abandon(run);
        return prover_set_error(CENO_HIP_ERR_OOM, "create_chip_proof: out of host memory");
    }
    // prove_tower_relation binds the out-evaluations into the transcript (r, w, lk: cpu/mod.rs:783-786), the tower prover follows
    out->n_r_out = tw.has_r ? 2 : 0;
    out->n_w_out = tw.has_w ? 2 : 0;
    out->n_lk_out = tw.has_lk ? 4 : 0;
    memcpy(out->r_out_evals, tw.r_out_evals, sizeof(out->r_out_evals));
    memcpy(out->w_out_evals, tw.w_out_evals, sizeof(out->w_out_evals));
    memcpy(out->lk_out_evals, tw.lk_out_evals, sizeof(out->lk_out_evals));
    if (tw.has_r) prover_tr_ext_words(tr, tw.r_out_evals, 2);
    if (tw.has_w) prover_tr_ext_words(tr, tw.w_out_evals, 2);
    if (tw.has_lk) prover_tr_ext_words(tr, tw.lk_out_evals, 4);
    rc = tower_state_init(run.st, ctx, tw.prod, tw.n_prod, tw.logup, tw.n_logup, tr, s, &out->tower, nullptr);
    if (rc) {
        chip_run_abandon(run);
        return rc;
    }
    return 0;
}

int chip_run_begin(ChipProofRun& run, ceno_hip_ctx* ctx, const ceno_chip_task* task, const uint64_t* challenges4, ceno_transcript* tr,
                   ceno_hip_stream s, ceno_chip_proof* out) {
    static const bool trace = getenv("CENO_PROVER_CHIP_TRACE") != nullptr;  // where a chip proof's wall time goes (host view, microseconds)
    auto now_us = []() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e6 + ts.tv_nsec / 1e3;
    };
    const double t0 = trace ? now_us() : 0;
    if (int rc = chip_run_records(run, ctx, task, challenges4, tr, s, out)) return rc;
    const double t1 = trace ? now_us() : 0;
    // ---- prove_tower_relation (prover.rs:747-755 -> cpu/mod.rs:765-797) ----
    int rc = ceno_prover_build_tower_witness(ctx, run.records.data(), task->num_reads, task->num_writes, task->num_lk_tables, task->num_lk,
                                             task->log2_num_instances, task->rotation_vars, challenges4, s, &run.tw);
    const double t2 = trace ? now_us() : 0;
    chip_run_free_records(run);
    if (rc) return rc;
    run.live = true;
    rc = chip_run_after_towers(run, s);
    if (trace)
        fprintf(stderr, "[ceno_prover] chip 2^%d: wit_infer %.0f us, tower witness (to out-evals) %.0f, to the first layer %.0f\n", task->log2_num_instances,
                t1 - t0, t2 - t1, now_us() - t2);
    return rc;
}

void chip_run_abandon(ChipProofRun& run) {
    chip_run_free_records(run);
    if (run.live) ceno_tower_witness_free(run.ctx, &run.tw);
    run.live = false;
    if (run.out) ceno_chip_proof_free(run.out);
}

int chip_run_finish(ChipProofRun& run, ceno_hip_stream s) {
    ceno_hip_ctx* ctx = run.ctx;
    const ceno_chip_task* task = run.task;
    ceno_chip_proof* out = run.out;
    ceno_transcript* tr = run.tr;
    run.st.s = s;
    while (!run.st.done())
        if (int rc = tower_state_step(run.st)) {
            chip_run_abandon(run);
            return rc;
        }
    tower_state_finish(run.st);
    ceno_tower_witness_free(ctx, &run.tw);
    run.live = false;
    const int max_nv = out->tower_num_vars, num_var_with_rotation = run.num_var_with_rotation;
    // rt_main = the LAST num_var_with_rotation coordinates of the tower point (prover.rs:758-764): the record selector
    // occupies the low variables of the interleaved layout (utils.rs:402-462)
    if (max_nv < num_var_with_rotation) {
        ceno_chip_proof_free(out);
        return prover_set_error(CENO_HIP_ERR_STATE, "tower challenge point is shorter than the main point");
    }
    out->num_var_with_rotation = num_var_with_rotation;
    memcpy(out->rt_main, out->tower.point + (size_t)2 * (max_nv - num_var_with_rotation), (size_t)16 * num_var_with_rotation);
    // ---- prove_rotation (prover.rs:771-776): keccak-style chips only ----
    if (task->n_rotation_pairs > 0) {
        const int n = num_var_with_rotation, np = task->n_rotation_pairs;
        out->n_rotation_pairs = np;
        out->rotation_msgs = (uint64_t*)calloc((size_t)n * 2 * 2, 8);
        out->rotation_evals = (uint64_t*)calloc((size_t)3 * np * 2, 8);
        out->rotation_points = (uint64_t*)calloc((size_t)3 * n * 2, 8);  // origin | left | right
        if (!out->rotation_msgs || !out->rotation_evals || !out->rotation_points) {
            ceno_chip_proof_free(out);
            return prover_set_error(CENO_HIP_ERR_OOM, "create_chip_proof: out of host memory");
        }
        int rc = ceno_prover_prove_rotation(ctx, task->mles, task->rotation_source_idx, task->rotation_target_idx, np, task->cyclic_subgroup_size,
                                            task->cyclic_group_log2, out->rt_main, n, tr, s, out->rotation_msgs, out->rotation_evals,
                                            out->rotation_points, out->rotation_points + (size_t)2 * n, out->rotation_points + (size_t)4 * n);
        if (rc) {
            ceno_chip_proof_free(out);
            return rc;
        }
    }
    return 0;
}
