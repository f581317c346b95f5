// Batched main-constraint sumcheck — host control flow of
// `BatchedMainConstraintProver::prove_batched_main_constraints` (ceno_zkvm/src/scheme/cpu/mod.rs:1052-1390),
// written against the device C ABI.  See include/ceno_prover.h for the job encoding.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <string>
#include <vector>

#include "../../include/ceno_prover.h"
#include "../csrc/gl64.hpp"
#include "tower_hook.hpp"

using gl::E2;

extern "C" const char* ceno_prover_last_error(void);
int prover_set_error(int code, const char* msg);  // prover.cpp

namespace {

// extrapolate_uni_poly: degree-d polynomial through (0, p0), (i, ev[i-1]) evaluated at x
E2 extrapolate(E2 p0, const std::vector<E2>& ev, E2 x) {
    const int d = (int)ev.size();
    E2 acc = gl::e2_zero();
    for (int i = 0; i <= d; i++) {
        E2 yi = i == 0 ? p0 : ev[i - 1];
        E2 num = gl::e2_one();
        uint64_t den = 1;
        for (int j = 0; j <= d; j++) {
            if (j == i) continue;
            num = num * (x - E2{(uint64_t)j, 0});
            uint64_t dij = i > j ? (uint64_t)(i - j) : gl::neg((uint64_t)(j - i));
            den = gl::mul(den, dij);
        }
        acc = acc + yi * gl::e2_mul_base(num, gl::inv(den));
    }
    return acc;
}

}  // namespace

extern "C" int ceno_prover_prove_batched_main_constraints(ceno_hip_ctx* ctx, const ceno_main_job* jobs, int n_jobs, const uint64_t* gc4,
                                                          ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_claimed_sum, uint64_t* out_msgs,
                                                          uint64_t* out_global_rt, uint64_t* out_evals, int* out_num_vars, int* out_degree) {
    return prover_main_constraints_sharded(ctx, jobs, n_jobs, gc4, tr, s, out_claimed_sum, out_msgs, out_global_rt, out_evals, out_num_vars, out_degree, nullptr);
}

// sh == NULL: the tables of the jobs are whole.  sh != NULL (ceno_dist_prove_batched_main_constraints, dist_gkr.cpp): every table holds THIS rank's
// rows in the block layout of the row-sharded chip proof (rows whose index bits [q, q + k) equal the rank; J.num_vars stays the GLOBAL number
// of variables, the tables have num_vars - k).  A Whole / Prefix selector is eq(x, point) on a row range: on a rank it is the same construction
// at the point without the rank coordinates, on the rank's part of the range (a prefix of its own rows), times the scalar
// eq(rank, point[q .. q + k)), which rides on the coefficients of the selector's terms.  The first q rounds run on the local tables (the d
// partial evaluations of a round are summed over the ranks), then every table — 2^(n_c - k - q) entries per rank — is gathered with the
// rank bits lowest and the remaining rounds run replicated; the evaluations of the columns that left the plan (below) are per-rank
// evaluations at the local point, weighted by eq over the rank coordinates and summed.  Every rank ends with the single-device outputs.
int prover_main_constraints_sharded(ceno_hip_ctx* ctx, const ceno_main_job* jobs, int n_jobs, const uint64_t* gc4, ceno_transcript* tr, ceno_hip_stream s,
                                    uint64_t* out_claimed_sum, uint64_t* out_msgs, uint64_t* out_global_rt, uint64_t* out_evals, int* out_num_vars,
                                    int* out_degree, const RotationShard* sh) {
    if (!ctx || !jobs || n_jobs < 1 || !gc4 || !tr) return prover_set_error(CENO_HIP_ERR_INVALID, "bad arguments");
    const int shk = sh ? sh->k : 0, shq = sh ? sh->q : 0;  // (constants of the layout; `sh` itself is dropped below when no chip is large enough to be sharded)
    auto eq_rank = [&](const uint64_t* p) {  // eq(rank, p[q .. q + k))
        E2 v = gl::e2_one();
        for (int j = 0; j < shk; j++) {
            const E2 c{p[2 * (shq + j)], p[2 * (shq + j) + 1]};
            v = v * (((sh->rank >> j) & 1) ? c : gl::e2_one() - c);
        }
        return v;
    };
    auto local_point = [&](const uint64_t* p, int n_glob) {  // the point without its rank coordinates
        std::vector<uint64_t> out;
        for (int j = 0; j < n_glob; j++)
            if (j < shq || j >= shq + shk) {
                out.push_back(p[2 * j]);
                out.push_back(p[2 * j + 1]);
            }
        return out;
    };
    auto local_rows = [&](size_t t) -> size_t {  // how many of this rank's rows have a global index below t
        const size_t lo = t & (((size_t)1 << shq) - 1), g = (t >> shq) & (((size_t)1 << shk) - 1), hi = t >> (shq + shk);
        return (hi << shq) + ((size_t)sh->rank < g ? (size_t)1 << shq : ((size_t)sh->rank == g ? lo : 0));
    };
    // a chip with fewer than 2^(q + k + 1) rows is not sharded: every rank holds its WHOLE tables (`big[c]` = 0) and proves it replicated — its part
    // of every round's message is the same on every rank and is added once
    std::vector<char> big((size_t)n_jobs, 0);
    bool any_big = false, any_small = false;
    for (int c = 0; c < n_jobs; c++) {
        big[(size_t)c] = sh && jobs[c].num_vars - shk >= shq + 1;
        any_big = any_big || big[(size_t)c];
        any_small = any_small || !big[(size_t)c];
    }
    if (sh && !any_big) sh = nullptr;  // nothing is sharded: every rank runs the whole batch (same transcript, same outputs)
    static const bool dbg = getenv("CENO_HIP_DEBUG") != nullptr;
    auto now_us = []() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e6 + ts.tv_nsec / 1e3;
    };
    const double t_start = dbg ? now_us() : 0;
    int max_nv = 0, max_deg = 0, total_exprs = 0;
    for (int c = 0; c < n_jobs; c++) {
        max_nv = std::max(max_nv, jobs[c].num_vars);       // cpu/mod.rs:1091-1099
        max_deg = std::max(max_deg, jobs[c].max_degree);
        total_exprs += jobs[c].n_exprs;
    }
    // ---- selector eq tables per chip (cpu/mod.rs:1200-1234): first selector per structural id wins ----
    std::vector<std::vector<ceno_hip_mle*>> sel_by_id(n_jobs);
    std::vector<std::vector<int>> sel_k_by_id(n_jobs);  // which selector of the job stands for a structural id
    std::vector<ceno_hip_mle*> owned;
    auto cleanup = [&]() { for (auto* m : owned) ceno_hip_mle_free(ctx, m); };
    // Whole / Prefix selectors (every chip's sel_all: the common case) are built together, two launches for the batch; the other kinds one
    // at a time.  First selector per structural id wins.
    struct Pending { int c, id, k; };
    std::vector<Pending> batch;
    for (int c = 0; c < n_jobs; c++) {
        const ceno_main_job& J = jobs[c];
        sel_by_id[c].assign(J.n_structural, nullptr);
        sel_k_by_id[c].assign(J.n_structural, -1);
        for (int k = 0; k < J.n_selectors; k++) {
            const int id = J.sel_structural_id[k];
            if (id < 0 || id >= J.n_structural) { cleanup(); return prover_set_error(CENO_HIP_ERR_INVALID, "selector wit id out of range"); }
            if (sel_k_by_id[c][id] >= 0) continue;
            sel_k_by_id[c][id] = k;
            if (J.sel_kind[k] == CENO_HIP_SEL_WHOLE || J.sel_kind[k] == CENO_HIP_SEL_PREFIX) {
                batch.push_back(Pending{c, id, k});
                continue;
            }
            if (sh && big[(size_t)c]) { cleanup(); return prover_set_error(CENO_HIP_ERR_UNSUPPORTED, "sharded main constraints: Whole and Prefix selectors only"); }
            ceno_hip_mle* m = nullptr;
            int rc = ceno_hip_selector_build(ctx, J.sel_kind[k], J.sel_points[k], J.num_vars, J.sel_offset[k], J.sel_num_instances[k],
                                             J.sel_sparse_indices ? J.sel_sparse_indices[k] : nullptr, J.sel_n_sparse ? J.sel_n_sparse[k] : 0,
                                             J.sel_sparse_num_vars ? J.sel_sparse_num_vars[k] : 0, s, &m);
            if (rc) { cleanup(); return prover_set_error(rc, ceno_hip_last_error(ctx)); }
            owned.push_back(m);
            sel_by_id[c][id] = m;
        }
    }
    std::vector<std::vector<uint64_t>> sel_loc_pt;        // sharded: per batched selector its point without the rank coordinates ...
    std::vector<std::vector<size_t>> sel_loc_pt_of(n_jobs);  // ... indexed [job][selector k] -> sel_loc_pt (SIZE_MAX: none)
    std::vector<std::vector<E2>> sel_eqg(n_jobs);         // ... and per structural id the rank's eq factor (1 when the tables are whole)
    for (int c = 0; c < n_jobs; c++) {
        sel_eqg[c].assign(jobs[c].n_structural, gl::e2_one());
        sel_loc_pt_of[c].assign(jobs[c].n_selectors, (size_t)-1);
    }
    if (!batch.empty()) {
        const int nb = (int)batch.size();
        std::vector<int> kinds(nb), nvs(nb);
        std::vector<const uint64_t*> pts(nb);
        std::vector<size_t> offs(nb), nins(nb);
        std::vector<ceno_hip_mle*> outs(nb, nullptr);
        sel_loc_pt.reserve((size_t)nb);
        for (int b = 0; b < nb; b++) {
            const ceno_main_job& J = jobs[batch[b].c];
            const int k = batch[b].k;
            kinds[b] = J.sel_kind[k];
            nvs[b] = J.num_vars;
            pts[b] = J.sel_points[k];
            offs[b] = J.sel_offset[k];
            nins[b] = J.sel_num_instances[k];
            if (sh && big[(size_t)batch[b].c]) {
                sel_loc_pt.push_back(local_point(J.sel_points[k], J.num_vars));
                sel_loc_pt_of[batch[b].c][k] = sel_loc_pt.size() - 1;
                pts[b] = sel_loc_pt.back().data();
                nvs[b] = J.num_vars - shk;
                sel_eqg[batch[b].c][batch[b].id] = eq_rank(J.sel_points[k]);
                if (kinds[b] == CENO_HIP_SEL_PREFIX) {
                    const size_t lo = local_rows(J.sel_offset[k]), hi = local_rows(J.sel_offset[k] + J.sel_num_instances[k]);
                    offs[b] = lo;
                    nins[b] = hi - lo;
                }
            }
        }
        int rc = ceno_hip_selector_build_batch(ctx, nb, kinds.data(), pts.data(), nvs.data(), offs.data(), nins.data(), s, outs.data());
        if (rc) { cleanup(); return prover_set_error(rc, ceno_hip_last_error(ctx)); }
        for (int b = 0; b < nb; b++) {
            owned.push_back(outs[b]);
            sel_by_id[batch[b].c][batch[b].id] = outs[b];
        }
    }
    const double t_sel = dbg ? (ceno_hip_stream_sync(ctx, s), now_us()) : 0;
    // ---- alpha powers (cpu/mod.rs:1254) ----
    static const char lbl[] = "combine subset evals";
    tr->append_label(tr->self, (const uint8_t*)lbl, sizeof(lbl) - 1);
    uint64_t a2[2];
    tr->sample_ext(tr->self, a2);
    std::vector<E2> alpha_pows(total_exprs);
    {
        E2 a{a2[0], a2[1]}, acc = gl::e2_one();
        for (int i = 0; i < total_exprs; i++) { alpha_pows[i] = acc; acc = acc * a; }
    }
    // ---- global MLE list and monomial terms (cpu/mod.rs:1255-1329) ----
    std::vector<ceno_hip_mle*> mles;
    std::vector<E2> mle_eqg;  // per table: the rank's eq factor of a sharded selector (1 otherwise)
    std::vector<int> mle_job;  // per table: its chip
    std::vector<int> mle_start(n_jobs), mle_nv;
    std::vector<uint64_t> coeffs;
    std::vector<uint32_t> toff{0}, tidx;
    // Whole / Prefix selectors ARE eq(., point) on a row range: declared to the sumcheck (ceno_hip_sumcheck_begin_eq), whose rounds then
    // evaluate a chip's quotient by eq(X, rt_i) at one point fewer (the k_tower idea applied to the main constraints)
    std::vector<int> eq_idx;
    std::vector<const uint64_t*> eq_pts;
    std::vector<size_t> eq_lo, eq_hi;
    int alpha_start = 0;
    for (int c = 0; c < n_jobs; c++) {
        const ceno_main_job& J = jobs[c];
        mle_start[c] = (int)mles.size();
        const int n_m = J.n_witin + J.n_fixed + J.n_structural;
        for (int j = 0; j < n_m; j++) {
            ceno_hip_mle* m = J.mles[j];
            if (j >= J.n_witin + J.n_fixed && sel_by_id[c][j - J.n_witin - J.n_fixed]) {
                const int id = j - J.n_witin - J.n_fixed, k = sel_k_by_id[c][id];
                m = sel_by_id[c][id];
                if (J.sel_kind[k] == CENO_HIP_SEL_WHOLE || J.sel_kind[k] == CENO_HIP_SEL_PREFIX) {
                    eq_idx.push_back((int)mles.size());
                    if (!sh || !big[(size_t)c]) {
                        eq_pts.push_back(J.sel_points[k]);
                        eq_lo.push_back(J.sel_kind[k] == CENO_HIP_SEL_WHOLE ? 0 : J.sel_offset[k]);
                        eq_hi.push_back(J.sel_kind[k] == CENO_HIP_SEL_WHOLE ? (size_t)1 << J.num_vars : J.sel_offset[k] + J.sel_num_instances[k]);
                    } else {  // this rank's rows of the table: eq at the local point on the rank's part of the range
                        eq_pts.push_back(sel_loc_pt[sel_loc_pt_of[c][k]].data());
                        eq_lo.push_back(J.sel_kind[k] == CENO_HIP_SEL_WHOLE ? 0 : local_rows(J.sel_offset[k]));
                        eq_hi.push_back(J.sel_kind[k] == CENO_HIP_SEL_WHOLE ? (size_t)1 << (J.num_vars - shk) : local_rows(J.sel_offset[k] + J.sel_num_instances[k]));
                    }
                }
            }
            if (!m) { cleanup(); return prover_set_error(CENO_HIP_ERR_INVALID, "structural witness without selector is NULL"); }
            const int shift = sh && big[(size_t)c] ? shk : 0;
            if (sh && ceno_hip_mle_num_vars(m) != J.num_vars - shift) { cleanup(); return prover_set_error(CENO_HIP_ERR_INVALID, "sharded main constraints: a table has the wrong height (rows of this rank for a sharded chip, all rows for a small one)"); }
            mles.push_back(m);
            mle_job.push_back(c);
            mle_nv.push_back(ceno_hip_mle_num_vars(m) + shift);  // (the GLOBAL number of variables)
            mle_eqg.push_back(j >= J.n_witin + J.n_fixed && sel_by_id[c][j - J.n_witin - J.n_fixed] ? sel_eqg[c][j - J.n_witin - J.n_fixed] : gl::e2_one());
        }
        std::vector<E2> ch{E2{gc4[0], gc4[1]}, E2{gc4[2], gc4[3]}};
        for (int i = 0; i < J.n_exprs; i++) ch.push_back(alpha_pows[alpha_start + i]);
        for (int i = 0; i < J.n_pi && J.pi; i++) ch.push_back(E2{J.pi[2 * i], J.pi[2 * i + 1]});  // Instance atoms (cpu/mod.rs:1241-1247,1297-1304)
        for (int t = 0; t < J.n_terms; t++) {
            E2 scalar = gl::e2_zero();
            for (uint32_t m = J.scalar_offsets[t]; m < J.scalar_offsets[t + 1]; m++) {
                E2 v{J.mono_coeffs[2 * m], J.mono_coeffs[2 * m + 1]};
                for (uint32_t k = J.mono_chal_offsets[m]; k < J.mono_chal_offsets[m + 1]; k++) {
                    if (J.mono_chal_idx[k] >= ch.size()) { cleanup(); return prover_set_error(CENO_HIP_ERR_INVALID, "challenge id out of range"); }
                    v = v * ch[J.mono_chal_idx[k]];
                }
                scalar = scalar + v;
            }
            if (scalar.c0 == 0 && scalar.c1 == 0) continue;  // cpu/mod.rs:1308-1314
            coeffs.push_back(scalar.c0);
            coeffs.push_back(scalar.c1);
            for (uint32_t k = J.term_offsets[t]; k < J.term_offsets[t + 1]; k++) tidx.push_back((uint32_t)(mle_start[c] + (int)J.term_mle_idx[k]));
            toff.push_back((uint32_t)tidx.size());
        }
        alpha_start += J.n_exprs;
    }
    const int n_terms_all = (int)toff.size() - 1;
    if (n_terms_all == 0) { cleanup(); return prover_set_error(CENO_HIP_ERR_INVALID, "all term scalars are zero"); }
    // ---- columns the batch reads only LINEARLY.  The monomial form of a chip is dominated by `selector x column` terms — the records are
    // RLCs of columns (zerocheck_layer.rs:118-140, instructions.rs:48-83) — and most columns occur in nothing else.  For such columns
    //     sum_j c_j sel(x) col_j(x) = sel(x) A(x) + X sel(x) B(x),   A = sum_j c_j.c0 col_j,  B = sum_j c_j.c1 col_j
    // with two BASE-field tables per selector built in one streaming pass (ceno_hip_lincomb_base_batch): the sumcheck then folds two tables
    // where it folded dozens, and the evaluations of the columns themselves at the sumcheck's point — which the proof carries — come from one
    // read-only pass afterwards (ceno_hip_mle_evaluate_prefix_batch).  Messages and evaluations are the same field elements either way.
    // CENO_PROVER_MAIN_LINCOMB: smallest group that is combined (default 3; 0: off).  Read per call. ----
    const int lin_min = getenv("CENO_PROVER_MAIN_LINCOMB") ? atoi(getenv("CENO_PROVER_MAIN_LINCOMB")) : 3;
    std::vector<int> plan_of(mles.size(), 0);          // index in the sumcheck's table list, -1: a combined column
    std::vector<ceno_hip_mle*> plan_mles;
    std::vector<uint64_t> p_coeffs, p_coeffs_loc;
    std::vector<uint32_t> p_toff{0}, p_tidx;
    std::vector<int> removed;                          // global ids of the combined columns
    std::vector<E2> plan_eqg;                          // per table of the plan: the rank's eq factor of a sharded selector (1 otherwise)
    std::vector<int> plan_job;                         // per table of the plan: its chip
    {
        const size_t nm = mles.size();
        std::vector<char> is_ext(nm), other(nm, 0);
        for (size_t j = 0; j < nm; j++) is_ext[j] = (char)ceno_hip_mle_is_ext(mles[j]);
        std::vector<int> lin_sel(n_terms_all, -1), lin_col(n_terms_all, -1);
        for (int t = 0; t < n_terms_all; t++) {
            int n_ext = 0, n_base = 0, sel = -1, col = -1;
            for (uint32_t k = toff[t]; k < toff[t + 1]; k++) {
                if (is_ext[tidx[k]]) n_ext++, sel = (int)tidx[k];
                else n_base++, col = (int)tidx[k];
            }
            if (lin_min > 0 && n_ext == 1 && n_base == 1 && mle_nv[sel] == mle_nv[col]) {
                lin_sel[t] = sel;
                lin_col[t] = col;
            } else {
                for (uint32_t k = toff[t]; k < toff[t + 1]; k++) other[tidx[k]] = 1;
            }
        }
        // groups per selector: EVERY `selector x column` monomial of a selector with at least lin_min of them moves into the selector's two
        // combined tables — also those of a column that products read as well (its monomial then costs one pass of the combination instead of
        // D - 2 multiply-accumulates per pair and round); a column leaves the plan when nothing else reads it and all its monomials moved
        std::map<int, std::map<int, E2>> by_sel;  // selector -> column -> summed coefficient
        for (int t = 0; t < n_terms_all; t++)
            if (lin_sel[t] >= 0) by_sel[lin_sel[t]].emplace(lin_col[t], gl::e2_zero());
        for (auto it = by_sel.begin(); it != by_sel.end();) it = (int)it->second.size() >= lin_min ? std::next(it) : by_sel.erase(it);
        for (int t = 0; t < n_terms_all; t++) {
            if (lin_sel[t] < 0) continue;
            auto g = by_sel.find(lin_sel[t]);
            if (g == by_sel.end()) {  // the monomial stays a term of the sumcheck: so does its column
                other[lin_col[t]] = 1;
                lin_sel[t] = lin_col[t] = -1;
                continue;
            }
            E2& c = g->second[lin_col[t]];
            c = c + E2{coeffs[2 * t], coeffs[2 * t + 1]};
        }
        std::vector<char> gone(nm, 0);
        std::vector<uint32_t> goff{0};
        std::vector<ceno_hip_mle*> gcols;
        std::vector<uint64_t> gco;
        std::vector<int> gsel;
        for (auto& g : by_sel) {
            gsel.push_back(g.first);
            for (auto& kv : g.second) {
                if (!other[kv.first]) gone[kv.first] = 1;
                gcols.push_back(mles[kv.first]);
                gco.push_back(kv.second.c0);
                gco.push_back(kv.second.c1);
            }
            goff.push_back((uint32_t)gcols.size());
        }
        std::vector<ceno_hip_mle*> la(gsel.size(), nullptr), lb(gsel.size(), nullptr);
        if (!gsel.empty()) {
            int rc = ceno_hip_lincomb_base_batch(ctx, (int)gsel.size(), goff.data(), gcols.data(), gco.data(), s, la.data(), lb.data());
            if (rc) { cleanup(); return prover_set_error(rc, ceno_hip_last_error(ctx)); }
            for (size_t g = 0; g < gsel.size(); g++) {
                owned.push_back(la[g]);
                owned.push_back(lb[g]);
            }
        }
        for (size_t j = 0; j < nm; j++) {
            if (gone[j]) {
                plan_of[j] = -1;
                removed.push_back((int)j);
            } else {
                plan_of[j] = (int)plan_mles.size();
                plan_mles.push_back(mles[j]);
            }
        }
        auto push_coeff = [&](E2 c, E2 rank_factor) {  // the term's coefficient; the LOCAL plan of a sharded run carries its selectors' eq factors
            p_coeffs.push_back(c.c0);
            p_coeffs.push_back(c.c1);
            const E2 cl = c * rank_factor;
            p_coeffs_loc.push_back(cl.c0);
            p_coeffs_loc.push_back(cl.c1);
        };
        for (int t = 0; t < n_terms_all; t++) {
            if (lin_col[t] >= 0) continue;  // moved into its selector's combination
            E2 f = gl::e2_one();
            for (uint32_t k = toff[t]; k < toff[t + 1]; k++) f = f * mle_eqg[tidx[k]];
            push_coeff(E2{coeffs[2 * t], coeffs[2 * t + 1]}, f);
            for (uint32_t k = toff[t]; k < toff[t + 1]; k++) p_tidx.push_back((uint32_t)plan_of[tidx[k]]);
            p_toff.push_back((uint32_t)p_tidx.size());
        }
        for (size_t g = 0; g < gsel.size(); g++) {
            for (int half = 0; half < 2; half++) {  // sel x A + X sel x B
                push_coeff(half == 0 ? E2{1, 0} : E2{0, 1}, mle_eqg[(size_t)gsel[g]]);
                p_tidx.push_back((uint32_t)plan_of[gsel[g]]);
                p_tidx.push_back((uint32_t)plan_mles.size());
                p_toff.push_back((uint32_t)p_tidx.size());
                plan_mles.push_back(half == 0 ? la[g] : lb[g]);
            }
        }
        for (int& e : eq_idx) e = plan_of[e];
        plan_eqg.assign(plan_mles.size(), gl::e2_one());
        plan_job.assign(plan_mles.size(), 0);
        for (size_t j = 0; j < nm; j++)
            if (plan_of[j] >= 0) {
                plan_eqg[(size_t)plan_of[j]] = mle_eqg[j];
                plan_job[(size_t)plan_of[j]] = mle_job[j];
            }
        for (size_t g = 0; g < gsel.size(); g++)  // the combined tables belong to their selector's chip
            for (int half = 0; half < 2; half++) plan_job[plan_mles.size() - 2 * (gsel.size() - g) + (size_t)half] = mle_job[(size_t)gsel[g]];
    }
    const int n_terms = (int)p_toff.size() - 1;
    // ---- common-factor plan (the role of CommonTermPlan in the reference's GPU arm, scheme/gpu/mod.rs:2811-2962, built
    // there from shared witness prefixes, gkr_iop/src/gkr/layer/zerocheck_layer.rs:389-513).  Any factoring gives the same
    // messages; here terms are grouped by their EXTENSION-field factors — the selectors (eq tables) that every constraint of
    // an expression group carries — so that a group is  selector(s) x sum_t c_t prod(base witness columns):  the selector is
    // multiplied in once per group instead of once per term, and the first round keeps the column products in the base
    // field (k_accum_base0). ----
    std::vector<uint32_t> r_toff{0}, r_tidx;                  // residual factor lists handed to the engine
    std::vector<uint32_t> g_toff{0}, g_tidx, g_coff{0}, g_cidx;
    {
        std::map<std::vector<uint32_t>, std::vector<int>> by_ext;  // (ext factor list incl. the chip's MLE base) -> terms
        std::vector<std::vector<uint32_t>> base_part(n_terms);
        for (int t = 0; t < n_terms; t++) {
            std::vector<uint32_t> ext_part;
            for (uint32_t k = p_toff[t]; k < p_toff[t + 1]; k++) (ceno_hip_mle_is_ext(plan_mles[p_tidx[k]]) ? ext_part : base_part[t]).push_back(p_tidx[k]);
            if (ext_part.size() == 1 && base_part[t].empty()) {
                // selector x constant (the constants of a chip's record RLCs, zerocheck_layer.rs:118-140): a term of the selector's group
                // with no factor of its own — left ungrouped it would read the selector as an ordinary factor and take the whole chip off
                // the eq-factored rounds
                by_ext[ext_part].push_back(t);
            } else if (ext_part.empty() || base_part[t].empty()) {
                base_part[t].assign(p_tidx.begin() + p_toff[t], p_tidx.begin() + p_toff[t + 1]);  // ungrouped: full product
            } else {
                std::sort(ext_part.begin(), ext_part.end());
                by_ext[ext_part].push_back(t);
            }
        }
        for (int t = 0; t < n_terms; t++) {
            r_tidx.insert(r_tidx.end(), base_part[t].begin(), base_part[t].end());
            r_toff.push_back((uint32_t)r_tidx.size());
        }
        for (auto& kv : by_ext) {
            for (int t : kv.second) g_tidx.push_back((uint32_t)t);
            g_toff.push_back((uint32_t)g_tidx.size());
            g_cidx.insert(g_cidx.end(), kv.first.begin(), kv.first.end());
            g_coff.push_back((uint32_t)g_cidx.size());
        }
    }
    ceno_hip_sumcheck_plan plan{};
    plan.num_mles = (int)plan_mles.size();
    plan.num_terms = n_terms;
    plan.term_coeffs = p_coeffs.data();
    plan.term_offsets = r_toff.data();
    plan.term_mle_idx = r_tidx.data();
    plan.num_groups = (int)g_toff.size() - 1;
    plan.group_term_offsets = g_toff.data();
    plan.group_term_idx = g_tidx.data();
    plan.common_offsets = g_coff.data();
    plan.common_mle_idx = g_cidx.data();
    plan.max_num_vars = max_nv;
    plan.max_degree = max_deg;
    std::vector<uint64_t> evals(2 * mles.size()), p_evals(2 * plan_mles.size());
    const double t_plan = dbg ? now_us() : 0;
    int rc = 0;
    if (!sh) {
        rc = ceno_prover_sumcheck_prove_eq(ctx, plan_mles.data(), &plan, (int)eq_idx.size(), eq_idx.data(), eq_pts.data(), eq_lo.data(), eq_hi.data(), tr, s,
                                           out_msgs, out_global_rt, p_evals.data());   // cpu/mod.rs:1332-1337
    } else {
        // ---- the same sumcheck over row-sharded tables: q local rounds, the gathered tail replicated; the chips that are too small to be
        // sharded run beside it in a sumcheck of their own, the same on every rank, whose messages are added once ----
        const int D = max_deg, W = sh->world;
        auto tr_usize = [&](uint64_t v) {
            uint8_t b[8];
            for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
            tr->append_label(tr->self, b, 8);
        };
        // the plan of a subset of the chips: its tables, terms and groups renumbered (a term, a group and an eq declaration belong to one chip)
        struct Pack {
            std::vector<ceno_hip_mle*> mles;
            std::vector<int> orig;  // index in the whole plan
            std::vector<uint64_t> coeffs, coeffs_loc;
            std::vector<uint32_t> toff{0}, tidx, gtoff{0}, gtidx, gcoff{0}, gcidx;
            std::vector<int> eq_i;
            std::vector<const uint64_t*> eq_p;
            std::vector<size_t> eq_l, eq_h;
            ceno_hip_sumcheck_plan plan{};
        };
        auto make_pack = [&](bool want_big, Pack& P) -> int {
            std::vector<int> remap(plan_mles.size(), -1), tmap((size_t)n_terms, -1);
            for (size_t i = 0; i < plan_mles.size(); i++)
                if ((bool)big[(size_t)plan_job[i]] == want_big) {
                    remap[i] = (int)P.mles.size();
                    P.mles.push_back(plan_mles[i]);
                    P.orig.push_back((int)i);
                }
            auto job_of_term = [&](int t) {  // (a term without residual factors sits in a group: its common factor names the chip)
                if (p_toff[t + 1] > p_toff[t]) return plan_job[p_tidx[p_toff[t]]];
                return -1;
            };
            std::vector<int> tjob((size_t)n_terms, -1);
            for (int t = 0; t < n_terms; t++) tjob[(size_t)t] = job_of_term(t);
            for (size_t g = 0; g + 1 < g_toff.size(); g++) {
                const int j = plan_job[g_cidx[g_coff[g]]];
                for (uint32_t x = g_toff[g]; x < g_toff[g + 1]; x++) tjob[g_tidx[x]] = j;
            }
            for (int t = 0; t < n_terms; t++) {
                // a monomial with no factor at all, in no group: nothing names its chip (the engine refuses such a plan as an "empty product",
                // but only at begin — this index comes first)
                if (tjob[(size_t)t] < 0) return CENO_HIP_ERR_INVALID;
                if ((bool)big[(size_t)tjob[(size_t)t]] != want_big) continue;
                tmap[(size_t)t] = (int)P.toff.size() - 1;
                P.coeffs.insert(P.coeffs.end(), {p_coeffs[2 * t], p_coeffs[2 * t + 1]});
                P.coeffs_loc.insert(P.coeffs_loc.end(), {p_coeffs_loc[2 * t], p_coeffs_loc[2 * t + 1]});
                for (uint32_t x = r_toff[t]; x < r_toff[t + 1]; x++) P.tidx.push_back((uint32_t)remap[r_tidx[x]]);
                P.toff.push_back((uint32_t)P.tidx.size());
            }
            for (size_t g = 0; g + 1 < g_toff.size(); g++) {
                if ((bool)big[(size_t)plan_job[g_cidx[g_coff[g]]]] != want_big) continue;
                for (uint32_t x = g_toff[g]; x < g_toff[g + 1]; x++) P.gtidx.push_back((uint32_t)tmap[g_tidx[x]]);
                P.gtoff.push_back((uint32_t)P.gtidx.size());
                for (uint32_t x = g_coff[g]; x < g_coff[g + 1]; x++) P.gcidx.push_back((uint32_t)remap[g_cidx[x]]);
                P.gcoff.push_back((uint32_t)P.gcidx.size());
            }
            for (size_t e = 0; e < eq_idx.size(); e++)
                if (remap[(size_t)eq_idx[e]] >= 0) {
                    P.eq_i.push_back(remap[(size_t)eq_idx[e]]);
                    P.eq_p.push_back(eq_pts[e]);
                    P.eq_l.push_back(eq_lo[e]);
                    P.eq_h.push_back(eq_hi[e]);
                }
            P.plan.num_mles = (int)P.mles.size();
            P.plan.num_terms = (int)P.toff.size() - 1;
            P.plan.term_coeffs = P.coeffs.data();
            P.plan.term_offsets = P.toff.data();
            P.plan.term_mle_idx = P.tidx.data();
            P.plan.num_groups = (int)P.gtoff.size() - 1;
            P.plan.group_term_offsets = P.gtoff.data();
            P.plan.group_term_idx = P.gtidx.data();
            P.plan.common_offsets = P.gcoff.data();
            P.plan.common_mle_idx = P.gcidx.data();
            P.plan.max_degree = D;
            return 0;
        };
        Pack PB, PS;
        if (make_pack(true, PB) || (any_small && make_pack(false, PS))) {
            cleanup();
            return prover_set_error(CENO_HIP_ERR_INVALID, "main constraints: a monomial without factors belongs to no chip (empty product)");
        }
        ceno_hip_sumcheck *sc = nullptr, *sc2 = nullptr, *scs = nullptr;
        auto drop = [&]() {
            if (sc) ceno_hip_sumcheck_free(ctx, sc);
            if (sc2) ceno_hip_sumcheck_free(ctx, sc2);
            if (scs) ceno_hip_sumcheck_free(ctx, scs);
            sc = sc2 = scs = nullptr;
        };
        PB.plan.term_coeffs = PB.coeffs_loc.data();
        PB.plan.max_num_vars = max_nv - shk;
        rc = ceno_hip_sumcheck_begin_eq(ctx, PB.mles.data(), &PB.plan, (int)PB.eq_i.size(), PB.eq_i.data(), PB.eq_p.data(), PB.eq_l.data(), PB.eq_h.data(), s, &sc);
        if (!rc && any_small && PS.plan.num_terms > 0) {  // the small chips: whole tables, all max_nv rounds in one handle (its own front-load rule)
            PS.plan.max_num_vars = max_nv;
            rc = ceno_hip_sumcheck_begin_eq(ctx, PS.mles.data(), &PS.plan, (int)PS.eq_i.size(), PS.eq_i.data(), PS.eq_p.data(), PS.eq_l.data(), PS.eq_h.data(), s, &scs);
        }
        if (rc) { drop(); cleanup(); return prover_set_error(rc, ceno_hip_last_error(ctx)); }
        tr_usize((uint64_t)max_nv);
        tr_usize((uint64_t)D);
        uint64_t ch[2] = {0, 0};
        std::vector<uint64_t> m((size_t)2 * D), ms((size_t)2 * D), all((size_t)W * 2 * D);
        auto publish = [&](int round, const E2* pv) {
            uint64_t* msg = out_msgs + (size_t)2 * D * round;
            for (int e = 0; e < D; e++) {
                msg[2 * e] = pv[e].c0;
                msg[2 * e + 1] = pv[e].c1;
                tr->append_ext(tr->self, msg + 2 * e);
            }
            static const char lbl_r[] = "Internal round";
            tr->append_label(tr->self, (const uint8_t*)lbl_r, sizeof(lbl_r) - 1);
            tr->sample_ext(tr->self, ch);
            out_global_rt[2 * round] = ch[0];
            out_global_rt[2 * round + 1] = ch[1];
        };
        std::vector<E2> pv((size_t)D);
        auto add_small = [&](int i) -> int {  // the small chips' part of round i, once
            if (!scs) return 0;
            int r2 = ceno_hip_sumcheck_round(ctx, scs, i == 0 ? nullptr : ch, ms.data());
            if (r2) return prover_set_error(r2, ceno_hip_last_error(ctx));
            for (int e = 0; e < D; e++) pv[(size_t)e] = pv[(size_t)e] + E2{ms[2 * e], ms[2 * e + 1]};
            return 0;
        };
        for (int i = 0; i < shq && !rc; i++) {
            rc = ceno_hip_sumcheck_round(ctx, sc, i == 0 ? nullptr : ch, m.data());
            if (rc) { rc = prover_set_error(rc, ceno_hip_last_error(ctx)); break; }
            rc = sh->allgather(sh->self, m.data(), m.size(), all.data());
            if (rc) break;
            for (int e = 0; e < D; e++) {
                pv[(size_t)e] = gl::e2_zero();
                for (int g = 0; g < W; g++) pv[(size_t)e] = pv[(size_t)e] + E2{all[(size_t)g * 2 * D + 2 * e], all[(size_t)g * 2 * D + 2 * e + 1]};
            }
            rc = add_small(i);
            if (rc) break;
            publish(i, pv.data());
        }
        if (rc) { drop(); cleanup(); return rc; }
        // every sharded table as the next round would read it (folded q - 1 times; once more here), gathered with the rank bits lowest
        const E2 r_last{ch[0], ch[1]};
        std::vector<ceno_hip_mle*> glob(PB.mles.size(), nullptr);
        {
            // ALL tables in three steps — one fetch (every copy queued, one wait), ONE exchange of the concatenated folded tables, one upload into one
            // device block that the gathered tables are views of — instead of a copy + wait, an exchange and an upload PER TABLE: a wide plan has
            // hundreds of tables in the sumcheck, and on eight ranks that was hundreds of exchanges in front of the tail (config #4's wide plan at
            // max_nv = 20 on 8 virtual ranks: 301 message exchanges per rank before, ~30 now; tests/test_gpu_dist_at_size.py)
            const size_t n_t = PB.mles.size();
            std::vector<size_t> len_loc(n_t), off_loc(n_t + 1, 0);
            for (size_t mi = 0; mi < n_t; mi++) {
                len_loc[mi] = (size_t)1 << (ceno_hip_mle_num_vars(PB.mles[mi]) - shq);
                off_loc[mi + 1] = off_loc[mi] + len_loc[mi];
            }
            const size_t total = off_loc[n_t];
            std::vector<E2> raw(2 * total);  // table mi: 2 len_loc entries (one fold still to do) at 2 off_loc[mi]
            std::vector<int> idx(n_t), nvs(n_t);
            std::vector<uint64_t*> outs(n_t);
            std::vector<size_t> caps(n_t);
            for (size_t mi = 0; mi < n_t; mi++) {
                idx[mi] = (int)mi;
                outs[mi] = reinterpret_cast<uint64_t*>(raw.data() + 2 * off_loc[mi]);
                caps[mi] = 2 * len_loc[mi];
            }
            rc = ceno_hip_sumcheck_tables_host(ctx, sc, (int)n_t, idx.data(), outs.data(), caps.data(), nvs.data());
            if (rc) rc = prover_set_error(rc, ceno_hip_last_error(ctx));
            for (size_t mi = 0; mi < n_t && !rc; mi++)
                if (nvs[mi] != ceno_hip_mle_num_vars(PB.mles[mi]) - shq + 1)
                    rc = prover_set_error(CENO_HIP_ERR_STATE, "sharded main constraints: unexpected table shape after the local rounds");
            std::vector<uint64_t> mine(2 * total), gathered;
            if (!rc) {
                for (size_t mi = 0; mi < n_t; mi++) {
                    const E2 scale = plan_eqg[(size_t)PB.orig[mi]];  // (a selector's rank factor went into the coefficients)
                    const E2* t = raw.data() + 2 * off_loc[mi];
                    uint64_t* dst = mine.data() + 2 * off_loc[mi];
                    for (size_t j = 0; j < len_loc[mi]; j++) {
                        const E2 v = (t[2 * j] + r_last * (t[2 * j + 1] - t[2 * j])) * scale;
                        dst[2 * j] = v.c0;
                        dst[2 * j + 1] = v.c1;
                    }
                }
                gathered.resize((size_t)W * 2 * total);
                rc = sh->allgather(sh->self, mine.data(), mine.size(), gathered.data());
            }
            if (!rc) {
                // one device block for all gathered tables (each a power of two long, tallest first is not needed: offsets are multiples of the
                // smallest table only when sorted — so every table is placed at a multiple of its own length)
                std::vector<size_t> order(n_t), off_g(n_t);
                for (size_t mi = 0; mi < n_t; mi++) order[mi] = mi;
                std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return len_loc[a] > len_loc[b]; });
                size_t tot_g = 0;
                for (size_t mi : order) {
                    off_g[mi] = tot_g;
                    tot_g += len_loc[mi] * (size_t)W;
                }
                int nv_all = 0;
                while (((size_t)1 << nv_all) < tot_g) nv_all++;
                std::vector<E2> g_all((size_t)1 << nv_all, gl::e2_zero());
                for (size_t mi = 0; mi < n_t; mi++)
                    for (int g = 0; g < W; g++) {
                        const E2* src = reinterpret_cast<const E2*>(gathered.data()) + (size_t)g * total + off_loc[mi];
                        E2* dst = g_all.data() + off_g[mi];
                        for (size_t j = 0; j < len_loc[mi]; j++) dst[(j << shk) | (size_t)g] = src[j];
                    }
                ceno_hip_mle* block = nullptr;
                rc = ceno_hip_mle_upload(ctx, reinterpret_cast<const uint64_t*>(g_all.data()), nv_all, 1, s, &block);
                if (rc) rc = prover_set_error(rc, ceno_hip_last_error(ctx));
                else {
                    owned.push_back(block);
                    for (size_t mi = 0; mi < n_t && !rc; mi++) {
                        rc = ceno_hip_mle_wrap(ctx, ceno_hip_mle_device_ptr(block) + 2 * off_g[mi], ceno_hip_mle_num_vars(PB.mles[mi]) - shq + shk, 1, &glob[mi]);
                        if (rc) rc = prover_set_error(rc, ceno_hip_last_error(ctx));
                        else owned.push_back(glob[mi]);
                    }
                }
            }
        }
        if (rc) { drop(); cleanup(); return rc; }
        ceno_hip_sumcheck_free(ctx, sc);
        sc = nullptr;
        PB.plan.term_coeffs = PB.coeffs.data();
        PB.plan.max_num_vars = max_nv - shq;
        rc = ceno_hip_sumcheck_begin(ctx, glob.data(), &PB.plan, s, &sc2);
        if (rc) { drop(); cleanup(); return prover_set_error(rc, ceno_hip_last_error(ctx)); }
        for (int i = shq; i < max_nv; i++) {
            rc = ceno_hip_sumcheck_round(ctx, sc2, i == shq ? nullptr : ch, m.data());
            if (rc) { drop(); cleanup(); return prover_set_error(rc, ceno_hip_last_error(ctx)); }
            for (int e = 0; e < D; e++) pv[(size_t)e] = E2{m[2 * e], m[2 * e + 1]};
            rc = add_small(i);
            if (rc) { drop(); cleanup(); return rc; }
            publish(i, pv.data());
        }
        std::vector<uint64_t> eb(2 * PB.mles.size()), es(2 * std::max<size_t>(PS.mles.size(), 1));
        rc = ceno_hip_sumcheck_finish(ctx, sc2, max_nv > shq ? ch : nullptr, eb.data());
        if (!rc && scs) rc = ceno_hip_sumcheck_finish(ctx, scs, ch, es.data());
        if (rc) rc = prover_set_error(rc, ceno_hip_last_error(ctx));
        for (size_t mi = 0; mi < PB.mles.size(); mi++) {
            p_evals[2 * (size_t)PB.orig[mi]] = eb[2 * mi];
            p_evals[2 * (size_t)PB.orig[mi] + 1] = eb[2 * mi + 1];
        }
        for (size_t mi = 0; scs && mi < PS.mles.size(); mi++) {
            p_evals[2 * (size_t)PS.orig[mi]] = es[2 * mi];
            p_evals[2 * (size_t)PS.orig[mi] + 1] = es[2 * mi + 1];
        }
        drop();
    }
    const double t_sc = dbg ? now_us() : 0;
    if (rc) { cleanup(); return rc; }
    // the evaluations of every table of the batch at (its prefix of) the sumcheck's point: the sumcheck's own for the tables it folded, one
    // read-only pass for the combined columns
    for (size_t j = 0; j < mles.size(); j++)
        if (plan_of[j] >= 0) {
            evals[2 * j] = p_evals[2 * (size_t)plan_of[j]];
            evals[2 * j + 1] = p_evals[2 * (size_t)plan_of[j] + 1];
        }
    if (!removed.empty()) {
        std::vector<ceno_hip_mle*> rc_cols(removed.size());
        std::vector<uint64_t> rc_out(2 * removed.size());
        for (size_t k = 0; k < removed.size(); k++) rc_cols[k] = mles[(size_t)removed[k]];
        if (!sh) {
            rc = ceno_hip_mle_evaluate_prefix_batch(ctx, (int)removed.size(), rc_cols.data(), out_global_rt, max_nv, s, rc_out.data());
            if (rc) { cleanup(); return prover_set_error(rc, ceno_hip_last_error(ctx)); }
        } else {
            // a sharded chip's column: per-rank evaluations at the point without the rank coordinates, weighted by eq over them, summed over the
            // ranks; a small chip's column is whole on every rank: evaluated at the point itself
            std::vector<size_t> kb, ks;
            for (size_t k = 0; k < removed.size(); k++) (big[(size_t)mle_job[(size_t)removed[k]]] ? kb : ks).push_back(k);
            std::vector<ceno_hip_mle*> cb(kb.size()), cs(ks.size());
            std::vector<uint64_t> ob(2 * kb.size()), os(2 * ks.size());
            for (size_t x = 0; x < kb.size(); x++) cb[x] = rc_cols[kb[x]];
            for (size_t x = 0; x < ks.size(); x++) cs[x] = rc_cols[ks[x]];
            if (!ks.empty()) {
                rc = ceno_hip_mle_evaluate_prefix_batch(ctx, (int)ks.size(), cs.data(), out_global_rt, max_nv, s, os.data());
                if (rc) { cleanup(); return prover_set_error(rc, ceno_hip_last_error(ctx)); }
                for (size_t x = 0; x < ks.size(); x++) {
                    rc_out[2 * ks[x]] = os[2 * x];
                    rc_out[2 * ks[x] + 1] = os[2 * x + 1];
                }
            }
            if (!kb.empty()) {
                const std::vector<uint64_t> rt_loc = local_point(out_global_rt, max_nv);
                rc = ceno_hip_mle_evaluate_prefix_batch(ctx, (int)kb.size(), cb.data(), rt_loc.data(), max_nv - shk, s, ob.data());
                if (rc) { cleanup(); return prover_set_error(rc, ceno_hip_last_error(ctx)); }
                const E2 w = eq_rank(out_global_rt);
                for (size_t x = 0; x < kb.size(); x++) {
                    const E2 v = E2{ob[2 * x], ob[2 * x + 1]} * w;
                    ob[2 * x] = v.c0;
                    ob[2 * x + 1] = v.c1;
                }
                std::vector<uint64_t> parts((size_t)sh->world * ob.size());
                rc = sh->allgather(sh->self, ob.data(), ob.size(), parts.data());
                if (rc) { cleanup(); return rc; }
                for (size_t x = 0; x < kb.size(); x++) {
                    E2 v = gl::e2_zero();
                    for (int g = 0; g < sh->world; g++) v = v + E2{parts[(size_t)g * ob.size() + 2 * x], parts[(size_t)g * ob.size() + 2 * x + 1]};
                    rc_out[2 * kb[x]] = v.c0;
                    rc_out[2 * kb[x] + 1] = v.c1;
                }
            }
        }
        for (size_t k = 0; k < removed.size(); k++) {
            evals[2 * (size_t)removed[k]] = rc_out[2 * k];
            evals[2 * (size_t)removed[k] + 1] = rc_out[2 * k + 1];
        }
    }
    if (dbg)
        fprintf(stderr, "[ceno_prover] batched main: selectors %.0f us, host plan + combination (%zu columns left the plan) %.0f us, sumcheck %.0f us, their evaluations %.0f us\n",
                t_sel - t_start, removed.size(), t_plan - t_sel, t_sc - t_plan, now_us() - t_sc);
    // ---- final claim by the front-load rule and the claimed sum recovered backwards (cpu/mod.rs:1338-1360,1393-1413) ----
    E2 final_claim = gl::e2_zero();
    for (int t = 0; t < n_terms_all; t++) {
        E2 v{coeffs[2 * t], coeffs[2 * t + 1]};
        for (uint32_t k = toff[t]; k < toff[t + 1]; k++) {
            const int j = (int)tidx[k];
            v = v * E2{evals[2 * j], evals[2 * j + 1]};
            for (int i = mle_nv[j]; i < max_nv; i++) v = v * E2{out_global_rt[2 * i], out_global_rt[2 * i + 1]};
        }
        final_claim = final_claim + v;
    }
    E2 expected = final_claim;
    for (int round = max_nv - 1; round >= 0; round--) {
        std::vector<E2> ev(max_deg), zeros(max_deg, gl::e2_zero());
        for (int t = 0; t < max_deg; t++) ev[t] = E2{out_msgs[2 * ((size_t)round * max_deg + t)], out_msgs[2 * ((size_t)round * max_deg + t) + 1]};
        const E2 r{out_global_rt[2 * round], out_global_rt[2 * round + 1]};
        const E2 hidden = extrapolate(gl::e2_one(), zeros, r);
        const E2 without = extrapolate(gl::e2_neg(ev[0]), ev, r);
        expected = (expected - without) * gl::e2_inv(hidden);
    }
    out_claimed_sum[0] = expected.c0;
    out_claimed_sum[1] = expected.c1;
    for (size_t j = 0; j < mles.size(); j++) tr->append_ext(tr->self, evals.data() + 2 * j);                      // cpu/mod.rs:1361
    memcpy(out_evals, evals.data(), evals.size() * 8);
    if (out_num_vars) *out_num_vars = max_nv;
    if (out_degree) *out_degree = max_deg;
    cleanup();
    return 0;
}
