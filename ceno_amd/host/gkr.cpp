// Multi-layer GKR circuit prover — host control flow of `GKRCircuit::prove` (gkr_iop/src/gkr.rs:72-115) and
// `Layer::prove` (gkr_iop/src/gkr/layer.rs:198-243, claims plumbing :289-322) over the device C ABI; the three layer provers
// follow gkr_iop/src/gkr/layer/cpu/mod.rs:47-66 (linear), :72-96 (sumcheck), :102-238 (zerocheck).  See include/ceno_prover.h.
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ceno_prover.h"
#include "../csrc/gl64.hpp"

using gl::E2;

int prover_set_error(int code, const char* msg);  // prover.cpp

namespace {

struct Claim {
    std::vector<E2> point;
    E2 eval = gl::e2_zero();
    bool set = false;
};

// EvalExpression::evaluate (gkr_iop/src/evaluation.rs:41-93)
bool eval_expr(const ceno_eval_expr& e, const std::vector<Claim>& claims, Claim& out) {
    switch (e.kind) {
    case CENO_EVAL_ZERO:
        out.point = claims[0].point;  // "for zero eval ... pick first point as representative"
        out.eval = gl::e2_zero();
        out.set = claims[0].set;
        return true;
    case CENO_EVAL_SINGLE:
        if (e.idx < 0 || e.idx >= (int)claims.size()) return false;
        out = claims[e.idx];
        return true;
    case CENO_EVAL_LINEAR:
        if (e.idx < 0 || e.idx >= (int)claims.size()) return false;
        out.point = claims[e.idx].point;
        out.eval = claims[e.idx].eval * E2{e.c0[0], e.c0[1]} + E2{e.c1[0], e.c1[1]};
        out.set = claims[e.idx].set;
        return true;
    default: return false;
    }
}

}  // namespace

extern "C" int ceno_prover_gkr_prove(ceno_hip_ctx* ctx, const ceno_gkr_layer* layers, int n_layers, int max_num_vars, int n_evaluations,
                                     const uint64_t* const* claim_points, const int* claim_point_len, const uint64_t* claim_evals,
                                     const uint64_t* pub_io, int n_pub_io, const uint64_t* gc4, ceno_transcript* tr, ceno_hip_stream s,
                                     uint64_t* const* out_msgs, uint64_t* const* out_evals, uint64_t* const* out_points, uint64_t* out_claim_points,
                                     int* out_claim_point_len, uint64_t* out_claim_evals) {
    if (!ctx || !layers || n_layers < 1 || n_evaluations < 1 || !claim_points || !claim_point_len || !claim_evals || !gc4 || !tr || !out_evals || !out_points)
        return prover_set_error(CENO_HIP_ERR_INVALID, "gkr_prove: bad arguments");
    // running evals: "a global referable within chip" (gkr.rs:83-85)
    std::vector<Claim> claims(n_evaluations);
    for (int i = 0; i < n_evaluations; i++) {
        if (claim_point_len[i] > 0) {
            claims[i].set = true;
            claims[i].point.resize(claim_point_len[i]);
            for (int k = 0; k < claim_point_len[i]; k++) claims[i].point[k] = E2{claim_points[i][2 * k], claim_points[i][2 * k + 1]};
        }
        claims[i].eval = E2{claim_evals[2 * i], claim_evals[2 * i + 1]};
    }
    for (int l = 0; l < n_layers; l++) {
        const ceno_gkr_layer& L = layers[l];
        const int n_mles = L.n_witin + L.n_fixed + L.n_structural;
        if (n_mles < 1 || !L.mles || L.n_groups < 1 || !L.group_expr_offsets || !L.out_exprs)
            return prover_set_error(CENO_HIP_ERR_INVALID, "gkr_prove: bad layer");
        // ---- extract_claim_and_point (layer.rs:289-313): one point per group = the point of its first expression ----
        std::vector<std::vector<uint64_t>> group_points(L.n_groups);
        for (int g = 0; g < L.n_groups; g++) {
            if (L.group_expr_offsets[g + 1] == L.group_expr_offsets[g]) continue;
            Claim c;
            if (!eval_expr(L.out_exprs[L.group_expr_offsets[g]], claims, c)) return prover_set_error(CENO_HIP_ERR_INVALID, "gkr_prove: unsupported out-evaluation expression");
            if (!c.set) return prover_set_error(CENO_HIP_ERR_STATE, "gkr_prove: a layer reads a claim no earlier layer produced");
            for (const E2& x : c.point) {
                group_points[g].push_back(x.c0);
                group_points[g].push_back(x.c1);
            }
        }
        std::vector<uint64_t> evals((size_t)2 * n_mles), point;
        if (L.type == CENO_LAYER_LINEAR) {
            // LinearLayerProver::prove: the witness evaluations at the out point are the proof (layer/cpu/mod.rs:47-66)
            if (L.n_groups != 1) return prover_set_error(CENO_HIP_ERR_INVALID, "gkr_prove: a linear layer has exactly one out point (layer.rs:226)");
            point = group_points[0];
            if ((int)point.size() != 2 * L.num_vars) return prover_set_error(CENO_HIP_ERR_INVALID, "gkr_prove: linear layer point / table size mismatch");
            for (int j = 0; j < n_mles; j++) {
                int rc = ceno_hip_mle_evaluate(ctx, L.mles[j], point.data(), evals.data() + 2 * j, s);
                if (rc) return prover_set_error(rc, ceno_hip_last_error(ctx));
            }
            for (int j = 0; j < n_mles; j++) tr->append_ext(tr->self, evals.data() + 2 * j);
        } else if (L.type == CENO_LAYER_ZEROCHECK) {
            // ZerocheckLayerProver::prove = the batched main-constraint prover with ONE job: alpha powers under the label
            // "combine subset evals", selector eq tables at the group points, the sumcheck, the final evaluations appended
            std::vector<int> sel_kind, sel_id, sel_n_sparse, sel_sparse_nv;
            std::vector<size_t> sel_off, sel_n;
            std::vector<const uint32_t*> sel_sparse;
            std::vector<const uint64_t*> sel_pts;
            for (int g = 0; g < L.n_groups; g++) {
                if (!L.group_sel_kind || L.group_sel_kind[g] < 0) continue;  // SelectorType::None
                if ((int)group_points[g].size() != 2 * L.num_vars) return prover_set_error(CENO_HIP_ERR_INVALID, "gkr_prove: out point / layer size mismatch");
                sel_kind.push_back(L.group_sel_kind[g]);
                sel_id.push_back(L.group_sel_structural_id[g]);
                sel_off.push_back(L.group_sel_offset ? L.group_sel_offset[g] : 0);
                sel_n.push_back(L.group_sel_num_instances ? L.group_sel_num_instances[g] : 0);
                sel_sparse.push_back(L.group_sel_sparse_indices ? L.group_sel_sparse_indices[g] : nullptr);
                sel_n_sparse.push_back(L.group_sel_n_sparse ? L.group_sel_n_sparse[g] : 0);
                sel_sparse_nv.push_back(L.group_sel_sparse_num_vars ? L.group_sel_sparse_num_vars[g] : 0);
                sel_pts.push_back(group_points[g].data());
            }
            ceno_main_job J{};
            J.circuit_idx = l;
            J.num_vars = L.num_vars;
            J.n_witin = L.n_witin;
            J.n_fixed = L.n_fixed;
            J.n_structural = L.n_structural;
            J.mles = L.mles;
            J.n_selectors = (int)sel_kind.size();
            J.sel_kind = sel_kind.data();
            J.sel_offset = sel_off.data();
            J.sel_num_instances = sel_n.data();
            J.sel_structural_id = sel_id.data();
            J.sel_sparse_indices = sel_sparse.data();
            J.sel_n_sparse = sel_n_sparse.data();
            J.sel_sparse_num_vars = sel_sparse_nv.data();
            J.sel_points = sel_pts.data();
            J.n_exprs = L.n_exprs;
            J.max_degree = L.max_degree;
            J.n_terms = L.n_terms;
            J.term_offsets = L.term_offsets;
            J.term_mle_idx = L.term_mle_idx;
            J.scalar_offsets = L.scalar_offsets;
            J.mono_coeffs = L.mono_coeffs;
            J.mono_chal_offsets = L.mono_chal_offsets;
            J.mono_chal_idx = L.mono_chal_idx;
            J.n_pi = n_pub_io;
            J.pi = pub_io;
            uint64_t claimed[2];
            point.assign((size_t)2 * L.num_vars, 0);
            std::vector<uint64_t> msgs((size_t)2 * L.num_vars * L.max_degree);
            int nv_o = 0, d_o = 0;
            int rc = ceno_prover_prove_batched_main_constraints(ctx, &J, 1, gc4, tr, s, claimed, out_msgs && out_msgs[l] ? out_msgs[l] : msgs.data(), point.data(),
                                                                evals.data(), &nv_o, &d_o);
            if (rc) return rc;
        } else if (L.type == CENO_LAYER_SUMCHECK) {
            // SumcheckLayerProver::prove: plain sumcheck of the layer expression over the layer witness (layer/cpu/mod.rs:72-96);
            // scalars are polynomials in [challenges, pub_io]
            std::vector<E2> ch{E2{gc4[0], gc4[1]}, E2{gc4[2], gc4[3]}};
            for (int i = 0; i < n_pub_io && pub_io; i++) ch.push_back(E2{pub_io[2 * i], pub_io[2 * i + 1]});
            std::vector<uint64_t> coeffs;
            std::vector<uint32_t> toff{0}, tidx;
            for (int t = 0; t < L.n_terms; t++) {
                E2 sc = gl::e2_zero();
                for (uint32_t m = L.scalar_offsets[t]; m < L.scalar_offsets[t + 1]; m++) {
                    E2 v{L.mono_coeffs[2 * m], L.mono_coeffs[2 * m + 1]};
                    for (uint32_t k = L.mono_chal_offsets[m]; k < L.mono_chal_offsets[m + 1]; k++) {
                        if (L.mono_chal_idx[k] >= ch.size()) return prover_set_error(CENO_HIP_ERR_INVALID, "gkr_prove: challenge id out of range");
                        v = v * ch[L.mono_chal_idx[k]];
                    }
                    sc = sc + v;
                }
                if (sc.c0 == 0 && sc.c1 == 0) continue;
                coeffs.push_back(sc.c0);
                coeffs.push_back(sc.c1);
                for (uint32_t k = L.term_offsets[t]; k < L.term_offsets[t + 1]; k++) tidx.push_back(L.term_mle_idx[k]);
                toff.push_back((uint32_t)tidx.size());
            }
            ceno_hip_sumcheck_plan plan{};
            plan.num_mles = n_mles;
            plan.num_terms = (int)toff.size() - 1;
            plan.term_coeffs = coeffs.data();
            plan.term_offsets = toff.data();
            plan.term_mle_idx = tidx.data();
            plan.max_num_vars = L.num_vars;
            plan.max_degree = L.max_degree;
            point.assign((size_t)2 * L.num_vars, 0);
            std::vector<uint64_t> msgs((size_t)2 * L.num_vars * L.max_degree);
            int rc = ceno_prover_sumcheck_prove(ctx, L.mles, &plan, tr, s, out_msgs && out_msgs[l] ? out_msgs[l] : msgs.data(), point.data(), evals.data());
            if (rc) return rc;
            for (int j = 0; j < n_mles; j++) tr->append_ext(tr->self, evals.data() + 2 * j);
        } else {
            return prover_set_error(CENO_HIP_ERR_INVALID, "gkr_prove: unknown layer type");
        }
        memcpy(out_evals[l], evals.data(), evals.size() * 8);
        memcpy(out_points[l], point.data(), point.size() * 8);
        // ---- update_claims (layer.rs:315-322): evals zipped with in_eval_expr ----
        for (int k = 0; k < L.n_in_evals && k < n_mles; k++) {
            const int pos = L.in_eval_pos[k];
            if (pos < 0 || pos >= n_evaluations) return prover_set_error(CENO_HIP_ERR_INVALID, "gkr_prove: in_eval position out of range");
            claims[pos].set = true;
            claims[pos].point.resize(point.size() / 2);
            for (size_t i = 0; i < point.size() / 2; i++) claims[pos].point[i] = E2{point[2 * i], point[2 * i + 1]};
            claims[pos].eval = E2{evals[2 * k], evals[2 * k + 1]};
        }
    }
    if (out_claim_points && out_claim_point_len && out_claim_evals) {
        for (int i = 0; i < n_evaluations; i++) {
            out_claim_point_len[i] = claims[i].set ? (int)claims[i].point.size() : 0;
            for (size_t k = 0; k < claims[i].point.size() && (int)k < max_num_vars; k++) {
                out_claim_points[((size_t)i * max_num_vars + k) * 2] = claims[i].point[k].c0;
                out_claim_points[((size_t)i * max_num_vars + k) * 2 + 1] = claims[i].point[k].c1;
            }
            out_claim_evals[2 * i] = claims[i].eval.c0;
            out_claim_evals[2 * i + 1] = claims[i].eval.c1;
        }
    }
    return 0;
}
