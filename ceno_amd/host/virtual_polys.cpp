// VirtualPolynomialsBuilder — host-side plan builder in front of the sumcheck operator.
//
// Reference: EXT `multilinear_extensions::virtual_polys::VirtualPolynomialsBuilder::{new_with_mles, lift,
// to_virtual_polys_with_monomial_terms}` and `Term{scalar, product}`; call sites gkr_iop/src/gkr/layer/cpu/mod.rs:80-88,
// 213-226, 295-321 and ceno_zkvm/src/scheme/cpu/mod.rs:98,135-137,413-418,1255-1334.  There the builder registers MLEs
// (Left = borrowed, Right = owned), hands out expression ids, and finally turns a list of monomial terms
// (scalar x product of registered MLEs) into the VirtualPolynomials that IOPProverState::prove consumes.
// Here it produces the CSR `ceno_hip_sumcheck_plan` of the C ABI; "owned" MLEs are freed with the builder.
#include <cstring>
#include <map>
#include <vector>

#include "../../include/ceno_prover.h"

int prover_set_error(int code, const char* msg);  // prover.cpp

struct ceno_vp_builder {
    int max_num_vars = 0;
    std::vector<ceno_hip_mle*> mles;
    std::vector<char> owned;
    std::map<ceno_hip_mle*, int> index;  // lift() of an already registered MLE returns the same id
    std::vector<uint64_t> coeffs;
    std::vector<uint32_t> term_off{0}, term_idx;
    int max_degree = 0;
};

extern "C" {

ceno_vp_builder* ceno_vp_builder_new(int max_num_vars) {
    auto* b = new ceno_vp_builder();
    b->max_num_vars = max_num_vars;
    return b;
}

void ceno_vp_builder_free(ceno_hip_ctx* ctx, ceno_vp_builder* b) {
    if (!b) return;
    for (size_t i = 0; i < b->mles.size(); i++)
        if (b->owned[i]) ceno_hip_mle_free(ctx, b->mles[i]);
    delete b;
}

int ceno_vp_builder_lift(ceno_vp_builder* b, ceno_hip_mle* mle, int take_ownership) {
    if (!b || !mle) return prover_set_error(CENO_HIP_ERR_INVALID, "vp_builder_lift: NULL argument");
    if (ceno_hip_mle_num_vars(mle) > b->max_num_vars)
        return prover_set_error(CENO_HIP_ERR_INVALID, "vp_builder_lift: MLE has more variables than the builder's max_num_vars");
    auto it = b->index.find(mle);
    if (it != b->index.end()) {
        if (take_ownership) b->owned[it->second] = 1;
        return it->second;
    }
    const int id = (int)b->mles.size();
    b->mles.push_back(mle);
    b->owned.push_back(take_ownership ? 1 : 0);
    b->index[mle] = id;
    return id;
}

int ceno_vp_builder_add_term(ceno_vp_builder* b, const uint64_t* scalar2, const int* product, int n) {
    if (!b || !scalar2 || !product || n < 1) return prover_set_error(CENO_HIP_ERR_INVALID, "vp_builder_add_term: bad argument");
    int nv = -1;
    for (int i = 0; i < n; i++) {
        if (product[i] < 0 || product[i] >= (int)b->mles.size()) return prover_set_error(CENO_HIP_ERR_INVALID, "vp_builder_add_term: unknown expression id");
        const int v = ceno_hip_mle_num_vars(b->mles[product[i]]);
        if (nv >= 0 && v != nv)  // same rule the reference's GPU arm asserts (gkr_iop/src/gkr/layer/gpu/utils.rs:54-63)
            return prover_set_error(CENO_HIP_ERR_INVALID, "vp_builder_add_term: factors of one term must have the same num_vars");
        nv = v;
        b->term_idx.push_back((uint32_t)product[i]);
    }
    b->term_off.push_back((uint32_t)b->term_idx.size());
    b->coeffs.push_back(scalar2[0]);
    b->coeffs.push_back(scalar2[1]);
    if (n > b->max_degree) b->max_degree = n;
    return (int)b->term_off.size() - 2;
}

int ceno_vp_builder_num_mles(const ceno_vp_builder* b) { return b ? (int)b->mles.size() : -1; }
int ceno_vp_builder_max_degree(const ceno_vp_builder* b) { return b ? b->max_degree : -1; }

int ceno_vp_builder_prove(ceno_hip_ctx* ctx, ceno_vp_builder* b, int max_degree, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_msgs,
                          uint64_t* out_challenges, uint64_t* out_final_evals) {
    if (!ctx || !b || !tr) return prover_set_error(CENO_HIP_ERR_INVALID, "vp_builder_prove: NULL argument");
    if (b->coeffs.empty()) return prover_set_error(CENO_HIP_ERR_INVALID, "vp_builder_prove: no terms");
    ceno_hip_sumcheck_plan plan;
    memset(&plan, 0, sizeof(plan));
    plan.num_mles = (int)b->mles.size();
    plan.num_terms = (int)b->term_off.size() - 1;
    plan.term_coeffs = b->coeffs.data();
    plan.term_offsets = b->term_off.data();
    plan.term_mle_idx = b->term_idx.data();
    plan.max_num_vars = b->max_num_vars;
    plan.max_degree = max_degree > 0 ? max_degree : b->max_degree;
    return ceno_prover_sumcheck_prove(ctx, b->mles.data(), &plan, tr, s, out_msgs, out_challenges, out_final_evals);
}

}  // extern "C"
