// Record inference shared between witinfer.hip (records materialised) and tower.hip (towers built straight from the record expressions):
// the plan of `wit_infer_by_monomial_expr` (EXT; gkr_iop/src/cpu/mod.rs:119-176) staged in LDS and one record evaluated at one row.
#pragma once
#include "common.hpp"

struct WiSlot {
    const uint64_t* ptr;
    int is_ext;
    int pad;
};

struct WiPlan {
    const WiSlot* mles;
    const E2* coeffs;
    const uint32_t* term_off;
    const uint32_t* term_idx;
    const uint32_t* out_term_off;
    E2* const* outs;
    int num_outs;
};


struct WiLds {
    WiSlot* slots;
    gl::E2* coeffs;
    gl::E2** outs;
    uint32_t *toff, *tidx, *ooff;
};
// the plan into LDS (every lane walks the same records: from global memory that walk is a chain of dependent loads); ends with a barrier
template <int NT_>
__device__ __forceinline__ WiLds wi_stage(char* dyn, const WiPlan& pl, int num_mles, int num_terms, int num_factors, bool with_outs) {
    WiLds L;
    L.slots = reinterpret_cast<WiSlot*>(dyn);
    L.coeffs = reinterpret_cast<gl::E2*>(L.slots + num_mles);
    L.outs = reinterpret_cast<gl::E2**>(L.coeffs + num_terms);
    L.toff = reinterpret_cast<uint32_t*>(L.outs + pl.num_outs);
    L.tidx = L.toff + num_terms + 1;
    L.ooff = L.tidx + num_factors;
    for (int i = threadIdx.x; i < num_mles; i += NT_) L.slots[i] = pl.mles[i];
    for (int i = threadIdx.x; i < num_terms; i += NT_) L.coeffs[i] = pl.coeffs[i];
    if (with_outs)
        for (int i = threadIdx.x; i < pl.num_outs; i += NT_) L.outs[i] = pl.outs[i];
    for (int i = threadIdx.x; i <= num_terms; i += NT_) L.toff[i] = pl.term_off[i];
    for (int i = threadIdx.x; i < num_factors; i += NT_) L.tidx[i] = pl.term_idx[i];
    for (int i = threadIdx.x; i <= pl.num_outs; i += NT_) L.ooff[i] = pl.out_term_off[i];
    __syncthreads();
    return L;
}
// record o at row x: sum_t c_t prod_j f_j[x]; the base-field factors of a term are multiplied together first and meet the extension-field
// coefficient once at the end
__device__ __forceinline__ gl::E2 wi_eval(const WiLds& L, int o, size_t x) {
    using namespace gl;
    E2 acc = e2_zero();
    for (uint32_t t = L.ooff[o]; t < L.ooff[o + 1]; t++) {
        E2 v = L.coeffs[t];
        uint64_t pb = 1;
        bool any_base = false;
        for (uint32_t k = L.toff[t]; k < L.toff[t + 1]; k++) {
            const WiSlot sl = L.slots[L.tidx[k]];
            if (sl.is_ext) {
                v = v * reinterpret_cast<const E2*>(sl.ptr)[x];
            } else {
                const uint64_t f = sl.ptr[x];
                pb = any_base ? mul(pb, f) : f;
                any_base = true;
            }
        }
        if (any_base) v = e2_mul_base(v, pb);
        acc = acc + v;
    }
    return acc;
}
inline size_t wit_infer_lds(int num_mles, int num_terms, int num_factors, int num_outs) {
    return (size_t)num_mles * sizeof(WiSlot) + (size_t)num_terms * sizeof(gl::E2) + (size_t)num_outs * sizeof(gl::E2*) +
           ((size_t)num_terms + 1 + (size_t)num_factors + (size_t)num_outs + 1) * 4 + 16;
}
