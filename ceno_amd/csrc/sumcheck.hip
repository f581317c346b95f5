// Sumcheck prover rounds on gfx950.
//
// Reference operator: `prove_generic_sumcheck_gpu(mles, mle_size_info, term_coefficients,
// mle_indices_per_term, max_num_var, max_degree, Option<&CommonTermPlan>, &mut transcript, stream)`
// (call sites gkr_iop/src/gkr/layer/gpu/mod.rs:259-271, ceno_zkvm/src/scheme/gpu/mod.rs:891-902,
// 2968-2982), i.e. EXT `IOPProverState::prove` (gkr_iop/src/gkr/layer/cpu/mod.rs:217-227).
// Semantics: round i binds variable i (LSB first), the message is p(1..d),
// MLEs with fewer variables are front-loaded (scheme/verifier.rs:233-237).
//
// Schedule.  Round 0 is one read-only pass.  Round i>0 is ONE fused pass per size class: read the
// table of round i-1, fold it with r_{i-1} in registers, write the half-size table and accumulate the
// message of round i on the folded values.  Every table is therefore read once and written once (half
// size) per round: HBM-streaming, 16 B per lane, no reuse -> HBM-roofline bound; no MFMA.
//   * dense class  (one product term of K<=4 tables): register-resident fused kernel `k_dense`.
//   * generic class (CSR term plan with optional common-factor groups): batched fold kernel + plan-
//     driven accumulate kernel (`k_fold_batch`, `k_accum`), factor re-reads served by L1/L2.
// Per-block partial sums (wave64 shuffle -> LDS) are combined inside the same launch by the last block to
// arrive (`epilogue`), which applies class coefficients and the host-computed front-load scalars and
// writes the d extension elements of the message straight into pinned host memory + a sequence flag the
// host spins on (or into a caller device buffer): one launch and no D2H copy per round.
#include "sumcheck_dev.hpp"
#include "sumcheck_dense.hpp"
#include "sumcheck_gen.hpp"
#include "sumcheck_tower.hpp"
#include "e2_host_avx512.hpp"
#include "sumcheck_small.hpp"

#include <algorithm>
#include <atomic>
#include <functional>
#include <numeric>
#include <ctime>

// ------------------------------------------------------------------------------------------------
// generic path
// ------------------------------------------------------------------------------------------------
// fold every MLE of a class: blockIdx.y = MLE
__global__ void __launch_bounds__(NT) k_fold_batch(const MleSlot* __restrict__ slots, size_t half, E2 r, const Bcast* bcast,
                                                    unsigned long long wait_seq) {
    if (wait_seq != 0) {  // pipelined: the challenge was relayed by the previous launch
        __shared__ unsigned long long s_c[3];
        if (threadIdx.x == 0) {
            const volatile Bcast* bc = bcast;
            s_c[2] = bc->ready_seq == (unsigned)wait_seq;
            s_c[0] = bc->chal[0];
            s_c[1] = bc->chal[1];
        }
        __syncthreads();
        if (s_c[2] == 0) return;
        r = E2{s_c[0], s_c[1]};
    }
    const MleSlot sl = slots[blockIdx.y];
    const size_t stride = (size_t)gridDim.x * NT;
    E2* out = reinterpret_cast<E2*>(sl.out);
    const E2Pre rp = e2_pre(r);
    if (sl.in_ext) {
        const E2* in = reinterpret_cast<const E2*>(sl.in);
        for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < half; j += stride) {
            E2 lo = in[2 * j], hi = in[2 * j + 1];
            out[j] = lo + e2_mul_pre(rp, hi - lo);
        }
    } else {
        for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < half; j += stride) {
            ulonglong2 v = *reinterpret_cast<const ulonglong2*>(sl.in + 2 * j);
            E2 t = e2_mul_base(r, sub(v.y, v.x));
            out[j] = E2{add(t.c0, v.x), t.c1};
        }
    }
}

// evaluations at points 1..D of one factor's pair, multiplied into the running products of a term / a group:
// base-field tables (round 0 reads the witness columns as they are) stay in the base field — one 64-bit multiply per
// point instead of an extension multiply — and are folded into the extension product once per term.
// `seed` (optional): the term's coefficient, folded into the FIRST extension factor as c*f(X) = c*hi + (X-1) * c*delta —
// two multiplications instead of one per evaluation point (used when D >= 3).
// ALLEXT: every table read is an extension table (all rounds after the first): the base-field arm is compiled out.
// `flags`: bit 0 = a base-field product is running in pb, bit 1 = an extension-field product is running in pe.  ONE word by
// reference: two separate bool references ended up in an 8-byte scratch frame (the compiler selected between their addresses:
// `scratch_store_byte off, v1, s4`, 13-14 scratch instructions per instantiation of k_accum).
static constexpr unsigned MF_BASE = 1u, MF_EXT = 2u;
template <int D, bool ALLEXT = false>
__device__ __forceinline__ void mul_factor(const MleSlot& sl, int use_out, size_t p, E2 (&pe)[D], uint64_t (&pb)[D], unsigned& flags,
                                           const E2* seed = nullptr) {
    const bool has_b = flags & MF_BASE, has_e = flags & MF_EXT;
    if (!ALLEXT && !use_out && !sl.in_ext) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(sl.in + 2 * p);
        const uint64_t delta = sub(v.y, v.x);
        uint64_t x = v.y;
        if (has_b) {
#pragma unroll
            for (int t = 0; t < D; t++) {
                pb[t] = mul_nc(pb[t], x);  // only multiplied again
                x = add(x, delta);
            }
        } else {
#pragma unroll
            for (int t = 0; t < D; t++) {
                pb[t] = x;
                x = add(x, delta);
            }
        }
        flags |= MF_BASE;
    } else {
        E2 lo, hi;
        if (ALLEXT) {
            const E2* q = reinterpret_cast<const E2*>(sl.out) + 2 * p;
            lo = q[0];
            hi = q[1];
        } else {
            load_pair(sl, use_out, p, lo, hi);
        }
        E2 delta = hi - lo;
        E2 x = hi;
        if (seed && !has_e) {
            x = *seed * x;
            delta = *seed * delta;
        }
        if (has_e) {
#pragma unroll
            for (int t = 0; t < D; t++) {
                pe[t] = e2_mul_nc(pe[t], x);  // only multiplied again
                x = x + delta;
            }
        } else {
#pragma unroll
            for (int t = 0; t < D; t++) {
                pe[t] = x;
                x = x + delta;
            }
        }
        flags |= MF_EXT;
    }
}

template <int D, bool LAZY>
__global__ void __launch_bounds__(NT) k_accum(DevPlan pl, size_t pairs, Epilogue ep) {
    __shared__ E2 smem[(NT / 64) * D];
    if (ep.wait_seq != 0) {  // pipelined: do nothing (and publish nothing) if the fold before us was aborted
        __shared__ unsigned long long s_c[3];
        E2 unused;
        if (!read_challenge(ep, unused, s_c)) return;
    }
    E2 acc[D];
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t p = (size_t)blockIdx.x * NT + threadIdx.x; p < pairs; p += stride) {
        if constexpr (LAZY) {
            for (int g = 0; g < pl.n_groups; g++) {
                // sum_t c_t * P_t(X): the factors of a term are multiplied first (one multiply fewer than seeding the product
                // with the coefficient), then c_t * P_t goes UNREDUCED into 160-bit accumulators, reduced once per group
                E2Acc wacc[D];
#pragma unroll
                for (int t = 0; t < D; t++) wacc[t] = e2acc_zero();
                for (uint32_t ti = pl.group_term_off[g]; ti < pl.group_term_off[g + 1]; ti++) {
                    const uint32_t term = pl.group_terms[ti];
                    const E2 c = pl.coeffs[term];
                    E2 pr[D];
                    uint64_t pb[D];
                    unsigned fl = 0;
                    for (uint32_t k = pl.term_off[term]; k < pl.term_off[term + 1]; k++)
                        mul_factor<D>(pl.slots[pl.term_idx[k]], pl.use_out, p, pr, pb, fl);
                    const bool has_e = fl & MF_EXT, has_b = fl & MF_BASE;
                    if (has_e) {
                        if (has_b) {
#pragma unroll
                            for (int t = 0; t < D; t++) pr[t] = e2_mul_base(pr[t], pb[t]);
                        }
#pragma unroll
                        for (int t = 0; t < D; t++) e2acc_mac(wacc[t], c, pr[t]);
                    } else if (has_b) {
#pragma unroll
                        for (int t = 0; t < D; t++) {
                            acc5_add(wacc[t].s00, mul_wide(c.c0, pb[t]));
                            acc5_add(wacc[t].s01, mul_wide(c.c1, pb[t]));
                        }
                    } else {  // coefficient times the group's common factors only
#pragma unroll
                        for (int t = 0; t < D; t++) e2acc_mac(wacc[t], c, e2_one());
                    }
                }
                E2 inner[D];
#pragma unroll
                for (int t = 0; t < D; t++) inner[t] = e2acc_reduce(wacc[t]);
                const uint32_t cb = pl.common_off[g], ce = pl.common_off[g + 1];
                if (ce > cb) {
                    E2 cm[D];
                    uint64_t cmb[D];
                    unsigned fl = 0;
                    for (uint32_t k = cb; k < ce; k++) mul_factor<D>(pl.slots[pl.common_idx[k]], pl.use_out, p, cm, cmb, fl);
                    const bool has_e = fl & MF_EXT, has_b = fl & MF_BASE;
#pragma unroll
                    for (int t = 0; t < D; t++) {
                        E2 v = inner[t];
                        if (has_e) v = cm[t] * v;
                        if (has_b) v = e2_mul_base(v, cmb[t]);
                        acc[t] = acc[t] + v;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < D; t++) acc[t] = acc[t] + inner[t];
                }
            }
        } else {
            for (int g = 0; g < pl.n_groups; g++) {
                E2 inner[D];
#pragma unroll
                for (int t = 0; t < D; t++) inner[t] = e2_zero();
                for (uint32_t ti = pl.group_term_off[g]; ti < pl.group_term_off[g + 1]; ti++) {
                    const uint32_t term = pl.group_terms[ti];
                    const E2 c = pl.coeffs[term];
                    E2 pr[D];
                    uint64_t pb[D];
                    // the coefficient seeds the extension product: as the initial value (one multiply per point with the
                    // first factor) or, from 3 evaluation points on, folded into the first extension factor (two multiplies)
                    constexpr bool FOLD_SEED = D >= 3;
#pragma unroll
                    for (int t = 0; t < D; t++) pr[t] = c;
                    unsigned fl = FOLD_SEED ? 0u : MF_EXT;
                    for (uint32_t k = pl.term_off[term]; k < pl.term_off[term + 1]; k++)
                        mul_factor<D>(pl.slots[pl.term_idx[k]], pl.use_out, p, pr, pb, fl, FOLD_SEED ? &c : nullptr);
                    // no extension factor at all: pr still holds the coefficient
                    if (fl & MF_BASE) {
#pragma unroll
                        for (int t = 0; t < D; t++) pr[t] = e2_mul_base(pr[t], pb[t]);
                    }
#pragma unroll
                    for (int t = 0; t < D; t++) inner[t] = inner[t] + pr[t];
                }
                const uint32_t cb = pl.common_off[g], ce = pl.common_off[g + 1];
                if (ce > cb) {
                    E2 cm[D];
                    uint64_t cmb[D];
                    unsigned fl = 0;
                    for (uint32_t k = cb; k < ce; k++) mul_factor<D>(pl.slots[pl.common_idx[k]], pl.use_out, p, cm, cmb, fl);
                    const bool has_e = fl & MF_EXT, has_b = fl & MF_BASE;
#pragma unroll
                    for (int t = 0; t < D; t++) {
                        E2 v = inner[t];
                        if (has_e) v = cm[t] * v;
                        if (has_b) v = e2_mul_base(v, cmb[t]);
                        acc[t] = acc[t] + v;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < D; t++) acc[t] = acc[t] + inner[t];
                }
            }
        }
    }
    __shared__ int s_flag;
    epilogue<D, NT>(acc, ep, smem, &s_flag);
}

// Rounds after the first read the folded tables, which are all extension tables: same plan walk as k_accum without the
// base-field arm and its registers (the generic kernel keeps base products, flags and both load forms alive).
template <int D>
__global__ void __launch_bounds__(NT) k_accum_ext(DevPlan pl, size_t pairs, Epilogue ep) {
    __shared__ E2 smem[(NT / 64) * D];
    if (ep.wait_seq != 0) {
        __shared__ unsigned long long s_c[3];
        E2 unused;
        if (!read_challenge(ep, unused, s_c)) return;
    }
    E2 acc[D];
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t p = (size_t)blockIdx.x * NT + threadIdx.x; p < pairs; p += stride) {
        for (int g = 0; g < pl.n_groups; g++) {
            E2 inner[D];
#pragma unroll
            for (int t = 0; t < D; t++) inner[t] = e2_zero();
            for (uint32_t ti = pl.group_term_off[g]; ti < pl.group_term_off[g + 1]; ti++) {
                const uint32_t term = pl.group_terms[ti];
                const E2 c = pl.coeffs[term];
                E2 pr[D];
#pragma unroll
                for (int t = 0; t < D; t++) pr[t] = c;
                bool seeded = false;
                // the pair of factor k+1 is requested before factor k is multiplied in (one L2 round trip hidden per factor)
                uint32_t k = pl.term_off[term];
                const uint32_t ke = pl.term_off[term + 1];
                E2 lo = e2_zero(), hi = e2_zero();
                if (k < ke) {
                    const E2* q = reinterpret_cast<const E2*>(pl.slots[pl.term_idx[k]].out) + 2 * p;
                    lo = q[0];
                    hi = q[1];
                }
                for (; k < ke; k++) {
                    E2 nlo = lo, nhi = hi;
                    if (k + 1 < ke) {
                        const E2* q = reinterpret_cast<const E2*>(pl.slots[pl.term_idx[k + 1]].out) + 2 * p;
                        nlo = q[0];
                        nhi = q[1];
                    }
                    mul_points<D>(pr, seeded, c, hi, hi - lo);
                    lo = nlo;
                    hi = nhi;
                }
#pragma unroll
                for (int t = 0; t < D; t++) inner[t] = inner[t] + pr[t];
            }
            const uint32_t cb = pl.common_off[g], ce = pl.common_off[g + 1];
            for (uint32_t k = cb; k < ce; k++) {
                const E2* q = reinterpret_cast<const E2*>(pl.slots[pl.common_idx[k]].out) + 2 * p;
                const E2 lo = q[0], hi = q[1], delta = hi - lo;
                E2 x = hi;
#pragma unroll
                for (int t = 0; t < D; t++) {
                    inner[t] = inner[t] * x;
                    x = x + delta;
                }
            }
#pragma unroll
            for (int t = 0; t < D; t++) acc[t] = acc[t] + inner[t];
        }
    }
    __shared__ int s_flag;
    epilogue<D, NT>(acc, ep, smem, &s_flag);
}

// First round of a main-constraint sumcheck: every term is a product of BASE-field witness columns times an extension
// coefficient, under extension common factors (selectors).  Specialised accumulate:
//   * the product of a term's columns stays in the base field, first factor peeled (no multiply by one);
//   * c_t * P_t(X) is NOT formed: its two 128-bit partial products go straight into unreduced 160-bit accumulators per
//     evaluation point, reduced once per group and pair before the selector multiplies in — the coefficient scaling was
//     half of all multiplications in this round (PMC: 15.7k VALU instructions per pair on the ADD-shaped plan).
template <int D>
__global__ void __launch_bounds__(NT) k_accum_base0(DevPlan pl, size_t pairs, Epilogue ep) {
    __shared__ E2 smem[(NT / 64) * D];
    E2 acc[D];
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t p = (size_t)blockIdx.x * NT + threadIdx.x; p < pairs; p += stride) {
        for (int g = 0; g < pl.n_groups; g++) {
            Acc5 w0[D], w1[D];
#pragma unroll
            for (int t = 0; t < D; t++) w0[t] = w1[t] = Acc5{0, 0, 0, 0, 0};
            for (uint32_t ti = pl.group_term_off[g]; ti < pl.group_term_off[g + 1]; ti++) {
                const uint32_t term = pl.group_terms[ti];
                const E2 c = pl.coeffs[term];
                uint64_t pb[D];
                const uint32_t kb = pl.term_off[term], ke = pl.term_off[term + 1];
                {   // first column of the term
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(pl.slots[pl.term_idx[kb]].in + 2 * p);
                    const uint64_t nd = sub(v.x, v.y);
                    uint64_t x = v.y;
#pragma unroll
                    for (int t = 0; t < D; t++) {
                        pb[t] = x;
                        if (t + 1 < D) x = sub(x, nd);
                    }
                }
                for (uint32_t k = kb + 1; k < ke; k++) {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(pl.slots[pl.term_idx[k]].in + 2 * p);
                    const uint64_t nd = sub(v.x, v.y);
                    uint64_t x = v.y;
#pragma unroll
                    for (int t = 0; t < D; t++) {
                        pb[t] = mul_nc(pb[t], x);  // only multiplied again: any 64-bit value will do
                        if (t + 1 < D) x = sub(x, nd);
                    }
                }
#pragma unroll
                for (int t = 0; t < D; t++) {
                    acc5_add(w0[t], mul_wide(c.c0, pb[t]));
                    acc5_add(w1[t], mul_wide(c.c1, pb[t]));
                }
            }
            E2 inner[D];
#pragma unroll
            for (int t = 0; t < D; t++) inner[t] = E2{acc5_reduce(w0[t]), acc5_reduce(w1[t])};
            const uint32_t cb = pl.common_off[g], ce = pl.common_off[g + 1];
            for (uint32_t k = cb; k < ce; k++) {  // extension (or base) common factors: selectors
                E2 lo, hi;
                load_pair(pl.slots[pl.common_idx[k]], 0, p, lo, hi);
                const E2 nd = lo - hi;
                E2 x = hi;
#pragma unroll
                for (int t = 0; t < D; t++) {
                    inner[t] = inner[t] * x;
                    if (t + 1 < D) x = x - nd;
                }
            }
#pragma unroll
            for (int t = 0; t < D; t++) acc[t] = acc[t] + inner[t];
        }
    }
    __shared__ int s_flag;
    epilogue<D, NT>(acc, ep, smem, &s_flag);
}

// ------------------------------------------------------------------------------------------------
// fused generic round: fold every MLE of the class with r_{i-1}, write the half-size tables, stage the
// folded (f(1), delta) of this lane's pair in LDS — lane-private columns, no barrier — and evaluate the CSR
// term plan from LDS.  One launch per round for any single-class plan whose staging fits in LDS
// (n_mles * TNT * 32 B); all LDS is dynamic so the 16-byte alignment of the b128 accesses is guaranteed.
// ------------------------------------------------------------------------------------------------
template <int D, int TNT>
__global__ void __launch_bounds__(TNT) k_fused(DevPlan pl, int n_mles, size_t pairs, E2 r, Epilogue ep) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    E2* stage = reinterpret_cast<E2*>(dyn);                               // [n_mles][2][TNT]
    E2* smem = stage + (size_t)n_mles * 2 * TNT;                          // [(TNT/64) * D]
    unsigned long long* s_chal = reinterpret_cast<unsigned long long*>(smem + (TNT / 64) * D);  // 3 words + flag
    int* s_flag = reinterpret_cast<int*>(s_chal + 4);
    if (ep.wait_seq != 0) {
        if (!read_challenge(ep, r, s_chal)) return;
    }
    const E2Pre rp = e2_pre(r);
    const bool fold = pl.use_out != 0;
    E2 acc[D];
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    const size_t stride = (size_t)gridDim.x * TNT;
    const int tid = threadIdx.x;
    for (size_t p = (size_t)blockIdx.x * TNT + tid; p < pairs; p += stride) {
        for (int m = 0; m < n_mles; m++) {
            const MleSlot sl = pl.slots[m];
            E2 lo, hi;
            if (fold) {
                if (sl.in_ext) {
                    const uint64_t* q = sl.in + 8 * p;
                    const E2 a0 = ld_e2(q), a1 = ld_e2(q + 2), a2 = ld_e2(q + 4), a3 = ld_e2(q + 6);
                    lo = a0 + e2_mul_pre(rp, a1 - a0);
                    hi = a2 + e2_mul_pre(rp, a3 - a2);
                } else {
                    const uint64_t* q = sl.in + 4 * p;
                    const ulonglong2 v0 = *reinterpret_cast<const ulonglong2*>(q);
                    const ulonglong2 v1 = *reinterpret_cast<const ulonglong2*>(q + 2);
                    const E2 t0 = e2_mul_base(r, sub(v0.y, v0.x)), t1 = e2_mul_base(r, sub(v1.y, v1.x));
                    lo = E2{add(t0.c0, v0.x), t0.c1};
                    hi = E2{add(t1.c0, v1.x), t1.c1};
                }
                st_e2(sl.out + 4 * p, lo);
                st_e2(sl.out + 4 * p + 2, hi);
            } else if (sl.in_ext) {
                lo = ld_e2(sl.in + 4 * p);
                hi = ld_e2(sl.in + 4 * p + 2);
            } else {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(sl.in + 2 * p);
                lo = E2{v.x, 0};
                hi = E2{v.y, 0};
            }
            stage[(size_t)(2 * m) * TNT + tid] = hi;           // f(1)
            stage[(size_t)(2 * m + 1) * TNT + tid] = hi - lo;  // delta
        }
        for (int g = 0; g < pl.n_groups; g++) {
            E2 inner[D];
#pragma unroll
            for (int t = 0; t < D; t++) inner[t] = e2_zero();
            for (uint32_t ti = pl.group_term_off[g]; ti < pl.group_term_off[g + 1]; ti++) {
                const uint32_t term = pl.group_terms[ti];
                const E2 c = pl.coeffs[term];
                E2 pr[D];
#pragma unroll
                for (int t = 0; t < D; t++) pr[t] = c;
                bool seeded = false;
                for (uint32_t k = pl.term_off[term]; k < pl.term_off[term + 1]; k++) {
                    const uint32_t m = pl.term_idx[k];
                    mul_points<D>(pr, seeded, c, stage[(size_t)(2 * m) * TNT + tid], stage[(size_t)(2 * m + 1) * TNT + tid]);
                }
#pragma unroll
                for (int t = 0; t < D; t++) inner[t] = inner[t] + pr[t];
            }
            const uint32_t cb = pl.common_off[g], ce = pl.common_off[g + 1];
            if (ce > cb) {
                E2 cm[D];
                for (uint32_t k = cb; k < ce; k++) {
                    const uint32_t m = pl.common_idx[k];
                    E2 x = stage[(size_t)(2 * m) * TNT + tid];
                    const E2 delta = stage[(size_t)(2 * m + 1) * TNT + tid];
#pragma unroll
                    for (int t = 0; t < D; t++) {
                        cm[t] = (k == cb) ? x : cm[t] * x;
                        x = x + delta;
                    }
                }
#pragma unroll
                for (int t = 0; t < D; t++) acc[t] = acc[t] + cm[t] * inner[t];
            } else {
#pragma unroll
                for (int t = 0; t < D; t++) acc[t] = acc[t] + inner[t];
            }
        }
    }
    epilogue<D, TNT>(acc, ep, smem, s_flag);
}

// gather element 0 of every listed table into out[i] (final evaluations)
struct GatherArgs {
    const MleSlot* slots;
};
__global__ void k_gather_first(const MleSlot* __restrict__ slots, int n, E2* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = *reinterpret_cast<const E2*>(slots[i].out);
}

// start of a sumcheck: zero the arrival counter / relay block and pull the (small) plan blob out of the pinned block in ONE
// launch — the runtime's fill and copy kernels cost ~9 and ~4 us of stream time each, in front of every tower layer
__global__ void __launch_bounds__(256) k_setup(uint64_t* __restrict__ zero, size_t zero_words, uint64_t* __restrict__ dst,
                                               const uint64_t* __restrict__ src_host_view, size_t words, uint64_t* __restrict__ ones = nullptr,
                                               size_t ones_words = 0) {
    for (size_t i = threadIdx.x; i < zero_words; i += 256) zero[i] = 0;
    for (size_t i = threadIdx.x; i < words; i += 256) dst[i] = src_host_view[i];
    for (size_t i = threadIdx.x; i < ones_words; i += 256) ones[i] = MSG_INVALID;  // the armed partial-sum rows of k_mid
}

// last fold of a sumcheck: table i has two elements left, its evaluation is lo + r (hi - lo).  One launch writes all of
// them straight into pinned host memory (no separate fold, gather and device-to-host blit on the way to the caller).
__global__ void k_finish_evals(const MleSlot* __restrict__ slots, int n, E2 r, E2* __restrict__ out_host) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MleSlot sl = slots[i];
    E2 lo, hi;
    if (sl.in_ext) {
        lo = ld_e2(sl.in);
        hi = ld_e2(sl.in + 2);
    } else {
        lo = E2{sl.in[0], 0};
        hi = E2{sl.in[1], 0};
    }
    // one 16-byte write-through system-scope store: the host watches these words (ceno_hip_sumcheck_finish)
    const E2 v = lo + r * (hi - lo);
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 w = {(unsigned)v.c0, (unsigned)(v.c0 >> 32), (unsigned)v.c1, (unsigned)(v.c1 >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out_host + i), "v"(w) : "memory");
}

// tables of a class as extension elements into pinned host memory (armed with MSG_INVALID): the host takes the sumcheck over from here
__global__ void __launch_bounds__(NT) k_export_tables(const MleSlot* __restrict__ slots, int n_mles, int len, E2* __restrict__ out_host) {
    const int total = n_mles * len;
    for (int idx = blockIdx.x * NT + threadIdx.x; idx < total; idx += gridDim.x * NT) {
        const int m = idx / len, j = idx - m * len;
        const MleSlot sl = slots[m];
        const E2 v = sl.in_ext ? ld_e2(sl.in + 2 * (size_t)j) : E2{sl.in[j], 0};
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 w = {(unsigned)v.c0, (unsigned)(v.c0 >> 32), (unsigned)v.c1, (unsigned)(v.c1 >> 32)};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out_host + idx), "v"(w) : "memory");
    }
}

// ------------------------------------------------------------------------------------------------
// host-side state
// ------------------------------------------------------------------------------------------------
// CENO_HIP_EQ_VERIFY=1: spot check of a declared eq table (ceno_hip_sumcheck_begin_eq) — rows at and around the ends of the declared range and a
// few inside must hold eq(row, point) resp. zero; *bad counts the rows that do not
struct EqVerifyArg {
    E2 pt[40];
    int nv;
    unsigned long long rows[8];
    unsigned long long lo, hi;
};
__global__ void k_eq_verify(const E2* __restrict__ table, EqVerifyArg a, unsigned* __restrict__ bad) {
    const int k = threadIdx.x;
    if (k >= 8) return;
    const unsigned long long x = a.rows[k];
    if (x >= (1ull << a.nv)) return;
    E2 want = e2_zero();
    if (x >= a.lo && x < a.hi) {
        want = e2_one();
        for (int v = 0; v < a.nv; v++) want = want * (((x >> v) & 1) ? a.pt[v] : (e2_one() - a.pt[v]));
    }
    const E2 got = table[x];
    if (got.c0 != want.c0 || got.c1 != want.c1) atomicAdd(bad, 1u);
}

namespace {

struct ScMle {
    const uint64_t* cur = nullptr;
    int cur_ext = 0;
    int nv = 0;
    uint64_t* buf[2] = {nullptr, nullptr};  // ping (2^(nv-1) ext), pong (2^(nv-2) ext)
    int which = 0;
    bool done = false;  // reduced to a single value
    E2 eval = e2_zero();
    E2 tail = e2_one();
    int cls = -1, local = -1;
};

struct ScTerm {
    E2 coeff;
    std::vector<int> idx;   // residual factors (what the plan lists for the term)
    std::vector<int> full;  // residual + common factors of its group
    int nv;
    // once the term's class has retired: coeff * prod_j eval_j (its factors are scalars from then on), and whether every factor sits in
    // that one class (then they share one tail: the product of the challenges since)
    bool front_ready = false, front_one_class = false;
    E2 front = gl::e2_zero();
};

struct ScClass {
    int nv = 0;
    std::vector<int> mles;    // global ids
    std::vector<int> terms;   // global term ids
    bool dense = false;       // single term, no groups, K == d <= MAXK, all same element kind
    // device plan (generic)
    MleSlot* d_slots = nullptr;
    uint32_t *d_group_term_off = nullptr, *d_group_terms = nullptr, *d_common_off = nullptr, *d_common_idx = nullptr;
    uint32_t *d_term_off = nullptr, *d_term_idx = nullptr;
    E2* d_coeffs = nullptr;
    int n_groups = 0;
    int n_flat = 0;           // entries of group_terms
    bool terms_all_base = false;  // every factor of every term is a base-field input table (round 0 may use k_accum_base0)
    uint32_t part_off = 0;    // offset (E2 units) into partials
    // host copies of the class plan (class-local MLE ids), from which the LDS-blocked kernel's component tables are built
    std::vector<uint32_t> h_gto, h_gt, h_co, h_ci, h_to, h_ti;
    std::vector<E2> h_coeffs;
    bool gen = false;         // rounds of this class run in the merged k_gen launch (sumcheck_gen.hip)
    // once the class is a scalar: its terms' constant parts c_t prod_j eval_j summed by number of factors (the class's tail and the point
    // enter as powers: sum_t c_t prod_j (eval_j tail (x + 1)) = sum_k front_sum[k] tail^k (x + 1)^k) and the terms that do not fit the form
    bool front_grouped = false;
    E2 front_sum[17];
    bool front_has[17] = {};
    std::vector<int> front_slow;
};

// one connected component of a class's plan (a chip of a batched main sumcheck: its columns, selectors and terms)
struct GenCompHost {
    int cls = 0;
    std::vector<int> mles;                 // class-local MLE ids in component order
    std::vector<char> writes;              // per MLE: this component writes its folded table (a column staged by several column blocks of a
                                           // wide chip is written by ONE of them; empty = all)
    std::vector<std::vector<uint32_t>> gts;  // per group of the class: the class-local terms this component evaluates
    int n_groups = 0, n_terms = 0;
    int tp_log = 8, wt_log = 0;
    bool base0_ok = false;
    size_t units[2] = {0, 0};              // stage rows: [0] all-extension layout, [1] first-round layout (base rows are one unit)
    size_t off_terms[3] = {0, 0, 0}, off_groups[3] = {0, 0, 0}, off_unit[3] = {0, 0, 0};  // offsets into the gen blob ([2]: factor bytes = MLE indices, k_eq_base0)
    size_t slot_off = 0;                   // first slot of the component inside a round's slot row
    int geq = -1;                          // >= 0: the component runs in eq-factored form (index into sumcheck::geq.comps)
    int deg = 0;                           // eq components: 1 + the most column factors of a term, at least 3 (the length of ITS round polynomial)
    double pair_cost[2] = {1.0, 1.0};      // relative cost of a pair in phase 1 / phase 2 (workgroup split of the component-aligned launch)
};
struct GenRound {
    size_t off_comps = 0;
    int n_comps = 0;
    unsigned total_tiles = 0;
    size_t stage_bytes = 0;
    bool base0 = false, has_terms = false;
    // the eq-factored components of the round (k_gen_eq, component-aligned launches): ONE launch at the sumcheck's degree, or — in the large
    // rounds of a batch whose components have polynomials of different lengths (round 5) — one launch per component degree
    struct EqLaunch {
        size_t off_comps = 0;
        int n_comps = 0;
        unsigned grid = 0;
        size_t off_wg_comp = 0;  // uint16 per workgroup of the component-aligned launch: its component
        bool slots = false;      // every component of the launch is one tile: one workgroup per component and slot (k_gen_eq_slots)
        size_t stage_bytes = 0;
        int D = 0;               // message length the launch's kernel is instantiated for
    };
    std::vector<EqLaunch> eq;
    bool by_degree = false;    // every eq component delivers the slots of ITS OWN degree this round (geq_collect extends its polynomial)
    bool direct0 = false;      // the eq lists are laid out for k_eq_base0 (first round, base-field columns, no LDS stage)
};
// a table declared as eq(., point) on the rows [lo, hi) (ceno_hip_sumcheck_begin_eq)
struct EqDecl {
    bool on = false;
    std::vector<E2> pt;
    uint64_t lo = 0, hi = 0;
};
// an eq-factored component: host side of the round protocol (sumcheck_gen.hip "EQ-FACTORED FORM")
struct GeqComp {
    int comp = -1;                 // index into gen_comps
    int nv = 0;
    int slot = 0;                  // index among the eq components
    int deg = 0;                   // the component's own message length (<= the sumcheck's)
    std::vector<E2> pt, inv1m;     // the component's eq point and 1 / (1 - pt_i)
    struct Grp {
        uint64_t lo, hi;
        int brow;                  // first of the group's two rows in the boundary block
    };
    std::vector<Grp> groups;
    std::vector<E2> P;             // p_c(0 .. D) of the round answered last
};

}  // namespace

struct ceno_hip_sumcheck {
    ceno_hip_ctx* ctx = nullptr;
    hipStream_t st = nullptr;
    int n = 0, d = 0;
    int round = 0;  // next message to produce
    bool finished = false;
    std::vector<ScMle> mles;
    std::vector<ScTerm> terms;
    std::vector<ScClass> classes;  // sorted by nv descending
    E2* d_partials = nullptr;
    E2* d_msg = nullptr;           // d ext (device)
    unsigned* d_counter = nullptr; // arrival counter of the in-kernel reduction
    E2* d_round_acc = nullptr;     // running message total across the classes of one round
    uint64_t* d_hmsg = nullptr;    // device view of h_pinned (message lands directly in host memory)
    uint64_t* d_mid_rows = nullptr;         // two sets of armed partial-sum rows + relay lines of the persistent mid-round kernel
    bool slots_preloaded = false;           // the slot rows of every round went to the device with the plan blob (single generic class)
    int mid_reserved = 0;                   // workgroups of a k_mid launch booked against the context's residency budget
    bool tail_evals = false;                // the persistent tail kernel also produces the final evaluations (finish posts the last challenge)
    // host-finished tail: rounds [host_from, n) are computed by the HOST on the tables the tail kernel exported with its last message
    int host_from = -1;
    int host_ht = 0;                        // host rounds this plan is worth (host_tail_rounds at begin)
    E2* h_tail = nullptr;                   // inside the pinned block: n_mles x (2 << host_ht) E2, armed with MSG_INVALID
    E2* d_tail_view = nullptr;
    int host_len0 = 0;                      // entries per exported table
    int host_len = 0;                       // entries per table in `host_tab` now (0 = not taken over yet)
    std::vector<E2> host_tab;               // [class-local mle][host_len0]
    unsigned long long* h_flag = nullptr;   // pinned sequence flag written by the kernel, polled by the host
    unsigned long long* d_hflag = nullptr;
    unsigned long long seq = 0;
    void* h_block = nullptr;       // one pinned allocation carved into flag | mailbox | message+evals | slot staging
    Mailbox* h_mailbox = nullptr;  // host -> device challenges (pipelined mode)
    Mailbox* d_mailbox = nullptr;
    void* vram_slot = nullptr;     // != NULL: the mailbox lives in host-writable device memory (h_mailbox == d_mailbox)
    Bcast* d_bcast = nullptr;      // device relay
    bool allow_pipeline = false;   // caller promised to drive the rounds back to back (ceno_hip_sumcheck_set_pipelined)
    bool pipelined = false;        // round kernels are enqueued ahead of their challenges, which travel through the mailbox
    bool live_counted = false;     // counted in ctx->pipelined_live (ctx_pipelined_begin / _end)
    PipelinedOwner live_owner;     // the beginning thread's counter (the handle may be released on another thread)
    int enq = 0;                   // pipelined: rounds [0, enq) are in the stream
    E2* d_evals = nullptr;         // gather scratch (device), num_mles
    E2* h_pinned = nullptr;        // pinned host staging: msg (MAXD) + evals (num_mles)
    MleSlot* h_slots = nullptr;    // pinned staging for slot tables, (n + 2) x total class mles
    size_t slots_per_round = 0;
    std::vector<void*> dev_allocs; // everything from ctx_alloc, freed on free()
    char* arena = nullptr;         // the current chunk of the handle's small-block arena (sc_dev_alloc; the chunk itself is in dev_allocs)
    size_t arena_used = 0;
    // tower layers (sumcheck_tower.hpp): rounds [0, fast_upto) run on k_tower, whose message the host completes (sc_tower_message)
    struct {
        bool on = false;
        int n_prod = 0, n_logup = 0, fast_upto = 0;
        TowerCoef coef;
        std::vector<E2> pt, inv1m;  // the eq point rt and 1 / (1 - rt_i) for the rounds that need it
        E2 q0, c1, c2;              // q_{i-1} of the round answered last
        bool has_claim = false;     // the caller stated the sum (ceno_hip_sumcheck_set_claim): round 0 needs two values instead of three
        E2 claim0;
    } eqf;
    bool owns_tower_eq = false;
    ceno_hip_mle* extra_owned = nullptr;  // eq table built by tower_layer_sumcheck_begin
    // LDS-blocked generic rounds (sumcheck_gen.hip): component tables and the slot schedule of every round, one device blob
    bool gen_on = false;
    std::vector<GenCompHost> gen_comps;
    std::vector<GenRound> gen_rounds;
    char* d_gen = nullptr;
    void* h_gen = nullptr;                // pinned staging of the blob (alive until the handle is freed)
    // eq-factored main-constraint rounds (ceno_hip_sumcheck_begin_eq; sumcheck_gen.hip "EQ-FACTORED FORM")
    struct {
        bool on = false;
        std::vector<EqDecl> decl;         // per MLE of the plan
        std::vector<GeqComp> comps;
        int n_brows = 0;
        void* h_block = nullptr;          // pinned: [slot][D] quotient sums, then [brow][D] boundary values
        E2 *h_q = nullptr, *d_q = nullptr, *h_b = nullptr, *d_b = nullptr;
        unsigned* d_counters = nullptr;   // device: one arrival counter per eq component
        unsigned max_grid = 0;            // largest eq launch of the sumcheck (rows of the components with several workgroups)
        uint64_t* d_rows = nullptr;       // device: [workgroup][D] partial sums of such components
        // interpolation weights (base field) per message length Dc (index Dc; 3 .. D), nodes -> target:  A: 1..Dc-1 -> 0, Dc   B: 0..Dc-2 -> Dc-1, Dc
        struct Wts {
            std::vector<uint64_t> wA0, wAD, wB1, wBD, pow_dm1;
            std::vector<std::vector<uint64_t>> ext;  // ext[t - Dc - 1][j]: value at t = Dc + 1 .. D of the polynomial through (j, v_j), j = 0 .. Dc
        };
        std::vector<Wts> w;
        std::vector<uint64_t> lag_den_inv;
    } geq;
};

static int host_tail_rounds(const ceno_hip_sumcheck* sc);  // below, next to the host rounds
static int sc_wait_words(ceno_hip_sumcheck* sc, const uint64_t* words, int n_words, const char* what);

// Device memory of a handle.  Every block lives until the handle is released, so the SMALL ones (a working buffer pair per table of a
// small layer, message / evaluation / counter / slot / plan blocks: ~25 of them for a tower layer) are cut from chunks of the handle's own
// arena: one trip to the pool per chunk instead of one per block — with four lanes beginning and releasing layers side by side the pool's
// lock was most of a layer's begin (tools/dev/lanes_sc.cpp: 96 us at four lanes against 8 us alone).  Large blocks stay pool blocks of
// their own (the pool can hand them to the next handle whatever its shape).
static constexpr size_t SC_ARENA_CHUNK = (size_t)2 << 20, SC_ARENA_ITEM_MAX = (size_t)512 << 10;
static int sc_dev_alloc(ceno_hip_sumcheck* sc, size_t bytes, void** out) {
    const size_t b = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    if (b <= SC_ARENA_ITEM_MAX) {
        if (!sc->arena || sc->arena_used + b > SC_ARENA_CHUNK) {
            void* c = nullptr;
            TRY(ctx_alloc(sc->ctx, SC_ARENA_CHUNK, &c));
            sc->dev_allocs.push_back(c);
            sc->arena = (char*)c;
            sc->arena_used = 0;
        }
        *out = sc->arena + sc->arena_used;
        sc->arena_used += b;
        return 0;
    }
    TRY(ctx_alloc(sc->ctx, bytes, out));
    sc->dev_allocs.push_back(*out);
    return 0;
}

template <typename T>
static int upload_vec(ceno_hip_sumcheck* sc, const std::vector<T>& v, T** out) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    TRY(sc_dev_alloc(sc, bytes, &p));
    if (!v.empty()) HIP_TRY(sc->ctx, hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, sc->st));
    *out = (T*)p;
    return 0;
}

// stores to a mailbox that may sit behind the PCIe BAR (write-combining mapping): sfence orders and flushes them
static inline void host_store_fence() {
#if defined(__x86_64__)
    __builtin_ia32_sfence();
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
static inline void mailbox_clear(Mailbox* m) {
    volatile Mailbox* v = m;
    v->chal_seq = 0;
    v->chal[0] = 0;
    v->chal[1] = 0;
    v->abort = 0;
    host_store_fence();
}

static void sc_release(ceno_hip_sumcheck* sc) {
    if (!sc) return;
    CENO_TIMED("sc_release");
    if (sc->pipelined && !sc->finished && sc->h_mailbox) {  // also after the last round: the tail may be waiting for the final challenge
        __atomic_store_n(&sc->h_mailbox->abort, 1ull, __ATOMIC_RELEASE);  // queued kernels exit at their wait
        host_store_fence();
    }
    // a finished sumcheck may still have its last kernel's trailing stores in flight (finish takes the evaluations from armed
    // pinned words and does not synchronise): no wait here either — it was ~20 us per tower layer — the buffers go back to the
    // pool tagged with THIS sumcheck's stream (ctx_free_on below), so another stream gets them only once it has drained
    if (!sc->finished) (void)hipStreamSynchronize(sc->st);
    if (sc->pipelined && getenv("CENO_HIP_DEBUG") && sc->d_bcast) {
        static Bcast hb;
        if (hipMemcpy(&hb, sc->d_bcast, sizeof(Bcast), hipMemcpyDeviceToHost) == hipSuccess) {
            for (int i = 1; i <= sc->n && i < 64; i++)
                fprintf(stderr, "[ceno_hip] dev round %d: start->publish %.1f us, publish %.1f us, poll %.1f us, prev poll end -> start %.1f us\n", i - 1,
                        (hb.dbg[i][1] - hb.dbg[i][0]) / 100.0, (hb.dbg[i][2] - hb.dbg[i][1]) / 100.0, (hb.dbg[i][3] - hb.dbg[i][2]) / 100.0,
                        i > 1 ? (hb.dbg[i][0] - hb.dbg[i - 1][3]) / 100.0 : 0.0);
        }
    }
    if (sc->mid_reserved) {
        sc->ctx->mid_wgs_in_flight.fetch_sub(sc->mid_reserved);
        sc->mid_reserved = 0;
    }
    if (sc->live_counted) ctx_pipelined_end(sc->ctx, sc->live_owner);
    ctx_free_many_on(sc->ctx, sc->dev_allocs.data(), sc->dev_allocs.size(), sc->st, false);
    ctx_pinned_free(sc->ctx, sc->h_block);
    ctx_pinned_free(sc->ctx, sc->h_gen);
    ctx_pinned_free(sc->ctx, sc->geq.h_block);
    ctx_vram_slot_free(sc->ctx, sc->vram_slot);
    if (sc->extra_owned) ceno_hip_mle_free(sc->ctx, sc->extra_owned);
    delete sc;
}

template <int K>
static void launch_dense(ceno_hip_sumcheck* sc, ScClass& cl, int mode, size_t pairs, E2 r, unsigned grid, const Epilogue& ep) {
    DenseTables tp{};
    const ScTerm& term = sc->terms[cl.terms[0]];
    for (int m = 0; m < K; m++) {
        ScMle& M = sc->mles[term.idx[m]];
        tp.in[m] = M.cur;
        tp.out[m] = M.buf[M.which];
    }
    launch_dense_tables(sc->ctx, K, mode, tp, pairs, r, ep, grid, sc->st);
}

// the same launch with the tables of one staged slot row (pipelined rounds: the ping-pong walk of every round is staged once)
template <int K>
static void launch_dense_row(ceno_hip_sumcheck* sc, ScClass& cl, int mode, size_t pairs, E2 r, unsigned grid, const Epilogue& ep, const MleSlot* row) {
    DenseTables tp{};
    const ScTerm& term = sc->terms[cl.terms[0]];
    for (int m = 0; m < K; m++) {
        const MleSlot& sl = row[sc->mles[term.idx[m]].local];
        tp.in[m] = sl.in;
        tp.out[m] = sl.out;
    }
    launch_dense_tables(sc->ctx, K, mode, tp, pairs, r, ep, grid, sc->st);
}

static bool accum_lazy() {
    static bool v = [] {
        const char* e = getenv("CENO_HIP_ACCUM_LAZY");  // 1: unreduced coefficient products (60 more VGPRs; measured +-1 %, off by default)
        return e && atoi(e) != 0;
    }();
    return v;
}
static bool accum_ext() {
    static bool v = [] {
        const char* e = getenv("CENO_HIP_ACCUM_EXT");  // 0: the generic accumulate kernel in every round (A/B measurements)
        return !(e && atoi(e) == 0);
    }();
    return v;
}
// (grids stay within the workgroups that are resident at once — k_accum<4> holds 3 per CU at 150 VGPRs: resident_grid, common.hpp)
template <int D>
static void launch_accum_d(ceno_hip_ctx* ctx, const DevPlan& pl, size_t pairs, const Epilogue& ep, unsigned grid, hipStream_t st, bool base0) {
    if (base0) hipLaunchKernelGGL((k_accum_base0<D>), dim3(resident_grid(ctx, k_accum_base0<D>, NT, 0, grid)), dim3(NT), 0, st, pl, pairs, ep);
    else if (accum_lazy()) hipLaunchKernelGGL((k_accum<D, true>), dim3(resident_grid(ctx, k_accum<D, true>, NT, 0, grid)), dim3(NT), 0, st, pl, pairs, ep);
    else if (pl.use_out && accum_ext()) hipLaunchKernelGGL((k_accum_ext<D>), dim3(resident_grid(ctx, k_accum_ext<D>, NT, 0, grid)), dim3(NT), 0, st, pl, pairs, ep);
    else hipLaunchKernelGGL((k_accum<D, false>), dim3(resident_grid(ctx, k_accum<D, false>, NT, 0, grid)), dim3(NT), 0, st, pl, pairs, ep);
}
// base0: round 0 of a class whose every term is a product of base-field tables (k_accum_base0)
static void launch_accum(ceno_hip_ctx* ctx, int d, const DevPlan& pl, size_t pairs, const Epilogue& ep, unsigned grid, hipStream_t st, bool base0 = false) {
    static const bool no_base0 = getenv("CENO_HIP_NO_BASE0") != nullptr;  // A/B switch
    if (no_base0) base0 = false;
    switch (d) {
    case 1: launch_accum_d<1>(ctx, pl, pairs, ep, grid, st, base0); break;
    case 2: launch_accum_d<2>(ctx, pl, pairs, ep, grid, st, base0); break;
    case 3: launch_accum_d<3>(ctx, pl, pairs, ep, grid, st, base0); break;
    case 4: launch_accum_d<4>(ctx, pl, pairs, ep, grid, st, base0); break;
    case 5: launch_accum_d<5>(ctx, pl, pairs, ep, grid, st, base0); break;
    case 6: launch_accum_d<6>(ctx, pl, pairs, ep, grid, st, base0); break;
    case 7: launch_accum_d<7>(ctx, pl, pairs, ep, grid, st, base0); break;
    default: launch_accum_d<8>(ctx, pl, pairs, ep, grid, st, base0); break;
    }
}

// fused generic kernel: threads per block chosen so that the LDS staging (n_mles * TNT * 32 B) stays <= 60 KB
// The staging costs 32 B of LDS per MLE per in-flight pair, which caps occupancy (160 KB / CU), so the fused
// kernel is used where a round is latency bound (one launch instead of two); large rounds keep the
// two-kernel path whose factor re-reads are served by L2 (measured on the 2^20-row chip flow).
static constexpr size_t FUSED_MAX_PAIRS = (size_t)1 << 14;
static int fused_tnt(size_t n_mles, size_t pairs) {
    if (pairs > FUSED_MAX_PAIRS) return 0;
    if (n_mles <= 7) return 256;
    if (n_mles <= 15) return 128;
    if (n_mles <= 30) return 64;
    return 0;  // does not fit: two-kernel path
}
template <int D, int TNT>
static void launch_fused_dt(const DevPlan& pl, int n_mles, size_t pairs, E2 r, const Epilogue& ep, unsigned grid, hipStream_t st) {
    const size_t lds = ((size_t)n_mles * 2 * TNT + (TNT / 64) * D) * sizeof(E2) + 64;
    hipLaunchKernelGGL((k_fused<D, TNT>), dim3(grid), dim3(TNT), lds, st, pl, n_mles, pairs, r, ep);
}
template <int D>
static void launch_fused_d(int tnt, const DevPlan& pl, int n_mles, size_t pairs, E2 r, const Epilogue& ep, unsigned grid, hipStream_t st) {
    if (tnt == 256) launch_fused_dt<D, 256>(pl, n_mles, pairs, r, ep, grid, st);
    else if (tnt == 128) launch_fused_dt<D, 128>(pl, n_mles, pairs, r, ep, grid, st);
    else launch_fused_dt<D, 64>(pl, n_mles, pairs, r, ep, grid, st);
}
static void launch_fused(int d, int tnt, const DevPlan& pl, int n_mles, size_t pairs, E2 r, const Epilogue& ep, unsigned grid, hipStream_t st) {
    switch (d) {
    case 1: launch_fused_d<1>(tnt, pl, n_mles, pairs, r, ep, grid, st); break;
    case 2: launch_fused_d<2>(tnt, pl, n_mles, pairs, r, ep, grid, st); break;
    case 3: launch_fused_d<3>(tnt, pl, n_mles, pairs, r, ep, grid, st); break;
    case 4: launch_fused_d<4>(tnt, pl, n_mles, pairs, r, ep, grid, st); break;
    case 5: launch_fused_d<5>(tnt, pl, n_mles, pairs, r, ep, grid, st); break;
    case 6: launch_fused_d<6>(tnt, pl, n_mles, pairs, r, ep, grid, st); break;
    case 7: launch_fused_d<7>(tnt, pl, n_mles, pairs, r, ep, grid, st); break;
    default: launch_fused_d<8>(tnt, pl, n_mles, pairs, r, ep, grid, st); break;
    }
}

// grid for `pairs` work items: enough blocks to fill 256 CUs x 8, at least 1
static unsigned sc_grid(size_t pairs) {
    static const unsigned cap = getenv("CENO_HIP_MAXB") ? (unsigned)atoi(getenv("CENO_HIP_MAXB")) : MAXB;
    return grid_for(pairs, NT, cap);
}


// ------------------------------------------------------------------------------------------------
// LDS-blocked generic rounds (sumcheck_gen.hip): split every generic class into the connected components of its plan, lay
// the terms / groups out per component, and precompute the slot tables and component lists of ALL rounds (buffer
// ping-pong is deterministic), so that a round is one launch and no copy.  Built only for sumchecks large enough to pay
// for the table upload; anything that does not fit (a component whose staged rows exceed the LDS budget) keeps the
// two-kernel path.
// ------------------------------------------------------------------------------------------------
// (read per call, not cached: the test-suite switches them between sumchecks)
static int gen_min_log() {
    const char* e = getenv("CENO_HIP_GEN_MIN_LOG");  // smallest class (variables) that gets component tables
    return e ? atoi(e) : 13;
}
// ... and, inside a sumcheck that has them, the smallest CLASS that goes through them.  A batch of many chips has many small classes (48 chips of
// 2^2 .. 2^14 rows: thirteen), and on the two-kernel path every one of them costs two launches per round: 5.35 ms for that batch against 3.95 with
// every class from 2^4 rows in the one launch of the round (2^16: 5.8 -> 4.5, 2^20: 7.7 -> 6.6, 2^24: unchanged; tools/dev/min_log_sweep.sh).
static int gen_class_min_log() {
    if (const char* e = getenv("CENO_HIP_GEN_CLASS_MIN_LOG")) return atoi(e);
    if (const char* e = getenv("CENO_HIP_GEN_MIN_LOG")) return atoi(e);
    return 4;
}
static size_t gen_stage_budget(int d) {
    const char* e = getenv("CENO_HIP_GEN_STAGE_KB");  // LDS the staged rows of one tile may take
    const int kb = e ? atoi(e) : 48;
    // the whole block (fixed part + stage + exchange of 4 KB per evaluation point) stays within 64 KB
    const size_t cap = 63 * 1024 - gen_lds_bytes(d, 0);
    return std::min((size_t)std::max(kb, 1) * 1024, cap);
}
// ... and what the staged rows may take when THREE workgroups are to share a CU's 160 KB (the round kernels are bound by VALU issue: at degree 5
// the 64 KB block held two workgroups per CU = two waves per SIMD, VALU busy 0.72; three: ~0.9).  The column blocks of wide components are cut
// to this size, and tiles larger than 64 pairs are only taken where they fit it.
static size_t gen_stage_budget3(ceno_hip_ctx* ctx, int d) {
    // (only where the REGISTERS allow a third workgroup: the eq-factored kernel of degree 5 takes 205 VGPRs + AGPRs = two waves per SIMD
    // whatever its LDS, and smaller blocks then only stage the selectors more often — 62.9 vs 61.1 ms for the wide batch)
    if (gen_resident_cap(ctx, d, false, 0) < 3u * (unsigned)ctx->num_cus) return gen_stage_budget(d);
    const size_t per_wg = (size_t)52 * 1024;  // 160 KB / 3 rounded down to the allocation granularity
    const size_t fixed = gen_lds_bytes(d, 0);
    return std::min(gen_stage_budget(d), per_wg > fixed ? per_wg - fixed : (size_t)0);
}
static size_t gen_pipe_min_pairs() {
    // pipelined single-class sumchecks (tower layers, a single chip's main sumcheck): rounds with at least 2^this pairs use
    // k_gen.  Off by default: these plans have few terms per column (memory-bound), where the two-kernel path's streaming
    // fold + L2-served accumulate measured faster than tiles separated by barriers (tower proof 6.1 vs 7.3 ms at 2^20 rows).
    const char* e = getenv("CENO_HIP_GEN_PIPE_MIN_LOG");
    return (size_t)1 << std::min(e ? atoi(e) : 62, 62);
}
static size_t gen_stage_bytes(size_t units, int tp_log) { return units * (((size_t)1 << tp_log) + GEN_PAD) * sizeof(E2); }

// ------------------------------------------------------------------------------------------------
// eq-factored main-constraint rounds: host side (kernel side and derivation: sumcheck_gen.hip "EQ-FACTORED FORM").
// Per eq component c and round i the device delivers   Q_c(1 .. D-2), [Q_c(D-1) in the first round], the leading coefficient of Q_c
// (up to the sign (-1)^(D-1)) and, per boundary pair, the same slots of (c G) at that pair; the host completes
//     p_c(X) = eq(X, rt_i) Q_c(X) + sum_pairs X (c G)(X)
// from the component's running claim  p_c^{(i-1)}(r_{i-1}) = p_c(0) + p_c(1)  and adds p_c(1 .. D) to the round's message.
// ------------------------------------------------------------------------------------------------
static int geq_prepare(ceno_hip_sumcheck* sc) {
    auto& Gq = sc->geq;
    const int D = sc->d;
    const size_t n_slots = Gq.comps.size();
    const size_t bytes = (n_slots + (size_t)std::max(Gq.n_brows, 1)) * (size_t)D * sizeof(E2);
    void *hb = nullptr, *db = nullptr;
    TRY(ctx_pinned_alloc(sc->ctx, bytes, &hb, &db));
    Gq.h_block = hb;
    Gq.h_q = (E2*)hb;
    Gq.d_q = (E2*)db;
    Gq.h_b = Gq.h_q + n_slots * D;
    Gq.d_b = Gq.d_q + n_slots * D;
    for (size_t k = 0; k < bytes / 8; k++) reinterpret_cast<uint64_t*>(hb)[k] = MSG_INVALID;
    {
        void* p = nullptr;
        TRY(sc_dev_alloc(sc, std::max<size_t>(n_slots, 1) * sizeof(unsigned), &p));
        Gq.d_counters = (unsigned*)p;
        HIP_TRY(sc->ctx, hipMemsetAsync(p, 0, std::max<size_t>(n_slots, 1) * sizeof(unsigned), sc->st));
    }
    // Lagrange weights over integer nodes: value at `x` of the polynomial of degree < m through (node0 + j, v_j)
    auto weights = [](int node0, int m, int x) {
        std::vector<uint64_t> w((size_t)m);
        for (int j = 0; j < m; j++) {
            uint64_t num = 1, den = 1;
            for (int k = 0; k < m; k++) {
                if (k == j) continue;
                const int a = x - (node0 + k), b = j - k;
                num = gl::mul(num, a >= 0 ? (uint64_t)a : gl::neg((uint64_t)(-a)));
                den = gl::mul(den, b >= 0 ? (uint64_t)b : gl::neg((uint64_t)(-b)));
            }
            w[(size_t)j] = gl::mul(num, gl::inv(den));
        }
        return w;
    };
    {
        void* p = nullptr;
        TRY(sc_dev_alloc(sc, (size_t)std::max(Gq.max_grid, 1u) * (size_t)D * sizeof(E2), &p));
        Gq.d_rows = (uint64_t*)p;
    }
    // (a component whose own polynomial is shorter than the sumcheck's delivers Dc < D slots in the large rounds: weights for every length)
    Gq.w.assign((size_t)D + 1, {});
    for (int Dc = 3; Dc <= D; Dc++) {
        auto& W = Gq.w[(size_t)Dc];
        W.wA0 = weights(1, Dc - 1, 0);
        W.wAD = weights(1, Dc - 1, Dc);
        W.wB1 = weights(0, Dc - 1, Dc - 1);
        W.wBD = weights(0, Dc - 1, Dc);
        W.pow_dm1.assign((size_t)Dc + 1, 0);
        for (int t = 0; t <= Dc; t++) W.pow_dm1[(size_t)t] = gl::pow((uint64_t)t, (uint64_t)(Dc - 1));
        for (int t = Dc + 1; t <= D; t++) W.ext.push_back(weights(0, Dc + 1, t));
    }
    // 1 / prod_{j != t} (t - j) over the nodes 0 .. D (Lagrange basis at a field point)
    Gq.lag_den_inv.assign((size_t)D + 1, 0);
    for (int t = 0; t <= D; t++) {
        uint64_t den = 1;
        for (int j = 0; j <= D; j++)
            if (j != t) den = gl::mul(den, t > j ? (uint64_t)(t - j) : gl::neg((uint64_t)(j - t)));
        Gq.lag_den_inv[(size_t)t] = gl::inv(den);
    }
    for (auto& Q : Gq.comps) Q.P.assign((size_t)D + 1, e2_zero());
    Gq.on = true;
    return 0;
}
// boundary pairs of a group in round i: is pair `P` one (the device's predicate, gen_group_eq), and which candidates exist
static bool geq_pair_irregular(uint64_t lo, uint64_t hi, int i, uint64_t P) {
    const uint64_t s0 = (2 * P) << i, s1 = (2 * P + 1) << i, s2 = (2 * P + 2) << i;
    const bool full0 = lo <= s0 && s1 <= hi, full1 = lo <= s1 && s2 <= hi;
    const bool empty0 = s1 <= lo || s0 >= hi, empty1 = s2 <= lo || s1 >= hi;
    return !((full0 && full1) || (empty0 && empty1));
}
// rows of the boundary block the device will write in round i for group g of a component with `pairs` pairs: out[side] = true
static void geq_boundary_rows(const GeqComp::Grp& g, int i, uint64_t pairs, bool (&out)[2]) {
    out[0] = out[1] = false;
    const uint64_t p_lo = (g.lo >> i) >> 1, p_hi = g.hi > 0 ? ((g.hi - 1) >> i) >> 1 : 0;
    if (p_lo < pairs && geq_pair_irregular(g.lo, g.hi, i, p_lo)) out[0] = true;
    if (p_hi != p_lo && p_hi < pairs && geq_pair_irregular(g.lo, g.hi, i, p_hi)) out[1] = true;
}
// before the launch of round i: arm the words the device is going to write
static void geq_arm(ceno_hip_sumcheck* sc, int i) {
    auto& Gq = sc->geq;
    const int D = sc->d;
    const bool by_degree = sc->gen_rounds[(size_t)i].by_degree;
    for (auto& Q : Gq.comps) {
        if (Q.nv <= i) continue;
        const int Dc = by_degree ? Q.deg : D;  // (the rows keep their stride D)
        uint64_t* w = reinterpret_cast<uint64_t*>(Gq.h_q + (size_t)Q.slot * D);
        for (int k = 0; k < 2 * Dc; k++) __atomic_store_n(&w[k], MSG_INVALID, __ATOMIC_RELAXED);
        const uint64_t pairs = 1ull << (Q.nv - i - 1);
        for (const auto& g : Q.groups) {
            bool rows[2];
            geq_boundary_rows(g, i, pairs, rows);
            for (int sd = 0; sd < 2; sd++) {
                if (!rows[sd]) continue;
                uint64_t* b = reinterpret_cast<uint64_t*>(Gq.h_b + (size_t)(g.brow + sd) * D);
                for (int k = 0; k < 2 * Dc; k++) __atomic_store_n(&b[k], MSG_INVALID, __ATOMIC_RELAXED);
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
}
// after the launch: take the per-component values, complete every p_c and add p_c(1 .. D) to the message `h_out` (D ext); `r` = the
// challenge of round i - 1 (the running claims are evaluated at it)
static int geq_collect(ceno_hip_sumcheck* sc, int i, E2 r, uint64_t* h_out) {
    auto& Gq = sc->geq;
    const int D = sc->d;
    const bool by_degree = sc->gen_rounds[(size_t)i].by_degree;
    // Lagrange basis over the nodes 0 .. D at r (claims of this round), shared by every component
    E2 L[MAXD + 1];
    if (i > 0) {
        E2 pre[MAXD + 2], suf[MAXD + 2];
        pre[0] = e2_one();
        for (int j = 0; j <= D; j++) pre[j + 1] = pre[j] * (r - E2{(uint64_t)j, 0});
        suf[D + 1] = e2_one();
        for (int j = D; j >= 0; j--) suf[j] = suf[j + 1] * (r - E2{(uint64_t)j, 0});
        for (int t = 0; t <= D; t++) L[t] = e2_mul_base(pre[t] * suf[t + 1], Gq.lag_den_inv[(size_t)t]);
    }
    E2 msg[MAXD];
    for (int t = 0; t < D; t++) msg[t] = E2{h_out[2 * t], h_out[2 * t + 1]};

    for (auto& Q : Gq.comps) {
        if (Q.nv <= i) continue;
        // Dc: the length of the message this component's workgroups were instantiated for — the sumcheck's, or in a large round of a batch of
        // mixed degrees the component's own (its polynomial has no higher coefficients: the nodes Dc + 1 .. D follow by extrapolation)
        const int Dc = by_degree ? Q.deg : D;
        const auto& W = Gq.w[(size_t)Dc];
        const bool neg_lead = ((Dc - 1) & 1) != 0;  // the device multiplies the (f(0) - f(1)): the leading coefficient carries (-1)^(Dc-1)
        auto lead_of = [&](E2 raw) { return neg_lead ? e2_neg(raw) : raw; };
        // value at 0 and at Dc of the polynomial of degree Dc - 1 with values v[0 .. Dc-2] at 1 .. Dc-1 and leading coefficient a
        auto ends_from_values = [&](const E2* v, E2 a, E2& at0, E2& atD) {
            E2 h0 = e2_zero(), hD = e2_zero();
            for (int j = 0; j < Dc - 1; j++) {
                const E2 h = v[j] - e2_mul_base(a, W.pow_dm1[(size_t)j + 1]);
                h0 = h0 + e2_mul_base(h, W.wA0[(size_t)j]);
                hD = hD + e2_mul_base(h, W.wAD[(size_t)j]);
            }
            at0 = h0;  // a * 0^(Dc-1) = 0
            atD = hD + e2_mul_base(a, W.pow_dm1[(size_t)Dc]);
        };
        // (all Dc slots are written every round; slot Dc - 2 means something in the first round only — later it holds a by-product of the
        // waves with a boundary pair)
        const E2* qv = Gq.h_q + (size_t)Q.slot * D;
        TRY(sc_wait_words(sc, reinterpret_cast<const uint64_t*>(qv), 2 * Dc, "the quotient sums of an eq-factored component"));
        const E2 rt = Q.pt[(size_t)i];
        // ---- boundary part B(t) = t (c G)(t), t = 0 .. Dc ----
        E2 B[MAXD + 1];
        for (int t = 0; t <= Dc; t++) B[t] = e2_zero();
        const uint64_t pairs = 1ull << (Q.nv - i - 1);
        for (const auto& g : Q.groups) {
            bool rows[2];
            geq_boundary_rows(g, i, pairs, rows);
            for (int sd = 0; sd < 2; sd++) {
                if (!rows[sd]) continue;
                const E2* bv = Gq.h_b + (size_t)(g.brow + sd) * D;
                TRY(sc_wait_words(sc, reinterpret_cast<const uint64_t*>(bv), 2 * Dc, "the boundary values of an eq-factored component"));
                E2 at0, atD;
                ends_from_values(bv, lead_of(bv[Dc - 1]), at0, atD);
                for (int t = 1; t < Dc; t++) B[t] = B[t] + e2_mul_base(bv[t - 1], (uint64_t)t);
                B[Dc] = B[Dc] + e2_mul_base(atD, (uint64_t)Dc);
            }
        }
        // ---- quotient Q(t), t = 0 .. Dc ----
        E2 Qt[MAXD + 1];
        const E2 a = lead_of(qv[Dc - 1]);
        if (i == 0) {
            for (int t = 1; t < Dc; t++) Qt[t] = qv[t - 1];
            ends_from_values(qv, a, Qt[0], Qt[Dc]);
        } else {
            E2 claim = e2_zero();
            for (int t = 0; t <= D; t++) claim = claim + L[t] * Q.P[(size_t)t];
            for (int t = 1; t <= Dc - 2; t++) Qt[t] = qv[t - 1];
            Qt[0] = (claim - rt * Qt[1] - B[1]) * Q.inv1m[(size_t)i];  // claim = (1 - rt) Q(0) + rt Q(1) + B(1)
            E2 h1 = e2_zero(), hD = e2_zero();
            for (int j = 0; j < Dc - 1; j++) {
                const E2 h = Qt[j] - e2_mul_base(a, W.pow_dm1[(size_t)j]);
                h1 = h1 + e2_mul_base(h, W.wB1[(size_t)j]);
                hD = hD + e2_mul_base(h, W.wBD[(size_t)j]);
            }
            Qt[Dc - 1] = h1 + e2_mul_base(a, W.pow_dm1[(size_t)Dc - 1]);
            Qt[Dc] = hD + e2_mul_base(a, W.pow_dm1[(size_t)Dc]);
        }
        // ---- p_c(t) = eq(t, rt) Q(t) + B(t),  eq(t, rt) = (1 - t) + (2 t - 1) rt ----
        for (int t = 0; t <= Dc; t++) {
            const E2 one_minus_t = t <= 1 ? E2{(uint64_t)(1 - t), 0} : E2{gl::neg((uint64_t)(t - 1)), 0};
            const E2 two_t_minus_1 = t >= 1 ? E2{(uint64_t)(2 * t - 1), 0} : E2{gl::neg(1), 0};
            const E2 eq = one_minus_t + e2_mul_base(rt, two_t_minus_1.c0);
            Q.P[(size_t)t] = eq * Qt[t] + B[t];
        }
        for (int t = Dc + 1; t <= D; t++) {
            const auto& we = W.ext[(size_t)(t - Dc - 1)];
            E2 v = e2_zero();
            for (int j = 0; j <= Dc; j++) v = v + e2_mul_base(Q.P[(size_t)j], we[(size_t)j]);
            Q.P[(size_t)t] = v;
        }
        for (int t = 1; t <= D; t++) msg[t - 1] = msg[t - 1] + Q.P[(size_t)t];
    }
    for (int t = 0; t < D; t++) {
        h_out[2 * t] = msg[t].c0;
        h_out[2 * t + 1] = msg[t].c1;
    }
    return 0;
}

// ---- column blocks of a WIDE component ----
// A chip of the reference has 22 .. 96 witness columns under one (two, three) selectors and 60 .. 250 monomials, most of them
// selector x column (ceno_zkvm/src/instructions.rs:48-83, gkr_iop/src/gkr/layer/zerocheck_layer.rs:86-207).  The LDS stage holds 32 bytes per
// MLE and pair: at 64 pairs per tile — the geometry at which every lane of phase 2 owns a pair — about 21 MLEs fit beside the exchange block
// with three workgroups per CU; a 26-MLE chip used to drop to 32-pair tiles (half the lanes of phase 2 idle), an 86-MLE chip to the two-kernel
// path.  A sum of monomials splits freely: the component is cut into BLOCKS of at most `cap` MLEs, every block with the selectors of its
// groups, a term goes to a block that holds all its columns (columns of the product terms are clustered first, the selector x column terms fill
// up), and every block is a component of its own from here on — its own tiles, its own eq slot and running claim (the round polynomial of a
// chip is the sum of its blocks').  A column that two blocks need is staged by both and written by the first.
static unsigned gen_oversub() {
    const char* e = getenv("CENO_HIP_GEN_OVERSUB");  // workgroups per resident slot in the large rounds of an eq-factored batch (read per build)
    return e ? (unsigned)std::min(std::max(atoi(e), 1), 64) : 16u;
}
static int gen_by_degree_mode() {
    // per-degree launches of the large eq-factored rounds (sc_build_gen): 0 off, 1 where a round has more tiles than resident workgroups, 2 in
    // every round (tests).  Read per build.
    return getenv("CENO_HIP_GEN_BY_DEGREE") ? atoi(getenv("CENO_HIP_GEN_BY_DEGREE")) : 1;
}
// MLEs per column block for a block whose own polynomial has length `deg` in a sumcheck of length d (0: no column blocks)
static int gen_split_cap(ceno_hip_ctx* ctx, int d, int deg) {
    const char* off = getenv("CENO_HIP_GEN_SPLIT");  // 0: no column blocks (A/B, tests)
    if (off && atoi(off) == 0) return 0;
    const char* e = getenv("CENO_HIP_GEN_SPLIT_MLES");
    if (e && atoi(e) > 0) return std::max(atoi(e), 3);
    // MLEs (two units each) whose rows fit at 64 pairs per tile: beside the exchange block of the block's own kernel with three workgroups per
    // CU where its registers allow them, and beside the sumcheck's in the one launch of a small round
    return std::max(3, (int)(std::min(gen_stage_budget3(ctx, deg), gen_stage_budget(d)) / gen_stage_bytes(2, 6)));
}
// caps[deg]: gen_split_cap per block degree (3 .. d; all equal = the blocks are cut without regard to their degree)
static void gen_split_component(const ScClass& cl, const GenCompHost& C, const int* caps, int d, bool by_degree, std::vector<GenCompHost>& out) {
    const int ng = (int)C.gts.size();
    struct Blk {
        std::vector<int> mles;
        std::vector<char> has;  // by class-local id
        std::vector<std::vector<uint32_t>> gts;
        int deg = 3;
    };
    std::vector<Blk> blocks;
    const size_t km = cl.mles.size();
    auto deg_of = [d](uint32_t nf) { return std::min(d, std::max(3, 1 + (int)nf)); };
    // terms in clustering order: products first (widest first), then single columns, then constants
    struct Item { int g; uint32_t t; uint32_t nf; };
    std::vector<Item> items;
    size_t max_need = 0;
    int cap_min = 1 << 30;
    for (int g = 0; g < ng; g++)
        for (uint32_t t : C.gts[(size_t)g]) {
            const uint32_t nf = cl.h_to[t + 1] - cl.h_to[t];
            items.push_back(Item{g, t, nf});
            max_need = std::max<size_t>(max_need, (size_t)nf + (cl.h_co[g + 1] - cl.h_co[g]));
            cap_min = std::min(cap_min, caps[by_degree ? deg_of(nf) : d]);
        }
    if (items.empty() || (int)max_need > cap_min) {  // a single term does not fit a block: leave the component as it is
        out.push_back(C);
        return;
    }
    std::vector<char> is_common(km, 0);
    int n_common = 0;
    for (int g = 0; g < ng; g++)
        if (!C.gts[(size_t)g].empty())
            for (uint32_t k = cl.h_co[g]; k < cl.h_co[g + 1]; k++)
                if (!is_common[cl.h_ci[k]]) is_common[cl.h_ci[k]] = 1, n_common++;
    // blocks of EQUAL size rather than full blocks and a remainder (23 tables at 20 per block were 20 + 4: the small block staged the selector
    // for three columns): `cols` columns with s selectors in every block -> n = ceil(cols / (cap - s)) blocks of ceil(cols / n) + s tables
    auto balanced = [&](int cap, int cols) {
        if (cap > n_common + 1 && cols > 0) {
            const int nb = (cols + (cap - n_common) - 1) / (cap - n_common);
            cap = std::max((int)max_need, std::min(cap, (cols + nb - 1) / nb + n_common + 1));  // (+1: room for a column two blocks share)
        }
        return cap;
    };
    std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.nf > b.nf; });
    auto wanted = [&](const Item& it, std::vector<int>& w) {
        w.clear();
        for (uint32_t k = cl.h_co[it.g]; k < cl.h_co[it.g + 1]; k++) w.push_back((int)cl.h_ci[k]);
        for (uint32_t k = cl.h_to[it.t]; k < cl.h_to[it.t + 1]; k++) w.push_back((int)cl.h_ti[k]);
        std::sort(w.begin(), w.end());
        w.erase(std::unique(w.begin(), w.end()), w.end());
    };
    std::vector<int> w;
    // `cap_of(block)`: what the block may hold; `foreign_ok(block, it)`: may the item bring NEW columns into it
    auto place = [&](const Item& it, auto&& cap_of, auto&& foreign_ok, int deg_new) {
        wanted(it, w);
        int best = -1, best_need = 1 << 30;
        for (size_t b = 0; b < blocks.size(); b++) {
            int need = 0;
            for (int m : w) need += blocks[b].has[(size_t)m] ? 0 : 1;
            if (need > 0 && !foreign_ok(blocks[b])) continue;
            if ((int)blocks[b].mles.size() + need <= cap_of(blocks[b]) && need < best_need) best = (int)b, best_need = need;
        }
        if (best < 0) {
            blocks.emplace_back();
            blocks.back().has.assign(km, 0);
            blocks.back().gts.assign((size_t)ng, {});
            blocks.back().deg = deg_new;
            best = (int)blocks.size() - 1;
        }
        Blk& B = blocks[(size_t)best];
        for (int m : w)
            if (!B.has[(size_t)m]) {
                B.has[(size_t)m] = 1;
                B.mles.push_back(m);
            }
        B.gts[(size_t)it.g].push_back(it.t);
    };
    if (!by_degree) {
        const int cap = balanced(caps[d], (int)C.mles.size() - n_common);
        for (const Item& it : items) place(it, [&](const Blk&) { return cap; }, [](const Blk&) { return true; }, d);
    } else {
        // Blocks by DEGREE (the large rounds launch every block on the kernel of its own message length, sc_build_gen): the products cluster
        // first, every block as large as its kernel allows; a selector x column term joins the block that already stages its column; the columns
        // that no product reads go to blocks of length 3 only — in a block of length 5 such a term is evaluated at four points instead of two,
        // by waves that hold five accumulators — of equal size.
        size_t k = 0;
        for (; k < items.size() && items[k].nf >= 2; k++)
            place(items[k], [&](const Blk& B) { return caps[B.deg]; }, [](const Blk&) { return true; }, deg_of(items[k].nf));
        std::vector<char> covered(km, 0);
        int cols3 = 0;
        for (const Blk& B : blocks)
            for (int m : B.mles)
                if (!covered[(size_t)m]) {
                    covered[(size_t)m] = 1;
                    if (B.deg == 3 && !is_common[(size_t)m]) cols3++;
                }
        for (int m : C.mles)
            if (!covered[(size_t)m] && !is_common[(size_t)m]) cols3++;
        const int cap3 = balanced(caps[3], cols3);
        for (; k < items.size(); k++)
            place(items[k], [&](const Blk& B) { return B.deg == 3 ? std::max(cap3, (int)B.mles.size()) : caps[B.deg]; }, [](const Blk& B) { return B.deg == 3; }, 3);
    }
    std::vector<char> owned(km, 0);
    for (Blk& B : blocks) {
        GenCompHost S;
        S.cls = C.cls;
        S.mles = B.mles;
        std::sort(S.mles.begin(), S.mles.end());
        for (int m : S.mles) {
            S.writes.push_back(owned[(size_t)m] ? 0 : 1);
            owned[(size_t)m] = 1;
        }
        for (auto& ts : B.gts) std::sort(ts.begin(), ts.end());  // (the per-component code orders them widest first)
        S.gts = std::move(B.gts);
        out.push_back(std::move(S));
    }
}

static int sc_build_gen(ceno_hip_sumcheck* sc) {
    CENO_TIMED("sc_build_gen");
    ceno_hip_ctx* ctx = sc->ctx;
    if (getenv("CENO_HIP_PLAN_REPORT")) {
        std::lock_guard<PoolMutex> g(ctx->mu);
        ctx->plan_report = "[]";
    }
    // (a batch of three and more size classes pays for its tables at any size — wide batch of 48 chips at max_nv 8 / 10 / 12 / 13: 3.30 -> 3.05,
    // 3.98 -> 3.44, 4.70 -> 3.68, 5.06 -> 3.81 ms, tools/dev/min_log_sweep2.sh; with one or two classes the table build is the larger cost)
    const int sc_min_log = sc->classes.size() >= 3 ? std::min(gen_min_log(), gen_class_min_log()) : gen_min_log();
    if (sc->n < sc_min_log) return 0;
    // a single class covering all variables runs pipelined (tower layers, one chip's main sumcheck), where k_gen is off unless
    // CENO_HIP_GEN_PIPE_MIN_LOG asks for it: do not build tables nobody reads
    if (sc->classes.size() == 1 && sc->classes[0].nv == sc->n && !getenv("CENO_HIP_GEN_PIPE_MIN_LOG")) return 0;
    bool any = false;
    for (auto& cl : sc->classes) any = any || (!cl.dense && cl.nv >= sc_min_log);
    if (!any) return 0;
    std::vector<char> blob;
    auto append = [&blob](const void* data, size_t bytes) {
        const size_t off = (blob.size() + 15) & ~(size_t)15;
        blob.resize(off + std::max<size_t>(bytes, 16));
        if (bytes && data) memcpy(blob.data() + off, data, bytes);
        return off;
    };
    std::vector<GenCompHost> comps;
    bool geq_wanted = false;
    {
        const char* e = getenv("CENO_HIP_GEN_EQF");  // 0: declared eq tables are treated like any other common factor (A/B, tests)
        for (const EqDecl& dcl : sc->geq.decl) geq_wanted = geq_wanted || dcl.on;
        if (e && atoi(e) == 0) geq_wanted = false;
    }
    sc->geq.comps.clear();
    sc->geq.max_grid = 0;
    sc->geq.n_brows = 0;
    for (size_t ci = 0; ci < sc->classes.size(); ci++) {
        ScClass& cl = sc->classes[ci];
        if (cl.dense || cl.nv < gen_class_min_log()) continue;
        const int km = (int)cl.mles.size();
        static HostTimeSlot* const hs_comp = host_time_slot("sc_build_gen: components of the classes");
        static HostTimeSlot* const hs_split = host_time_slot("sc_build_gen: column blocks of the classes");
        static HostTimeSlot* const hs_rec = host_time_slot("sc_build_gen: component records of the classes");
        timespec ht = host_time_mark();
        // ---- connected components over class-local MLE ids (a group ties its common factors and its terms' factors) ----
        std::vector<int> uf(km);
        std::iota(uf.begin(), uf.end(), 0);
        std::function<int(int)> find = [&](int x) { return uf[x] == x ? x : uf[x] = find(uf[x]); };
        std::vector<char> used(km, 0);
        const int ng = cl.n_groups;
        bool ok = true;
        for (int g = 0; g < ng && ok; g++) {
            int anchor = -1;
            auto touch = [&](uint32_t m) {
                used[m] = 1;
                if (anchor < 0) anchor = (int)m;
                else uf[find((int)m)] = find(anchor);
            };
            const bool free_group = cl.h_co[g + 1] == cl.h_co[g];
            for (uint32_t k = cl.h_co[g]; k < cl.h_co[g + 1]; k++) touch(cl.h_ci[k]);
            if (cl.h_co[g + 1] - cl.h_co[g] > 8) ok = false;
            for (uint32_t ti = cl.h_gto[g]; ti < cl.h_gto[g + 1]; ti++) {
                const uint32_t t = cl.h_gt[ti];
                if (free_group) anchor = -1;  // ungrouped terms share nothing: every term may be its own component
                if (cl.h_to[t + 1] - cl.h_to[t] > 8) ok = false;
                for (uint32_t k = cl.h_to[t]; k < cl.h_to[t + 1]; k++) touch(cl.h_ti[k]);
            }
        }
        if (!ok) continue;  // a term / group too wide for the flat records: two-kernel path
        std::map<int, int> comp_of_root;
        std::vector<GenCompHost> mine;
        for (int m = 0; m < km; m++) {
            if (!used[m]) continue;
            const int r = find(m);
            auto it = comp_of_root.find(r);
            if (it == comp_of_root.end()) {
                it = comp_of_root.emplace(r, (int)mine.size()).first;
                mine.emplace_back();
                mine.back().cls = (int)ci;
            }
            mine[it->second].mles.push_back(m);
        }
        // the terms of every component, group by group (a common-factor group lies in one component entirely; the ungrouped terms of the
        // class are spread over theirs), then the column blocks of the wide ones
        // (one pass over the terms: a batch of ~50 chips is ~50 components per class, and a pass per component made this quadratic)
        {
            std::vector<int> comp_of((size_t)km, -1);
            for (size_t c = 0; c < mine.size(); c++) {
                for (int m : mine[c].mles) comp_of[(size_t)m] = (int)c;
                mine[c].gts.assign((size_t)ng, {});
            }
            for (int g = 0; g < ng; g++) {
                const bool free_group = cl.h_co[g + 1] == cl.h_co[g];
                for (uint32_t ti = cl.h_gto[g]; ti < cl.h_gto[g + 1]; ti++) {
                    const uint32_t t = cl.h_gt[ti];
                    const uint32_t probe = cl.h_to[t + 1] > cl.h_to[t] ? cl.h_ti[cl.h_to[t]] : (free_group ? UINT32_MAX : cl.h_ci[cl.h_co[g]]);
                    if (probe != UINT32_MAX && comp_of[probe] >= 0) mine[(size_t)comp_of[probe]].gts[(size_t)g].push_back(t);
                }
            }
        }
        // (CENO_HIP_GEN_SPLIT_MIN_LOG: smallest class that gets column blocks.  Measured: blocks for every class are best — the wide batch 60.2 ms
        // against 69.9 with blocks from 2^18 rows up and 96.8 without; the latency-bound batch of the 2^20-cycle shard does not care, 2.03-2.19 ms)
        static const int split_min_log = getenv("CENO_HIP_GEN_SPLIT_MIN_LOG") ? atoi(getenv("CENO_HIP_GEN_SPLIT_MIN_LOG")) : 0;
        ht = host_time_add(hs_comp, ht);
        if (cl.nv >= split_min_log && gen_split_cap(ctx, sc->d, sc->d) > 0) {
            const bool by_degree = geq_wanted && sc->d >= 3 && gen_by_degree_mode() >= 1;
            int caps[MAXD + 1] = {};
            int cap_min = 1 << 30;
            for (int dg = std::min(3, sc->d); dg <= sc->d; dg++) {
                caps[dg] = gen_split_cap(ctx, sc->d, by_degree ? dg : sc->d);
                cap_min = std::min(cap_min, caps[dg]);
            }
            std::vector<GenCompHost> cut;
            for (auto& C : mine) {
                if ((int)C.mles.size() <= cap_min) cut.push_back(std::move(C));
                else gen_split_component(cl, C, caps, sc->d, by_degree, cut);
            }
            mine = std::move(cut);
        }
        GenCompHost fold_only;  // tables no term reads: folded only
        fold_only.cls = (int)ci;
        for (int m = 0; m < km; m++)
            if (!used[m]) fold_only.mles.push_back(m);
        // ---- per component: units, terms, groups in both layouts ----
        ht = host_time_add(hs_split, ht);
        std::vector<GeqComp> geq_new;  // eq-factored components of this class (committed with the class)
        const int brows_before = sc->geq.n_brows;
        for (auto& C : mine) {
            std::map<int, int> pos;  // class-local id -> component-local id
            for (size_t k = 0; k < C.mles.size(); k++) pos[C.mles[k]] = (int)k;
            std::vector<uint16_t> unit[3];
            std::vector<char> is_base(C.mles.size());
            size_t u0 = 0, u1 = 0;
            for (size_t k = 0; k < C.mles.size(); k++) {
                is_base[k] = !sc->mles[cl.mles[C.mles[k]]].cur_ext;
                unit[0].push_back((uint16_t)u0);
                unit[1].push_back((uint16_t)u1);
                unit[2].push_back((uint16_t)k);
                u0 += 2;
                u1 += is_base[k] ? 1 : 2;
            }
            C.units[0] = u0;
            C.units[1] = u1;
            std::vector<GenTerm> terms[3];
            std::vector<GenGroup> groups[3];
            std::vector<int> grp_common;  // per emitted group: global id of its ONE common factor, or -1
            C.base0_ok = true;
            size_t max_terms = 0;
            for (int g = 0; g < ng; g++) {
                std::vector<uint32_t> ts = C.gts[(size_t)g];  // the terms of this group that this component evaluates
                if (ts.empty()) continue;
                // widest terms first: the round-robin split over the waves stays balanced
                std::stable_sort(ts.begin(), ts.end(), [&](uint32_t a, uint32_t b) { return cl.h_to[a + 1] - cl.h_to[a] > cl.h_to[b + 1] - cl.h_to[b]; });
                max_terms = std::max(max_terms, ts.size());
                for (int lay = 0; lay < 3; lay++) {
                    GenGroup G{};
                    G.term_begin = (uint32_t)terms[lay].size();
                    for (uint32_t t : ts) {
                        GenTerm T{};
                        T.c = cl.h_coeffs[t];
                        T.nf = cl.h_to[t + 1] - cl.h_to[t];
                        for (uint32_t k = 0; k < T.nf; k++) {
                            const int cm = pos[(int)cl.h_ti[cl.h_to[t] + k]];
                            T.idx8 |= (uint64_t)(unit[lay][cm] & 0xff) << (8 * k);
                            if (lay == 1 && !is_base[cm]) C.base0_ok = false;  // an extension factor inside a term: no base-field product
                        }
                        terms[lay].push_back(T);
                    }
                    G.term_end = (uint32_t)terms[lay].size();
                    G.n_common = cl.h_co[g + 1] - cl.h_co[g];
                    if (lay == 0) grp_common.push_back(G.n_common == 1 ? cl.mles[cl.h_ci[cl.h_co[g]]] : -1);
                    for (uint32_t k = 0; k < G.n_common; k++) {
                        const int cm = pos[(int)cl.h_ci[cl.h_co[g] + k]];
                        G.common8 |= (uint64_t)(unit[lay][cm] & 0xff) << (8 * k);
                        if (lay == 1 && is_base[cm]) G.base_mask |= 1u << k;
                    }
                    groups[lay].push_back(G);
                }
            }
            C.n_groups = (int)groups[0].size();
            C.n_terms = (int)terms[0].size();
            // ---- eq-factored form: every group = ONE common factor, a table declared as eq(., point) on a row range, one point for the
            // whole component, and no term of the component reads such a table as an ordinary factor ----
            if (geq_wanted && sc->d >= 3 && C.n_groups > 0) {
                bool eq_ok = true;
                const EqDecl* first = nullptr;
                for (int gid : grp_common) {
                    if (gid < 0 || !sc->geq.decl[gid].on) { eq_ok = false; break; }
                    const EqDecl& dcl = sc->geq.decl[gid];
                    if (!first) first = &dcl;
                    else if (dcl.pt.size() != first->pt.size() || memcmp(dcl.pt.data(), first->pt.data(), dcl.pt.size() * sizeof(E2)) != 0) eq_ok = false;
                }
                for (const GenTerm& T : terms[0]) {
                    if ((int)T.nf > sc->d - 1) eq_ok = false;
                    for (uint32_t k = 0; k < T.nf && eq_ok; k++) {
                        const unsigned u = (unsigned)((T.idx8 >> (8 * k)) & 0xff);
                        const size_t m = u / 2;  // layout 0 gives table m the unit 2m (a walk over the component's tables per factor was its largest loop)
                        if (m < C.mles.size() && unit[0][m] == u && sc->geq.decl[cl.mles[C.mles[m]]].on) eq_ok = false;
                    }
                }
                if (eq_ok && first) {
                    for (const E2& v : first->pt)
                        if (v.c0 == 1 && v.c1 == 0) eq_ok = false;  // 1 - rt_i = 0: the running claim does not determine Q(0)
                }
                if (eq_ok && first && (int)first->pt.size() == cl.nv) {
                    GeqComp Q;
                    Q.nv = cl.nv;
                    Q.pt = first->pt;
                    // the length of the component's OWN round polynomial: eq x (products of at most nf columns); the kernels exist from 3 on
                    uint32_t nf_max = 0;
                    for (const GenTerm& T : terms[0]) nf_max = std::max(nf_max, T.nf);
                    Q.deg = C.deg = std::min(sc->d, std::max(3, 1 + (int)nf_max));
                    Q.slot = (int)sc->geq.comps.size() + (int)geq_new.size();
                    for (size_t k = 0; k < grp_common.size(); k++) {
                        const EqDecl& dcl = sc->geq.decl[grp_common[k]];
                        const int brow = sc->geq.n_brows;
                        sc->geq.n_brows += 2;
                        Q.groups.push_back(GeqComp::Grp{dcl.lo, dcl.hi, brow});
                        for (int lay = 0; lay < 3; lay++) {
                            groups[lay][k].eq = 1;
                            groups[lay][k].brow = (uint32_t)brow;
                            groups[lay][k].lo = dcl.lo;
                            groups[lay][k].hi = dcl.hi;
                        }
                    }
                    C.geq = (int)geq_new.size();  // index into geq_new for now; fixed up when the class is accepted
                    geq_new.push_back(std::move(Q));
                }
            }
            {   // relative cost of a pair (the component-aligned launch splits its workgroups by it)
                double c2 = 0.0;
                const double w_nl = getenv("CENO_HIP_GEN_W_NL") ? atof(getenv("CENO_HIP_GEN_W_NL")) : 1.0;      // (calibration knobs)
                const double w_fold = getenv("CENO_HIP_GEN_W_FOLD") ? atof(getenv("CENO_HIP_GEN_W_FOLD")) : 1.0;
                for (const GenTerm& T : terms[0])
                    c2 += T.nf > 1 ? w_nl * (2.0 + (T.nf - 1) * (double)std::max(sc->d - 2, 1) + ((int)T.nf == sc->d - 1 ? T.nf - 1.0 : 0.0)) : (double)std::max(sc->d - 2, 1);
                C.pair_cost[0] = 2.0 * w_fold * (double)C.mles.size();
                C.pair_cost[1] = c2 + (double)(sc->d - 1) * C.n_groups;
            }
            // geometry: waves sharing a group's terms, pairs per tile, shrunk until the staged rows fit the LDS budget
            C.wt_log = max_terms >= 4 ? 2 : (max_terms >= 2 ? 1 : 0);
            C.tp_log = 8 - C.wt_log;
            while (C.tp_log > 6 && gen_stage_bytes(C.units[0], C.tp_log) > gen_stage_budget3(ctx, sc->d)) C.tp_log--;
            while (C.tp_log > 4 && gen_stage_bytes(C.units[0], C.tp_log) > gen_stage_budget(sc->d)) C.tp_log--;
            if (gen_stage_bytes(C.units[0], C.tp_log) > gen_stage_budget(sc->d) || C.units[0] > GEN_MAX_UNITS) { ok = false; break; }
            C.wt_log = std::min(2, 8 - C.tp_log);
            for (int lay = 0; lay < 3; lay++) {
                C.off_terms[lay] = append(terms[lay].data(), terms[lay].size() * sizeof(GenTerm));
                C.off_groups[lay] = append(groups[lay].data(), groups[lay].size() * sizeof(GenGroup));
                C.off_unit[lay] = append(unit[lay].data(), unit[lay].size() * sizeof(uint16_t));
            }
        }
        (void)host_time_add(hs_rec, ht);
        if (!ok) {
            sc->geq.n_brows = brows_before;
            continue;
        }
        for (auto& C : mine)
            if (C.geq >= 0) C.geq += (int)sc->geq.comps.size();
        for (auto& Q : geq_new) sc->geq.comps.push_back(std::move(Q));
        if (!fold_only.mles.empty()) {
            std::vector<uint16_t> zeros(fold_only.mles.size(), 0);
            fold_only.off_unit[0] = fold_only.off_unit[1] = fold_only.off_unit[2] = append(zeros.data(), zeros.size() * sizeof(uint16_t));
            fold_only.base0_ok = true;
            mine.push_back(fold_only);
        }
        cl.gen = true;
        comps.insert(comps.end(), mine.begin(), mine.end());
    }
    if (getenv("CENO_HIP_PLAN_REPORT")) {
        std::string rep = "[";
        for (size_t ci = 0; ci < sc->classes.size(); ci++) {
            const ScClass& cl = sc->classes[ci];
            int n_comp = 0, n_eq = 0, staged = 0, tp_min = 99, tp_max = 0;
            size_t units_max = 0;
            for (const auto& C : comps) {
                if (C.cls != (int)ci || C.n_groups == 0) continue;
                n_comp++;
                n_eq += C.geq >= 0 ? 1 : 0;
                staged += (int)C.mles.size();
                tp_min = std::min(tp_min, C.tp_log);
                tp_max = std::max(tp_max, C.tp_log);
                units_max = std::max(units_max, C.units[0]);
            }
            const char* path = cl.dense ? "dense" : (!cl.gen ? "two-kernel" : (n_comp > 0 && n_eq == n_comp ? "eq-factored" : (n_eq > 0 ? "eq-factored+generic" : "generic")));
            char buf[512];
            snprintf(buf, sizeof buf, "%s{\"num_vars\": %d, \"tables\": %zu, \"terms\": %zu, \"groups\": %d, \"path\": \"%s\", \"components\": %d, "
                     "\"eq_components\": %d, \"tables_staged\": %d, \"pairs_per_tile_log2\": [%d, %d], \"max_stage_units\": %zu}",
                     ci ? ", " : "", cl.nv, cl.mles.size(), cl.terms.size(), cl.n_groups, path, n_comp, n_eq, staged, n_comp ? tp_min : 0, tp_max, units_max);
            rep += buf;
        }
        rep += "]";
        std::lock_guard<PoolMutex> g(ctx->mu);
        ctx->plan_report = rep;
    }
    if (comps.empty()) return 0;
    CENO_TIMED("sc_build_gen: from the slot schedule on");
    // ---- slot schedule of every round (simulation of the buffer ping-pong of sc_round / sc_advance) ----
    size_t slots_per_round = 0;
    for (auto& C : comps) {
        C.slot_off = slots_per_round;
        slots_per_round += C.mles.size();
    }
    const int n = sc->n;
    const size_t off_slots = append(nullptr, (size_t)n * slots_per_round * sizeof(MleSlot));
    {
        std::vector<ScMle> sim = sc->mles;
        for (int i = 0; i < n; i++) {
            MleSlot* row = reinterpret_cast<MleSlot*>(blob.data() + off_slots) + (size_t)i * slots_per_round;
            for (auto& C : comps) {
                const ScClass& cl = sc->classes[C.cls];
                if (cl.nv <= i) continue;
                for (size_t k = 0; k < C.mles.size(); k++) {
                    const ScMle& M = sim[cl.mles[C.mles[k]]];
                    row[C.slot_off + k] = MleSlot{M.cur, C.writes.empty() || C.writes[k] ? M.buf[M.which] : nullptr, M.cur_ext, 0};
                }
            }
            if (i > 0)
                for (auto& cl : sc->classes)
                    if (cl.nv > i)
                        for (int j : cl.mles) {
                            ScMle& M = sim[j];
                            M.cur = M.buf[M.which];
                            M.cur_ext = 1;
                            M.which ^= 1;
                        }
        }
    }
    CENO_TIMED("sc_build_gen: from the component lists on");
    // ---- component lists per round ----
    for (size_t k = 0; k < comps.size(); k++)
        if (comps[k].geq >= 0) sc->geq.comps[comps[k].geq].comp = (int)k;
    const bool geq_on = !sc->geq.comps.empty();
    if (geq_on) {
        // 1 / (1 - pt_i) for every eq component and round: ONE inversion (prefix products)
        std::vector<E2*> where;
        std::vector<E2> vals;
        for (auto& Q : sc->geq.comps) {
            Q.inv1m.assign(Q.pt.size(), e2_zero());
            for (size_t i = 0; i < Q.pt.size(); i++) {
                where.push_back(&Q.inv1m[i]);
                vals.push_back(e2_one() - Q.pt[i]);
            }
        }
        std::vector<E2> pre(vals.size());
        E2 run = e2_one();
        for (size_t k = 0; k < vals.size(); k++) {
            pre[k] = run;
            run = run * vals[k];
        }
        E2 inv = e2_inv(run);
        for (size_t k = vals.size(); k-- > 0;) {
            *where[k] = inv * pre[k];
            inv = inv * vals[k];
        }
    }
    sc->gen_rounds.assign(n, GenRound{});
    std::vector<size_t> comp_fix;  // offsets of GenComp records whose pointers still hold blob offsets
    for (int i = 0; i < n; i++) {
        GenRound& R = sc->gen_rounds[i];
        std::vector<GenComp> list, list_eq;
        std::vector<double> weight_eq;
        std::vector<int> deg_eq;         // per entry of list_eq: the component's own degree (0: folded only)
        std::vector<size_t> stage_eq;    // ... and the bytes of its LDS stage
        bool base0 = i == 0;
        for (auto& C : comps)
            if (sc->classes[C.cls].nv > i && C.n_groups > 0 && !C.base0_ok) base0 = false;
        // first round of an eq-factored batch over base-field columns: the direct kernel (k_eq_base0, no LDS stage), third record layout
        const bool no_direct0 = getenv("CENO_HIP_EQ_DIRECT0") && atoi(getenv("CENO_HIP_EQ_DIRECT0")) == 0;  // A/B, tests: the staged kernel (read per call)
        const bool direct0 = geq_on && i == 0 && base0 && !no_direct0;
        R.direct0 = direct0;
        unsigned tiles = 0;
        for (auto& C : comps) {
            const ScClass& cl = sc->classes[C.cls];
            if (cl.nv <= i) continue;
            const int lay = direct0 && C.geq >= 0 ? 2 : (base0 ? 1 : 0);
            GenComp G{};
            G.slots = reinterpret_cast<const MleSlot*>(off_slots + ((size_t)i * slots_per_round + C.slot_off) * sizeof(MleSlot));
            G.terms = reinterpret_cast<const GenTerm*>(C.off_terms[lay]);
            G.groups = reinterpret_cast<const GenGroup*>(C.off_groups[lay]);
            G.unit = reinterpret_cast<const uint16_t*>(C.off_unit[lay]);
            G.pairs = 1ull << (cl.nv - i - 1);
            G.n_mles = (uint32_t)C.mles.size();
            G.n_groups = (uint32_t)C.n_groups;
            G.tp_log = (uint32_t)C.tp_log;
            G.wt_log = (uint32_t)C.wt_log;
            G.fold = i > 0 ? 1u : 0u;
            G.tile_begin = tiles;
            G.n_tiles = (uint32_t)((G.pairs + ((1ull << C.tp_log) - 1)) >> C.tp_log);
            if (i == 0 && C.n_groups == 0) continue;  // nothing to fold and nothing to evaluate in the first round
            if (geq_on && (C.geq >= 0 || C.n_groups == 0)) {
                // the eq-factored launch of the round (fold-only components ride along: no second launch for them)
                G.tile_begin = 0;
                double w = (double)G.n_tiles * C.pair_cost[0] * (i > 0 ? 1.0 : 0.1);
                if (C.geq >= 0) {
                    const GeqComp& Q = sc->geq.comps[C.geq];
                    G.eqf = 1u | (i == 0 ? 2u : 0u);
                    G.eq_slot = (uint32_t)Q.slot;
                    G.shift = (uint32_t)i;
                    G.rt = Q.pt[i];
                    G.inv1m = Q.inv1m[i];
                    // pairs that meet the row range of some group: [floor(lo / 2^(i+1)), ceil(hi / 2^(i+1)))
                    uint64_t pb = ~0ull, pe = 0;
                    for (const auto& g : Q.groups) {
                        if (g.hi <= g.lo) continue;
                        pb = std::min(pb, g.lo >> (i + 1));
                        pe = std::max(pe, ((g.hi - 1) >> (i + 1)) + 1);
                    }
                    if (pe > pb) {
                        G.p2_tile_begin = (uint32_t)(pb >> C.tp_log);
                        G.p2_tile_end = (uint32_t)std::min<uint64_t>(((pe - 1) >> C.tp_log) + 1, G.n_tiles);
                    }
                    w += (double)(G.p2_tile_end - G.p2_tile_begin) * C.pair_cost[1];
                }
                weight_eq.push_back(w * (double)(1u << C.tp_log));
                deg_eq.push_back(C.geq >= 0 ? C.deg : 0);
                stage_eq.push_back(C.geq >= 0 ? gen_stage_bytes(C.units[lay == 2 ? 1 : lay], C.tp_log) : 0);
                list_eq.push_back(G);
                continue;
            }
            tiles += G.n_tiles;
            if (C.n_groups > 0) {
                R.has_terms = true;
                R.stage_bytes = std::max(R.stage_bytes, gen_stage_bytes(C.units[lay], C.tp_log));
            }
            list.push_back(G);
        }
        R.base0 = base0;
        R.n_comps = (int)list.size();
        R.total_tiles = tiles;
        R.off_comps = append(list.data(), list.size() * sizeof(GenComp));
        for (size_t k = 0; k < list.size(); k++) comp_fix.push_back(R.off_comps + k * sizeof(GenComp));
        // one component-aligned launch over `sel` (indices into list_eq) with the kernels instantiated for messages of length Dl
        auto emit_eq_launch = [&](const std::vector<size_t>& sel, int Dl) {
            GenRound::EqLaunch EL;
            EL.D = Dl;
            std::vector<GenComp> lst;
            std::vector<double> wts;
            for (size_t k : sel) {
                lst.push_back(list_eq[k]);
                wts.push_back(weight_eq[k]);
                EL.stage_bytes = std::max(EL.stage_bytes, stage_eq[k]);
            }
            // component-aligned workgroup split: one workgroup per tile while all of them are resident at once, else every component
            // gets one workgroup and the rest in proportion to its estimated work
            uint64_t total = 0;
            double wsum = 0.0;
            for (size_t k = 0; k < lst.size(); k++) {
                total += lst[k].n_tiles;
                wsum += wts[k];
            }
            const unsigned cap = std::max<unsigned>(direct0 ? eq_base0_resident_cap(ctx, Dl) : gen_resident_cap(ctx, Dl, base0, EL.stage_bytes), (unsigned)lst.size());
            // small rounds -> one workgroup per tile and SLOT (the four waves of a workgroup walking 4-8 terms each, one pair per lane, were
            // the longest phase of such a round: CENO_HIP_GEN_PHASE_DBG=1).  CENO_HIP_EQ_SLOTS=0: off (A/B, read per build)
            const bool no_slots = getenv("CENO_HIP_EQ_SLOTS") && atoi(getenv("CENO_HIP_EQ_SLOTS")) == 0;
            // (one workgroup per TILE and slot as long as the whole launch is resident at once: a workgroup then has one tile, as before)
            const bool slots = !no_slots && i > 0 && !direct0 && Dl >= 3 && total * (uint64_t)Dl <= cap;
            EL.slots = slots;
            unsigned wg = 0;
            for (size_t k = 0; k < lst.size(); k++) {
                GenComp& G = lst[k];
                unsigned cnt = G.n_tiles;
                if (slots && G.n_groups > 0) {
                    cnt = G.n_tiles * (unsigned)Dl;
                    G.eqf |= 4u;
                    G.wg_begin = wg;
                    G.wg_count = cnt;
                    wg += cnt;
                    continue;
                }
                if (total > cap) {
                    // more workgroups than are resident at once, each with 1 / oversub of a resident slot's share: the dispatcher hands the
                    // next one to whichever slot frees up, so a component whose cost the estimate got wrong no longer sets the launch's time
                    // (with one workgroup per slot a 2x error in the weight of the product terms cost the wide batch 106 -> 157 ms,
                    // tools/dev/wide_sweep.sh) — and the tail of the launch is 1 / oversub of a slot's time
                    const double share = wsum > 0 ? wts[k] / wsum : 0.0;
                    // ... as long as a workgroup still has ~8 tiles to amortise its epilogue over (a row, an atomic, a share of the last workgroup's
                    // sum): the medium rounds of a small batch lost 0.13 ms of 1.9 to sixteen workgroups per slot (tools/dev/shard_ab.sh)
                    const unsigned os = (unsigned)std::min<uint64_t>(gen_oversub(), std::max<uint64_t>(1, total / ((uint64_t)cap * 8)));
                    const unsigned budget = std::max(cap * os, (unsigned)lst.size());
                    cnt = 1u + (unsigned)(share * (double)(budget - (unsigned)lst.size()));
                    cnt = std::min(cnt, G.n_tiles);
                }
                G.wg_begin = wg;
                G.wg_count = std::max(cnt, 1u);
                wg += G.wg_count;
            }
            if (i == 1 && getenv("CENO_HIP_PLAN_REPORT") && atoi(getenv("CENO_HIP_PLAN_REPORT")) >= 2) {
                fprintf(stderr, "[ceno_hip] eq launch of round 1 at degree %d: cap %u, %zu components, %u workgroups, stage %zu B, LDS per workgroup %zu B\n", Dl, cap,
                        lst.size(), wg, EL.stage_bytes, gen_lds_bytes(Dl, EL.stage_bytes));
                for (size_t k = 0; k < lst.size(); k++)
                    fprintf(stderr, "[ceno_hip]   comp %3zu: pairs 2^%d mles %3u groups %u tp_log %u tiles %7u p2 tiles %7u weight %.3e wgs %4u weight/wg %.3e\n", k,
                            (int)(63 - __builtin_clzll(lst[k].pairs)), lst[k].n_mles, lst[k].n_groups, lst[k].tp_log, lst[k].n_tiles,
                            lst[k].p2_tile_end - lst[k].p2_tile_begin, wts[k], lst[k].wg_count, wts[k] / lst[k].wg_count);
            }
            EL.grid = wg;
            sc->geq.max_grid = std::max(sc->geq.max_grid, wg);
            EL.n_comps = (int)lst.size();
            EL.off_comps = append(lst.data(), lst.size() * sizeof(GenComp));
            for (size_t k = 0; k < lst.size(); k++) comp_fix.push_back(EL.off_comps + k * sizeof(GenComp));
            // workgroup -> component, one scalar load (walking the list cost the last of 24 components ~10 us of dependent loads in
            // every round: CENO_HIP_GEN_PHASE_DBG=1)
            std::vector<uint16_t> wg_comp(wg);
            for (size_t k = 0; k < lst.size(); k++)
                for (unsigned x = 0; x < lst[k].wg_count; x++) wg_comp[lst[k].wg_begin + x] = (uint16_t)k;
            EL.off_wg_comp = append(wg_comp.data(), wg_comp.size() * sizeof(uint16_t));
            R.eq.push_back(EL);
        };
        if (!list_eq.empty()) {
            // A batch's sumcheck has the length of its LONGEST polynomial (degree 5 for the wide batch: eq x four columns) while most column
            // blocks hold selector x column terms only — a polynomial of length 3 (clamped: 2 + 1) that the kernel for length 5 evaluates at
            // four points instead of two, with the registers of five accumulators (two waves per SIMD).  In the LARGE rounds (more tiles than
            // resident workgroups: throughput, not latency) the components are launched per degree, each with the kernel of its own length;
            // the host completes every component's polynomial at ITS length and extends it to the sumcheck's nodes (geq_collect).  The small
            // rounds keep the one launch.  CENO_HIP_GEN_BY_DEGREE=0: off; =2: in every round (tests).  Read per build.
            const int by_deg_mode = gen_by_degree_mode();
            uint64_t total = 0;
            size_t stage_all = 0;
            int deg_min = sc->d;
            for (size_t k = 0; k < list_eq.size(); k++) {
                total += list_eq[k].n_tiles;
                stage_all = std::max(stage_all, stage_eq[k]);
                if (deg_eq[k] > 0) deg_min = std::min(deg_min, deg_eq[k]);
            }
            const unsigned cap_all = direct0 ? eq_base0_resident_cap(ctx, sc->d) : gen_resident_cap(ctx, sc->d, base0, stage_all);
            R.by_degree = deg_min < sc->d && (by_deg_mode >= 2 || (by_deg_mode == 1 && total > cap_all));
            std::vector<size_t> sel;
            if (!R.by_degree) {
                for (size_t k = 0; k < list_eq.size(); k++) sel.push_back(k);
                emit_eq_launch(sel, sc->d);
            } else {
                bool first = true;
                for (int Dl = sc->d; Dl >= 3; Dl--) {  // (longest first; the folded-only tables ride with the first launch)
                    sel.clear();
                    for (size_t k = 0; k < list_eq.size(); k++)
                        if (deg_eq[k] == Dl || (first && deg_eq[k] == 0)) sel.push_back(k);
                    bool any = false;
                    for (size_t k : sel) any = any || deg_eq[k] == Dl;
                    if (!any) continue;
                    emit_eq_launch(sel, Dl);
                    first = false;
                }
            }
        }
    }
    CENO_TIMED("sc_build_gen: allocation + upload");
    // ---- one device allocation, pointers fixed up, one copy from pinned staging ----
    void* d = nullptr;
    TRY(sc_dev_alloc(sc, blob.size(), &d));
    sc->d_gen = (char*)d;
    for (size_t off : comp_fix) {
        GenComp* G = reinterpret_cast<GenComp*>(blob.data() + off);
        G->slots = reinterpret_cast<const MleSlot*>(sc->d_gen + reinterpret_cast<size_t>(G->slots));
        G->terms = reinterpret_cast<const GenTerm*>(sc->d_gen + reinterpret_cast<size_t>(G->terms));
        G->groups = reinterpret_cast<const GenGroup*>(sc->d_gen + reinterpret_cast<size_t>(G->groups));
        G->unit = reinterpret_cast<const uint16_t*>(sc->d_gen + reinterpret_cast<size_t>(G->unit));
    }
    void *hb = nullptr, *db = nullptr;
    TRY(ctx_pinned_alloc(ctx, blob.size(), &hb, &db));
    sc->h_gen = hb;
    memcpy(hb, blob.data(), blob.size());
    HIP_TRY(ctx, hipMemcpyAsync(d, hb, blob.size(), hipMemcpyHostToDevice, sc->st));
    sc->gen_comps = std::move(comps);
    sc->gen_on = true;
    if (geq_on) TRY(geq_prepare(sc));
    return 0;
}

struct EqDeclArgs {
    int n;
    const int* mle_idx;
    const uint64_t* const* points;
    const size_t *lo, *hi;
};
static int sc_build(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan, hipStream_t st,
                    ceno_hip_sumcheck** out, SetupJob* defer_setup = nullptr, const EqDeclArgs* eqd = nullptr) {
    CHECK_ARG(ctx, mles && plan && out, "NULL argument");
    const int n = plan->max_num_vars, d = plan->max_degree;
    CHECK_ARG(ctx, n >= 0 && n < 40, "max_num_vars %d out of range", n);
    CHECK_ARG(ctx, d >= 1 && d <= MAXD, "max_degree %d unsupported (1..%d)", d, MAXD);
    CHECK_ARG(ctx, plan->num_mles >= 1 && plan->num_terms >= 1, "empty plan");
    CENO_TIMED("sc_build (whole)");
    auto* sc = new ceno_hip_sumcheck();
    sc->ctx = ctx;
    sc->st = st;
    sc->n = n;
    sc->d = d;
    sc->mles.resize(plan->num_mles);
    for (int j = 0; j < plan->num_mles; j++) {
        if (!mles[j]) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "mle %d is NULL", j); }
        if (mles[j]->num_vars > n) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "mle %d has %d vars > max_num_vars %d", j, mles[j]->num_vars, n); }
        sc->mles[j].cur = mles[j]->d;
        sc->mles[j].cur_ext = mles[j]->is_ext;
        sc->mles[j].nv = mles[j]->num_vars;
    }
    sc->geq.decl.assign((size_t)plan->num_mles, EqDecl{});
    for (int k = 0; eqd && k < eqd->n; k++) {
        const int j = eqd->mle_idx[k];
        if (j < 0 || j >= plan->num_mles || !eqd->points[k]) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "eq declaration %d: bad table index or point", k); }
        EqDecl& dcl = sc->geq.decl[(size_t)j];
        const int nv = sc->mles[j].nv;
        dcl.pt.resize((size_t)nv);
        for (int i = 0; i < nv; i++) {
            dcl.pt[(size_t)i] = E2{eqd->points[k][2 * i], eqd->points[k][2 * i + 1]};
            if (dcl.pt[(size_t)i].c0 >= gl::P || dcl.pt[(size_t)i].c1 >= gl::P) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "eq declaration %d: point is not canonical", k); }
        }
        dcl.lo = std::min<uint64_t>(eqd->lo[k], (uint64_t)1 << nv);
        dcl.hi = std::min<uint64_t>(eqd->hi[k], (uint64_t)1 << nv);
        if (dcl.hi <= dcl.lo) dcl.lo = dcl.hi = 0;
        dcl.on = sc->mles[j].cur_ext != 0 && nv >= 1;
        if (dcl.on && getenv("CENO_HIP_EQ_VERIFY") && atoi(getenv("CENO_HIP_EQ_VERIFY")) != 0) {
            // the caller vouches for a declaration; this switch (tests, bring-up of a new caller) checks it on a handful of rows
            EqVerifyArg a{};
            a.nv = nv;
            a.lo = dcl.lo;
            a.hi = dcl.hi;
            for (int i = 0; i < nv; i++) a.pt[i] = dcl.pt[(size_t)i];
            const unsigned long long len = 1ull << nv, mid = dcl.lo + (dcl.hi - dcl.lo) / 2;
            const unsigned long long rows[8] = {dcl.lo, dcl.hi ? dcl.hi - 1 : 0, dcl.lo ? dcl.lo - 1 : len, dcl.hi, mid, mid ^ 1, 0, len - 1};
            for (int i = 0; i < 8; i++) a.rows[i] = rows[i];
            void* p = nullptr;
            int rc = ctx_alloc(ctx, 256, &p);
            unsigned bad = ~0u;
            if (!rc) {
                hipError_t e = hipMemsetAsync(p, 0, 4, st);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL(k_eq_verify, dim3(1), dim3(64), 0, st, reinterpret_cast<const E2*>(sc->mles[j].cur), a, (unsigned*)p);
                    e = hipMemcpyAsync(&bad, p, 4, hipMemcpyDeviceToHost, st);
                }
                if (e == hipSuccess) e = hipStreamSynchronize(st);
                ctx_free(ctx, p);
                if (e != hipSuccess) bad = ~0u;
            }
            if (bad != 0) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "eq declaration %d: table %d is not eq(., point) on the rows [%llu, %llu) (CENO_HIP_EQ_VERIFY)", k, j, (unsigned long long)dcl.lo, (unsigned long long)dcl.hi); }
        }
    }
    sc->terms.resize(plan->num_terms);
    // the group of every term first: a term of a common-factor group may have NO factor of its own (coefficient x the group's common
    // factors: the `selector x constant` monomials of a chip's records, zerocheck_layer.rs:118-140) and takes its size from the group
    std::vector<int> term_group(plan->num_terms, -1);
    for (int g = 0; g < plan->num_groups; g++)
        for (uint32_t k = plan->group_term_offsets[g]; k < plan->group_term_offsets[g + 1]; k++) {
            const uint32_t t = plan->group_term_idx[k];
            if ((int)t >= plan->num_terms || term_group[t] != -1) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "bad common-term plan (term %u)", t); }
            term_group[t] = g;
        }
    for (int t = 0; t < plan->num_terms; t++) {
        uint32_t b = plan->term_offsets[t], e = plan->term_offsets[t + 1];
        ScTerm& T = sc->terms[t];
        T.coeff = E2{plan->term_coeffs[2 * t], plan->term_coeffs[2 * t + 1]};
        // all factors of a term share num_vars (layer/gpu/utils.rs:54-63); empty products are not sumcheck terms
        const int g = term_group[t];
        const bool has_common = g >= 0 && plan->common_offsets[g + 1] > plan->common_offsets[g];
        if (e <= b && !has_common) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "term %d has an empty product", t); }
        for (uint32_t k = b; k < e; k++) {
            uint32_t j = plan->term_mle_idx[k];
            if ((int)j >= plan->num_mles) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "term %d references mle %u", t, j); }
            T.idx.push_back((int)j);
        }
        if (T.idx.empty()) {
            const uint32_t j = plan->common_mle_idx[plan->common_offsets[g]];
            if ((int)j >= plan->num_mles) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "group %d references mle %u", g, j); }
            T.nv = sc->mles[j].nv;
        } else {
            T.nv = sc->mles[T.idx[0]].nv;
        }
        for (int j : T.idx)
            if (sc->mles[j].nv != T.nv) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "term %d mixes MLEs of %d and %d variables", t, T.nv, sc->mles[j].nv); }
    }
    // degree check including common factors
    for (int g = 0; g < plan->num_groups; g++) {
        int ncommon = (int)(plan->common_offsets[g + 1] - plan->common_offsets[g]);
        for (uint32_t k = plan->group_term_offsets[g]; k < plan->group_term_offsets[g + 1]; k++) {
            uint32_t t = plan->group_term_idx[k];
            if ((int)sc->terms[t].idx.size() + ncommon > d) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "term %u exceeds max_degree %d", t, d); }
            for (uint32_t c = plan->common_offsets[g]; c < plan->common_offsets[g + 1]; c++) {
                uint32_t j = plan->common_mle_idx[c];
                if ((int)j >= plan->num_mles || sc->mles[j].nv != sc->terms[t].nv) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "group %d common factor / term size mismatch", g); }
                sc->terms[t].full.push_back((int)j);
            }
        }
    }
    for (int t = 0; t < plan->num_terms; t++) sc->terms[t].full.insert(sc->terms[t].full.end(), sc->terms[t].idx.begin(), sc->terms[t].idx.end());
    for (int t = 0; t < plan->num_terms; t++)
        if (term_group[t] < 0 && (int)sc->terms[t].idx.size() > d) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "term %d exceeds max_degree %d", t, d); }

    CENO_TIMED("sc_build: from the size classes on");
    // ---- size classes (descending nv); every MLE belongs to the class of its nv ----
    std::vector<int> nvs;
    for (auto& M : sc->mles) nvs.push_back(M.nv);
    std::sort(nvs.begin(), nvs.end(), std::greater<int>());
    nvs.erase(std::unique(nvs.begin(), nvs.end()), nvs.end());
    if ((int)nvs.size() > MAX_CLASSES) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_UNSUPPORTED, "too many distinct sizes"); }
    sc->classes.resize(nvs.size());
    for (size_t c = 0; c < nvs.size(); c++) sc->classes[c].nv = nvs[c];
    auto class_of = [&](int nv) { return (int)(std::find(nvs.begin(), nvs.end(), nv) - nvs.begin()); };
    for (int j = 0; j < plan->num_mles; j++) {
        ScClass& cl = sc->classes[class_of(sc->mles[j].nv)];
        sc->mles[j].cls = (int)(&cl - sc->classes.data());
        sc->mles[j].local = (int)cl.mles.size();
        cl.mles.push_back(j);
    }
    for (int t = 0; t < plan->num_terms; t++) sc->classes[class_of(sc->terms[t].nv)].terms.push_back(t);

    CENO_TIMED("sc_build: from the working buffers on");
    // ---- working buffers per MLE ----
    for (auto& M : sc->mles) {
        if (M.nv >= 1) {
            void* p = nullptr;
            int rc = sc_dev_alloc(sc, ((size_t)1 << (M.nv - 1)) * sizeof(E2), &p);
            if (rc) { sc_release(sc); return rc; }
            M.buf[0] = (uint64_t*)p;
        }
        if (M.nv >= 2) {
            void* p = nullptr;
            int rc = sc_dev_alloc(sc, ((size_t)1 << (M.nv - 2)) * sizeof(E2), &p);
            if (rc) { sc_release(sc); return rc; }
            M.buf[1] = (uint64_t*)p;
        }
    }

    CENO_TIMED("sc_build: from the per-class plans on");
    // ---- per-class plans: every array goes into ONE blob, uploaded with a single copy from pinned memory ----
    uint32_t part_off = 0;
    size_t total_slots = 0;
    std::vector<char> blob;
    auto append = [&blob](const void* data, size_t bytes) {
        size_t off = (blob.size() + 15) & ~(size_t)15;
        blob.resize(off + std::max<size_t>(bytes, 16));
        if (bytes) memcpy(blob.data() + off, data, bytes);
        return off;
    };
    struct PlanOff { size_t gto, gt, co, ci, to, ti, cf; };
    std::vector<PlanOff> plan_offs;
    for (auto& cl : sc->classes) {
        total_slots += cl.mles.size();
        cl.part_off = part_off;
        part_off += MAXB * MAXD;
        // dense: exactly one term, not grouped, K == d <= MAXK, distinct MLEs, every MLE of the class used, one element kind
        cl.dense = false;
        if (cl.terms.size() == 1 && term_group[cl.terms[0]] < 0) {
            const ScTerm& T = sc->terms[cl.terms[0]];
            bool ok = (int)T.idx.size() == d && d <= MAXK && T.idx.size() == cl.mles.size();
            std::vector<int> s = T.idx;
            std::sort(s.begin(), s.end());
            ok = ok && std::unique(s.begin(), s.end()) == s.end();
            for (int j : T.idx) ok = ok && sc->mles[j].cur_ext == sc->mles[T.idx[0]].cur_ext;
            cl.dense = ok;
        }
        if (cl.dense || cl.terms.empty()) {
            // terms-less classes still need slots to fold their MLEs
        }
        // generic device plan with class-local MLE ids
        std::vector<uint32_t> g_term_off{0}, g_terms, c_off{0}, c_idx, t_off{0}, t_idx;
        std::vector<E2> coeffs;
        std::map<int, int> local_term;  // global term -> class-local term id
        for (int t : cl.terms) {
            local_term[t] = (int)coeffs.size();
            coeffs.push_back(sc->terms[t].coeff);
            for (int j : sc->terms[t].idx) t_idx.push_back((uint32_t)sc->mles[j].local);
            t_off.push_back((uint32_t)t_idx.size());
        }
        // groups of this class first, then every ungrouped term as its own group without common factors
        for (int g = 0; g < plan->num_groups; g++) {
            uint32_t b = plan->group_term_offsets[g], e = plan->group_term_offsets[g + 1];
            if (e <= b || sc->terms[plan->group_term_idx[b]].nv != cl.nv) continue;
            for (uint32_t k = b; k < e; k++) g_terms.push_back((uint32_t)local_term[(int)plan->group_term_idx[k]]);
            g_term_off.push_back((uint32_t)g_terms.size());
            for (uint32_t c = plan->common_offsets[g]; c < plan->common_offsets[g + 1]; c++) c_idx.push_back((uint32_t)sc->mles[plan->common_mle_idx[c]].local);
            c_off.push_back((uint32_t)c_idx.size());
        }
        bool any_free = false;
        for (int t : cl.terms)
            if (term_group[t] < 0) { g_terms.push_back((uint32_t)local_term[t]); any_free = true; }
        if (any_free) {
            g_term_off.push_back((uint32_t)g_terms.size());
            c_off.push_back((uint32_t)c_idx.size());
        }
        cl.n_groups = (int)g_term_off.size() - 1;
        cl.n_flat = (int)g_terms.size();
        cl.terms_all_base = !cl.terms.empty();
        for (int t : cl.terms) {
            if (sc->terms[t].idx.empty()) cl.terms_all_base = false;  // coefficient x common factors only
            for (int j : sc->terms[t].idx)
                if (sc->mles[j].cur_ext) cl.terms_all_base = false;
        }
        int rc = 0;
        cl.h_gto = g_term_off; cl.h_gt = g_terms; cl.h_co = c_off; cl.h_ci = c_idx; cl.h_to = t_off; cl.h_ti = t_idx; cl.h_coeffs = coeffs;
        plan_offs.push_back(PlanOff{append(g_term_off.data(), g_term_off.size() * 4), append(g_terms.data(), g_terms.size() * 4),
                                    append(c_off.data(), c_off.size() * 4), append(c_idx.data(), c_idx.size() * 4),
                                    append(t_off.data(), t_off.size() * 4), append(t_idx.data(), t_idx.size() * 4),
                                    append(coeffs.data(), coeffs.size() * sizeof(E2))});
        if (!rc) {
            void* p = nullptr;
            rc = sc_dev_alloc(sc, std::max<size_t>(cl.mles.size(), 1) * sizeof(MleSlot) * (size_t)(n + 2), &p);
            if (!rc) cl.d_slots = (MleSlot*)p;
        }
        if (rc) { sc_release(sc); return rc; }
    }
    sc->slots_per_round = total_slots;
    {
        void* p = nullptr;
        int rc = sc_dev_alloc(sc, (size_t)part_off * sizeof(E2), &p);
        if (!rc) { sc->d_partials = (E2*)p; rc = sc_dev_alloc(sc, MAXD * sizeof(E2), &p); }
        if (!rc) { sc->d_msg = (E2*)p; rc = sc_dev_alloc(sc, (size_t)plan->num_mles * sizeof(E2), &p); }
        if (!rc) { sc->d_evals = (E2*)p; rc = sc_dev_alloc(sc, 4096, &p); }
        if (!rc) { sc->d_counter = (unsigned*)p; sc->d_bcast = (Bcast*)((char*)p + 64); rc = sc_dev_alloc(sc, MAXD * sizeof(E2), &p); }
        if (!rc) sc->d_round_acc = (E2*)p;
        if (rc) { sc_release(sc); return rc; }
        // zeroed together with the plan upload below (k_setup), or by a memset when the plan takes the copy path
    }
    // A single generic class over all n variables is what the pipelined driver runs: its slot table of every round is
    // deterministic (ping-pong), so it travels inside the plan blob — one set-up kernel, not a second one at round 0
    size_t off_pre_slots = (size_t)-1;
    // (a dense class too: its LARGE rounds run the register-resident fused kernel, its small rounds join the same latency
    // ladder — k_mid / k_tail — as every other pipelined sumcheck: sc_pipeline_enqueue)
    if (sc->classes.size() == 1 && sc->classes[0].nv == n && n >= 1 && !sc->classes[0].terms.empty()) {
        const ScClass& cl = sc->classes[0];
        const size_t k = cl.mles.size();
        std::vector<MleSlot> rows((size_t)(n + 2) * k);
        std::vector<const uint64_t*> cur(k);
        std::vector<int> cur_ext(k), which(k);
        for (size_t m = 0; m < k; m++) {
            const ScMle& M = sc->mles[cl.mles[m]];
            cur[m] = M.cur;
            cur_ext[m] = M.cur_ext;
            which[m] = M.which;
        }
        for (int i = 0; i < n; i++)
            for (size_t m = 0; m < k; m++) {
                const ScMle& M = sc->mles[cl.mles[m]];
                rows[(size_t)i * k + m] = MleSlot{cur[m], M.buf[which[m]], cur_ext[m], 0};
                if (i > 0) {  // the same walk as sc_advance
                    cur[m] = M.buf[which[m]];
                    cur_ext[m] = 1;
                    which[m] ^= 1;
                }
            }
        off_pre_slots = append(rows.data(), rows.size() * sizeof(MleSlot));
    }
    hipError_t e = hipSuccess;
    {
        CENO_TIMED("sc_build: pinned block + plan upload");
        // pinned block: [flag 64 B][mailbox 64 B][message MAXD + evals num_mles (E2)][slot staging]
        const size_t msg_bytes = (MAXD + (size_t)plan->num_mles) * sizeof(E2);
        const size_t slot_bytes = std::max<size_t>(total_slots, 1) * sizeof(MleSlot) * (size_t)(n + 2);
        // ... [plan blob][tables of a host-finished tail]: reserved HERE — a pinned allocation at enqueue time may call into the
        // driver (hipHostMalloc) while this sumcheck's own persistent round kernels wait for this very host thread
        size_t tail_bytes = 0;
        if (sc->classes.size() == 1 && sc->classes[0].nv == n && n >= 2 && !sc->classes[0].terms.empty()) {
            sc->host_ht = host_tail_rounds(sc);
            if (sc->host_ht > 0) tail_bytes = sc->classes[0].mles.size() * ((size_t)2 << sc->host_ht) * sizeof(E2);
        }
        const size_t tail_off = (128 + msg_bytes + slot_bytes + blob.size() + 16 + 15) & ~(size_t)15;
        void *hb = nullptr, *db = nullptr;
        int rc = ctx_pinned_alloc(ctx, tail_off + tail_bytes, &hb, &db);
        if (rc) { sc_release(sc); return rc; }
        if (tail_bytes) {
            sc->h_tail = reinterpret_cast<E2*>((char*)hb + tail_off);
            sc->d_tail_view = reinterpret_cast<E2*>((char*)db + tail_off);
            for (size_t x = 0; x < tail_bytes / 8; x++) reinterpret_cast<uint64_t*>(sc->h_tail)[x] = MSG_INVALID;
        }
        sc->h_block = hb;
        sc->h_flag = (unsigned long long*)hb;
        sc->d_hflag = (unsigned long long*)db;
        sc->h_mailbox = (Mailbox*)((char*)hb + 64);
        sc->d_mailbox = (Mailbox*)((char*)db + 64);
        // On large-BAR boxes the mailbox moves into fine-grained DEVICE memory: the finishing workgroup then polls HBM
        // (~0.13 us per poll, measured) instead of reading host memory across PCIe (~2 us per poll), and the host's
        // challenge is one posted write through the BAR.
        if ((sc->vram_slot = ctx_vram_slot_alloc(ctx)) != nullptr) sc->h_mailbox = sc->d_mailbox = (Mailbox*)sc->vram_slot;
        sc->h_pinned = (E2*)((char*)hb + 128);
        sc->d_hmsg = (uint64_t*)((char*)db + 128);
        sc->h_slots = (MleSlot*)((char*)hb + 128 + msg_bytes);
        *sc->h_flag = 0;
        for (size_t k = 0; k < 2 * (MAXD + (size_t)plan->num_mles); k++) reinterpret_cast<uint64_t*>(sc->h_pinned)[k] = MSG_INVALID;
        mailbox_clear(sc->h_mailbox);
        // plan blob: pinned staging -> one device allocation, one copy (the pinned block outlives the copy)
        char* h_blob = (char*)hb + ((128 + msg_bytes + slot_bytes + 15) & ~(size_t)15);
        memcpy(h_blob, blob.data(), blob.size());
        void* d_blob = nullptr;
        rc = sc_dev_alloc(sc, std::max<size_t>(blob.size(), 16), &d_blob);
        if (rc) { sc_release(sc); return rc; }
        // rows of the persistent mid-round kernel (2 sets x 256 workgroups x MAXD ext, then 256 relay lines): armed HERE, by the
        // set-up kernel — a fill queued right in front of k_mid sat on the critical path of its first round (~9 us)
        static constexpr size_t MID_ROWS_BYTES = (size_t)2 * 256 * MAXD * sizeof(E2), MID_BLOCK_BYTES = MID_ROWS_BYTES + 256 * 64;
        if (off_pre_slots != (size_t)-1 && n >= 9) {
            void* p = nullptr;
            rc = sc_dev_alloc(sc, MID_BLOCK_BYTES, &p);
            if (rc) { sc_release(sc); return rc; }
            sc->d_mid_rows = (uint64_t*)p;
        }
        if (blob.size() <= 16 * 1024 && blob.size() % 8 == 0) {
            const uint64_t* d_view = reinterpret_cast<const uint64_t*>((char*)db + (h_blob - (char*)hb));
            SetupJob job;
            job.zero = reinterpret_cast<uint64_t*>(sc->d_counter);
            job.zero_words = (size_t)4096 / 8;
            job.dst = (uint64_t*)d_blob;
            job.src_host_view = d_view;
            job.words = blob.size() / 8;
            job.ones = sc->d_mid_rows;
            job.ones_words = sc->d_mid_rows ? MID_ROWS_BYTES / 8 : (size_t)0;
            if (defer_setup) {
                *defer_setup = job;  // the caller queues it with a kernel of its own (a tower layer: its eq table)
            } else {
                CENO_TIMED("sc_build: launch_setup_job");
                launch_setup_job(job, st);
                if (hipGetLastError() != hipSuccess) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_HIP, "plan upload failed"); }
            }
        } else {
            if (sc->d_mid_rows && hipMemsetAsync(sc->d_mid_rows, 0xFF, MID_ROWS_BYTES, st) != hipSuccess) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_HIP, "memset failed"); }
            if (hipMemsetAsync(sc->d_counter, 0, 4096, st) != hipSuccess) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_HIP, "memset failed"); }
            if (hipMemcpyAsync(d_blob, h_blob, blob.size(), hipMemcpyHostToDevice, st) != hipSuccess) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_HIP, "plan upload failed"); }
        }
        for (size_t c = 0; c < sc->classes.size(); c++) {
            ScClass& cl = sc->classes[c];
            const PlanOff& po = plan_offs[c];
            char* base = (char*)d_blob;
            cl.d_group_term_off = (uint32_t*)(base + po.gto);
            cl.d_group_terms = (uint32_t*)(base + po.gt);
            cl.d_common_off = (uint32_t*)(base + po.co);
            cl.d_common_idx = (uint32_t*)(base + po.ci);
            cl.d_term_off = (uint32_t*)(base + po.to);
            cl.d_term_idx = (uint32_t*)(base + po.ti);
            cl.d_coeffs = (E2*)(base + po.cf);
        }
        if (off_pre_slots != (size_t)-1) {
            sc->classes[0].d_slots = reinterpret_cast<MleSlot*>((char*)d_blob + off_pre_slots);
            sc->slots_preloaded = true;
        }
    }

    // zero-variable MLEs are already scalars: fetch their values
    for (auto& M : sc->mles) {
        if (M.nv == 0) {
            uint64_t h[2] = {0, 0};
            e = hipMemcpyAsync(h, M.cur, M.cur_ext ? 16 : 8, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { sc_release(sc); return ctx_fail(ctx, CENO_HIP_ERR_HIP, "begin: %s", hipGetErrorString(e)); }
            M.eval = E2{h[0], M.cur_ext ? h[1] : 0};
            M.done = true;
        }
    }
    {
        static const bool no_gen = getenv("CENO_HIP_NO_GEN") != nullptr;  // A/B switch: two-kernel generic path everywhere
        if (!no_gen) {
            int rc = sc_build_gen(sc);
            if (rc) { sc_release(sc); return rc; }
        }
    }
    *out = sc;
    return 0;
}

// write this round's slot table of class `cl` (device copy queued on the stream); returns device ptr
static int sc_push_slots(ceno_hip_sumcheck* sc, ScClass& cl, int slot_round, size_t& h_cursor, const MleSlot** d_out) {
    MleSlot* h = sc->h_slots + (size_t)slot_round * sc->slots_per_round + h_cursor;
    for (size_t k = 0; k < cl.mles.size(); k++) {
        ScMle& M = sc->mles[cl.mles[k]];
        h[k].in = M.cur;
        h[k].out = M.buf[M.which];
        h[k].in_ext = M.cur_ext;
        h[k].pad = 0;
    }
    MleSlot* d = cl.d_slots + (size_t)slot_round * std::max<size_t>(cl.mles.size(), 1);
    HIP_TRY(sc->ctx, hipMemcpyAsync(d, h, cl.mles.size() * sizeof(MleSlot), hipMemcpyHostToDevice, sc->st));
    h_cursor += cl.mles.size();
    *d_out = d;
    return 0;
}

static void sc_advance(ceno_hip_sumcheck* sc, ScClass& cl) {
    for (int j : cl.mles) {
        ScMle& M = sc->mles[j];
        M.cur = M.buf[M.which];
        M.cur_ext = 1;
        M.which ^= 1;
    }
}

// wait until `n_words` pinned words (pre-filled with MSG_INVALID) have all been written by the device
static int sc_wait_words(ceno_hip_sumcheck* sc, const uint64_t* words, int n_words, const char* what = "its message") {
    unsigned long long spins = 0;
    for (;;) {
        int k = 0;
        while (k < n_words && __atomic_load_n(&words[k], __ATOMIC_ACQUIRE) != MSG_INVALID) k++;
        if (k == n_words) return 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        __builtin_ia32_pause();  // several lanes may be spinning at once
#endif
        if ((++spins & 0xFFFFF) == 0) {
            // every ~1M polls make sure the stream is still alive (a faulted kernel never writes its message)
            hipError_t q = hipStreamQuery(sc->st);
            if (q != hipSuccess && q != hipErrorNotReady)
                return ctx_fail(sc->ctx, CENO_HIP_ERR_HIP, "sumcheck round kernel failed: %s", hipGetErrorString(q));
            if (q == hipSuccess) {
                k = 0;
                while (k < n_words && __atomic_load_n(&words[k], __ATOMIC_ACQUIRE) != MSG_INVALID) k++;
                if (k == n_words) return 0;
                return ctx_fail(sc->ctx, CENO_HIP_ERR_HIP, "sumcheck round finished without publishing %s (round %d of %d, %d of %d words, host rounds from %d)",
                                what, sc->round, sc->n, k, n_words, sc->host_from);
            }
        }
    }
}
// take the round message out of the pinned block and arm the words for the next one (the reset is ordered before whatever
// releases the next kernel: the mailbox post fences, a launch rings a doorbell)
static int sc_take_message(ceno_hip_sumcheck* sc, uint64_t* h_out) {
    CENO_TIMED("sc_take_message (wait)");
    uint64_t* w = reinterpret_cast<uint64_t*>(sc->h_pinned);
    const int n_words = 2 * sc->d;
    TRY(sc_wait_words(sc, w, n_words));
    for (int k = 0; k < n_words; k++) {
        h_out[k] = w[k];
        __atomic_store_n(&w[k], MSG_INVALID, __ATOMIC_RELAXED);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Pipelined rounds (single dense class covering all n variables, host-side transcript):
// every round kernel is enqueued when round 0 is requested; the finishing workgroup of kernel i-1 fetches
// challenge i-1 from the host mailbox and relays it, kernel i starts at the kernel boundary and reads it,
// so neither the launch latency nor a stream synchronisation sits between a message and the next pass —
// the host only moves 16-byte challenges and d*16-byte messages through pinned memory.
// ------------------------------------------------------------------------------------------------
static bool sc_pipeline_eligible(const ceno_hip_sumcheck* sc) {
    if (!sc->allow_pipeline) return false;
    static const bool disabled = getenv("CENO_HIP_NO_PIPELINE") != nullptr;  // A/B switch for measurements
    if (disabled) return false;
    if (sc->ctx->prof_on && !sc->ctx->prof_pipelined) return false;  // per-launch timing (mode 1) wants the kernels free of mailbox waits
    if (sc->classes.size() != 1) return false;
    const ScClass& cl = sc->classes[0];
    return cl.nv == sc->n && sc->n >= 2 && !cl.terms.empty();
}

template <int K>
static void pipe_launch(ceno_hip_sumcheck* sc, ScClass& cl, int mode, size_t pairs, unsigned grid, const Epilogue& ep) {
    launch_dense<K>(sc, cl, mode, pairs, e2_zero(), grid, ep);
}

static Epilogue pipe_epilogue(ceno_hip_sumcheck* sc, ScClass& cl, int i) {
    Epilogue ep{};
    ep.partials = reinterpret_cast<uint64_t*>(sc->d_partials + cl.part_off);
    ep.counter = sc->d_counter;
    ep.round_acc = sc->d_round_acc;
    ep.out_msg = sc->d_hmsg;
    ep.flag = sc->d_hflag;
    ep.seq = (unsigned long long)(i + 1);
    ep.coeff = e2_one();
    ep.first_class = 1;
    ep.last_class = 1;
    ep.d = sc->d;
    ep.mailbox = sc->d_mailbox;
    ep.bcast = sc->d_bcast;
    static const int dbg_on = getenv("CENO_HIP_DEBUG") != nullptr;
    ep.dbg = dbg_on;
    // A queued round gives up when its challenge does not arrive in time (the host died, or forgot the handle); a live host
    // aborts the pipeline explicitly (sc_release), so the limit only has to outlast the slowest legitimate gap between two
    // rounds — ranks of a sharded run wait for each other here (first-use initialisation, a descheduled peer): 60 s by
    // default, CENO_HIP_PIPE_TIMEOUT_S to change it.
    static const unsigned long long ticks = [] {
        const char* e = getenv("CENO_HIP_PIPE_TIMEOUT_S");
        const double sec = e && atof(e) > 0 ? atof(e) : 60.0;
        return (unsigned long long)(sec * 1e8);
    }();
    ep.poll_ticks = ticks;
    ep.wait_seq = (unsigned long long)i;                              // round 0 takes no challenge
    ep.next_seq = (i + 1 < sc->n) ? (unsigned long long)(i + 1) : 0;  // fetch challenge i for round i+1
    return ep;
}

// Rounds are enqueued a few ahead of the one being answered instead of all at once: launching ~2n kernels costs
// 100-300 us of host time, and a round-0 kernel shorter than that would sit in its challenge poll until the host
// got around to reading its message (measured: 20 us per tiny round instead of 12, 800 us for a 160 us round 0).
// Rounds at the end of a pipelined sumcheck that the HOST computes.  A device round of the persistent tail costs ~9.6 us whatever
// its size (publish, PCIe, challenger, poll); a host round costs its arithmetic: pairs x sum_T |T| x d extension multiplications of
// ~7 ns with every factor of every term multiplied out (measured: 0.31 us per pair for a tower layer's 5 terms of degree 3, 2.5 us per pair for a
// chip's 33 constraint terms; sc_host_round now walks the plan group by group as the device does — 1.85 us per pair for that chip — and the
// model below, still counting the old way, hands over a fraction of a round late: a sweep of the budget from 10.5 to 24 us moved nothing).
// The host takes over at the first round it computes faster than the device would: 32 pairs for a tower layer, 4 for
// the chip's main sumcheck, 128 for a plain product of three tables.  CENO_HIP_HOST_TAIL caps the number of host rounds (0 = the
// device runs every round), CENO_HIP_HOST_TAIL_NS is the per-round budget.
static int host_tail_rounds(const ceno_hip_sumcheck* sc) {
    const char* e = getenv("CENO_HIP_HOST_TAIL");  // (read per call: the test-suite switches it between sumchecks)
    int cap = e ? atoi(e) : 8;
    cap = cap < 0 ? 0 : (cap > 12 ? 12 : cap);
    const char* b = getenv("CENO_HIP_HOST_TAIL_NS");
    const double budget = b && atof(b) > 0 ? atof(b) : 10500.0;
    size_t mults = 0;
    for (int ti : sc->classes[0].terms) mults += sc->terms[ti].full.size();
    const double per_pair = 7.0 * (double)sc->d * (double)std::max<size_t>(mults, 1);
    int ht = 1;  // the last round (one pair) is always cheaper on the host
    while (ht < cap && (double)((size_t)1 << ht) * per_pair <= budget) ht++;  // first host round of `ht` rounds: 2^(ht-1) pairs
    return std::min(ht, cap);
}
// One host round of a host-finished tail: fold the tables with the challenge of round i - 1, then the message of round i,
// p(1) .. p(d) of sum_pairs sum_terms c_T prod_{j in T} f_j(X).  Field arithmetic is exact, so the order of the sums is free and the
// words equal what the device's kernels would have published.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// eight pairs per pass (AVX-512): entries [first, first + 8) of the folded table from the sixteen entries they come from
__attribute__((target("avx512f,avx512dq"))) static void host_fold8(E2* t, int first, E2 r) {
    const e2v::VE2 lo = e2v::load(t, 2 * (size_t)first, 2), hi = e2v::load(t, 2 * (size_t)first + 1, 2);
    e2v::store(t, (size_t)first, e2v::add(lo, e2v::mul(e2v::bcast(r), e2v::sub(hi, lo))));
}
#endif
static void host_fold(ceno_hip_sumcheck* sc, E2 r) {
    const size_t k = sc->classes[0].mles.size();
    const int half = sc->host_len / 2;
    for (size_t m = 0; m < k; m++) {
        E2* t = sc->host_tab.data() + m * (size_t)sc->host_len0;
        int j0 = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        if (p2host::have_avx512())
            for (; j0 + 8 <= half; j0 += 8) host_fold8(t, j0, r);  // in place: pass j0 reads entries [2 j0, 2 j0 + 16), all at or beyond what it writes
#endif
        for (int j = j0; j < half; j++) {
            const E2 lo = t[2 * j], hi = t[2 * j + 1];
            t[j] = lo + r * (hi - lo);
        }
    }
    sc->host_len = half;
}
static int host_take_over(ceno_hip_sumcheck* sc) {
    if (sc->host_len) return 0;
    const size_t k = sc->classes[0].mles.size(), words = k * (size_t)sc->host_len0 * 2;
    TRY(sc_wait_words(sc, reinterpret_cast<const uint64_t*>(sc->h_tail), (int)words, "the tables of the host-finished tail"));
    sc->host_tab.resize(k * (size_t)sc->host_len0);
    memcpy(sc->host_tab.data(), sc->h_tail, words * 8);
    sc->host_len = sc->host_len0;
    return 0;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
// pairs [p, p + 8) of the grouped plan walk of sc_host_round, one pair per lane: adds the eight pairs' contribution to acc[0 .. d)
__attribute__((target("avx512f,avx512dq"))) static void host_eval8(const ScClass& cl, const E2* tab, size_t len0, int d, int p, E2* acc) {
    using namespace e2v;
    const int n_groups = (int)cl.h_gto.size() - 1;
    VE2 tot[MAXD];
    for (int t = 0; t < d; t++) tot[t] = bcast(e2_zero());
    for (int g = 0; g < n_groups; g++) {
        VE2 inner[MAXD];
        for (int t = 0; t < d; t++) inner[t] = bcast(e2_zero());
        for (uint32_t ti = cl.h_gto[g]; ti < cl.h_gto[g + 1]; ti++) {
            const uint32_t term = cl.h_gt[ti];
            const VE2 c = bcast(cl.h_coeffs[term]);
            const uint32_t k0 = cl.h_to[term], k1 = cl.h_to[term + 1];
            VE2 pr[MAXD];
            if (k0 == k1) {
                for (int t = 0; t < d; t++) pr[t] = c;
            } else {
                const E2* q = tab + (size_t)cl.h_ti[k0] * len0;
                const VE2 lo = load(q, 2 * (size_t)p, 2), hi = load(q, 2 * (size_t)p + 1, 2);
                VE2 x = mul(c, hi);
                const VE2 delta = mul(c, sub(hi, lo));
                for (int t = 0; t < d; t++) {
                    pr[t] = x;
                    x = add(x, delta);
                }
                for (uint32_t k = k0 + 1; k < k1; k++) {
                    const E2* f = tab + (size_t)cl.h_ti[k] * len0;
                    const VE2 flo = load(f, 2 * (size_t)p, 2), fhi = load(f, 2 * (size_t)p + 1, 2);
                    VE2 y = fhi;
                    const VE2 dy = sub(fhi, flo);
                    for (int t = 0; t < d; t++) {
                        pr[t] = mul(pr[t], y);
                        y = add(y, dy);
                    }
                }
            }
            for (int t = 0; t < d; t++) inner[t] = add(inner[t], pr[t]);
        }
        for (uint32_t k = cl.h_co[g]; k < cl.h_co[g + 1]; k++) {
            const E2* f = tab + (size_t)cl.h_ci[k] * len0;
            const VE2 flo = load(f, 2 * (size_t)p, 2), fhi = load(f, 2 * (size_t)p + 1, 2);
            VE2 y = fhi;
            const VE2 dy = sub(fhi, flo);
            for (int t = 0; t < d; t++) {
                inner[t] = mul(inner[t], y);
                y = add(y, dy);
            }
        }
        for (int t = 0; t < d; t++) tot[t] = add(tot[t], inner[t]);
    }
    for (int t = 0; t < d; t++) acc[t] = acc[t] + hsum(tot[t]);
}
#endif
static int sc_host_round(ceno_hip_sumcheck* sc, E2 r, uint64_t* h_out, bool fold = true) {
    static const bool dbg = getenv("CENO_HIP_DEBUG") != nullptr;
    timespec ta, tb, tc;
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &ta);
    TRY(host_take_over(sc));
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &tb);
    if (fold) host_fold(sc, r);
    const ScClass& cl = sc->classes[0];
    const int d = sc->d, pairs = sc->host_len / 2;
    E2 acc[MAXD];
    for (int t = 0; t < d; t++) acc[t] = e2_zero();
    const E2* tab = sc->host_tab.data();
    const size_t len0 = (size_t)sc->host_len0;
    if (cl.h_gto.size() >= 2) {
        // the class plan as the device kernels walk it: per group  prod(common factors) * sum_terms c_T prod(residual factors); the coefficient
        // rides on a term's first factor (c f(X) = c f(1) + (X - 1) c delta: two multiplications instead of d), the common factors multiply the
        // group's sum once — 28 multiplications per pair of a tower layer instead of the 45 of coefficient x every factor of every term
        const int n_groups = (int)cl.h_gto.size() - 1;
        int p0 = 0;
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        if (p2host::have_avx512())
            for (; p0 + 8 <= pairs; p0 += 8) host_eval8(cl, tab, len0, d, p0, acc);  // eight pairs per pass; the rest below, one by one
#endif
        for (int p = p0; p < pairs; p++) {
            for (int g = 0; g < n_groups; g++) {
                E2 inner[MAXD];
                for (int t = 0; t < d; t++) inner[t] = e2_zero();
                for (uint32_t ti = cl.h_gto[g]; ti < cl.h_gto[g + 1]; ti++) {
                    const uint32_t term = cl.h_gt[ti];
                    const E2 c = cl.h_coeffs[term];
                    const uint32_t k0 = cl.h_to[term], k1 = cl.h_to[term + 1];
                    E2 pr[MAXD];
                    if (k0 == k1) {
                        for (int t = 0; t < d; t++) pr[t] = c;
                    } else {
                        const E2* q = tab + (size_t)cl.h_ti[k0] * len0 + 2 * p;
                        E2 x = c * q[1];
                        const E2 delta = c * (q[1] - q[0]);
                        for (int t = 0; t < d; t++) {
                            pr[t] = x;
                            x = x + delta;
                        }
                        for (uint32_t k = k0 + 1; k < k1; k++) {
                            const E2* f = tab + (size_t)cl.h_ti[k] * len0 + 2 * p;
                            E2 y = f[1];
                            const E2 dy = f[1] - f[0];
                            for (int t = 0; t < d; t++) {
                                pr[t] = pr[t] * y;
                                y = y + dy;
                            }
                        }
                    }
                    for (int t = 0; t < d; t++) inner[t] = inner[t] + pr[t];
                }
                for (uint32_t k = cl.h_co[g]; k < cl.h_co[g + 1]; k++) {
                    const E2* f = tab + (size_t)cl.h_ci[k] * len0 + 2 * p;
                    E2 y = f[1];
                    const E2 dy = f[1] - f[0];
                    for (int t = 0; t < d; t++) {
                        inner[t] = inner[t] * y;
                        y = y + dy;
                    }
                }
                for (int t = 0; t < d; t++) acc[t] = acc[t] + inner[t];
            }
        }
    } else {
        for (int ti : cl.terms) {
            const ScTerm& T = sc->terms[ti];
            for (int p = 0; p < pairs; p++) {
                E2 pr[MAXD];
                for (int t = 0; t < d; t++) pr[t] = T.coeff;
                for (int j : T.full) {
                    const E2* q = tab + (size_t)sc->mles[j].local * len0 + 2 * p;
                    E2 x = q[1];
                    const E2 delta = q[1] - q[0];
                    for (int t = 0; t < d; t++) {
                        pr[t] = pr[t] * x;
                        x = x + delta;
                    }
                }
                for (int t = 0; t < d; t++) acc[t] = acc[t] + pr[t];
            }
        }
    }
    for (int t = 0; t < d; t++) {
        h_out[2 * t] = acc[t].c0;
        h_out[2 * t + 1] = acc[t].c1;
    }
    if (dbg) {
        clock_gettime(CLOCK_MONOTONIC, &tc);
        fprintf(stderr, "[ceno_hip] host round %d: %d pairs, %zu terms, waited %.1f us for the tables, fold + message %.1f us\n", sc->round, pairs,
                cl.terms.size(), (tb.tv_sec - ta.tv_sec) * 1e6 + (tb.tv_nsec - ta.tv_nsec) / 1e3, (tc.tv_sec - tb.tv_sec) * 1e6 + (tc.tv_nsec - tb.tv_nsec) / 1e3);
    }
    return 0;
}

static int pipe_lookahead() {
    static int v = [] {
        const char* e = getenv("CENO_HIP_PIPE_LOOKAHEAD");  // large value = enqueue everything at round 0 (A/B measurements)
        return e ? atoi(e) : 3;
    }();
    return v;
}
static int sc_pipeline_enqueue(ceno_hip_sumcheck* sc, int upto) {
    CENO_TIMED("sc_pipeline_enqueue");
    ceno_hip_ctx* ctx = sc->ctx;
    ScClass& cl = sc->classes[0];
    const bool dbg = getenv("CENO_HIP_DEBUG") != nullptr;
    timespec ts0, ts1;
    if (dbg) clock_gettime(CLOCK_MONOTONIC, &ts0);
    upto = std::min(upto, sc->n);
    const int from = sc->enq;
    if (from >= upto) return 0;
    if (!sc->live_counted) {  // (once per handle, BEFORE the first kernel that waits for this host is queued) the pool returns nothing
        sc->live_owner = ctx_pipelined_begin(ctx);  // to the driver while such kernels exist, and none is queued while a trim is under way (common.hpp)
        sc->live_counted = true;
    }
    // A dense class (one product of K <= 4 tables) runs its LARGE rounds on the register-resident fused kernel and hands over to
    // the latency ladder below (k_mid / k_tail: persistent kernels, no kernel boundary and no cross-workgroup counter per round)
    // once a round has at most 2^15 pairs: the last ~15 rounds of every dense sumcheck cost ~9-13 us instead of ~15 us each
    // (CENO_HIP_DENSE_LADDER=0 keeps one k_dense launch per round: A/B measurements).
    static const size_t dense_small_pairs = [] {
        const char* e = getenv("CENO_HIP_DENSE_LADDER");
        if (e && atoi(e) == 0) return (size_t)0;
        const char* l = getenv("CENO_HIP_DENSE_LADDER_LOG");  // hand-over point (A/B measurements)
        return (size_t)1 << (l ? std::max(10, std::min(atoi(l), 20)) : 15);
    }();
    {
        // slot tables of every round are deterministic: stage them all, one upload
        const size_t k = cl.mles.size();
        for (int i = 0; i < sc->n && from == 0; i++) {
            MleSlot* h = sc->h_slots + (size_t)i * sc->slots_per_round;
            for (size_t m = 0; m < k; m++) {
                ScMle& M = sc->mles[cl.mles[m]];
                h[m].in = M.cur;
                h[m].out = M.buf[M.which];
                h[m].in_ext = M.cur_ext;
                h[m].pad = 0;
            }
            if (i > 0) sc_advance(sc, cl);
        }
        if (from == 0 && !sc->slots_preloaded) {
            const size_t bytes = (size_t)sc->n * k * sizeof(MleSlot);
            if (bytes <= 16 * 1024) {  // small: one tiny kernel reads the pinned block instead of a copy-engine blit
                const uint64_t* d_view = reinterpret_cast<const uint64_t*>(reinterpret_cast<char*>(sc->d_hflag) +
                                                                           (reinterpret_cast<char*>(sc->h_slots) - reinterpret_cast<char*>(sc->h_block)));
                hipLaunchKernelGGL(k_setup, dim3(1), dim3(256), 0, sc->st, (uint64_t*)nullptr, (size_t)0, reinterpret_cast<uint64_t*>(cl.d_slots), d_view, bytes / 8);
            } else {
                HIP_TRY(ctx, hipMemcpyAsync(cl.d_slots, sc->h_slots, bytes, hipMemcpyHostToDevice, sc->st));
            }
        }
        for (int i = from; i < upto; i++) {
            const size_t pairs = (size_t)1 << (cl.nv - i - 1);
            Epilogue ep = pipe_epilogue(sc, cl, i);
            if (cl.dense && pairs > dense_small_pairs) {
                const ScTerm& T = sc->terms[cl.terms[0]];
                const MleSlot* row = sc->h_slots + (size_t)i * sc->slots_per_round;
                ep.coeff = T.coeff;
                const int mode = (i == 0 ? 0 : 2) + (row[0].in_ext ? 0 : 1);
                const unsigned grid = sc_grid(pairs);
                // ceno_hip_prof_enable(ctx, 2): ONE event pair around the large rounds of a sumcheck as queued — they run back to back on the
                // stream, so the span is the sum of their durations, the waits for the next challenge included (an event pair per launch
                // cost the timed steps 2.6 %)
                const bool last_large = ((size_t)1 << (cl.nv - i - 1)) / 2 <= dense_small_pairs || i + 1 >= sc->n;
                if (i == 0) prof_begin(ctx, sc->st);
                switch ((int)T.idx.size()) {
                case 1: launch_dense_row<1>(sc, cl, mode, pairs, e2_zero(), grid, ep, row); break;
                case 2: launch_dense_row<2>(sc, cl, mode, pairs, e2_zero(), grid, ep, row); break;
                case 3: launch_dense_row<3>(sc, cl, mode, pairs, e2_zero(), grid, ep, row); break;
                default: launch_dense_row<4>(sc, cl, mode, pairs, e2_zero(), grid, ep, row); break;
                }
                {
                    const double in_el = row[0].in_ext ? 16.0 : 8.0, kk = (double)T.idx.size();
                    prof_count(ctx, i == 0 ? kk * 2.0 * pairs * in_el : kk * (4.0 * pairs * in_el + 2.0 * pairs * 16.0));
                    if (last_large) prof_end(ctx, sc->st, 0.0, 0);
                }
                continue;
            }
            if (sc->eqf.on && i < sc->eqf.fast_upto) {  // a tower layer's large rounds: one fused pass, one evaluation point fewer
                launch_tower_round(sc->ctx, sc->eqf.n_prod, sc->eqf.n_logup, i > 0 ? 2 : sc->eqf.has_claim ? 1 : 0, cl.d_slots + (size_t)i * k, sc->eqf.coef, pairs, ep, sc_grid(pairs), sc->st);
                continue;
            }
            DevPlan pl;
            pl.slots = cl.d_slots + (size_t)i * k;
            pl.use_out = i > 0 ? 1 : 0;
            pl.n_groups = cl.n_groups;
            pl.group_term_off = cl.d_group_term_off;
            pl.group_terms = cl.d_group_terms;
            pl.common_off = cl.d_common_off;
            pl.common_idx = cl.d_common_idx;
            pl.coeffs = cl.d_coeffs;
            pl.term_off = cl.d_term_off;
            pl.term_idx = cl.d_term_idx;
            const int tnt = fused_tnt(k, pairs);
            if (tail_eligible(k, pairs, sc->d, (size_t)cl.n_flat)) {  // this launch produces rounds i .. n-1
                static const bool tail_evals = !(getenv("CENO_HIP_TAIL_EVALS") && atoi(getenv("CENO_HIP_TAIL_EVALS")) == 0);  // A/B switch
                // The last rounds belong to the host (sc_host_round): the device stops after round n_stop - 1 and ships its tables.
                int n_stop = sc->n;
                E2* export_view = nullptr;
                const int ht = sc->h_tail ? sc->host_ht : 0;
                if (ht > 0 && std::max(i + 1, sc->n - ht) < sc->n) {
                    n_stop = std::max(i + 1, sc->n - ht);
                    sc->host_from = n_stop;
                    sc->host_len0 = 2 << (sc->n - n_stop);  // round n_stop - 1 has 2^(n - n_stop) <= 2^ht pairs
                    export_view = sc->d_tail_view;
                }
                sc->tail_evals = tail_evals && !export_view;
                launch_tail(sc->d, pl, cl.d_slots + (size_t)(sc->n - 1) * k, (int)k, cl.n_flat, pairs, i, n_stop, ep,
                            sc->tail_evals ? reinterpret_cast<E2*>(sc->d_hmsg) + MAXD : nullptr, export_view, sc->st);
                upto = sc->n;
                break;
            }
            {
                // Residency budget: the workgroups of a k_mid launch wait for each other, so all of them must fit on the chip
                // next to every other such launch in flight (lanes).  Capacity and cost come from the runtime's occupancy for
                // the launch's own dynamic LDS and the device's CU count (units of 1/64 CU, 1/8 of the chip kept as headroom
                // for everything else that is resident); a launch that does not fit falls back to the per-round kernels.
                const int capacity = ctx->num_cus * 64 - ctx->num_cus * 8;
                int W = 0, S0 = 0, cost = 0;
                if (!sc->mid_reserved && sc->d_mid_rows) {
                    mid_geometry(k, pairs, sc->d, (size_t)cl.n_flat, 1 << 30, &W, &S0);  // slice size first: it fixes the LDS
                    int nb = W ? mid_blocks_per_cu(sc->d, k, (size_t)S0, (size_t)cl.n_flat) : 0;
                    if (nb > 0) {
                        const int free_units = capacity - ctx->mid_wgs_in_flight.load();
                        mid_geometry(k, pairs, sc->d, (size_t)cl.n_flat, (int)std::min<long long>((long long)free_units * nb / 64, 1 << 20), &W, &S0);
                        nb = W ? mid_blocks_per_cu(sc->d, k, (size_t)S0, (size_t)cl.n_flat) : 0;  // fewer workgroups = larger slices
                    }
                    if (nb > 0) cost = (W * 64 + nb - 1) / nb;
                    else W = 0;
                }
                // rounds i .. i1 in one launch of W resident workgroups, i1 = the last round too large for the tail kernel
                int i1 = i;
                while (i1 + 1 < sc->n && !tail_eligible(k, pairs >> (i1 + 1 - i), sc->d, (size_t)cl.n_flat)) i1++;
                bool booked = false;
                if (W >= 4 && i1 > i && i1 + 1 < sc->n && (pairs >> (i1 - i)) >= (size_t)W) {
                    int cur = ctx->mid_wgs_in_flight.load();
                    while (cur + cost <= capacity && !booked) booked = ctx->mid_wgs_in_flight.compare_exchange_weak(cur, cur + cost);
                }
                if (booked) {
                    sc->mid_reserved = cost;
                    static std::atomic<unsigned long long> nonce_src{0};
                    const unsigned long long nonce = (++nonce_src) & ((1ull << 56) - 1);
                    // armed rows (every word MSG_INVALID since the set-up kernel; the reducer re-arms what it reads) + relay lines
                    Epilogue epm = ep;
                    epm.partials = sc->d_mid_rows;
                    MidRelay* relay = reinterpret_cast<MidRelay*>(reinterpret_cast<char*>(sc->d_mid_rows) + (size_t)2 * 256 * MAXD * sizeof(E2));
                    const bool relay_only = getenv("CENO_HIP_MID_RELAY") && atoi(getenv("CENO_HIP_MID_RELAY")) != 0;  // A/B switch
                    launch_mid(sc->d, pl, cl.d_slots + (size_t)i1 * k, (int)k, cl.n_flat, W, S0, i, i1, epm, relay, nonce,
                               sc->vram_slot != nullptr && !relay_only, sc->st);
                    i = i1;          // the loop continues with round i1 + 1: the persistent tail
                    upto = sc->n;
                    continue;
                }
            }
            if (sc->gen_on && cl.gen && sc->gen_rounds[i].n_comps > 0 && pairs >= gen_pipe_min_pairs()) {
                const GenRound& R = sc->gen_rounds[i];
                launch_gen(sc->ctx, sc->d, R.base0, reinterpret_cast<const GenComp*>(sc->d_gen + R.off_comps), R.n_comps, R.total_tiles, e2_zero(), ep,
                           R.stage_bytes, sc->st);
            } else if (tile_eligible(k, pairs)) {
                launch_tile(sc->d, pl, (int)k, cl.n_flat, pairs, e2_zero(), ep, sc->st);
            } else if (tnt) {
                launch_fused(sc->d, tnt, pl, (int)k, pairs, e2_zero(), ep, grid_for(pairs, (unsigned)tnt, MAXB), sc->st);
            } else {
                if (i > 0)
                    hipLaunchKernelGGL(k_fold_batch, dim3(grid_for(2 * pairs, NT, 1024), (unsigned)k), dim3(NT), 0, sc->st, pl.slots, 2 * pairs,
                                       e2_zero(), (const Bcast*)sc->d_bcast, (unsigned long long)i);
                launch_accum(sc->ctx, sc->d, pl, pairs, ep, sc_grid(pairs), sc->st, i == 0 && cl.terms_all_base);
            }
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    if (dbg) {
        clock_gettime(CLOCK_MONOTONIC, &ts1);
        fprintf(stderr, "[ceno_hip] enqueued pipelined rounds %d..%d in %.1f us\n", from, upto - 1,
                (ts1.tv_sec - ts0.tv_sec) * 1e6 + (ts1.tv_nsec - ts0.tv_nsec) / 1e3);
    }
    sc->enq = upto;
    sc->pipelined = true;
    sc->seq = (unsigned long long)sc->n;
    return 0;
}

// A round of k_tower (sumcheck_tower.hpp) delivers q_i(1), q_i's leading coefficient and, in round 0, q_i(0); `h` leaves as the
// message p_i(1), p_i(2), p_i(3) with p_i(X) = eq(X, rt_i) q_i(X).  `r` = the challenge of round i - 1.
static void sc_tower_message(ceno_hip_sumcheck* sc, int i, E2 r, uint64_t* h) {
    auto& F = sc->eqf;
    const E2 rt = F.pt[i];
    const E2 q1{h[0], h[1]}, c2{h[2], h[3]};
    E2 q0;
    if (i == 0 && !F.has_claim) {
        q0 = E2{h[4], h[5]};
    } else if (i == 0) {
        q0 = (F.claim0 - rt * q1) * F.inv1m[0];
    } else {
        // the claim this round answers: p_{i-1}(r) = eq(r, rt_{i-1}) q_{i-1}(r) = (1 - rt_i) q_i(0) + rt_i q_i(1)
        const E2 rp = F.pt[i - 1];
        const E2 eq_r = e2_one() - rp - r + e2_mul_base(rp * r, 2);
        const E2 claim = eq_r * (F.q0 + r * (F.c1 + r * F.c2));
        q0 = (claim - rt * q1) * F.inv1m[i];
    }
    const E2 c1 = q1 - q0 - c2;
    const E2 qa = q0 + e2_mul_base(c1, 2) + e2_mul_base(c2, 4), qb = q0 + e2_mul_base(c1, 3) + e2_mul_base(c2, 9);  // q_i(2), q_i(3)
    const E2 p1 = rt * q1;
    const E2 p2 = (e2_mul_base(rt, 3) - e2_one()) * qa;        // eq(2, rt) = 3 rt - 1
    const E2 p3 = (e2_mul_base(rt, 5) - E2{2, 0}) * qb;        // eq(3, rt) = 5 rt - 2
    h[0] = p1.c0; h[1] = p1.c1; h[2] = p2.c0; h[3] = p2.c1; h[4] = p3.c0; h[5] = p3.c1;
    F.q0 = q0;
    F.c1 = c1;
    F.c2 = c2;
}

// one round; out goes to host (h_out != NULL, waits for the message) or to device memory d_out
static int sc_round(ceno_hip_sumcheck* sc, const uint64_t* challenge2, uint64_t* h_out, uint64_t* d_out) {
    ceno_hip_ctx* ctx = sc->ctx;
    if (sc->finished || sc->round >= sc->n) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "sumcheck: all %d rounds already produced", sc->n);
    const int i = sc->round;
    if (i > 0 && !challenge2) return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "sumcheck round %d needs the challenge of round %d", i, i - 1);
    if (i == 0 && challenge2) return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "sumcheck round 0 takes no challenge");
    const E2 r = i > 0 ? E2{challenge2[0], challenge2[1]} : e2_zero();
    const int d = sc->d;
    size_t h_cursor = 0;

    // ---- 0. pipelined fast path ----
    if (i == 0 && h_out && !d_out && sc_pipeline_eligible(sc)) TRY(sc_pipeline_enqueue(sc, 1 + pipe_lookahead()));
    if (sc->pipelined) {
        if (d_out) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "sumcheck: device-output rounds cannot follow host-output rounds");
        if (sc->host_from >= 0 && i >= sc->host_from) {  // the device has shipped its tables and left: this round is the host's
            TRY(sc_host_round(sc, r, h_out));
            sc->round++;
            return 0;
        }
        if (i > 0) {
            volatile Mailbox* mb = sc->h_mailbox;
            mb->chal[0] = r.c0;
            mb->chal[1] = r.c1;
            host_store_fence();  // BAR mappings are write-combining: order the words before the sequence number ...
            __atomic_store_n(&sc->h_mailbox->chal_seq, (unsigned long long)i, __ATOMIC_RELEASE);
            host_store_fence();  // ... and push the sequence number out of the write-combining buffer now
            TRY(sc_pipeline_enqueue(sc, i + 1 + pipe_lookahead()));  // the device is busy with round i meanwhile
        }
        static const bool dbg = getenv("CENO_HIP_DEBUG") != nullptr;
        timespec ta, tb;
        if (dbg) clock_gettime(CLOCK_MONOTONIC, &ta);
        TRY(sc_take_message(sc, h_out));
        if (sc->eqf.on && i < sc->eqf.fast_upto) sc_tower_message(sc, i, r, h_out);
        if (dbg) {
            clock_gettime(CLOCK_MONOTONIC, &tb);
            static timespec last_ret = {0, 0};
            fprintf(stderr, "[ceno_hip] round %d: host away %.1f us, waited %.1f us\n", i,
                    (ta.tv_sec - last_ret.tv_sec) * 1e6 + (ta.tv_nsec - last_ret.tv_nsec) / 1e3,
                    (tb.tv_sec - ta.tv_sec) * 1e6 + (tb.tv_nsec - ta.tv_nsec) / 1e3);
            last_ret = tb;
        }
        sc->round++;
        return 0;
    }

    // ---- 0b. a single-class sumcheck driven round by round (no persistent kernels): once the remaining rounds are worth more on
    // the host than a launch and a wait each (host_tail_rounds), the live tables are shipped to pinned memory ONCE and every
    // further round — and the final evaluations — is the host's.  (The opening's height groups: ~18 us per device round.) ----
    if (sc->host_from < 0 && h_out && !d_out && sc->h_tail && sc->classes.size() == 1 && sc->classes[0].nv == sc->n && i >= sc->n - sc->host_ht) {
        ScClass& cl = sc->classes[0];
        const MleSlot* d_slots = nullptr;
        size_t cur = 0;
        TRY(sc_push_slots(sc, cl, i, cur, &d_slots));
        const int len = 1 << (i == 0 ? sc->n : sc->n - i + 1);  // the tables message i - 1 was computed on (round 0: the inputs)
        const int total = (int)cl.mles.size() * len;
        hipLaunchKernelGGL(k_export_tables, dim3((unsigned)std::min((total + NT - 1) / NT, 64)), dim3(NT), 0, sc->st, d_slots, (int)cl.mles.size(), len,
                           sc->d_tail_view);
        HIP_TRY(ctx, hipGetLastError());
        sc->host_from = i;
        sc->host_len0 = len;
        sc->host_len = 0;
    }
    if (sc->host_from >= 0) {
        if (d_out) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "sumcheck: the host has taken this sumcheck over; device-output rounds cannot follow");
        TRY(sc_host_round(sc, r, h_out, i > 0));
        sc->round++;
        return 0;
    }
    static const bool phases = getenv("CENO_HIP_ROUND_PHASES") != nullptr;  // where a round of the launch-per-round path spends its time
    timespec tp0, tp1, tp2, tp3;
    if (phases) clock_gettime(CLOCK_MONOTONIC, &tp0);
    // ---- 1. classes that are (or just become) scalars: update tails / bind their last variable ----
    ScClass* became_scalar = nullptr;
    for (auto& cl : sc->classes) {
        if (cl.nv < i) {
            for (int j : cl.mles) sc->mles[j].tail = sc->mles[j].tail * r;  // exhausted earlier
        } else if (cl.nv == i && i > 0) {
            const MleSlot* d_slots = nullptr;
            TRY(sc_push_slots(sc, cl, i, h_cursor, &d_slots));
            // one launch folds the last variable of every table of the class and writes the values straight into the armed pinned
            // words the host watches (as ceno_hip_sumcheck_finish does): a fold, a gather, a device-to-host copy and a stream
            // synchronisation cost ~45 us on each of the rounds in which a class of a mixed-size batch retires
            E2* h_ev = sc->h_pinned + MAXD;
            for (size_t k = 0; k < 2 * cl.mles.size(); k++) reinterpret_cast<uint64_t*>(h_ev)[k] = MSG_INVALID;
            hipLaunchKernelGGL(k_finish_evals, dim3((unsigned)((cl.mles.size() + 63) / 64)), dim3(64), 0, sc->st, d_slots, (int)cl.mles.size(), r,
                               reinterpret_cast<E2*>(sc->d_hmsg) + MAXD);
            became_scalar = &cl;  // at most one class reaches its last variable per round
        }
    }
    auto retire_wait = [&]() -> int {
        if (!became_scalar) return 0;
        ScClass* cl = became_scalar;
        E2* h_ev = sc->h_pinned + MAXD;
        HIP_TRY(ctx, hipGetLastError());
        TRY(sc_wait_words(sc, reinterpret_cast<const uint64_t*>(h_ev), 2 * (int)cl->mles.size(), "the evaluations of a class that reached its last variable"));
        for (size_t k = 0; k < cl->mles.size(); k++) {  // evaluations are in class-local order
            ScMle& M = sc->mles[cl->mles[k]];
            M.eval = h_ev[k];
            M.done = true;
            M.cur = M.buf[M.which];
            M.cur_ext = 1;
        }
        return 0;
    };
    // A round whose only launch is the eq-factored one hands nothing of the retired classes to the device (the host adds their scalars to
    // the message): the wait for a retiring class's evaluations (~10 us) and the scalars then overlap the round's kernel
    bool defer_retire = sc->geq.on && h_out && !d_out && sc->gen_on && i > 0 && sc->gen_rounds[i].n_comps == 0;
    for (auto& cl : sc->classes)
        if (cl.nv > i && (cl.dense || !cl.gen)) defer_retire = false;
    if (!defer_retire) TRY(retire_wait());
    if (phases) clock_gettime(CLOCK_MONOTONIC, &tp1);
    // ---- 2. front-loaded contributions: c_t * prod_j (eval_j * tail_j * t)   (scheme/verifier.rs:233-237) ----
    E2 scalars[MAXD];
    for (int x = 0; x < MAXD; x++) scalars[x] = e2_zero();
    auto compute_scalars = [&]() {
    // c_t prod_j (eval_j tail_j (x + 1)) = (x + 1)^|T| c_t prod_j (eval_j tail_j): the product once per term and round, a small base-field
    // power per point (the retired chips of a mixed-size batch cost 20-40 us of host arithmetic per round with one product per point)
    // (per retired class ONE product per distinct term size and round — the 16 terms of each of ~20 retired chips cost 4-12 us per round
    // term by term: CENO_HIP_ROUND_PHASES=1)
    uint64_t pw[MAXD][17];  // (x + 1)^k
    for (int x = 0; x < d; x++) {
        pw[x][0] = 1;
        for (int k = 1; k < 17; k++) pw[x][k] = gl::mul(pw[x][k - 1], (uint64_t)(x + 1));
    }
    for (auto& cl : sc->classes) {
        if (cl.nv > i) continue;
        if (!cl.front_grouped) {
            const int this_cls = (int)(&cl - sc->classes.data());
            for (int k = 0; k < 17; k++) cl.front_sum[k] = e2_zero();
            for (int t : cl.terms) {
                ScTerm& T = sc->terms[t];
                T.front = T.coeff;
                T.front_one_class = !T.full.empty() && T.full.size() <= 16;
                for (int j : T.full) {
                    T.front = T.front * sc->mles[j].eval;
                    if (sc->mles[j].cls != this_cls) T.front_one_class = false;  // a factor that retired earlier carries another tail
                }
                T.front_ready = true;
                if (T.front_one_class) {
                    cl.front_sum[T.full.size()] = cl.front_sum[T.full.size()] + T.front;
                    cl.front_has[T.full.size()] = true;
                } else {
                    cl.front_slow.push_back(t);
                }
            }
            cl.front_grouped = true;
        }
        if (!cl.mles.empty()) {
            const E2 tl = sc->mles[cl.mles[0]].tail;  // every table of a class has the class's number of variables, hence its tail
            E2 tp_ = e2_one();
            for (int k = 1; k < 17; k++) {
                tp_ = tp_ * tl;
                if (!cl.front_has[k]) continue;
                const E2 pv = cl.front_sum[k] * tp_;
                for (int x = 0; x < d; x++) scalars[x] = scalars[x] + e2_mul_base(pv, pw[x][k]);
            }
        }
        for (int t : cl.front_slow) {
            ScTerm& T = sc->terms[t];
            E2 pv = T.coeff;
            for (int j : T.full) pv = pv * (sc->mles[j].eval * sc->mles[j].tail);
            for (int x = 0; x < d; x++) {
                uint64_t w = 1;
                for (size_t k = 0; k < T.full.size(); k++) w = gl::mul(w, (uint64_t)(x + 1));
                scalars[x] = scalars[x] + e2_mul_base(pv, w);
            }
        }
    }
    };
    if (!defer_retire) compute_scalars();
    if (phases) clock_gettime(CLOCK_MONOTONIC, &tp2);
    // ---- 3. live classes: dense classes launch their fused kernel, all classes with component tables share ONE k_gen
    // launch, the rest take the two-kernel path; the last launch that accumulates finishes the message ----
    std::vector<ScClass*> live;
    for (auto& cl : sc->classes)
        if (cl.nv > i) live.push_back(&cl);
    enum { U_DENSE, U_GEN, U_LEGACY, U_GENEQ };
    struct Unit {
        int kind;
        ScClass* cl;
        bool accumulates;
    };
    std::vector<Unit> units;
    bool gen_added = false;
    // first round of base-field columns: the per-class base-field kernel (k_accum_base0) executes fewer instructions than the
    // blocked kernel's first-round form (no staging, one reduction per pair): taken when EVERY class of the merged launch
    // qualifies (the component list of a round is all or nothing); CENO_HIP_GEN_ROUND0=1 keeps k_gen
    static const bool gen_round0 = getenv("CENO_HIP_GEN_ROUND0") != nullptr && atoi(getenv("CENO_HIP_GEN_ROUND0")) != 0;
    bool legacy_round0 = i == 0 && !gen_round0 && sc->gen_on && !sc->geq.on;  // (eq-factored components need their per-component sums)
    if (sc->geq.on && d_out) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "sumcheck: an eq-factored plan (ceno_hip_sumcheck_begin_eq) produces host messages only");
    for (ScClass* cl : live)
        if (cl->gen && !cl->dense && !cl->terms_all_base) legacy_round0 = false;
    for (ScClass* cl : live) {
        if (cl->dense) units.push_back(Unit{U_DENSE, cl, true});
        else if (cl->gen && sc->gen_on && !legacy_round0) {
            if (!gen_added && sc->gen_rounds[i].n_comps > 0) units.push_back(Unit{U_GEN, nullptr, sc->gen_rounds[i].has_terms});
            // the eq-factored components: their own (component-aligned) launch; it takes no part in the classic message chain — its
            // per-component sums go to the host, which completes and adds them (geq_collect)
            if (!gen_added && !sc->gen_rounds[i].eq.empty()) units.push_back(Unit{U_GENEQ, nullptr, false});
            gen_added = true;
        } else units.push_back(Unit{U_LEGACY, cl, !cl->terms.empty()});
    }
    int last_acc = -1, first_acc = -1;
    for (size_t u = 0; u < units.size(); u++)
        if (units[u].accumulates) {
            if (first_acc < 0) first_acc = (int)u;
            last_acc = (int)u;
        }
    const unsigned long long seq = ++sc->seq;
    for (size_t u = 0; u < units.size(); u++) {
        const Unit& U = units[u];
        Epilogue ep{};
        ep.counter = sc->d_counter;
        ep.round_acc = sc->d_round_acc;
        ep.out_msg = d_out ? d_out : sc->d_hmsg;
        ep.flag = d_out ? nullptr : sc->d_hflag;
        ep.seq = seq;
        ep.coeff = e2_one();
        ep.first_class = ((int)u == first_acc) ? 1 : 0;
        ep.last_class = ((int)u == last_acc) ? 1 : 0;
        ep.d = U.accumulates ? d : 0;
        if (ep.last_class)
            for (int x = 0; x < MAXD; x++) ep.scalars[x] = scalars[x];
        double bytes = 0.0;
        if (U.kind == U_GENEQ) {
            const GenRound& R = sc->gen_rounds[i];
            geq_arm(sc, i);
            ep.partials = sc->geq.d_rows;
            ep.d = 0;
            static const bool phase_dbg = getenv("CENO_HIP_GEN_PHASE_DBG") != nullptr;  // device-side phase stamps of the launch's last workgroup
            if (phase_dbg) {
                ep.bcast = sc->d_bcast;
                ep.dbg = 1;
            }
            // (launches of one round follow each other on the stream: the rows of the first are summed before the second starts)
            for (const GenRound::EqLaunch& EL : R.eq) {
                ctx->eq_launches.fetch_add(1, std::memory_order_relaxed);
                const GenEqArgs ea{1, sc->geq.d_q, sc->geq.d_b, sc->geq.d_counters, reinterpret_cast<const uint16_t*>(sc->d_gen + EL.off_wg_comp), (unsigned)d};
                const GenComp* cs = reinterpret_cast<const GenComp*>(sc->d_gen + EL.off_comps);
                prof_begin(ctx, sc->st);
                if (R.direct0) launch_eq_base0(ctx, EL.D, cs, EL.n_comps, ep, ea, EL.grid, sc->st);
                else if (EL.slots) launch_gen_eq_slots(EL.D, cs, EL.n_comps, r, ep, EL.stage_bytes, sc->st, ea, EL.grid);
                else launch_gen(ctx, EL.D, R.base0, cs, EL.n_comps, 0, r, ep, EL.stage_bytes, sc->st, &ea, EL.grid);
                prof_end(ctx, sc->st, 0.0);
            }
            continue;
        }
        if (U.kind == U_GEN) {
            const GenRound& R = sc->gen_rounds[i];
            ep.partials = reinterpret_cast<uint64_t*>(sc->d_partials);  // the merged launch uses the first class's partial rows
            prof_begin(ctx, sc->st);
            launch_gen(ctx, d, R.base0, reinterpret_cast<const GenComp*>(sc->d_gen + R.off_comps), R.n_comps, R.total_tiles, r, ep, R.stage_bytes, sc->st);
            for (ScClass* cl : live) {
                if (!(cl->gen && !cl->dense)) continue;
                const size_t pairs = (size_t)1 << (cl->nv - i - 1);
                for (int j : cl->mles) {
                    const double in_el = sc->mles[j].cur_ext ? 16.0 : 8.0;
                    if (i == 0) bytes += 2.0 * pairs * in_el;
                    else bytes += 4.0 * pairs * in_el + 2.0 * pairs * 16.0;
                }
            }
            prof_end(ctx, sc->st, bytes);
            continue;
        }
        ScClass& cl = *U.cl;
        const size_t pairs = (size_t)1 << (cl.nv - i - 1);
        const unsigned grid = sc_grid(pairs);
        ep.partials = reinterpret_cast<uint64_t*>(sc->d_partials + cl.part_off);
        if (U.kind == U_DENSE) {
            const ScTerm& T = sc->terms[cl.terms[0]];
            const int K = (int)T.idx.size();
            const bool base_in = !sc->mles[T.idx[0]].cur_ext;
            const int mode = (i == 0 ? 0 : 2) + (base_in ? 1 : 0);
            ep.coeff = T.coeff;
            prof_begin(ctx, sc->st);
            switch (K) {
            case 1: launch_dense<1>(sc, cl, mode, pairs, r, grid, ep); break;
            case 2: launch_dense<2>(sc, cl, mode, pairs, r, grid, ep); break;
            case 3: launch_dense<3>(sc, cl, mode, pairs, r, grid, ep); break;
            default: launch_dense<4>(sc, cl, mode, pairs, r, grid, ep); break;
            }
            const double in_el = base_in ? 8.0 : 16.0;
            if (i == 0) bytes += (double)K * 2.0 * pairs * in_el;
            else bytes += (double)K * (4.0 * pairs * in_el + 2.0 * pairs * 16.0);
            prof_end(ctx, sc->st, bytes);
        } else {
            const MleSlot* d_slots = nullptr;
            TRY(sc_push_slots(sc, cl, i, h_cursor, &d_slots));
            prof_begin(ctx, sc->st);
            const bool tile = !cl.terms.empty() && tile_eligible(cl.mles.size(), pairs);
            const int tnt = cl.terms.empty() || tile ? 0 : fused_tnt(cl.mles.size(), pairs);
            if (i > 0 && tnt == 0 && !tile)
                hipLaunchKernelGGL(k_fold_batch, dim3(grid_for(2 * pairs, NT, 1024), (unsigned)cl.mles.size()), dim3(NT), 0, sc->st, d_slots,
                                   2 * pairs, r, (const Bcast*)nullptr, 0ull);
            if (!cl.terms.empty()) {
                DevPlan pl;
                pl.slots = d_slots;
                pl.use_out = i > 0 ? 1 : 0;
                pl.n_groups = cl.n_groups;
                pl.group_term_off = cl.d_group_term_off;
                pl.group_terms = cl.d_group_terms;
                pl.common_off = cl.d_common_off;
                pl.common_idx = cl.d_common_idx;
                pl.coeffs = cl.d_coeffs;
                pl.term_off = cl.d_term_off;
                pl.term_idx = cl.d_term_idx;
                if (tile) launch_tile(d, pl, (int)cl.mles.size(), cl.n_flat, pairs, r, ep, sc->st);
                else if (tnt) launch_fused(d, tnt, pl, (int)cl.mles.size(), pairs, r, ep, grid_for(pairs, (unsigned)tnt, MAXB), sc->st);
                else launch_accum(ctx, d, pl, pairs, ep, grid, sc->st, i == 0 && cl.terms_all_base);
            }
            for (int j : cl.mles) {
                const double in_el = sc->mles[j].cur_ext ? 16.0 : 8.0;
                if (i == 0) bytes += 2.0 * pairs * in_el;
                else bytes += 4.0 * pairs * in_el + 2.0 * pairs * 16.0;
            }
            prof_end(ctx, sc->st, bytes);
        }
    }
    if (i > 0)
        for (ScClass* cl : live) sc_advance(sc, *cl);
    HIP_TRY(ctx, hipGetLastError());
    if (defer_retire) {
        if (last_acc >= 0) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "sumcheck: a deferred retirement met a round that accumulates on the device");
        TRY(retire_wait());
        compute_scalars();
    }
    if (last_acc < 0) {
        // no term is live in this round: the message consists of the front-loaded scalars only
        if (d_out) {
            memcpy(sc->h_pinned, scalars, (size_t)d * sizeof(E2));
            HIP_TRY(ctx, hipMemcpyAsync(d_out, sc->h_pinned, (size_t)d * sizeof(E2), hipMemcpyHostToDevice, sc->st));
            HIP_TRY(ctx, hipStreamSynchronize(sc->st));
        } else {
            memcpy(h_out, scalars, (size_t)d * sizeof(E2));
            if (phases) clock_gettime(CLOCK_MONOTONIC, &tp3);
            if (sc->geq.on) TRY(geq_collect(sc, i, r, h_out));
            if (phases) {
                timespec tp4;
                clock_gettime(CLOCK_MONOTONIC, &tp4);
                auto us = [](const timespec& a, const timespec& b) { return (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) / 1e3; };
                static timespec last_ret = tp4;
                fprintf(stderr, "[ceno_hip] round %d of %d (eq-factored): caller %.1f us, retire %.1f us, scalars %.1f us, arm + launch %.1f us, wait + complete %.1f us\n", i, sc->n,
                        us(last_ret, tp0), us(tp0, tp1), us(tp1, tp2), us(tp2, tp3), us(tp3, tp4));
                last_ret = tp4;
            }
            if (sc->geq.on && getenv("CENO_HIP_GEN_PHASE_DBG") && sc->d_bcast) {
                static Bcast hb;
                (void)hipStreamSynchronize(sc->st);
                if (hipMemcpy(&hb, sc->d_bcast, sizeof(Bcast), hipMemcpyDeviceToHost) == hipSuccess) {
                    const unsigned long long* a = hb.dbg[seq & 31];
                    fprintf(stderr, "[ceno_hip] eq round %d, last workgroup: find component %.1f us, phase 1 %.1f us, groups %.1f us, epilogue %.1f us\n", i,
                            (a[1] - a[0]) / 100.0, (a[2] - a[1]) / 100.0, (a[3] - a[2]) / 100.0, (hb.dbg[32 + (seq & 31)][0] - a[3]) / 100.0);
                }
            }
        }
    } else if (h_out) {
        if (phases) clock_gettime(CLOCK_MONOTONIC, &tp3);
        TRY(sc_take_message(sc, h_out));
        if (sc->geq.on) TRY(geq_collect(sc, i, r, h_out));
        if (phases) {
            timespec tp4;
            clock_gettime(CLOCK_MONOTONIC, &tp4);
            auto us = [](const timespec& a, const timespec& b) { return (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) / 1e3; };
            fprintf(stderr, "[ceno_hip] round %d of %d: retire %.1f us, scalars %.1f us, launch %.1f us, wait %.1f us\n", i, sc->n, us(tp0, tp1), us(tp1, tp2),
                    us(tp2, tp3), us(tp3, tp4));
        }
    }
    sc->round++;
    return 0;
}

void launch_setup_job(const SetupJob& job, hipStream_t st) {
    hipLaunchKernelGGL(k_setup, dim3(1), dim3(256), 0, st, job.zero, job.zero_words, job.dst, job.src_host_view, job.words, job.ones, job.ones_words);
}
int sumcheck_begin_deferred(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan, hipStream_t st, ceno_hip_sumcheck** out,
                            SetupJob* job) {
    *job = SetupJob{};
    return sc_build(ctx, mles, plan, st, out, job);
}

extern "C" {

int ceno_hip_sumcheck_begin(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan, ceno_hip_stream s,
                            ceno_hip_sumcheck** out) {
    return sc_build(ctx, mles, plan, ctx_stream(ctx, s), out);
}

int ceno_hip_sumcheck_begin_eq(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan, int num_eq, const int* eq_mle_idx,
                               const uint64_t* const* eq_points, const size_t* eq_lo, const size_t* eq_hi, ceno_hip_stream s, ceno_hip_sumcheck** out) {
    CHECK_ARG(ctx, num_eq >= 0 && (num_eq == 0 || (eq_mle_idx && eq_points && eq_lo && eq_hi)), "eq declarations: NULL array");
    const EqDeclArgs ea{num_eq, eq_mle_idx, eq_points, eq_lo, eq_hi};
    return sc_build(ctx, mles, plan, ctx_stream(ctx, s), out, nullptr, num_eq > 0 ? &ea : nullptr);
}

uint64_t ceno_hip_stat_eq_launches(const ceno_hip_ctx* ctx) { return ctx ? (uint64_t)ctx->eq_launches.load() : 0ull; }

const char* ceno_hip_plan_report(const ceno_hip_ctx* ctx) { return ctx ? ctx->plan_report.c_str() : ""; }
int ceno_hip_sumcheck_eq_components(const ceno_hip_sumcheck* sc) { return sc && sc->geq.on ? (int)sc->geq.comps.size() : 0; }

int ceno_hip_sumcheck_round(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, const uint64_t* challenge2, uint64_t* out_evals) {
    CHECK_ARG(ctx, sc && out_evals, "NULL argument");
    return sc_round(sc, challenge2, out_evals, nullptr);
}

int ceno_hip_sumcheck_round_dev(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, const uint64_t* challenge2, uint64_t* dev_out_evals) {
    CHECK_ARG(ctx, sc && dev_out_evals, "NULL argument");
    return sc_round(sc, challenge2, nullptr, dev_out_evals);
}

int ceno_hip_sumcheck_finish(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, const uint64_t* last_challenge2, uint64_t* final_evals) {
    CHECK_ARG(ctx, sc && final_evals, "NULL argument");
    CENO_TIMED("sumcheck_finish");
    if (sc->finished) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "sumcheck already finished");
    if (sc->round != sc->n) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "sumcheck finish after %d of %d rounds", sc->round, sc->n);
    if (sc->n > 0) {
        CHECK_ARG(ctx, last_challenge2, "last challenge is NULL");
        const E2 r{last_challenge2[0], last_challenge2[1]};
        size_t h_cursor = 0;
        if (sc->host_from >= 0) {
            // host-finished tail: the last fold happens here
            ScClass& cl = sc->classes[0];
            TRY(host_take_over(sc));
            host_fold(sc, r);
            for (size_t k = 0; k < cl.mles.size(); k++) {
                ScMle& M = sc->mles[cl.mles[k]];
                M.eval = sc->host_tab[k * (size_t)sc->host_len0];
                M.done = true;
            }
        } else if (sc->pipelined && sc->tail_evals) {
            // the persistent tail kernel is waiting for this challenge and writes the evaluations itself
            ScClass& cl = sc->classes[0];
            volatile Mailbox* mb = sc->h_mailbox;
            mb->chal[0] = r.c0;
            mb->chal[1] = r.c1;
            host_store_fence();
            __atomic_store_n(&sc->h_mailbox->chal_seq, (unsigned long long)sc->n, __ATOMIC_RELEASE);
            host_store_fence();
            E2* h_ev = sc->h_pinned + MAXD;
            TRY(sc_wait_words(sc, reinterpret_cast<const uint64_t*>(h_ev), 2 * (int)cl.mles.size()));
            for (size_t k = 0; k < cl.mles.size(); k++) {
                ScMle& M = sc->mles[cl.mles[k]];
                M.eval = h_ev[k];
                M.done = true;
            }
        } else
        for (auto& cl : sc->classes) {
            if (cl.nv != sc->n) continue;
            // the slot table is read by the kernel from the pinned block itself (device view of the same words): no
            // host-to-device blit in front of a kernel that reads a few hundred bytes once
            MleSlot* h = sc->h_slots + (size_t)(sc->n + 1) * sc->slots_per_round + h_cursor;
            for (size_t k = 0; k < cl.mles.size(); k++) {
                ScMle& M = sc->mles[cl.mles[k]];
                h[k].in = M.cur;
                h[k].out = M.buf[M.which];
                h[k].in_ext = M.cur_ext;
                h[k].pad = 0;
            }
            h_cursor += cl.mles.size();
            const MleSlot* d_slots = reinterpret_cast<const MleSlot*>(reinterpret_cast<char*>(sc->d_hflag) + (reinterpret_cast<char*>(h) - reinterpret_cast<char*>(sc->h_block)));
            E2* h_ev = sc->h_pinned + MAXD;
            E2* d_ev = reinterpret_cast<E2*>(sc->d_hmsg) + MAXD;  // device view of the same pinned words
            for (size_t k = 0; k < 2 * cl.mles.size(); k++) reinterpret_cast<uint64_t*>(h_ev)[k] = MSG_INVALID;  // arm (several classes share the words)
            hipLaunchKernelGGL(k_finish_evals, dim3((unsigned)((cl.mles.size() + 63) / 64)), dim3(64), 0, sc->st, d_slots, (int)cl.mles.size(), r, d_ev);
            HIP_TRY(ctx, hipGetLastError());
            // the evaluation words were armed with MSG_INVALID at begin: watching them costs the kernel's own time, a stream
            // synchronisation ~10 us more (completion signal + runtime).  Everything queued before this kernel has completed
            // when its stores are visible, and they are its last instructions.
            TRY(sc_wait_words(sc, reinterpret_cast<const uint64_t*>(h_ev), 2 * (int)cl.mles.size()));
            for (size_t k = 0; k < cl.mles.size(); k++) {
                ScMle& M = sc->mles[cl.mles[k]];
                M.eval = h_ev[k];
                M.done = true;
            }
        }
    }
    for (size_t j = 0; j < sc->mles.size(); j++) {
        final_evals[2 * j] = sc->mles[j].eval.c0;
        final_evals[2 * j + 1] = sc->mles[j].eval.c1;
    }
    sc->finished = true;
    return 0;
}

int ceno_hip_sumcheck_rounds_done(const ceno_hip_sumcheck* sc) { return sc ? sc->round : -1; }

int ceno_hip_sumcheck_table(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int mle_index, uint64_t** device_ptr, int* is_ext, int* num_vars) {
    CHECK_ARG(ctx, sc && device_ptr && is_ext && num_vars, "NULL argument");
    CHECK_ARG(ctx, mle_index >= 0 && mle_index < (int)sc->mles.size(), "mle index out of range");
    if (sc->pipelined) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "tables of a pipelined sumcheck are not observable between rounds");
    if (sc->host_from >= 0) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "the host has taken this sumcheck over: its tables are no longer on the device");
    const ScMle& M = sc->mles[mle_index];
    // after r rounds the live table is the one round r-1 was computed on: nv - (r - 1) variables (r >= 1), nv before round 0
    const int folds = sc->round > 0 ? sc->round - 1 : 0;
    if (M.nv < folds) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "table %d is already a scalar", mle_index);
    *device_ptr = const_cast<uint64_t*>(M.cur);
    *is_ext = M.cur_ext;
    *num_vars = M.nv - folds;
    return 0;
}

int ceno_hip_sumcheck_table_host(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int mle_index, uint64_t* out_host, size_t cap_ext, int* num_vars) {
    CHECK_ARG(ctx, sc && out_host && num_vars, "NULL argument");
    CHECK_ARG(ctx, mle_index >= 0 && mle_index < (int)sc->mles.size(), "mle index out of range");
    if (sc->pipelined) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "tables of a pipelined sumcheck are not observable between rounds");
    if (sc->host_from >= 0) {  // the host has taken the last rounds over: the live tables are in its copy
        if (!sc->host_len) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "the host copy of the tables is not there yet");
        const ScMle& M = sc->mles[mle_index];
        if (M.cls != 0 || M.local < 0) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "table %d is not part of the host-finished class", mle_index);
        const size_t len = (size_t)sc->host_len;
        if (len > cap_ext) return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "table_host: buffer of %zu elements for a table of %zu", cap_ext, len);
        memcpy(out_host, sc->host_tab.data() + (size_t)M.local * (size_t)sc->host_len0, len * sizeof(E2));
        int nv = 0;
        while (((size_t)1 << nv) < len) nv++;
        *num_vars = nv;
        return 0;
    }
    uint64_t* d = nullptr;
    int is_ext = 0, nv = 0;
    TRY(ceno_hip_sumcheck_table(ctx, sc, mle_index, &d, &is_ext, &nv));
    const size_t len = (size_t)1 << nv;
    if (len > cap_ext) return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "table_host: buffer of %zu elements for a table of %zu", cap_ext, len);
    if (is_ext) {
        HIP_TRY(ctx, hipMemcpyAsync(out_host, d, len * sizeof(E2), hipMemcpyDeviceToHost, sc->st));
        HIP_TRY(ctx, hipStreamSynchronize(sc->st));
    } else {  // a base-field table (before its first fold): widened on the way out
        std::vector<uint64_t> b(len);
        HIP_TRY(ctx, hipMemcpyAsync(b.data(), d, len * 8, hipMemcpyDeviceToHost, sc->st));
        HIP_TRY(ctx, hipStreamSynchronize(sc->st));
        for (size_t j = 0; j < len; j++) {
            out_host[2 * j] = b[j];
            out_host[2 * j + 1] = 0;
        }
    }
    *num_vars = nv;
    return 0;
}

// every listed table in one go: all device-to-host copies queued, ONE synchronisation (a wide chip's handle has hundreds of tables; the sharded main
// constraints fetch all of them before their gathered tail: a copy + wait per table was most of that phase)
int ceno_hip_sumcheck_tables_host(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int n, const int* mle_indices, uint64_t* const* outs_host, const size_t* caps_ext,
                                  int* num_vars) {
    CHECK_ARG(ctx, sc && n >= 0 && (n == 0 || (mle_indices && outs_host && caps_ext && num_vars)), "NULL argument");
    if (sc->pipelined) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "tables of a pipelined sumcheck are not observable between rounds");
    if (sc->host_from >= 0) {  // the host's copy: nothing to wait for
        for (int k = 0; k < n; k++) TRY(ceno_hip_sumcheck_table_host(ctx, sc, mle_indices[k], outs_host[k], caps_ext[k], &num_vars[k]));
        return 0;
    }
    std::vector<std::pair<int, std::vector<uint64_t>>> base;  // (position, staging) of base-field tables: widened after the wait
    for (int k = 0; k < n; k++) {
        CHECK_ARG(ctx, mle_indices[k] >= 0 && mle_indices[k] < (int)sc->mles.size() && outs_host[k], "mle index out of range");
        uint64_t* d = nullptr;
        int is_ext = 0, nv = 0;
        TRY(ceno_hip_sumcheck_table(ctx, sc, mle_indices[k], &d, &is_ext, &nv));
        const size_t len = (size_t)1 << nv;
        if (len > caps_ext[k]) return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "tables_host: buffer of %zu elements for a table of %zu", caps_ext[k], len);
        num_vars[k] = nv;
        if (is_ext) {
            HIP_TRY(ctx, hipMemcpyAsync(outs_host[k], d, len * sizeof(E2), hipMemcpyDeviceToHost, sc->st));
        } else {
            base.emplace_back(k, std::vector<uint64_t>(len));
            HIP_TRY(ctx, hipMemcpyAsync(base.back().second.data(), d, len * 8, hipMemcpyDeviceToHost, sc->st));
        }
    }
    HIP_TRY(ctx, hipStreamSynchronize(sc->st));
    for (auto& b : base)
        for (size_t j = 0; j < b.second.size(); j++) {
            outs_host[b.first][2 * j] = b.second[j];
            outs_host[b.first][2 * j + 1] = 0;
        }
    return 0;
}

int ceno_hip_sumcheck_set_pipelined(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int on) {
    CHECK_ARG(ctx, sc, "NULL argument");
    if (sc->round != 0) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "sumcheck: pipelining must be chosen before round 0");
    sc->allow_pipeline = on != 0;
    return 0;
}

int ceno_hip_sumcheck_free(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc) {
    (void)ctx;
    sc_release(sc);
    return 0;
}


// estimate_sumcheck_memory (EXT ceno_gpu; call sites ceno_zkvm/src/scheme/gpu/memory.rs:413-433,768,1114): device bytes a
// sumcheck over MLEs with the given numbers of variables allocates ON TOP of its (borrowed) inputs — what a scheduler
// books before it starts the task.  Mirrors sc_build: per MLE a ping (half size) and a pong (quarter size) ext buffer,
// the per-workgroup partials, message / evaluation / counter blocks and the plan blob; every block rounded up to the
// pool's bucket size.
size_t ceno_hip_sumcheck_estimate_memory(int max_num_vars, int max_degree, const int* mle_num_vars, int num_mles, int num_terms) {
    auto bucket = [](size_t b) {
        size_t r = 256;
        while (r < b) r <<= 1;
        return r;
    };
    (void)max_num_vars;
    size_t total = 0, small = 0;  // blocks of at most SC_ARENA_ITEM_MAX come out of the handle's arena chunks (sc_dev_alloc)
    auto block = [&](size_t b) {
        if (b <= SC_ARENA_ITEM_MAX) small += (b + 255) & ~(size_t)255;
        else total += bucket(b);
    };
    for (int j = 0; j < num_mles; j++) {
        const int nv = mle_num_vars ? mle_num_vars[j] : max_num_vars;
        if (nv >= 1) block(((size_t)1 << (nv - 1)) * sizeof(E2));
        if (nv >= 2) block(((size_t)1 << (nv - 2)) * sizeof(E2));
    }
    total += (small / (SC_ARENA_CHUNK - SC_ARENA_ITEM_MAX) + 2) * SC_ARENA_CHUNK;  // (a chunk is left when the next block does not fit)
    total += bucket((size_t)MAXB * MAXD * sizeof(E2) * 4);                       // partials (per size class, a few classes)
    total += 3 * bucket(MAXD * sizeof(E2)) + bucket((size_t)std::max(num_mles, 1) * sizeof(E2)) + bucket(4096);
    total += bucket((size_t)std::max(num_terms, 1) * (sizeof(E2) + 8 * (size_t)std::max(max_degree, 1)) + (size_t)num_mles * 64);  // plan blob
    total += bucket((size_t)std::max(num_mles, 1) * sizeof(MleSlot) * (size_t)(std::max(max_num_vars, 0) + 2));               // slot tables
    return total;
}

}  // extern "C"

// used by tower.hip: attach an MLE whose lifetime is tied to the sumcheck handle
// Called by ceno_hip_tower_layer_sumcheck_begin on the handle it built (MLE 0 = eq(., rt), one common-factor group over every term):
// turns the k_tower rounds on when the plan has exactly the shape the kernel is written for.  Anything else keeps the generic rounds.
void sumcheck_enable_tower_fast(ceno_hip_sumcheck* sc, const uint64_t* rt, int n_prod, int n_logup) {
    // (read per call, not cached: the test-suite switches it between sumchecks)
    const char* e = getenv("CENO_HIP_TOWER_FAST_MIN_LOG");  // rounds with at least 2^this pairs run on k_tower
    const int min_log = e ? std::max(atoi(e), 6) : 16;
    if (!tower_fast_shape(n_prod, n_logup) || sc->d != 3 || sc->classes.size() != 1) return;
    const ScClass& cl = sc->classes[0];
    const int k = 1 + 2 * n_prod + 4 * n_logup;
    if (cl.nv != sc->n || cl.dense || (int)cl.mles.size() != k || (int)cl.terms.size() != n_prod + 3 * n_logup || sc->n - min_log < 1) return;
    for (int m = 0; m < k; m++)
        if (cl.mles[m] != m || !sc->mles[m].cur_ext) return;
    auto term_is = [&](int t, int a, int b) {
        const ScTerm& T = sc->terms[cl.terms[t]];
        return T.idx.size() == 2 && T.idx[0] == a && T.idx[1] == b && T.full.size() == 3;
    };
    auto& F = sc->eqf;
    for (int i = 0; i < n_prod; i++) {
        if (!term_is(i, 1 + 2 * i, 2 + 2 * i)) return;
        F.coef.prod[i] = sc->terms[cl.terms[i]].coeff;
    }
    for (int j = 0; j < n_logup; j++) {
        const int t = n_prod + 3 * j, b = 1 + 2 * n_prod + 4 * j;
        if (!term_is(t, b, b + 3) || !term_is(t + 1, b + 1, b + 2) || !term_is(t + 2, b + 2, b + 3)) return;
        const E2 an = sc->terms[cl.terms[t]].coeff, an2 = sc->terms[cl.terms[t + 1]].coeff;
        if (an.c0 != an2.c0 || an.c1 != an2.c1) return;
        F.coef.logup[j][0] = an;
        F.coef.logup[j][1] = sc->terms[cl.terms[t + 2]].coeff;
    }
    F.fast_upto = sc->n - min_log;  // round i has 2^(n - 1 - i) pairs
    F.pt.resize((size_t)F.fast_upto);
    F.inv1m.assign((size_t)F.fast_upto, e2_zero());
    // 1 / (1 - rt_i) for the rounds 0 .. fast_upto - 1, one inversion (prefix products)
    std::vector<E2> pre((size_t)F.fast_upto, e2_one());
    E2 run = e2_one();
    for (int i = 0; i < F.fast_upto; i++) {
        F.pt[i] = E2{rt[2 * i], rt[2 * i + 1]};
        const E2 v = e2_one() - F.pt[i];
        if (v.c0 == 0 && v.c1 == 0) return;  // rt_i = 1: the claim does not determine q_i(0)
        pre[i] = run;
        run = run * v;
    }
    E2 inv = e2_inv(run);
    for (int i = F.fast_upto - 1; i >= 0; i--) {
        F.inv1m[i] = inv * pre[i];
        inv = inv * (e2_one() - F.pt[i]);
    }
    F.n_prod = n_prod;
    F.n_logup = n_logup;
    F.on = true;
}

extern "C" int ceno_hip_sumcheck_set_claim(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, const uint64_t* claim2) {
    CHECK_ARG(ctx, sc && claim2, "NULL argument");
    CHECK_ARG(ctx, claim2[0] < gl::P && claim2[1] < gl::P, "sumcheck claim is not canonical");
    if (sc->round != 0 || sc->enq != 0) return ctx_fail(ctx, CENO_HIP_ERR_STATE, "sumcheck: the claim must be stated before round 0");
    sc->eqf.has_claim = true;  // only the fused tower rounds use it; every other handle computes its messages without the sum
    sc->eqf.claim0 = E2{claim2[0], claim2[1]};
    return 0;
}

extern "C" int ceno_hip_sumcheck_fused_eq_rounds(const ceno_hip_sumcheck* sc) { return sc && sc->eqf.on ? sc->eqf.fast_upto : 0; }

void sumcheck_adopt_mle(ceno_hip_sumcheck* sc, ceno_hip_mle* m) { sc->extra_owned = m; }
void sumcheck_adopt_alloc(ceno_hip_sumcheck* sc, void* p) { if (p) sc->dev_allocs.push_back(p); }
