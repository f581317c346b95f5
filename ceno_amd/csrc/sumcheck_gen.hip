// Generic sumcheck round, LDS-blocked: ONE launch per round for every size class with a CSR term plan.
//
// Reference operator: `prove_generic_sumcheck_gpu(_v2)` with a `CommonTermPlan` (gkr_iop/src/gkr/layer/gpu/mod.rs:259-271,
// ceno_zkvm/src/scheme/gpu/mod.rs:2811-2982) = EXT `IOPProverState::prove` over monomial terms
// (ceno_zkvm/src/scheme/cpu/mod.rs:1332-1337).  Same messages as k_fold_batch + k_accum (sumcheck.hip); what changes is
// how the work meets the machine:
//   * fold and accumulate are ONE pass: a table of round i-1 is read once, folded with r_{i-1}, written half-size and used
//     for the message of round i (the two-kernel path wrote the folded table and read it back: +33 % traffic per round);
//   * every column is loaded ONCE per pair and shared by all terms that reference it: a workgroup owns a tile of 64..256
//     consecutive pairs of one connected COMPONENT of the plan (a chip: its columns, its selectors, its terms), phase 1
//     folds (MLE x pair) items — wave-uniform MLE, 64 B contiguous per lane, all loads of a tile in flight together — and
//     stages (f(1), f(1) - f(0)) in LDS; phase 2 evaluates the plan from LDS.  The two-kernel path issued one dependent L2
//     load pair per term factor (PMC: 35 % VALU active, 31 % instruction wait at 5 waves per SIMD);
//   * phase 2 splits the TERMS of a group over the waves of the workgroup (wave-uniform term: plan data through the scalar
//     cache, no divergence, conflict-free LDS rows) and keeps one pair per lane; every wave multiplies its partial sum by
//     the group's common factors (the selectors) — the sum over terms is linear;
//   * all live size classes of a round share the launch (a tile list over all components), so a batched main sumcheck is
//     one kernel per round instead of two per class.
// Round 0 of a main-constraint sumcheck (base-field witness columns under extension-field selectors) has its own phase 2:
// column products stay in the base field and c_t * P_t goes unreduced into 160-bit accumulators (as k_accum_base0).
//
// EQ-FACTORED FORM (k_gen_eq; the main-constraint sumcheck of ceno_zkvm/src/scheme/cpu/mod.rs:1052-1390, whose every group is
// selector x sum_t c_t prod(columns) with the selector = eq(., rt) on a row range, gkr_iop/src/selector.rs:131-245).  After i folds a
// selector table is EQ_i[2y + b] = w[y] eq(b, rt_i) on every pair that lies inside (or outside) the row range, w[y] = EQ_i[2y] +
// EQ_i[2y+1], so the component's round polynomial is
//     p(X) = eq(X, rt_i) Q(X) + B(X),   Q(X) = sum_y w[y] G(X, y)  of degree D - 1,   G = sum_t c_t prod_j f_j(X, y),
// and the kernel evaluates G at D - 2 points and its leading coefficient (only the terms of full degree have one) instead of the product
// at D points: the host gets Q(0) from the component's running claim and completes the message (sumcheck.hip).  The at most two pairs per
// group that straddle an end of the row range are written as  w' eq(X, rt_i) + c X  with  w' = EQ_i[2y] / (1 - rt_i): w' joins Q, and the
// lane hands c G — at one more point, X = D - 1 — to the host, which forms B(X) = X (c G)(X).  Exact field arithmetic: the words of the
// message are the generic kernel's.
#include "sumcheck_dev.hpp"
#include "sumcheck_gen.hpp"

#include <algorithm>
#include <cstdlib>

static constexpr unsigned GEN_FIXED = 640;  // bytes in front of the stage: block-sum scratch (4 x MAXD E2), challenge words, flag

// Plan records are read through the CONSTANT address space: their addresses are wave-uniform (the wave index comes from
// readfirstlane), so these become scalar loads served by the scalar cache instead of per-lane flat loads in front of every
// LDS read.
#define GEN_CONST __attribute__((address_space(4)))
typedef unsigned int gen_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int gen_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gen_u4 ldc4(const void* p) { return *reinterpret_cast<const GEN_CONST gen_u4*>(reinterpret_cast<uintptr_t>(p)); }
__device__ __forceinline__ unsigned ldc_u16(const uint16_t* p) { return *reinterpret_cast<const GEN_CONST uint16_t*>(reinterpret_cast<uintptr_t>(p)); }
__device__ __forceinline__ uint64_t u64_of(unsigned lo, unsigned hi) { return ((uint64_t)hi << 32) | lo; }


// ---- phase 1 of one tile: (MLE x pair) items: fold, write the half-size tables, stage (f(1), f(0) - f(1)) ----
template <bool BASE0>
__device__ __forceinline__ void gen_phase1(const GenComp& C, size_t p0, E2* stage, const E2Pre& rp, const E2& r, int wave, unsigned lane) {
    const unsigned tp = 1u << C.tp_log, tpp = tp + GEN_PAD;
    const bool staged = C.n_groups != 0;
    auto item = [&](const uint64_t* s_in, uint64_t* s_out, bool in_ext, unsigned row, unsigned q) {
        const size_t p = p0 + q;
        if (p >= C.pairs) return;
        if (BASE0 && !in_ext) {
            // first round, base-field column: stays in the base field (f(1), f(0) - f(1))
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(s_in + 2 * p);
            reinterpret_cast<ulonglong2*>(stage)[row + q] = ulonglong2{v.y, sub(v.x, v.y)};
            return;
        }
        E2 lo, hi;
        if (C.fold) {
            if (in_ext) {
                const uint64_t* qq = s_in + 8 * p;
                const E2 a0 = ld_e2(qq), a1 = ld_e2(qq + 2), a2 = ld_e2(qq + 4), a3 = ld_e2(qq + 6);
                lo = e2_fma_pre(rp, a1 - a0, a0);
                hi = e2_fma_pre(rp, a3 - a2, a2);
            } else {
                const uint64_t* qq = s_in + 4 * p;
                const ulonglong2 v0 = *reinterpret_cast<const ulonglong2*>(qq);
                const ulonglong2 v1 = *reinterpret_cast<const ulonglong2*>(qq + 2);
                const E2 t0 = e2_mul_base(r, sub(v0.y, v0.x)), t1 = e2_mul_base(r, sub(v1.y, v1.x));
                lo = E2{add(t0.c0, v0.x), t0.c1};
                hi = E2{add(t1.c0, v1.x), t1.c1};
            }
            if (s_out) {  // (a column staged by several column blocks of one chip is written by the block that owns it)
                st_e2(s_out + 4 * p, lo);
                st_e2(s_out + 4 * p + 2, hi);
            }
        } else if (in_ext) {
            lo = ld_e2(s_in + 4 * p);
            hi = ld_e2(s_in + 4 * p + 2);
        } else {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(s_in + 2 * p);
            lo = E2{v.x, 0};
            hi = E2{v.y, 0};
        }
        if (staged) {
            stage[row + q] = hi;             // f(1)
            stage[row + tpp + q] = lo - hi;  // f(0) - f(1): evaluation points step by subtraction (two instructions shorter)
        }
    };
    const size_t left = C.pairs - p0 < (size_t)tp ? C.pairs - p0 : (size_t)tp;  // pairs of this tile
    if (left <= 32) {
        // A SMALL tile (the last rounds of a chip: a handful of pairs) leaves most lanes of a wave idle when every wave takes one MLE at a
        // time, and the MLEs of a wave then queue behind each other's slot and table loads (~3 dependent round trips each: 44 us for a round
        // of eight chips with <= 64 pairs).  Here a wave takes 64 / w MLEs at once, w = the tile's pair count rounded up to a power of two.
        const unsigned w = left <= 1 ? 1u : 1u << (32 - __builtin_clz((unsigned)left - 1));
        const unsigned sub = 64u / w, ml = lane / w, q = lane % w;
        for (unsigned m0 = (unsigned)wave * sub; m0 < C.n_mles; m0 += (NT / 64) * sub) {
            const unsigned m = m0 + ml;
            if (m < C.n_mles && q < left) {
                const MleSlot sl = C.slots[m];
                item(sl.in, sl.out, sl.in_ext != 0, (unsigned)C.unit[m] * tpp, q);
            }
        }
    } else {
        // MLE uniform per wave
        for (unsigned m = (unsigned)wave; m < C.n_mles; m += NT / 64) {
            // MleSlot is 24 bytes: three 8-byte scalar loads
            const GEN_CONST gen_u2* sp = reinterpret_cast<const GEN_CONST gen_u2*>(reinterpret_cast<uintptr_t>(C.slots + m));
            const gen_u2 sw0 = sp[0], sw1 = sp[1], sw2 = sp[2];
            const uint64_t* s_in = reinterpret_cast<const uint64_t*>(u64_of(sw0.x, sw0.y));
            uint64_t* s_out = reinterpret_cast<uint64_t*>(u64_of(sw1.x, sw1.y));
            const bool in_ext = sw2.x != 0;
            const unsigned row = ldc_u16(C.unit + m) * tpp;
            for (unsigned q = lane; q < tp; q += 64) {
                if (p0 + q >= C.pairs) break;
                item(s_in, s_out, in_ext, row, q);
            }
        }
    }
}

template <int D, bool BASE0>
__global__ void __launch_bounds__(NT) k_gen(const GenComp* __restrict__ comps, int n_comps, unsigned total_tiles, E2 r, Epilogue ep,
                                            unsigned xch_off /* byte offset of the exchange block behind the stage */) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    E2* smem = reinterpret_cast<E2*>(dyn);
    unsigned long long* s_chal = reinterpret_cast<unsigned long long*>(dyn + (NT / 64) * MAXD * sizeof(E2));
    int* s_flag = reinterpret_cast<int*>(s_chal + 4);
    E2* stage = reinterpret_cast<E2*>(dyn + GEN_FIXED);
    E2* xch = reinterpret_cast<E2*>(dyn + GEN_FIXED + xch_off);  // [wave][D][64]
    if (ep.dbg && ep.bcast && blockIdx.x == 0 && threadIdx.x == 0) ep.bcast->dbg[ep.seq & 63][0] = wall_clock64();
    if (ep.wait_seq != 0) {
        if (!read_challenge(ep, r, s_chal)) return;
    }
    const E2Pre rp = e2_pre(r);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63;
    E2 acc[D];
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    int c = 0;
    for (unsigned tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        while (c + 1 < n_comps && comps[c + 1].tile_begin <= tile) c++;  // tiles are visited in increasing order
        const GenComp& C = comps[c];
        const unsigned tp = 1u << C.tp_log, tpp = tp + GEN_PAD;
        const size_t p0 = (size_t)(tile - C.tile_begin) << C.tp_log;
        const bool staged = C.n_groups != 0;
        // ---- phase 1: (MLE x pair) items ----
        auto item = [&](const uint64_t* s_in, uint64_t* s_out, bool in_ext, unsigned row, unsigned q) {
            const size_t p = p0 + q;
            if (p >= C.pairs) return;
            if (BASE0 && !in_ext) {
                // first round, base-field column: stays in the base field (f(1), f(0) - f(1))
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(s_in + 2 * p);
                reinterpret_cast<ulonglong2*>(stage)[row + q] = ulonglong2{v.y, sub(v.x, v.y)};
                return;
            }
            E2 lo, hi;
            if (C.fold) {
                if (in_ext) {
                    const uint64_t* qq = s_in + 8 * p;
                    const E2 a0 = ld_e2(qq), a1 = ld_e2(qq + 2), a2 = ld_e2(qq + 4), a3 = ld_e2(qq + 6);
                    lo = e2_fma_pre(rp, a1 - a0, a0);
                    hi = e2_fma_pre(rp, a3 - a2, a2);
                } else {
                    const uint64_t* qq = s_in + 4 * p;
                    const ulonglong2 v0 = *reinterpret_cast<const ulonglong2*>(qq);
                    const ulonglong2 v1 = *reinterpret_cast<const ulonglong2*>(qq + 2);
                    const E2 t0 = e2_mul_base(r, sub(v0.y, v0.x)), t1 = e2_mul_base(r, sub(v1.y, v1.x));
                    lo = E2{add(t0.c0, v0.x), t0.c1};
                    hi = E2{add(t1.c0, v1.x), t1.c1};
                }
                if (s_out) {  // (a column staged by several column blocks of one chip is written by the block that owns it)
                    st_e2(s_out + 4 * p, lo);
                    st_e2(s_out + 4 * p + 2, hi);
                }
            } else if (in_ext) {
                lo = ld_e2(s_in + 4 * p);
                hi = ld_e2(s_in + 4 * p + 2);
            } else {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(s_in + 2 * p);
                lo = E2{v.x, 0};
                hi = E2{v.y, 0};
            }
            if (staged) {
                stage[row + q] = hi;             // f(1)
                stage[row + tpp + q] = lo - hi;  // f(0) - f(1): evaluation points step by subtraction (two instructions shorter)
            }
        };
        const size_t left = C.pairs - p0 < (size_t)tp ? C.pairs - p0 : (size_t)tp;  // pairs of this tile
        if (left <= 32) {
            // A SMALL tile (the last rounds of a chip: a handful of pairs) leaves most lanes of a wave idle when every wave takes one MLE at a
            // time, and the MLEs of a wave then queue behind each other's slot and table loads (~3 dependent round trips each: 44 us for a round
            // of eight chips with <= 64 pairs).  Here a wave takes 64 / w MLEs at once, w = the tile's pair count rounded up to a power of two.
            const unsigned w = left <= 1 ? 1u : 1u << (32 - __builtin_clz((unsigned)left - 1));
            const unsigned sub = 64u / w, ml = lane / w, q = lane % w;
            for (unsigned m0 = (unsigned)wave * sub; m0 < C.n_mles; m0 += (NT / 64) * sub) {
                const unsigned m = m0 + ml;
                if (m < C.n_mles && q < left) {
                    const MleSlot sl = C.slots[m];
                    item(sl.in, sl.out, sl.in_ext != 0, (unsigned)C.unit[m] * tpp, q);
                }
            }
        } else {
        // MLE uniform per wave
        for (unsigned m = (unsigned)wave; m < C.n_mles; m += NT / 64) {
            // MleSlot is 24 bytes: three 8-byte scalar loads
            const GEN_CONST gen_u2* sp = reinterpret_cast<const GEN_CONST gen_u2*>(reinterpret_cast<uintptr_t>(C.slots + m));
            const gen_u2 sw0 = sp[0], sw1 = sp[1], sw2 = sp[2];
            const uint64_t* s_in = reinterpret_cast<const uint64_t*>(u64_of(sw0.x, sw0.y));
            uint64_t* s_out = reinterpret_cast<uint64_t*>(u64_of(sw1.x, sw1.y));
            const bool in_ext = sw2.x != 0;
            const unsigned row = ldc_u16(C.unit + m) * tpp;
            for (unsigned q = lane; q < tp; q += 64) {
                if (p0 + q >= C.pairs) break;
                item(s_in, s_out, in_ext, row, q);
            }
        }
        }
        if (!staged) continue;  // fold-only component (tables that no term reads in this class)
        __syncthreads();
        // ---- phase 2: the terms of every group are split over `wt` waves, one pair per lane.  The partial sums of the waves
        // that share a pair meet in LDS; then wave `ts` owns the evaluation points t = ts (mod wt): it adds the partials of
        // its points and multiplies the group's common factors (the selectors) in — once per pair and point ----
        const unsigned wt = 1u << C.wt_log;
        const unsigned ts = (unsigned)wave & (wt - 1), q = (((unsigned)wave >> C.wt_log) << 6) + lane;
        const bool valid = q < tp && p0 + q < C.pairs;
        for (unsigned g = 0; g < C.n_groups; g++) {
            const gen_u4 gw = ldc4(C.groups + g);
            const gen_u2 gw2 = *reinterpret_cast<const GEN_CONST gen_u2*>(reinterpret_cast<uintptr_t>(C.groups + g) + 16);
            const unsigned term_begin = gw.x, term_end = gw.y, n_common = gw.z, base_mask = gw.w;
            const uint64_t common8 = u64_of(gw2.x, gw2.y);
            E2 inner[D];
            if (BASE0) {
                Acc5 w0[D], w1[D];
#pragma unroll
                for (int t = 0; t < D; t++) w0[t] = w1[t] = Acc5{0, 0, 0, 0, 0};
                for (unsigned ti = term_begin + ts; ti < term_end; ti += wt) {
                    const gen_u4 t0 = ldc4(C.terms + ti), t1 = ldc4(reinterpret_cast<const char*>(C.terms + ti) + 16);
                    const E2 cf{u64_of(t0.x, t0.y), u64_of(t0.z, t0.w)};
                    const unsigned nf = t1.x;
                    uint64_t idx8 = u64_of(t1.z, t1.w);
                    uint64_t pb[D];
                    if (nf == 0) {  // coefficient x the group's common factors only
#pragma unroll
                        for (int t = 0; t < D; t++) pb[t] = 1;
                    } else {
                        const ulonglong2 v = reinterpret_cast<const ulonglong2*>(stage)[(unsigned)(idx8 & 0xff) * tpp + q];
                        uint64_t x = v.x;
#pragma unroll
                        for (int t = 0; t < D; t++) {
                            pb[t] = x;
                            if (t + 1 < D) x = sub(x, v.y);
                        }
                    }
                    for (unsigned k = 1; k < nf; k++) {
                        idx8 >>= 8;
                        const ulonglong2 v = reinterpret_cast<const ulonglong2*>(stage)[(unsigned)(idx8 & 0xff) * tpp + q];
                        uint64_t x = v.x;
#pragma unroll
                        for (int t = 0; t < D; t++) {
                            pb[t] = mul_nc(pb[t], x);  // only multiplied again: any 64-bit representative will do
                            if (t + 1 < D) x = sub(x, v.y);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < D; t++) {
                        acc5_add(w0[t], mul_wide(cf.c0, pb[t]));
                        acc5_add(w1[t], mul_wide(cf.c1, pb[t]));
                    }
                }
#pragma unroll
                for (int t = 0; t < D; t++) inner[t] = E2{acc5_reduce(w0[t]), acc5_reduce(w1[t])};
            } else {
                // the LAST factor of every term is multiplied in UNREDUCED (160-bit accumulators per point, one reduction per
                // group); products that are only multiplied again skip canonicalisation
                E2Acc wacc[D];
#pragma unroll
                for (int t = 0; t < D; t++) wacc[t] = e2acc_zero();
                for (unsigned ti = term_begin + ts; ti < term_end; ti += wt) {
                    const gen_u4 t0 = ldc4(C.terms + ti), t1 = ldc4(reinterpret_cast<const char*>(C.terms + ti) + 16);
                    const E2 cf{u64_of(t0.x, t0.y), u64_of(t0.z, t0.w)};
                    const unsigned nf = t1.x;
                    uint64_t idx8 = u64_of(t1.z, t1.w);
                    E2 pr[D];
                    if (nf == 0) {
#pragma unroll
                        for (int t = 0; t < D; t++) e2acc_mac(wacc[t], cf, e2_one());
                        continue;
                    }
                    unsigned row = (unsigned)(idx8 & 0xff) * tpp + q;
                    E2 x = stage[row], nd = stage[row + tpp];
                    if (nf == 1) {
#pragma unroll
                        for (int t = 0; t < D; t++) {
                            e2acc_mac(wacc[t], cf, x);
                            if (t + 1 < D) x = x - nd;
                        }
                        continue;
                    }
                    // the coefficient rides on the first factor: c f(X) = c f(1) - (X - 1) c (f(0) - f(1)), two products instead
                    // of one per evaluation point
                    x = cf * x;
                    nd = cf * nd;
#pragma unroll
                    for (int t = 0; t < D; t++) {
                        pr[t] = x;
                        if (t + 1 < D) x = x - nd;
                    }
                    for (unsigned k = 1; k + 1 < nf; k++) {
                        idx8 >>= 8;
                        row = (unsigned)(idx8 & 0xff) * tpp + q;
                        x = stage[row];
                        nd = stage[row + tpp];
#pragma unroll
                        for (int t = 0; t < D; t++) {
                            pr[t] = e2_mul_nc(pr[t], x);
                            if (t + 1 < D) x = x - nd;
                        }
                    }
                    idx8 >>= 8;
                    row = (unsigned)(idx8 & 0xff) * tpp + q;
                    x = stage[row];
                    nd = stage[row + tpp];
#pragma unroll
                    for (int t = 0; t < D; t++) {
                        e2acc_mac(wacc[t], pr[t], x);
                        if (t + 1 < D) x = x - nd;
                    }
                }
#pragma unroll
                for (int t = 0; t < D; t++) inner[t] = e2acc_reduce(wacc[t]);
            }
            if (wt > 1) {
                // partial sums of the waves that share this pair -> LDS; the previous group's reads must be over before
                __syncthreads();
#pragma unroll
                for (int t = 0; t < D; t++) xch[((unsigned)wave * D + t) * 64 + lane] = inner[t];
                __syncthreads();
            }
            // common factors at the points this wave owns
#pragma unroll
            for (int t = 0; t < D; t++) {
                if (wt > 1 && ((unsigned)t & (wt - 1)) != ts) continue;  // uniform per wave
                E2 v = inner[t];
                if (wt > 1) {
                    const unsigned w0_ = (unsigned)wave & ~(wt - 1);
                    v = xch[((w0_)*D + t) * 64 + lane];
                    for (unsigned s = 1; s < wt; s++) v = v + xch[((w0_ + s) * D + t) * 64 + lane];
                }
                uint64_t c8 = common8;
                for (unsigned k = 0; k < n_common; k++, c8 >>= 8) {
                    const unsigned row = (unsigned)(c8 & 0xff) * tpp + q;
                    if (BASE0 && ((base_mask >> k) & 1)) {
                        const ulonglong2 b = reinterpret_cast<const ulonglong2*>(stage)[row];
                        uint64_t xb = b.x;
                        for (int j = 0; j < t; j++) xb = sub(xb, b.y);
                        v = e2_mul_base(v, xb);
                    } else {
                        E2 x = stage[row];
                        const E2 nd = stage[row + tpp];
                        for (int j = 0; j < t; j++) x = x - nd;
                        v = v * x;
                    }
                }
                if (valid) acc[t] = acc[t] + v;
            }
        }
        __syncthreads();  // the stage (and the exchange block) are reused by the next tile
    }
    if (ep.d == 0) return;  // a launch that only folds (no live term): nothing to publish
    epilogue<D, NT>(acc, ep, smem, s_flag);
}


// ------------------------------------------------------------------------------------------------------------------------------------
// eq-factored form
// ------------------------------------------------------------------------------------------------------------------------------------
// sum over the terms [term_begin + ts, term_end) step wt of  c_t prod_j f_j(X)  at the NV points X = 1 + SKIP .. NV + SKIP (out[0 .. NV)) and,
// with LEAD, the product of the c_t (f_j(0) - f_j(1)) over the terms of DEG - 1 factors (out[NV]: the leading coefficient up to its sign).
// Extension-field tables; the last factor of a term goes in unreduced, as in the generic group.
template <int NV, bool LEAD, int SKIP, int DEG>
__device__ __forceinline__ void gen_eq_terms(const GenComp& C, unsigned term_begin, unsigned term_end, unsigned ts, unsigned wt, const E2* stage, unsigned tpp,
                                             unsigned q, E2 (&out)[NV + (LEAD ? 1 : 0)]) {
    constexpr int NS = NV + (LEAD ? 1 : 0);
    E2Acc wacc[NS];
#pragma unroll
    for (int t = 0; t < NS; t++) wacc[t] = e2acc_zero();
    auto first_point = [](E2 x, const E2& nd) {  // f at X = 1 + SKIP from the staged (f(1), f(0) - f(1))
#pragma unroll
        for (int j = 0; j < SKIP; j++) x = x - nd;
        return x;
    };
    for (unsigned ti = term_begin + ts; ti < term_end; ti += wt) {
        const gen_u4 t0 = ldc4(C.terms + ti), t1 = ldc4(reinterpret_cast<const char*>(C.terms + ti) + 16);
        const E2 cf{u64_of(t0.x, t0.y), u64_of(t0.z, t0.w)};
        const unsigned nf = t1.x;
        const bool lead = LEAD && nf == (unsigned)(DEG - 1);
        uint64_t idx8 = u64_of(t1.z, t1.w);
        if (nf == 0) {  // a constant: every value slot, no leading coefficient (that takes DEG - 1 >= 2 factors)
#pragma unroll
            for (int t = 0; t < NV; t++) e2acc_mac(wacc[t], cf, e2_one());
            continue;
        }
        unsigned row = (unsigned)(idx8 & 0xff) * tpp + q;
        E2 nd = stage[row + tpp], x = first_point(stage[row], nd);
        if (nf == 1) {  // (DEG >= 3: one factor never reaches the leading coefficient)
#pragma unroll
            for (int t = 0; t < NV; t++) {
                e2acc_mac(wacc[t], cf, x);
                if (t + 1 < NV) x = x - nd;
            }
            continue;
        }
        // the coefficient rides on the first factor: c f(X) = c f(1) - (X - 1) c (f(0) - f(1))
        x = cf * x;
        nd = cf * nd;
        E2 pr[NS];
#pragma unroll
        for (int t = 0; t < NV; t++) {
            pr[t] = x;
            if (t + 1 < NV) x = x - nd;
        }
        if (LEAD) pr[NS - 1] = nd;
        for (unsigned k = 1; k + 1 < nf; k++) {
            idx8 >>= 8;
            row = (unsigned)(idx8 & 0xff) * tpp + q;
            nd = stage[row + tpp];
            x = first_point(stage[row], nd);
#pragma unroll
            for (int t = 0; t < NV; t++) {
                pr[t] = e2_mul_nc(pr[t], x);
                if (t + 1 < NV) x = x - nd;
            }
            if (lead) pr[NS - 1] = e2_mul_nc(pr[NS - 1], nd);
        }
        idx8 >>= 8;
        row = (unsigned)(idx8 & 0xff) * tpp + q;
        nd = stage[row + tpp];
        x = first_point(stage[row], nd);
#pragma unroll
        for (int t = 0; t < NV; t++) {
            e2acc_mac(wacc[t], pr[t], x);
            if (t + 1 < NV) x = x - nd;
        }
        if (lead) e2acc_mac(wacc[NS - 1], pr[NS - 1], nd);
    }
#pragma unroll
    for (int t = 0; t < NS; t++) out[t] = e2acc_reduce(wacc[t]);
}

// (A/B, tools/dev/ab_eq5.sh: -DGEN_EQ5_SPLIT=1 -DGEN_EQ5_WAVES=3 = two passes at 168 registers, 7 spilled, three waves per SIMD: 31.7-32.6 ms for the
// wide batch against 32.2 in one pass at two waves — the second walk over the terms costs what the third wave returns; two passes at two waves: 33.8)
#ifndef GEN_EQ5_SPLIT
#define GEN_EQ5_SPLIT 0
#endif
// Evaluation SLOTS of an eq group (D = the message length, G has degree <= D - 1):
//   slot s < D - 1: X = s + 1  (slot D - 2, X = D - 1, only where it is wanted: the first round, and the waves that hold a boundary pair)
//   slot D - 1:     the coefficient of X^(D-1), up to the sign (-1)^(D-1) the host applies: only terms of D - 1 factors have one
template <int D, bool BASE0>
__device__ __forceinline__ void gen_group_eq(const GenComp& C, unsigned g, const E2* stage, E2* xch, unsigned tpp, unsigned q, unsigned ts, unsigned wt,
                                             bool valid, size_t pair, int wave, unsigned lane, E2 (&acc)[D], E2* b_out, unsigned bstride) {
    static_assert(D >= 3, "eq-factored groups need a message of at least three points");
    const gen_u4 gw = ldc4(C.groups + g);
    const gen_u4 gw3 = ldc4(reinterpret_cast<const char*>(C.groups + g) + 16);
    const gen_u4 gw4 = ldc4(reinterpret_cast<const char*>(C.groups + g) + 32);
    const unsigned term_begin = gw.x, term_end = gw.y;
    const unsigned sel_row = (unsigned)(gw3.x & 0xff) * tpp + q, brow = gw3.w;
    const uint64_t lo = u64_of(gw4.x, gw4.y), hi = u64_of(gw4.z, gw4.w);
    // is this lane's pair a boundary pair of the group's row range?  An entry of this round's tables stands for 2^shift rows.
    const unsigned sh = C.shift;
    const uint64_t s0 = (uint64_t)(2 * pair) << sh, s1 = (uint64_t)(2 * pair + 1) << sh, s2 = (uint64_t)(2 * pair + 2) << sh;
    const bool full0 = lo <= s0 && s1 <= hi, full1 = lo <= s1 && s2 <= hi;
    const bool empty0 = s1 <= lo || s0 >= hi, empty1 = s2 <= lo || s1 >= hi;
    const bool irr = valid && !((full0 && full1) || (empty0 && empty1));
    const bool extra = (C.eqf & 2) != 0 || __builtin_amdgcn_ballot_w64(irr) != 0;  // uniform over the waves that share these pairs
    E2 inner[D];
    if (BASE0) {
        Acc5 w0[D], w1[D];
#pragma unroll
        for (int t = 0; t < D; t++) w0[t] = w1[t] = Acc5{0, 0, 0, 0, 0};
        for (unsigned ti = term_begin + ts; ti < term_end; ti += wt) {
            const gen_u4 t0 = ldc4(C.terms + ti), t1 = ldc4(reinterpret_cast<const char*>(C.terms + ti) + 16);
            const E2 cf{u64_of(t0.x, t0.y), u64_of(t0.z, t0.w)};
            const unsigned nf = t1.x;
            const bool lead = nf == (unsigned)(D - 1);
            uint64_t idx8 = u64_of(t1.z, t1.w);
            uint64_t pb[D];  // pb[D - 1]: the product of the (f(0) - f(1))
            if (nf == 0) {  // a constant under the selector: every value slot, no leading coefficient (D - 1 >= 2 factors make one)
#pragma unroll
                for (int t = 0; t < D; t++) pb[t] = 1;
            } else {
                const ulonglong2 v = reinterpret_cast<const ulonglong2*>(stage)[(unsigned)(idx8 & 0xff) * tpp + q];
                uint64_t x = v.x;
#pragma unroll
                for (int t = 0; t < D - 1; t++) {
                    pb[t] = x;
                    if (t + 3 < D || (t + 2 < D && extra)) x = sub(x, v.y);
                }
                pb[D - 1] = v.y;
            }
            for (unsigned k = 1; k < nf; k++) {
                idx8 >>= 8;
                const ulonglong2 v = reinterpret_cast<const ulonglong2*>(stage)[(unsigned)(idx8 & 0xff) * tpp + q];
                uint64_t x = v.x;
#pragma unroll
                for (int t = 0; t < D - 1; t++) {
                    if (t < D - 2 || extra) pb[t] = mul_nc(pb[t], x);
                    if (t + 3 < D || (t + 2 < D && extra)) x = sub(x, v.y);
                }
                if (lead) pb[D - 1] = mul_nc(pb[D - 1], v.y);
            }
#pragma unroll
            for (int t = 0; t < D - 1; t++) {
                if (t < D - 2 || extra) {
                    acc5_add(w0[t], mul_wide(cf.c0, pb[t]));
                    acc5_add(w1[t], mul_wide(cf.c1, pb[t]));
                }
            }
            if (lead) {
                acc5_add(w0[D - 1], mul_wide(cf.c0, pb[D - 1]));
                acc5_add(w1[D - 1], mul_wide(cf.c1, pb[D - 1]));
            }
        }
#pragma unroll
        for (int t = 0; t < D; t++) inner[t] = E2{acc5_reduce(w0[t]), acc5_reduce(w1[t])};
    } else {
        // two passes keep the registers of the common case at those of a message one point shorter (three waves per SIMD at degree 4):
        // the slots every wave needs — X = 1 .. D - 2 and the leading coefficient — and, where it is wanted, X = D - 1 on its own
        if constexpr (D == 5 && GEN_EQ5_SPLIT) {
            // degree 5: the one-pass form holds four product chains and four wide accumulators — 205 registers, two waves per SIMD.  Walking the
            // terms GEN_EQ5_SPLIT + 1 times with fewer chains each (the staged values are re-read from LDS, the term words from the scalar cache)
            // keeps every pass at the registers of a shorter message
            if constexpr (GEN_EQ5_SPLIT == 1) {
                E2 in_a[3], in_b[1];
                gen_eq_terms<2, true, 0, D>(C, term_begin, term_end, ts, wt, stage, tpp, q, in_a);   // X = 1, 2 and the leading coefficient
                inner[0] = in_a[0];
                inner[1] = in_a[1];
                inner[4] = in_a[2];
                gen_eq_terms<1, false, 2, D>(C, term_begin, term_end, ts, wt, stage, tpp, q, in_b);  // X = 3
                inner[2] = in_b[0];
            } else {
                E2 in_a[2], in_b[1], in_c[1];
                gen_eq_terms<1, true, 0, D>(C, term_begin, term_end, ts, wt, stage, tpp, q, in_a);   // X = 1 and the leading coefficient
                inner[0] = in_a[0];
                inner[4] = in_a[1];
                gen_eq_terms<1, false, 1, D>(C, term_begin, term_end, ts, wt, stage, tpp, q, in_b);  // X = 2
                inner[1] = in_b[0];
                gen_eq_terms<1, false, 2, D>(C, term_begin, term_end, ts, wt, stage, tpp, q, in_c);  // X = 3
                inner[2] = in_c[0];
            }
            inner[D - 2] = e2_zero();
        } else {
        E2 in_a[D - 1];
        gen_eq_terms<D - 2, true, 0, D>(C, term_begin, term_end, ts, wt, stage, tpp, q, in_a);
#pragma unroll
        for (int t = 0; t < D - 2; t++) inner[t] = in_a[t];
        inner[D - 1] = in_a[D - 2];
        inner[D - 2] = e2_zero();
        }
        if (extra) {
            E2 in_b[1];
            gen_eq_terms<1, false, D - 2, D>(C, term_begin, term_end, ts, wt, stage, tpp, q, in_b);
            inner[D - 2] = in_b[0];
        }
    }
    if (wt > 1) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < D; t++) xch[((unsigned)wave * D + t) * 64 + lane] = inner[t];
        __syncthreads();
    }
    // the selector's weight of this pair: w = EQ[2y] + EQ[2y+1]; a boundary pair: w' = EQ[2y] / (1 - rt), remainder c = EQ[2y+1] - w' rt
    const E2 e1 = stage[sel_row], e0 = stage[sel_row + tpp] + e1;  // staged: (f(1), f(0) - f(1))
    E2 w = e0 + e1, cb = e2_zero();
    if (irr) {
        w = e0 * C.inv1m;
        cb = e1 - w * C.rt;
    }
    const unsigned side = pair == (size_t)((lo >> sh) >> 1) ? 0u : 1u;
#pragma unroll
    for (int t = 0; t < D; t++) {
        if (wt > 1 && ((unsigned)t & (wt - 1)) != ts) continue;  // uniform per wave
        if (t == D - 2 && !extra) continue;
        E2 v = inner[t];
        if (wt > 1) {
            const unsigned w0_ = (unsigned)wave & ~(wt - 1);
            v = xch[((w0_)*D + t) * 64 + lane];
            for (unsigned s = 1; s < wt; s++) v = v + xch[((w0_ + s) * D + t) * 64 + lane];
        }
        if (irr) {
            // one lane per boundary pair and slot, once per round: straight into the host's (armed) words
            const E2 bv = cb * v;
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            const u4 ww = {(unsigned)bv.c0, (unsigned)(bv.c0 >> 32), (unsigned)bv.c1, (unsigned)(bv.c1 >> 32)};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(b_out + (size_t)(brow + side) * bstride + t), "v"(ww) : "memory");
        }
        if (valid) acc[t] = acc[t] + w * v;
    }
}

// end of a component-aligned launch.  A component with ONE workgroup has its sums right there: they go straight into the host's armed
// words (the small rounds of a batch are all of this kind: no counter, no second trip through memory).  The workgroups of a larger
// component leave their rows in device memory and count themselves in at the COMPONENT's counter; the last of them adds the rows — all
// 256 lanes — and writes the sums to the host, while the other components are still running.  (One last workgroup for the whole launch
// cost ~3 us per component at the end of every round; handing all rows to the host ~0.2 us per row — each 64-byte row is a cache line
// the device's write has just taken away from the CPU.)
template <int D>
__device__ __forceinline__ void epilogue_eq(E2 (&acc)[D], const GenComp& C, const Epilogue& ep, const GenEqArgs& eqa, E2* smem, int* s_flag) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    auto to_host = [&](const E2 (&v)[D], unsigned slot) {
#pragma unroll
        for (int t = 0; t < D; t++) {
            const u4 ww = {(unsigned)v[t].c0, (unsigned)(v[t].c0 >> 32), (unsigned)v[t].c1, (unsigned)(v[t].c1 >> 32)};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(eqa.q_out + (size_t)slot * eqa.stride + t), "v"(ww) : "memory");
        }
    };
    if (C.n_groups == 0) return;  // folded only: nothing to report
    red::block_sum<D, NT>(acc, smem);
    if (C.wg_count == 1) {
        if (threadIdx.x == 0) to_host(acc, C.eq_slot);
        return;
    }
    int& s_is_last = *s_flag;
    if (threadIdx.x == 0) {
        uint64_t* row = ep.partials + (size_t)blockIdx.x * D * 2;
#pragma unroll
        for (int t = 0; t < D; t++) {
            st_agent(row + 2 * t, acc[t].c0);
            st_agent(row + 2 * t + 1, acc[t].c1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(eqa.counters + C.eq_slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_is_last = (prev == C.wg_count - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_is_last) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(eqa.counters + C.eq_slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    E2 tot[D];
#pragma unroll
    for (int t = 0; t < D; t++) tot[t] = e2_zero();
    for (unsigned b = threadIdx.x; b < C.wg_count; b += NT) {
        const uint64_t* row = ep.partials + (size_t)(C.wg_begin + b) * D * 2;
#pragma unroll
        for (int t = 0; t < D; t++) tot[t] = tot[t] + E2{ld_agent(row + 2 * t), ld_agent(row + 2 * t + 1)};
    }
    __syncthreads();  // smem is reused
    red::block_sum<D, NT>(tot, smem);
    if (threadIdx.x == 0) to_host(tot, C.eq_slot);
}

// (the first-round form at degree 4 sits four registers above the four-waves-per-SIMD line: the allocator is asked for it)
#ifndef GEN_EQ5_WAVES
#define GEN_EQ5_WAVES 1
#endif
constexpr int gen_eq_min_waves(int d, bool base0) { return base0 && d == 4 ? 4 : (!base0 && d == 5 ? GEN_EQ5_WAVES : 1); }
template <int D, bool BASE0>
__global__ void __launch_bounds__(NT, gen_eq_min_waves(D, BASE0)) k_gen_eq(const GenComp* __restrict__ comps, int n_comps, E2 r, Epilogue ep, unsigned xch_off, GenEqArgs eqa) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    E2* smem = reinterpret_cast<E2*>(dyn);
    int* s_flag = reinterpret_cast<int*>(dyn + (NT / 64) * MAXD * sizeof(E2) + 32);
    E2* stage = reinterpret_cast<E2*>(dyn + GEN_FIXED);
    E2* xch = reinterpret_cast<E2*>(dyn + GEN_FIXED + xch_off);  // [wave][D][64]
    const E2Pre rp = e2_pre(r);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63;
    E2 acc[D];
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    // CENO_HIP_GEN_PHASE_DBG=1: the last workgroup of the launch stamps its phases (100 MHz wall clock) into bcast->dbg rows seq and seq + 32
    const bool stamp = ep.dbg && ep.bcast && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0;
    if (stamp) ep.bcast->dbg[ep.seq & 31][0] = wall_clock64();
    // the component this workgroup belongs to (component-aligned launch: comps are sorted by wg_begin, every one owns >= 1 workgroup)
    int c = 0;
    if (eqa.wg_comp) c = (int)ldc_u16(eqa.wg_comp + blockIdx.x);
    else
        while (c + 1 < n_comps && comps[c + 1].wg_begin <= blockIdx.x) c++;
    const GenComp& C = comps[c];
    const unsigned tp = 1u << C.tp_log, tpp = tp + GEN_PAD;
    const unsigned wt = 1u << C.wt_log;
    const unsigned ts = (unsigned)wave & (wt - 1), q = (((unsigned)wave >> C.wt_log) << 6) + lane;
    if (stamp) ep.bcast->dbg[ep.seq & 31][1] = wall_clock64();
    for (unsigned tile = blockIdx.x - C.wg_begin; tile < C.n_tiles; tile += C.wg_count) {
        const size_t p0 = (size_t)tile << C.tp_log;
        gen_phase1<BASE0>(C, p0, stage, rp, r, wave, lane);
        if (C.n_groups == 0) continue;
        if (tile < C.p2_tile_begin || tile >= C.p2_tile_end) {  // no pair of this tile lies inside a group's row range: folded only
            __syncthreads();  // (the stage is rewritten by the next tile's phase 1)
            continue;
        }
        __syncthreads();
        if (stamp) ep.bcast->dbg[ep.seq & 31][2] = wall_clock64();
        const bool valid = q < tp && p0 + q < C.pairs;
        for (unsigned g = 0; g < C.n_groups; g++) gen_group_eq<D, BASE0>(C, g, stage, xch, tpp, q, ts, wt, valid, p0 + q, wave, lane, acc, eqa.b_out, eqa.stride);
        __syncthreads();  // the stage (and the exchange block) are reused by the next tile
    }
    if (stamp) ep.bcast->dbg[ep.seq & 31][3] = wall_clock64();
    epilogue_eq<D>(acc, C, ep, eqa, smem, s_flag);
    if (stamp) ep.bcast->dbg[32 + (ep.seq & 31)][0] = wall_clock64();
}

// ONE slot of an eq group (small rounds: a component of a single tile is given D workgroups, one per slot — its four waves walking 4-8 terms
// each, one pair per lane, were the longest phase of a small round, and the slots are independent chains of the same length).  Slot numbering
// as in gen_group_eq; extension-field tables only.
template <int D, int S>
__device__ __forceinline__ E2 gen_eq_slot_terms(const GenComp& C, unsigned term_begin, unsigned term_end, unsigned ts, unsigned wt, const E2* stage, unsigned tpp,
                                                unsigned q, unsigned slot, bool extra) {
    if constexpr (S < D - 1) {
        if (slot == (unsigned)S) {
            if (S == D - 2 && !extra) return e2_zero();
            E2 o[1];
            gen_eq_terms<1, false, S, D>(C, term_begin, term_end, ts, wt, stage, tpp, q, o);  // X = S + 1
            return o[0];
        }
        return gen_eq_slot_terms<D, S + 1>(C, term_begin, term_end, ts, wt, stage, tpp, q, slot, extra);
    } else {
        E2 o[1];
        gen_eq_terms<0, true, 0, D>(C, term_begin, term_end, ts, wt, stage, tpp, q, o);  // the leading coefficient
        return o[0];
    }
}
template <int D>
__device__ __forceinline__ void gen_group_eq_slot(const GenComp& C, unsigned g, const E2* stage, E2* xch, unsigned tpp, unsigned q, unsigned ts, unsigned wt,
                                                  bool valid, size_t pair, int wave, unsigned lane, E2 (&acc)[D], E2* b_out, unsigned bstride, unsigned slot) {
    const gen_u4 gw = ldc4(C.groups + g);
    const gen_u4 gw3 = ldc4(reinterpret_cast<const char*>(C.groups + g) + 16);
    const gen_u4 gw4 = ldc4(reinterpret_cast<const char*>(C.groups + g) + 32);
    const unsigned term_begin = gw.x, term_end = gw.y;
    const unsigned sel_row = (unsigned)(gw3.x & 0xff) * tpp + q, brow = gw3.w;
    const uint64_t lo = u64_of(gw4.x, gw4.y), hi = u64_of(gw4.z, gw4.w);
    const unsigned sh = C.shift;
    const uint64_t s0 = (uint64_t)(2 * pair) << sh, s1 = (uint64_t)(2 * pair + 1) << sh, s2 = (uint64_t)(2 * pair + 2) << sh;
    const bool full0 = lo <= s0 && s1 <= hi, full1 = lo <= s1 && s2 <= hi;
    const bool empty0 = s1 <= lo || s0 >= hi, empty1 = s2 <= lo || s1 >= hi;
    const bool irr = valid && !((full0 && full1) || (empty0 && empty1));
    const bool extra = (C.eqf & 2) != 0 || __builtin_amdgcn_ballot_w64(irr) != 0;
    const E2 inner = gen_eq_slot_terms<D, 0>(C, term_begin, term_end, ts, wt, stage, tpp, q, slot, extra);
    if (wt > 1) {
        __syncthreads();
        xch[(unsigned)wave * 64 + lane] = inner;
        __syncthreads();
    }
    if (wt > 1 && ts != 0) return;  // the first wave of every group of wt finishes the slot
    if (slot == (unsigned)(D - 2) && !extra) return;
    const E2 e1 = stage[sel_row], e0 = stage[sel_row + tpp] + e1;  // staged: (f(1), f(0) - f(1))
    E2 w = e0 + e1, cb = e2_zero();
    if (irr) {
        w = e0 * C.inv1m;
        cb = e1 - w * C.rt;
    }
    const unsigned side = pair == (size_t)((lo >> sh) >> 1) ? 0u : 1u;
    E2 v = inner;
    if (wt > 1)
        for (unsigned s = 1; s < wt; s++) v = v + xch[((unsigned)wave + s) * 64 + lane];
    if (irr) {
        const E2 bv = cb * v;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 ww = {(unsigned)bv.c0, (unsigned)(bv.c0 >> 32), (unsigned)bv.c1, (unsigned)(bv.c1 >> 32)};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(b_out + (size_t)(brow + side) * bstride + slot), "v"(ww) : "memory");
    }
    if (valid) {
        const E2 add = w * v;
#pragma unroll
        for (int t = 0; t < D; t++)
            if ((unsigned)t == slot) acc[t] = acc[t] + add;
    }
}

// The small rounds of an eq-factored batch: every tile of every component gets D workgroups — one per slot (gen_group_eq_slot) — where the whole
// launch still fits the chip at once.  A kernel of its own: the slot forms inlined into k_gen_eq cost it its third wave per SIMD at degree 3.
// The D workgroups of a tile all stage it and write the same folded tables (identical words).  A component of ONE tile: each workgroup writes
// its own word of the component's row straight into the host's armed words; more tiles: the rows-and-counter epilogue of k_gen_eq (a row
// holds one slot, the others are zero).
template <int D>
__global__ void __launch_bounds__(NT) k_gen_eq_slots(const GenComp* __restrict__ comps, int n_comps, E2 r, Epilogue ep, unsigned xch_off, GenEqArgs eqa) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    E2* smem = reinterpret_cast<E2*>(dyn);
    int* s_flag = reinterpret_cast<int*>(dyn + (NT / 64) * MAXD * sizeof(E2) + 32);
    E2* stage = reinterpret_cast<E2*>(dyn + GEN_FIXED);
    E2* xch = reinterpret_cast<E2*>(dyn + GEN_FIXED + xch_off);  // [wave][64]
    const E2Pre rp = e2_pre(r);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lane = threadIdx.x & 63;
    E2 acc[D];
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    const int c = (int)ldc_u16(eqa.wg_comp + blockIdx.x);
    const GenComp& C = comps[c];
    const unsigned k = blockIdx.x - C.wg_begin;
    const bool split = (C.eqf & 4) != 0;            // (a fold-only component rides along with one workgroup per tile)
    const unsigned tile = split ? k / D : k, slot = split ? k % D : 0;
    const unsigned tp = 1u << C.tp_log, tpp = tp + GEN_PAD;
    const unsigned wt = 1u << C.wt_log;
    const unsigned ts = (unsigned)wave & (wt - 1), q = (((unsigned)wave >> C.wt_log) << 6) + lane;
    const size_t p0 = (size_t)tile << C.tp_log;
    gen_phase1<false>(C, p0, stage, rp, r, wave, lane);
    if (C.n_groups == 0) return;  // folded only
    __syncthreads();
    if (tile >= C.p2_tile_begin && tile < C.p2_tile_end) {
        const bool valid = q < tp && p0 + q < C.pairs;
        for (unsigned g = 0; g < C.n_groups; g++) gen_group_eq_slot<D>(C, g, stage, xch, tpp, q, ts, wt, valid, p0 + q, wave, lane, acc, eqa.b_out, eqa.stride, slot);
        __syncthreads();
    }
    if (C.n_tiles > 1) {
        epilogue_eq<D>(acc, C, ep, eqa, smem, s_flag);  // (wg_count = n_tiles x D > 1: rows and the component's counter)
        return;
    }
    red::block_sum<D, NT>(acc, smem);
    if (threadIdx.x == 0) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int t = 0; t < D; t++) {
            if ((unsigned)t != slot) continue;
            const u4 ww = {(unsigned)acc[t].c0, (unsigned)(acc[t].c0 >> 32), (unsigned)acc[t].c1, (unsigned)(acc[t].c1 >> 32)};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(eqa.q_out + (size_t)C.eq_slot * eqa.stride + t), "v"(ww) : "memory");
        }
    }
}

// First round of an eq-factored batch whose terms are products of BASE-field columns (the main-constraint sumcheck before any fold):
// nothing is folded and nothing is shared through LDS — every lane walks the terms of its pair with the column pairs straight from
// L1 / L2 (k_accum_base0's form: the staged kernel pays two barriers and an LDS round trip per tile for columns that are read once or
// twice), products in the base field, c_t P_t unreduced.  Plan records of the third layout: a factor byte = the MLE's index in the
// component's slot row.  Slots as in gen_group_eq with all D wanted.
template <int D>
__global__ void __launch_bounds__(NT) k_eq_base0(const GenComp* __restrict__ comps, int n_comps, Epilogue ep, GenEqArgs eqa) {
    static_assert(D >= 3, "eq-factored groups need a message of at least three points");
    __shared__ E2 smem[(NT / 64) * D];
    __shared__ int s_flag;
    E2 acc[D];
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    int c = 0;
    if (eqa.wg_comp) c = (int)ldc_u16(eqa.wg_comp + blockIdx.x);
    else
        while (c + 1 < n_comps && comps[c + 1].wg_begin <= blockIdx.x) c++;
    const GenComp& C = comps[c];
    const size_t p_begin = (size_t)C.p2_tile_begin << C.tp_log;
    const size_t p_end = std::min<size_t>((size_t)C.p2_tile_end << C.tp_log, (size_t)C.pairs);
    auto table = [&](unsigned m) {  // the input table of MLE m of the component: a scalar load (m is wave-uniform)
        const GEN_CONST gen_u2* sp = reinterpret_cast<const GEN_CONST gen_u2*>(reinterpret_cast<uintptr_t>(C.slots + m));
        const gen_u2 w = sp[0];
        return reinterpret_cast<const uint64_t*>(u64_of(w.x, w.y));
    };
    for (size_t p = p_begin + (size_t)(blockIdx.x - C.wg_begin) * NT + threadIdx.x; p < p_end; p += (size_t)C.wg_count * NT) {
        for (unsigned g = 0; g < C.n_groups; g++) {
            const gen_u4 gw = ldc4(C.groups + g);
            const gen_u4 gw3 = ldc4(reinterpret_cast<const char*>(C.groups + g) + 16);
            const gen_u4 gw4 = ldc4(reinterpret_cast<const char*>(C.groups + g) + 32);
            const unsigned term_begin = gw.x, term_end = gw.y, brow = gw3.w;
            const uint64_t lo = u64_of(gw4.x, gw4.y), hi = u64_of(gw4.z, gw4.w);
            const uint64_t s0 = 2 * p, s1 = 2 * p + 1, s2 = 2 * p + 2;  // round 0: an entry is a row
            const bool full0 = lo <= s0 && s1 <= hi, full1 = lo <= s1 && s2 <= hi;
            const bool empty0 = s1 <= lo || s0 >= hi, empty1 = s2 <= lo || s1 >= hi;
            if (empty0 && empty1) continue;  // the selector is zero on this pair
            const bool irr = !(full0 && full1);
            Acc5 w0[D], w1[D];
#pragma unroll
            for (int t = 0; t < D; t++) w0[t] = w1[t] = Acc5{0, 0, 0, 0, 0};
            for (unsigned ti = term_begin; ti < term_end; ti++) {
                const gen_u4 t0 = ldc4(C.terms + ti), t1 = ldc4(reinterpret_cast<const char*>(C.terms + ti) + 16);
                const E2 cf{u64_of(t0.x, t0.y), u64_of(t0.z, t0.w)};
                const unsigned nf = t1.x;
                const bool lead = nf == (unsigned)(D - 1);
                uint64_t idx8 = u64_of(t1.z, t1.w);
                uint64_t pb[D];  // pb[D - 1]: the product of the (f(0) - f(1))
                if (nf == 0) {  // a constant under the selector
#pragma unroll
                    for (int t = 0; t < D; t++) pb[t] = 1;
                } else {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(table((unsigned)(idx8 & 0xff)) + 2 * p);
                    const uint64_t nd = sub(v.x, v.y);
                    uint64_t x = v.y;
#pragma unroll
                    for (int t = 0; t < D - 1; t++) {
                        pb[t] = x;
                        if (t + 2 < D) x = sub(x, nd);
                    }
                    pb[D - 1] = nd;
                }
                for (unsigned k = 1; k < nf; k++) {
                    idx8 >>= 8;
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(table((unsigned)(idx8 & 0xff)) + 2 * p);
                    const uint64_t nd = sub(v.x, v.y);
                    uint64_t x = v.y;
#pragma unroll
                    for (int t = 0; t < D - 1; t++) {
                        pb[t] = mul_nc(pb[t], x);  // only multiplied again: any 64-bit value will do
                        if (t + 2 < D) x = sub(x, nd);
                    }
                    if (lead) pb[D - 1] = mul_nc(pb[D - 1], nd);
                }
#pragma unroll
                for (int t = 0; t < D - 1; t++) {
                    acc5_add(w0[t], mul_wide(cf.c0, pb[t]));
                    acc5_add(w1[t], mul_wide(cf.c1, pb[t]));
                }
                if (lead) {
                    acc5_add(w0[D - 1], mul_wide(cf.c0, pb[D - 1]));
                    acc5_add(w1[D - 1], mul_wide(cf.c1, pb[D - 1]));
                }
            }
            // the selector's weight of this pair (gen_group_eq)
            const uint64_t* sel = table((unsigned)(gw3.x & 0xff)) + 4 * p;
            const E2 e0 = ld_e2(sel), e1 = ld_e2(sel + 2);
            E2 w = e0 + e1, cb = e2_zero();
            if (irr) {
                w = e0 * C.inv1m;
                cb = e1 - w * C.rt;
            }
            const unsigned side = p == (size_t)(lo >> 1) ? 0u : 1u;
#pragma unroll
            for (int t = 0; t < D; t++) {
                const E2 v{acc5_reduce(w0[t]), acc5_reduce(w1[t])};
                if (irr) {
                    const E2 bv = cb * v;
                    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                    const u4 ww = {(unsigned)bv.c0, (unsigned)(bv.c0 >> 32), (unsigned)bv.c1, (unsigned)(bv.c1 >> 32)};
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(eqa.b_out + (size_t)(brow + side) * eqa.stride + t), "v"(ww) : "memory");
                }
                acc[t] = acc[t] + w * v;
            }
        }
    }
    epilogue_eq<D>(acc, C, ep, eqa, smem, &s_flag);
}

template <int D>
static void launch_eq_base0_d(ceno_hip_ctx* ctx, const GenComp* comps, int n_comps, const Epilogue& ep, const GenEqArgs& eq, unsigned grid, hipStream_t st) {
    (void)ctx;
    hipLaunchKernelGGL((k_eq_base0<D>), dim3(grid), dim3(NT), 0, st, comps, n_comps, ep, eq);
}
void launch_eq_base0(ceno_hip_ctx* ctx, int d, const GenComp* comps, int n_comps, const Epilogue& ep, const GenEqArgs& eq, unsigned grid, hipStream_t st) {
    switch (d) {
    case 3: launch_eq_base0_d<3>(ctx, comps, n_comps, ep, eq, grid, st); break;
    case 4: launch_eq_base0_d<4>(ctx, comps, n_comps, ep, eq, grid, st); break;
    case 5: launch_eq_base0_d<5>(ctx, comps, n_comps, ep, eq, grid, st); break;
    case 6: launch_eq_base0_d<6>(ctx, comps, n_comps, ep, eq, grid, st); break;
    case 7: launch_eq_base0_d<7>(ctx, comps, n_comps, ep, eq, grid, st); break;
    default: launch_eq_base0_d<8>(ctx, comps, n_comps, ep, eq, grid, st); break;
    }
}
unsigned eq_base0_resident_cap(ceno_hip_ctx* ctx, int d) {
    switch (d) {
    case 3: return resident_grid(ctx, k_eq_base0<3>, NT, 0, MAXB);
    case 4: return resident_grid(ctx, k_eq_base0<4>, NT, 0, MAXB);
    case 5: return resident_grid(ctx, k_eq_base0<5>, NT, 0, MAXB);
    case 6: return resident_grid(ctx, k_eq_base0<6>, NT, 0, MAXB);
    case 7: return resident_grid(ctx, k_eq_base0<7>, NT, 0, MAXB);
    default: return resident_grid(ctx, k_eq_base0<8>, NT, 0, MAXB);
    }
}

size_t gen_lds_bytes(int d, size_t stage_bytes) { return GEN_FIXED + ((stage_bytes + 15) & ~(size_t)15) + (size_t)(NT / 64) * d * 64 * sizeof(E2); }

// grid: one workgroup per tile up to the number of workgroups that are resident at once (the kernel is bound by VALU issue: a
// launch beyond that runs a second, partly filled dispatch wave — 1024 instead of 768 workgroups at degree 4 cost 9 % of the batched
// main sumcheck, tools/dev/ab_gen_maxb.sh); CENO_HIP_GEN_MAXB overrides the cap
template <int D>
static unsigned gen_cap_d(ceno_hip_ctx* ctx, bool base0, bool eq, size_t lds) {
    static const unsigned forced = getenv("CENO_HIP_GEN_MAXB") ? (unsigned)std::max(atoi(getenv("CENO_HIP_GEN_MAXB")), 0) : 0u;
    if (forced) return std::min(forced, MAXB);
    if constexpr (D >= 3) {
        if (eq) return base0 ? resident_grid(ctx, k_gen_eq<D, true>, NT, lds, MAXB) : resident_grid(ctx, k_gen_eq<D, false>, NT, lds, MAXB);
    }
    return base0 ? resident_grid(ctx, k_gen<D, true>, NT, lds, MAXB) : resident_grid(ctx, k_gen<D, false>, NT, lds, MAXB);
}
template <int D>
static void launch_gen_d(ceno_hip_ctx* ctx, bool base0, const GenComp* comps, int n_comps, unsigned total_tiles, E2 r, const Epilogue& ep, size_t stage_bytes,
                         hipStream_t st, const GenEqArgs* eq, unsigned aligned_grid) {
    const size_t lds = gen_lds_bytes(D, stage_bytes);
    const unsigned xch_off = (unsigned)((stage_bytes + 15) & ~(size_t)15);
    if constexpr (D >= 3) {
        if (eq && eq->aligned) {
            if (base0) hipLaunchKernelGGL((k_gen_eq<D, true>), dim3(aligned_grid), dim3(NT), lds, st, comps, n_comps, r, ep, xch_off, *eq);
            else hipLaunchKernelGGL((k_gen_eq<D, false>), dim3(aligned_grid), dim3(NT), lds, st, comps, n_comps, r, ep, xch_off, *eq);
            return;
        }
    }
    const unsigned grid = std::max(1u, std::min(total_tiles, gen_cap_d<D>(ctx, base0, false, lds)));
    if (base0) hipLaunchKernelGGL((k_gen<D, true>), dim3(grid), dim3(NT), lds, st, comps, n_comps, total_tiles, r, ep, xch_off);
    else hipLaunchKernelGGL((k_gen<D, false>), dim3(grid), dim3(NT), lds, st, comps, n_comps, total_tiles, r, ep, xch_off);
}

unsigned gen_resident_cap(ceno_hip_ctx* ctx, int d, bool base0, size_t stage_bytes) {
    switch (d) {
    case 3: return gen_cap_d<3>(ctx, base0, true, gen_lds_bytes(3, stage_bytes));
    case 4: return gen_cap_d<4>(ctx, base0, true, gen_lds_bytes(4, stage_bytes));
    case 5: return gen_cap_d<5>(ctx, base0, true, gen_lds_bytes(5, stage_bytes));
    case 6: return gen_cap_d<6>(ctx, base0, true, gen_lds_bytes(6, stage_bytes));
    case 7: return gen_cap_d<7>(ctx, base0, true, gen_lds_bytes(7, stage_bytes));
    case 8: return gen_cap_d<8>(ctx, base0, true, gen_lds_bytes(8, stage_bytes));
    default: return MAXB;
    }
}

void launch_gen_eq_slots(int d, const GenComp* comps, int n_comps, E2 r, const Epilogue& ep, size_t stage_bytes, hipStream_t st, const GenEqArgs& eq,
                          unsigned grid) {
    const size_t lds = gen_lds_bytes(d, stage_bytes);
    const unsigned xch_off = (unsigned)((stage_bytes + 15) & ~(size_t)15);
    switch (d) {
    case 3: hipLaunchKernelGGL((k_gen_eq_slots<3>), dim3(grid), dim3(NT), lds, st, comps, n_comps, r, ep, xch_off, eq); break;
    case 4: hipLaunchKernelGGL((k_gen_eq_slots<4>), dim3(grid), dim3(NT), lds, st, comps, n_comps, r, ep, xch_off, eq); break;
    case 5: hipLaunchKernelGGL((k_gen_eq_slots<5>), dim3(grid), dim3(NT), lds, st, comps, n_comps, r, ep, xch_off, eq); break;
    case 6: hipLaunchKernelGGL((k_gen_eq_slots<6>), dim3(grid), dim3(NT), lds, st, comps, n_comps, r, ep, xch_off, eq); break;
    case 7: hipLaunchKernelGGL((k_gen_eq_slots<7>), dim3(grid), dim3(NT), lds, st, comps, n_comps, r, ep, xch_off, eq); break;
    default: hipLaunchKernelGGL((k_gen_eq_slots<8>), dim3(grid), dim3(NT), lds, st, comps, n_comps, r, ep, xch_off, eq); break;
    }
}

void launch_gen(ceno_hip_ctx* ctx, int d, bool base0, const GenComp* comps, int n_comps, unsigned total_tiles, E2 r, const Epilogue& ep, size_t stage_bytes,
                hipStream_t st, const GenEqArgs* eq, unsigned aligned_grid) {
    switch (d) {
    case 1: launch_gen_d<1>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st, eq, aligned_grid); break;
    case 2: launch_gen_d<2>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st, eq, aligned_grid); break;
    case 3: launch_gen_d<3>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st, eq, aligned_grid); break;
    case 4: launch_gen_d<4>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st, eq, aligned_grid); break;
    case 5: launch_gen_d<5>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st, eq, aligned_grid); break;
    case 6: launch_gen_d<6>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st, eq, aligned_grid); break;
    case 7: launch_gen_d<7>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st, eq, aligned_grid); break;
    default: launch_gen_d<8>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st, eq, aligned_grid); break;
    }
}
