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
                st_e2(s_out + 4 * p, lo);
                st_e2(s_out + 4 * p + 2, hi);
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
                    {
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

size_t gen_lds_bytes(int d, size_t stage_bytes) { return GEN_FIXED + ((stage_bytes + 15) & ~(size_t)15) + (size_t)(NT / 64) * d * 64 * sizeof(E2); }

// grid: one workgroup per tile up to the number of workgroups that are resident at once (the kernel is bound by VALU issue: a
// launch beyond that runs a second, partly filled dispatch wave — 1024 instead of 768 workgroups at degree 4 cost 9 % of the batched
// main sumcheck, tools/dev/ab_gen_maxb.sh); CENO_HIP_GEN_MAXB overrides the cap
template <int D>
static void launch_gen_d(ceno_hip_ctx* ctx, bool base0, const GenComp* comps, int n_comps, unsigned total_tiles, E2 r, const Epilogue& ep, size_t stage_bytes,
                         hipStream_t st) {
    const size_t lds = gen_lds_bytes(D, stage_bytes);
    const unsigned xch_off = (unsigned)((stage_bytes + 15) & ~(size_t)15);
    static const unsigned forced = getenv("CENO_HIP_GEN_MAXB") ? (unsigned)std::max(atoi(getenv("CENO_HIP_GEN_MAXB")), 0) : 0u;
    unsigned cap = base0 ? resident_grid(ctx, k_gen<D, true>, NT, lds, MAXB) : resident_grid(ctx, k_gen<D, false>, NT, lds, MAXB);
    if (forced) cap = std::min(forced, MAXB);
    const unsigned grid = std::max(1u, std::min(total_tiles, cap));
    if (base0) hipLaunchKernelGGL((k_gen<D, true>), dim3(grid), dim3(NT), lds, st, comps, n_comps, total_tiles, r, ep, xch_off);
    else hipLaunchKernelGGL((k_gen<D, false>), dim3(grid), dim3(NT), lds, st, comps, n_comps, total_tiles, r, ep, xch_off);
}

void launch_gen(ceno_hip_ctx* ctx, int d, bool base0, const GenComp* comps, int n_comps, unsigned total_tiles, E2 r, const Epilogue& ep, size_t stage_bytes,
                hipStream_t st) {
    switch (d) {
    case 1: launch_gen_d<1>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st); break;
    case 2: launch_gen_d<2>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st); break;
    case 3: launch_gen_d<3>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st); break;
    case 4: launch_gen_d<4>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st); break;
    case 5: launch_gen_d<5>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st); break;
    case 6: launch_gen_d<6>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st); break;
    case 7: launch_gen_d<7>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st); break;
    default: launch_gen_d<8>(ctx, base0, comps, n_comps, total_tiles, r, ep, stage_bytes, st); break;
    }
}
