// Dense sumcheck rounds on gfx950: the register-resident fused kernels of the headline (`k_dense`: fold with r_{i-1}, write the half-size table,
// accumulate the message of round i on the folded values — every table read once and written once per round, 16 B per lane, HBM-streaming),
// their software-pipelined and LDS-prefetched variants (A/B: profiles/r03_dense_kernel_ab.json, r05_dense_overlap.json) and the launcher.
// Split out of sumcheck.hip in round 6 (the host driver stays there).  Reference operator: see sumcheck.hip.
#include "sumcheck_dense.hpp"

#include <algorithm>

#include <type_traits>

// ------------------------------------------------------------------------------------------------
// dense fused kernel
// ------------------------------------------------------------------------------------------------
#ifndef CENO_DENSE_FMA
#define CENO_DENSE_FMA 0
#endif
template <int K>
struct TabPtrs {
    const uint64_t* in[K];
    uint64_t* out[K];
};

// MODE 0: accumulate only, ext input      MODE 1: accumulate only, base input
// MODE 2: fold + accumulate, ext input    MODE 3: fold + accumulate, base input (output ext)
template <int K, int MODE, bool WIDE_>
__global__ void __launch_bounds__(NT) k_dense(TabPtrs<K> tp, size_t pairs, E2 r, Epilogue ep) {
    __shared__ E2 smem[(NT / 64) * K];
    // Unreduced accumulation costs 9 more registers per evaluation point but ~15% fewer VALU instructions.
    // Measured on MI355X: a clear win in the read-only round (ALU-bound, 750 -> 600 us at nv=26); in the
    // folding rounds (HBM-bound) it is within run-to-run noise of the reduced form, which keeps 81 VGPRs and
    // two more resident waves, so the default (CENO_HIP_DENSE_WIDE=1) uses it in the read-only round only.
    constexpr bool WIDE = WIDE_ && (MODE == 0 || MODE == 2) && K > 1;
    E2 acc[K];
    E2Acc wacc[K];
#pragma unroll
    for (int t = 0; t < K; t++) {
        acc[t] = e2_zero();
        wacc[t] = e2acc_zero();
    }
    const size_t stride = (size_t)gridDim.x * NT;
    if (ep.dbg && ep.bcast && blockIdx.x == 0 && threadIdx.x == 0) ep.bcast->dbg[ep.seq & 63][0] = wall_clock64();
    __shared__ unsigned long long s_chal[3];
    __shared__ int s_flag;
    if (ep.wait_seq != 0) {
        if (!read_challenge(ep, r, s_chal)) return;  // pipeline aborted / timed out: leave everything untouched
    }
    const E2Pre rp = e2_pre(r);
    for (size_t p = (size_t)blockIdx.x * NT + threadIdx.x; p < pairs; p += stride) {
        if (MODE == 1) {
            // all-base first round: the product of base values stays in the base field
            uint64_t pr[K];
#pragma unroll
            for (int m = 0; m < K; m++) {
                ulonglong2 v = *reinterpret_cast<const ulonglong2*>(tp.in[m] + 2 * p);
                uint64_t delta = sub(v.y, v.x), x = v.y;
#pragma unroll
                for (int t = 0; t < K; t++) {
                    pr[t] = (m == 0) ? x : mul(pr[t], x);
                    x = add(x, delta);
                }
            }
#pragma unroll
            for (int t = 0; t < K; t++) acc[t].c0 = add(acc[t].c0, pr[t]);
        } else {
            E2 pr[K];
#pragma unroll
            for (int m = 0; m < K; m++) {
                E2 lo, hi;
                if (MODE == 0) {
                    lo = ld_e2(tp.in[m] + 4 * p);
                    hi = ld_e2(tp.in[m] + 4 * p + 2);
                } else if (MODE == 2) {
                    const uint64_t* q = tp.in[m] + 8 * p;
                    E2 a0 = ld_e2(q), a1 = ld_e2(q + 2), a2 = ld_e2(q + 4), a3 = ld_e2(q + 6);
#if CENO_DENSE_FMA  // the addend inside the 129-bit sum of the product (one reduction, no separate modular add): A/B tools/dev/ab_dense_fma.sh
                    lo = e2_fma_pre(rp, a1 - a0, a0);
                    hi = e2_fma_pre(rp, a3 - a2, a2);
#else
                    lo = a0 + e2_mul_pre(rp, a1 - a0);
                    hi = a2 + e2_mul_pre(rp, a3 - a2);
#endif
                    st_e2(tp.out[m] + 4 * p, lo);
                    st_e2(tp.out[m] + 4 * p + 2, hi);
                } else {
                    const uint64_t* q = tp.in[m] + 4 * p;
                    ulonglong2 v0 = *reinterpret_cast<const ulonglong2*>(q);
                    ulonglong2 v1 = *reinterpret_cast<const ulonglong2*>(q + 2);
                    E2 t0 = e2_mul_base(r, sub(v0.y, v0.x));
                    E2 t1 = e2_mul_base(r, sub(v1.y, v1.x));
                    lo = E2{add(t0.c0, v0.x), t0.c1};
                    hi = E2{add(t1.c0, v1.x), t1.c1};
                    st_e2(tp.out[m] + 4 * p, lo);
                    st_e2(tp.out[m] + 4 * p + 2, hi);
                }
                // evaluation points 1..K: x_t = hi + (t-1)(hi - lo), stepped by subtracting (lo - hi)
                // (a modular subtract is two instructions shorter than a modular add)
                E2 nd = lo - hi, x = hi;
#pragma unroll
                for (int t = 0; t < K; t++) {
                    if (m == 0) pr[t] = x;
                    else if (m < K - 1) pr[t] = e2_mul_nc(pr[t], x);  // only multiplied again: skip canonicalisation
                    else if (!WIDE) pr[t] = pr[t] * x;
                    else e2acc_mac(wacc[t], pr[t], x);  // last factor: accumulate the product unreduced
                    if (t + 1 < K) x = x - nd;
                }
            }
            if (!WIDE) {
#pragma unroll
                for (int t = 0; t < K; t++) acc[t] = acc[t] + pr[t];
            }
        }
    }
    if (WIDE) {
#pragma unroll
        for (int t = 0; t < K; t++) acc[t] = e2acc_reduce(wacc[t]);
    }
    epilogue<K, NT>(acc, ep, smem, &s_flag);
}

// Software-pipelined form of the extension-field modes (MODE 0 / 2).  The plain kernel leaves the placement of the loads to the
// compiler, which issues a table's four loads and waits for them at once in two of the three table steps: the six waves of a SIMD
// start together, do identical work and stay in lock step, so the SIMD idles through every such wait (round 1 of the nv=26
// sumcheck: 4.96 TB/s against a 5.79 TB/s ceiling of the same access pattern without arithmetic, tools/ubench_bw.hip, with
// ~0.68 ms of VALU issue time under 0.83 ms of memory time).  Here the loads of the NEXT (table, pair) step are issued before the
// arithmetic of the current one — across the back edge of the pair loop too — and scheduling barriers keep them there; the cost
// is one more 64-byte buffer per lane.
template <int K, int MODE, bool WIDE_>
__global__ void __launch_bounds__(NT) k_dense_pf(TabPtrs<K> tp, size_t pairs, E2 r, Epilogue ep) {
    static_assert(MODE == 0 || MODE == 2, "extension-field modes only");
    __shared__ E2 smem[(NT / 64) * K];
    constexpr bool WIDE = WIDE_ && K > 1;
    constexpr int NL = MODE == 2 ? 4 : 2;  // extension elements per lane, table and step
    E2 acc[K];
    E2Acc wacc[K];
#pragma unroll
    for (int t = 0; t < K; t++) {
        acc[t] = e2_zero();
        wacc[t] = e2acc_zero();
    }
    const size_t stride = (size_t)gridDim.x * NT;
    if (ep.dbg && ep.bcast && blockIdx.x == 0 && threadIdx.x == 0) ep.bcast->dbg[ep.seq & 63][0] = wall_clock64();
    __shared__ unsigned long long s_chal[3];
    __shared__ int s_flag;
    size_t p = (size_t)blockIdx.x * NT + threadIdx.x;
    E2 nxt[NL];
    // the first loads do not depend on the challenge: they are in flight while a pipelined launch waits for it
    if (p < pairs) {
#pragma unroll
        for (int k = 0; k < NL; k++) nxt[k] = ld_e2(tp.in[0] + 2 * NL * p + 2 * k);
    }
    if (ep.wait_seq != 0) {
        if (!read_challenge(ep, r, s_chal)) return;  // pipeline aborted / timed out: leave everything untouched
    }
    const E2Pre rp = e2_pre(r);
    for (; p < pairs; p += stride) {
        E2 pr[K];
#pragma unroll
        for (int m = 0; m < K; m++) {
            E2 cur[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) cur[k] = nxt[k];
            if (m + 1 < K) {
#pragma unroll
                for (int k = 0; k < NL; k++) nxt[k] = ld_e2(tp.in[m + 1] + 2 * NL * p + 2 * k);
            } else {  // the next pair's first table (the last iteration re-reads its own: an L2 hit, and no branch around the loads)
                const size_t pn = p + stride < pairs ? p + stride : p;
#pragma unroll
                for (int k = 0; k < NL; k++) nxt[k] = ld_e2(tp.in[0] + 2 * NL * pn + 2 * k);
            }
            __builtin_amdgcn_sched_barrier(0);
            E2 lo, hi;
            if (MODE == 0) {
                lo = cur[0];
                hi = cur[1];
            } else {
                lo = cur[0] + e2_mul_pre(rp, cur[1] - cur[0]);
                hi = cur[2] + e2_mul_pre(rp, cur[3] - cur[2]);
                st_e2(tp.out[m] + 4 * p, lo);
                st_e2(tp.out[m] + 4 * p + 2, hi);
            }
            E2 nd = lo - hi, x = hi;
#pragma unroll
            for (int t = 0; t < K; t++) {
                if (m == 0) pr[t] = x;
                else if (m < K - 1) pr[t] = e2_mul_nc(pr[t], x);
                else if (!WIDE) pr[t] = pr[t] * x;
                else e2acc_mac(wacc[t], pr[t], x);
                if (t + 1 < K) x = x - nd;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!WIDE) {
#pragma unroll
            for (int t = 0; t < K; t++) acc[t] = acc[t] + pr[t];
        }
    }
    if (WIDE) {
#pragma unroll
        for (int t = 0; t < K; t++) acc[t] = e2acc_reduce(wacc[t]);
    }
    epilogue<K, NT>(acc, ep, smem, &s_flag);
}

// LDS-prefetched form of the folding rounds (MODE 2: extension tables in, fold + accumulate).  The plain kernel's waves issue the twelve
// 16-byte loads of an iteration, wait, multiply, store — the waves of a SIMD run in near lock step (they start together and do identical
// work), so the memory pipe idles while they multiply and the VALU idles while they wait: the fold rounds of the nv = 26 sumcheck take
// ~1.94 ms against 1.67 ms of memory time at the ceiling of their access pattern and 1.34 ms of VALU issue time
// (profiles/r03_dense_kernel_ab.json).  Here every wave keeps the NEXT iteration's 3 x 4 KB in flight through the LDS-DMA path
// (global_load_lds_dwordx4: no registers, lane l's 16 bytes land at base + 16 l) while it multiplies the current one: per table step
//     wait for this step's block (counted vmcnt: everything issued after it stays in flight) -> four ds_read_b128 -> re-issue the block's
//     loads for the next iteration into the same LDS lines -> fold, store, multiply.
// The compiler does not see the LDS-DMA loads (inline assembly), so it inserts no conservative vmcnt(0) in front of the LDS reads; the
// waits are written here.  vmcnt counts loads and stores in issue order on gfx9-family parts: between a block's loads and their use lie the
// 2 stores of its own step and the 6 operations (4 loads + 2 stores) of each of the other K - 1 steps.
template <int K>
__global__ void __launch_bounds__(NT) k_dense_lds(TabPtrs<K> tp, size_t pairs, E2 r, Epilogue ep) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];  // [wave][K][4][64] x 16 B
    __shared__ E2 smem[(NT / 64) * K];
    __shared__ unsigned long long s_chal[3];
    __shared__ int s_flag;
    E2 acc[K];
#pragma unroll
    for (int t = 0; t < K; t++) acc[t] = e2_zero();
    const size_t stride = (size_t)gridDim.x * NT;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    typedef __attribute__((address_space(3))) char lds_char;
    typedef unsigned int lds_u4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const lds_u4 lds_vec;
    auto e2_of = [](const lds_u4 v) { return E2{((uint64_t)v.y << 32) | v.x, ((uint64_t)v.w << 32) | v.z}; };
    lds_char* const wbase = (lds_char*)dyn + (size_t)wave * K * 4096;
    const unsigned lds0 = (unsigned)(uintptr_t)wbase;  // LDS byte address of this wave's region (wave-uniform)
    size_t p = (size_t)blockIdx.x * NT + threadIdx.x;
    // four 16-byte LDS-DMA loads of one (table, pair) block: element k of every lane -> line k of the block
    auto issue = [&](int m, size_t pp) {
        const uint64_t* q = tp.in[m] + 8 * pp;
        const unsigned l = lds0 + (unsigned)m * 4096u;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
            "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
            "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(q), "v"(q + 2), "v"(q + 4), "v"(q + 6), "s"(l), "s"(l + 1024u), "s"(l + 2048u), "s"(l + 3072u)
            : "memory");
    };
    const bool live = p < pairs;  // (pairs is a multiple of 64: a wave is live or idle as a whole)
    if (live) {
#pragma unroll
        for (int m = 0; m < K; m++) issue(m, p);
    }
    if (ep.wait_seq != 0) {
        if (!read_challenge(ep, r, s_chal)) {  // pipeline aborted / timed out: leave everything untouched
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    }
    const E2Pre rp = e2_pre(r);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first iteration's blocks (the loop's counted waits assume a full pipeline behind them)
    for (; p < pairs; p += stride) {
        const size_t pn = p + stride < pairs ? p + stride : p;  // (the last iteration re-reads its own blocks: no branch, uniform counts)
        E2 pr[K];
#pragma unroll
        for (int m = 0; m < K; m++) {
            constexpr int AFTER = 2 + 6 * (K - 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AFTER) : "memory");
            lds_vec* blk = (lds_vec*)(wbase + (size_t)m * 4096) + lane;
            const lds_u4 v0 = blk[0], v1 = blk[64], v2 = blk[128], v3 = blk[192];
            const E2 a0 = e2_of(v0), a1 = e2_of(v1), a2 = e2_of(v2), a3 = e2_of(v3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the lines are free again
            issue(m, pn);
            const E2 lo = a0 + e2_mul_pre(rp, a1 - a0);
            const E2 hi = a2 + e2_mul_pre(rp, a3 - a2);
            st_e2(tp.out[m] + 4 * p, lo);
            st_e2(tp.out[m] + 4 * p + 2, hi);
            E2 nd = lo - hi, x = hi;
#pragma unroll
            for (int t = 0; t < K; t++) {
                if (m == 0) pr[t] = x;
                else if (m < K - 1) pr[t] = e2_mul_nc(pr[t], x);
                else pr[t] = pr[t] * x;
                if (t + 1 < K) x = x - nd;
            }
        }
#pragma unroll
        for (int t = 0; t < K; t++) acc[t] = acc[t] + pr[t];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last iteration's spare loads must not land in LDS the epilogue reuses
    epilogue<K, NT>(acc, ep, smem, &s_flag);
}

// ------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------
static int dense_wide_mode() {  // tuning switch: 0 = never, 1 = read-only round only, 2 = every ext round
    static int m = [] {
        const char* e = getenv("CENO_HIP_DENSE_WIDE");
        return e ? atoi(e) : 1;
    }();
    return m;
}

static int dense_pf_mode() {  // tuning switch (bit 0: read-only round, bit 1: folding rounds): software-pipelined loads.  Measured
    // neutral on MI355X (profiles/r03_dense_kernel_ab.json: 2.88 vs 2.89 ms per nv=26 sumcheck on one box), so the default stays
    // with the compiler-scheduled form
    static int m = [] {
        const char* e = getenv("CENO_HIP_DENSE_PF");
        return e ? atoi(e) : 0;
    }();
    return m;
}

static int dense_lds_mode() {  // folding rounds on the LDS-prefetched kernel (k_dense_lds)
    static int m = [] {
        const char* e = getenv("CENO_HIP_DENSE_LDS");
        return e ? atoi(e) : 0;
    }();
    return m;
}

template <int K>
static void launch_dense_k(ceno_hip_ctx* ctx, int mode, const TabPtrs<K>& tp, size_t pairs, E2 r, const Epilogue& ep, unsigned grid, hipStream_t st) {
    const int wm = dense_wide_mode(), pf = dense_pf_mode();
    switch (mode) {
    case 0:
        if (pf & 1) {
            if (wm >= 1) hipLaunchKernelGGL((k_dense_pf<K, 0, true>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep);
            else hipLaunchKernelGGL((k_dense_pf<K, 0, false>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep);
        } else if (wm >= 1) hipLaunchKernelGGL((k_dense<K, 0, true>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep);
        else hipLaunchKernelGGL((k_dense<K, 0, false>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep);
        break;
    case 1: hipLaunchKernelGGL((k_dense<K, 1, false>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep); break;
    case 2:
        if constexpr (K >= 2 && K <= 3) {
            if (dense_lds_mode() && pairs >= ((size_t)1 << 16)) {
                // K x 16 KB of LDS per workgroup: three workgroups per CU, all resident (768 on 256 CUs)
                const unsigned g = std::min(grid, resident_grid(ctx, k_dense_lds<K>, NT, (size_t)K * 16384, MAXB));
                hipLaunchKernelGGL((k_dense_lds<K>), dim3(g), dim3(NT), (size_t)K * 16384, st, tp, pairs, r, ep);
                break;
            }
        }
        if (pf & 2) {
            if (wm >= 2) hipLaunchKernelGGL((k_dense_pf<K, 2, true>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep);
            else hipLaunchKernelGGL((k_dense_pf<K, 2, false>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep);
        } else if (wm >= 2) hipLaunchKernelGGL((k_dense<K, 2, true>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep);
        else hipLaunchKernelGGL((k_dense<K, 2, false>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep);
        break;
    default: hipLaunchKernelGGL((k_dense<K, 3, false>), dim3(grid), dim3(NT), 0, st, tp, pairs, r, ep); break;
    }
}


void launch_dense_tables(ceno_hip_ctx* ctx, int K, int mode, const DenseTables& t, size_t pairs, E2 r, const Epilogue& ep, unsigned grid, hipStream_t st) {
    auto go = [&](auto kc) {
        constexpr int KK = decltype(kc)::value;
        TabPtrs<KK> tp;
        for (int m = 0; m < KK; m++) {
            tp.in[m] = t.in[m];
            tp.out[m] = t.out[m];
        }
        launch_dense_k<KK>(ctx, mode, tp, pairs, r, ep, grid, st);
    };
    switch (K) {
    case 1: go(std::integral_constant<int, 1>{}); break;
    case 2: go(std::integral_constant<int, 2>{}); break;
    case 3: go(std::integral_constant<int, 3>{}); break;
    default: go(std::integral_constant<int, 4>{}); break;
    }
}
