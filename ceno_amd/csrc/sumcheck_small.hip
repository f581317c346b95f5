// Small and mid-size rounds of a pipelined generic sumcheck: see sumcheck_small.hpp.  Reference semantics: the rounds of
// IOPProverState::prove (EXT sumcheck crate; call sites gkr_iop/src/gkr/layer/cpu/mod.rs:80-96, ceno_zkvm/src/scheme/cpu/mod.rs:409-494)
// on tables that fit in LDS or nearly so, where a round is a latency chain and not a bandwidth problem.
#include <map>
#include <mutex>

#include "sumcheck_small.hpp"

#include <algorithm>
#include <cstdlib>

#ifndef CENO_SMALL_SETPRIO
#define CENO_SMALL_SETPRIO 1
#endif

// a plan term flattened for the term-parallel kernels: coefficient, then its own factors followed by its group's common
// factors (their total is <= D <= 8, checked at begin)
struct alignas(16) TailTerm {
    E2 c;
    uint32_t nf;
    uint16_t idx[8];
    uint32_t pad[3];
};
static_assert(sizeof(TailTerm) == 48, "TailTerm layout");
__device__ __forceinline__ void flatten_term(const DevPlan& pl, int ti, TailTerm& t) {
    int g = 0;
    while ((int)pl.group_term_off[g + 1] <= ti) g++;
    const uint32_t term = pl.group_terms[ti];
    t.c = pl.coeffs[term];
    uint32_t nf = 0;
    for (uint32_t k = pl.term_off[term]; k < pl.term_off[term + 1] && nf < 8; k++) t.idx[nf++] = (uint16_t)pl.term_idx[k];
    for (uint32_t k = pl.common_off[g]; k < pl.common_off[g + 1] && nf < 8; k++) t.idx[nf++] = (uint16_t)pl.common_idx[k];
    t.nf = nf;
}

// ------------------------------------------------------------------------------------------------
// term-parallel generic round for small / mid-size rounds.  k_fused gives one pair to one lane, which then walks the
// whole plan serially: for a 33-term degree-4 layer that is ~400 dependent ext multiplies = 120 us per round however
// small the round is.  Here a workgroup owns a tile of TP pairs: phase 1 spreads the folds (MLE x pair) over the
// lanes and stages (f(1), delta) in LDS, phase 2 spreads (term x pair) over the lanes; a term of a group with common
// factors multiplies them in itself (the sum over terms is linear).  Rounds become 10-20 us.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(NT) k_tile(DevPlan pl, int n_mles, int n_flat, int TP, size_t pairs, E2 r, Epilogue ep, int flat_in_lds) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    if (CENO_SMALL_SETPRIO) __builtin_amdgcn_s_setprio(3);  // latency chain: win issue arbitration against bulk kernels of other lanes
    E2* stage = reinterpret_cast<E2*>(dyn);                               // [n_mles][2][TP]
    E2* smem = stage + (size_t)n_mles * 2 * TP;                    // [(NT/64) * D]
    unsigned long long* s_chal = reinterpret_cast<unsigned long long*>(smem + (NT / 64) * D);  // 3 words + flag
    int* s_flag = reinterpret_cast<int*>(s_chal + 4);
    // the flattened plan in LDS (when it fits): its walk in global memory is a chain of dependent loads that phase 2 would
    // otherwise start only after phase 1; issued here it overlaps the challenge relay, the cold table loads and the folds
    TailTerm* ft = reinterpret_cast<TailTerm*>(s_chal + 6);
    if (flat_in_lds)
        for (int ti = threadIdx.x; ti < n_flat; ti += NT) flatten_term(pl, ti, ft[ti]);
    if (ep.dbg && ep.bcast && blockIdx.x == 0 && threadIdx.x == 0) ep.bcast->dbg[ep.seq & 63][0] = wall_clock64();
    if (ep.wait_seq != 0) {
        if (!read_challenge(ep, r, s_chal)) return;
    }
    const E2Pre rp = e2_pre(r);
    const bool fold = pl.use_out != 0;
    E2 acc[D];
#pragma unroll
    for (int t = 0; t < D; t++) acc[t] = e2_zero();
    const size_t n_tiles = (pairs + TP - 1) / TP;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t p0 = tile * TP;
        for (int idx = threadIdx.x; idx < n_mles * TP; idx += NT) {
            const int m = idx / TP, q = idx - m * TP;
            const size_t p = p0 + q;
            if (p >= pairs) continue;
            const MleSlot sl = pl.slots[m];
            E2 lo, hi;
            if (fold) {
                if (sl.in_ext) {
                    const uint64_t* qq = sl.in + 8 * p;
                    const E2 a0 = ld_e2(qq), a1 = ld_e2(qq + 2), a2 = ld_e2(qq + 4), a3 = ld_e2(qq + 6);
                    lo = a0 + e2_mul_pre(rp, a1 - a0);
                    hi = a2 + e2_mul_pre(rp, a3 - a2);
                } else {
                    const uint64_t* qq = sl.in + 4 * p;
                    const ulonglong2 v0 = *reinterpret_cast<const ulonglong2*>(qq);
                    const ulonglong2 v1 = *reinterpret_cast<const ulonglong2*>(qq + 2);
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
            stage[(size_t)(2 * m) * TP + q] = hi;           // f(1)
            stage[(size_t)(2 * m + 1) * TP + q] = hi - lo;  // delta
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < n_flat * TP; idx += NT) {
            const int ti = idx / TP, q = idx - ti * TP;
            if (p0 + q >= pairs) continue;
            E2 pr[D];
            bool seeded = false;
            if (flat_in_lds) {
                const TailTerm& tt = ft[ti];
                const E2 c = tt.c;
#pragma unroll
                for (int t = 0; t < D; t++) pr[t] = c;
                for (uint32_t k = 0; k < tt.nf; k++) {
                    const uint32_t m = tt.idx[k];
                    mul_points<D>(pr, seeded, c, stage[(size_t)(2 * m) * TP + q], stage[(size_t)(2 * m + 1) * TP + q]);
                }
            } else {
                int g = 0;
                while ((int)pl.group_term_off[g + 1] <= ti) g++;
                const uint32_t term = pl.group_terms[ti];
                const E2 c = pl.coeffs[term];
#pragma unroll
                for (int t = 0; t < D; t++) pr[t] = c;
                for (uint32_t k = pl.term_off[term]; k < pl.term_off[term + 1]; k++) {
                    const uint32_t m = pl.term_idx[k];
                    mul_points<D>(pr, seeded, c, stage[(size_t)(2 * m) * TP + q], stage[(size_t)(2 * m + 1) * TP + q]);
                }
                for (uint32_t k = pl.common_off[g]; k < pl.common_off[g + 1]; k++) {
                    const uint32_t m = pl.common_idx[k];
                    mul_points<D>(pr, seeded, c, stage[(size_t)(2 * m) * TP + q], stage[(size_t)(2 * m + 1) * TP + q]);
                }
            }
#pragma unroll
            for (int t = 0; t < D; t++) acc[t] = acc[t] + pr[t];
        }
        __syncthreads();  // the stage is reused by the next tile
    }
    epilogue<D, NT>(acc, ep, smem, s_flag);
}

// ------------------------------------------------------------------------------------------------
// persistent tail: ALL remaining rounds of a pipelined single-class sumcheck in ONE launch once the tables fit in LDS.
// A small round is ~3 us of work wrapped in ~10 us of kernel boundary, relay and cold loads; here the workgroup keeps
// the tables in LDS, publishes each message, polls the mailbox for the challenge itself and folds in place
// (ping-pong), so a round costs the publish + the host round trip + the arithmetic.  The tables of the last round
// go back to the buffer ceno_hip_sumcheck_finish expects.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(NT) k_tail(DevPlan pl, const MleSlot* __restrict__ last_slots, int n_mles, int n_flat, int pairs0, int i0, int n,
                                             E2 r, Epilogue ep, E2* __restrict__ out_evals, E2* __restrict__ export_host) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    if (CENO_SMALL_SETPRIO) __builtin_amdgcn_s_setprio(3);  // latency chain: win issue arbitration against bulk kernels of other lanes
    E2* bufA = reinterpret_cast<E2*>(dyn);                 // [n_mles][2 * pairs0]
    E2* bufB = bufA + (size_t)n_mles * 2 * pairs0;         // [n_mles][pairs0]
    E2* smem = bufB + (size_t)n_mles * pairs0;             // [(NT/64) * D]
    unsigned long long* s_chal = reinterpret_cast<unsigned long long*>(smem + (NT / 64) * D);  // c0, c1, ok
    // the plan, flattened once into LDS: every round walks it again, and from global memory that walk is a chain of 4-5
    // dependent loads (group -> term -> offsets -> indices) on the critical path of a ~10 us round.  A flat term carries its
    // own factors followed by its group's common factors (their total is <= D, checked at begin).
    TailTerm* ft = reinterpret_cast<TailTerm*>(s_chal + 4);
    // filled in place: a local struct indexed at run time would sit in scratch memory
    for (int ti = threadIdx.x; ti < n_flat; ti += NT) flatten_term(pl, ti, ft[ti]);
    if (ep.wait_seq != 0) {
        if (!read_challenge(ep, r, s_chal)) return;
    }
    const int sa = 2 * pairs0, sb = pairs0;  // per-MLE strides of the two LDS images
    {   // stage the tables of round i0 (folded with r_{i0-1} unless i0 == 0)
        const E2Pre rp = e2_pre(r);
        const bool fold = pl.use_out != 0;
        for (int idx = threadIdx.x; idx < n_mles * sa; idx += NT) {
            const int m = idx / sa, j = idx - m * sa;
            const MleSlot sl = pl.slots[m];
            E2 v;
            if (fold) {
                if (sl.in_ext) {
                    const E2 a = ld_e2(sl.in + 4 * (size_t)j), b = ld_e2(sl.in + 4 * (size_t)j + 2);
                    v = a + e2_mul_pre(rp, b - a);
                } else {
                    const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(sl.in + 2 * (size_t)j);
                    const E2 t = e2_mul_base(r, sub(w.y, w.x));
                    v = E2{add(t.c0, w.x), t.c1};
                }
            } else {
                v = sl.in_ext ? ld_e2(sl.in + 2 * (size_t)j) : E2{sl.in[j], 0};
            }
            bufA[idx] = v;
        }
    }
    __syncthreads();
    E2 *cur = bufA, *nxt = bufB;
    int sc_ = sa, sn = sb, pairs = pairs0;
    for (int i = i0; i < n; i++) {
        if (ep.dbg && ep.bcast && threadIdx.x == 0) ep.bcast->dbg[(i + 1) & 63][0] = wall_clock64();
        E2 acc[D];
#pragma unroll
        for (int t = 0; t < D; t++) acc[t] = e2_zero();
        for (int idx = threadIdx.x; idx < n_flat * pairs; idx += NT) {
            const int ti = idx / pairs, p = idx - ti * pairs;
            const TailTerm& tt = ft[ti];
            const E2 c = tt.c;
            E2 pr[D];
#pragma unroll
            for (int t = 0; t < D; t++) pr[t] = c;
            bool seeded = false;
            for (uint32_t k = 0; k < tt.nf; k++) {
                const E2* q = cur + (size_t)tt.idx[k] * sc_ + 2 * p;
                mul_points<D>(pr, seeded, c, q[1], q[1] - q[0]);
            }
#pragma unroll
            for (int t = 0; t < D; t++) acc[t] = acc[t] + pr[t];
        }
        red::block_sum<D, NT>(acc, smem);
        if (threadIdx.x == 0) {
            // next_seq = 0: the challenge is fetched right here, not relayed to another launch
            finish_message<D, false>(acc, ep, (unsigned long long)(i + 1), 0ull);
            if (i + 1 < n || out_evals) {  // after the last message: the challenge the final evaluations are taken at
                unsigned long long c0 = 0, c1 = 0;
                const bool ok = poll_challenge(ep.mailbox, (unsigned long long)(i + 1), c0, c1, ep.poll_ticks);
                s_chal[0] = c0;
                s_chal[1] = c1;
                s_chal[2] = ok ? 1ull : 0ull;
            }
        }
        if (i + 1 == n) break;
        __syncthreads();
        if (s_chal[2] == 0) return;  // aborted / timed out: leave everything as it is
        const E2Pre rp = e2_pre(E2{s_chal[0], s_chal[1]});
        for (int idx = threadIdx.x; idx < n_mles * pairs; idx += NT) {
            const int m = idx / pairs, j = idx - m * pairs;
            const E2 lo = cur[(size_t)m * sc_ + 2 * j], hi = cur[(size_t)m * sc_ + 2 * j + 1];
            nxt[(size_t)m * sn + j] = lo + e2_mul_pre(rp, hi - lo);
        }
        __syncthreads();
        E2* tp_ = cur; cur = nxt; nxt = tp_;
        const int ts_ = sc_; sc_ = sn; sn = ts_;
        pairs >>= 1;
    }
    // Host-finished tail (`n` is then the first round the HOST computes): the tables round n-1 was computed on — still unfolded,
    // 2 * pairs entries each — go to pinned host memory right behind the message and the kernel is done.  The host folds with the
    // challenge it samples next and runs the remaining rounds itself: a round of <= 64 pairs is a few microseconds of host arithmetic
    // against ~10 us of PCIe round trip per round here.  (The words were armed with MSG_INVALID: the host waits for all of them.)
    if (export_host) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const int len = 2 * pairs;
        for (int idx = threadIdx.x; idx < n_mles * len; idx += NT) {
            const int m = idx / len, j = idx - m * len;
            const E2 v = cur[(size_t)m * sc_ + j];
            const u4 w = {(unsigned)v.c0, (unsigned)(v.c0 >> 32), (unsigned)v.c1, (unsigned)(v.c1 >> 32)};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(export_host + idx), "v"(w) : "memory");
        }
        return;
    }
    // the final evaluations f_m(r_0..r_{n-1}) = lo + r_{n-1} (hi - lo), straight into the pinned words the host watches: no
    // separate launch (and kernel boundary) for them at the end of every sumcheck
    if (out_evals) {
        __syncthreads();
        if (s_chal[2] != 0) {
            const E2Pre rp = e2_pre(E2{s_chal[0], s_chal[1]});
            for (int m = threadIdx.x; m < n_mles; m += NT) {
                const E2 lo = cur[(size_t)m * sc_], hi = cur[(size_t)m * sc_ + 1];
                const E2 v = lo + e2_mul_pre(rp, hi - lo);
                typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                const u4 w = {(unsigned)v.c0, (unsigned)(v.c0 >> 32), (unsigned)v.c1, (unsigned)(v.c1 >> 32)};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out_evals + m), "v"(w) : "memory");
            }
        }
    }
    // the two elements per table that the last round was computed on (ceno_hip_sumcheck_finish folds them when the
    // evaluations were not produced above)
    for (int idx = threadIdx.x; idx < n_mles * 2; idx += NT) {
        const int m = idx >> 1, j = idx & 1;
        st_e2(last_slots[m].out + 2 * j, cur[(size_t)m * sc_ + j]);
    }
}

// ------------------------------------------------------------------------------------------------
// persistent MID rounds: the rounds between the streaming kernels and the single-workgroup tail, in ONE launch of W
// workgroups and without a kernel boundary.  LSB-first folding pairs neighbours, so workgroup b keeps ITS slice of every
// table (elements [b 2S0, (b+1) 2S0) of round i0) in LDS for good and folds it in place — no table data ever crosses
// workgroups.  Per round the workgroups only send their D partial sums to workgroup 0 (armed rows, see below), which
// adds them and publishes the message; the challenge is read from the mailbox by every workgroup itself when the mailbox
// lives in device memory (large-BAR boxes), otherwise workgroup 0 fetches it across PCIe and passes it on through one
// 64-byte line PER WORKGROUP (pollers of a shared line serialise at ~90 loads per us).  The launch ends with the
// last round too large for the single-workgroup tail: the slices go back to memory and the tail kernel, already queued, takes
// over (its rounds cost ~9.5 us against ~15 us here, so nothing that fits the tail is kept).
// Relay words carry a per-sumcheck nonce, so the lines need no clearing.  All W workgroups must be resident at once
// (they wait for each other): the host books every launch against a residency budget computed from the runtime's occupancy
// for the launch's own dynamic LDS and the device's CU count (sumcheck.hip, mid_blocks_per_cu), W <= 256.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(NT) k_mid(DevPlan pl, const MleSlot* __restrict__ out_slots, int n_mles, int n_flat, int S0, int i0, int i1, E2 r,
                                            Epilogue ep, MidRelay* __restrict__ relay, unsigned long long nonce, int direct_poll) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    if (CENO_SMALL_SETPRIO) __builtin_amdgcn_s_setprio(3);  // latency chain: win issue arbitration against bulk kernels of other lanes
    const int stride = 2 * S0;
    E2* tab = reinterpret_cast<E2*>(dyn);                  // [n_mles][2 * S0], folded in place
    E2* smem = tab + (size_t)n_mles * stride;              // [(NT/64) * D]
    unsigned long long* s_chal = reinterpret_cast<unsigned long long*>(smem + (NT / 64) * D);  // c0, c1, ok, last
    TailTerm* ft = reinterpret_cast<TailTerm*>(s_chal + 4);
    const int W = gridDim.x, b = blockIdx.x;
    for (int ti = threadIdx.x; ti < n_flat; ti += NT) flatten_term(pl, ti, ft[ti]);
    if (ep.dbg && ep.bcast && b == 0 && threadIdx.x == 0) ep.bcast->dbg[(i0 + 1) & 63][0] = wall_clock64();
    if (ep.wait_seq != 0) {
        if (!read_challenge(ep, r, s_chal)) return;
    }
    {   // stage this workgroup's slice of the tables of round i0 (folded with r_{i0-1} unless i0 == 0)
        const E2Pre rp = e2_pre(r);
        const bool fold = pl.use_out != 0;
        for (int idx = threadIdx.x; idx < n_mles * stride; idx += NT) {
            const int m = idx / stride, j = idx - m * stride;
            const size_t g = (size_t)b * stride + j;
            const MleSlot sl = pl.slots[m];
            E2 v;
            if (fold) {
                if (sl.in_ext) {
                    const E2 a = ld_e2(sl.in + 4 * g), c = ld_e2(sl.in + 4 * g + 2);
                    v = a + e2_mul_pre(rp, c - a);
                } else {
                    const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(sl.in + 2 * g);
                    const E2 t = e2_mul_base(r, sub(w.y, w.x));
                    v = E2{add(t.c0, w.x), t.c1};
                }
            } else {
                v = sl.in_ext ? ld_e2(sl.in + 2 * g) : E2{sl.in[g], 0};
            }
            tab[idx] = v;
        }
    }
    __syncthreads();
    int pairs = S0;
    for (int i = i0; i <= i1; i++) {
        if (ep.dbg && ep.bcast && b == 0 && threadIdx.x == 0 && i > i0) ep.bcast->dbg[(i + 1) & 63][0] = wall_clock64();
        E2 acc[D];
#pragma unroll
        for (int t = 0; t < D; t++) acc[t] = e2_zero();
        for (int idx = threadIdx.x; idx < n_flat * pairs; idx += NT) {
            const int ti = idx / pairs, p = idx - ti * pairs;
            const TailTerm& tt = ft[ti];
            const E2 c = tt.c;
            E2 pr[D];
#pragma unroll
            for (int t = 0; t < D; t++) pr[t] = c;
            bool seeded = false;
            for (uint32_t k = 0; k < tt.nf; k++) {
                const E2* q = tab + (size_t)tt.idx[k] * stride + 2 * p;
                mul_points<D>(pr, seeded, c, q[1], q[1] - q[0]);
            }
#pragma unroll
            for (int t = 0; t < D; t++) acc[t] = acc[t] + pr[t];
        }
        red::block_sum<D, NT>(acc, smem);
        // Exchange of the partial sums without a counter: every workgroup fires its D partial sums at its row (write-through,
        // no wait) and workgroup 0 watches the rows — they were armed with MSG_INVALID, which no canonical element equals —
        // one lane per row, adds them, re-arms the rows and publishes.  (An arrival counter costs the writer a store drain and
        // an atomic round trip and the last arriver an acquire + reload: three trips through memory instead of one.)
        // two sets of rows, used alternately: a set is re-armed right AFTER its message has gone out (stores in flight, nobody
        // waits) and is next written two rounds later, behind a drain that by then costs nothing
        uint64_t* const rows = ep.partials + (size_t)(i & 1) * W * D * 2;
        if (threadIdx.x == 0) {
            uint64_t* row = rows + (size_t)b * D * 2;
#pragma unroll
            for (int t = 0; t < D; t++) {
                typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                const u4 w = {(unsigned)acc[t].c0, (unsigned)(acc[t].c0 >> 32), (unsigned)acc[t].c1, (unsigned)(acc[t].c1 >> 32)};
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(row + 2 * t), "v"(w) : "memory");
            }
        }
        if (i == i1) {  // the tables this round was computed on go back to memory for the tail kernel: `pairs` pairs per workgroup
            const int ne = 2 * pairs;
            for (int idx = threadIdx.x; idx < n_mles * ne; idx += NT) {
                const int m = idx / ne, j = idx - m * ne;
                st_e2(out_slots[m].out + 2 * ((size_t)b * ne + j), tab[(size_t)m * stride + j]);
            }
        }
        if (b == 0) {
            if (threadIdx.x == 0) s_chal[3] = 1;
            __syncthreads();  // smem is reused
            E2 tot[D];
#pragma unroll
            for (int t = 0; t < D; t++) tot[t] = e2_zero();
            bool got = true;
            for (int bb = threadIdx.x; bb < W; bb += NT) {
                uint64_t* row = rows + (size_t)bb * D * 2;
                const unsigned long long t0 = wall_clock64();
                unsigned spins = 0;
                uint64_t w[2 * D];
                for (;;) {
                    bool all = true;
#pragma unroll
                    for (int k = 0; k < 2 * D; k++) {
                        w[k] = ld_agent(row + k);
                        all = all && w[k] != MSG_INVALID;
                    }
                    if (all) break;
                    if ((++spins & 63u) == 0 && wall_clock64() - t0 > ep.poll_ticks) { got = false; break; }
                }
                if (!got) {
                    s_chal[3] = 0;  // a workgroup never delivered (it was never resident, or the device is going down)
                    break;
                }
#pragma unroll
                for (int t = 0; t < D; t++) tot[t] = tot[t] + E2{w[2 * t], w[2 * t + 1]};
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-arming stores of the previous round (other set): long done
            __syncthreads();
            if (s_chal[3] == 0) return;  // nothing is published: the host sees the stream drain without a message
            red::block_sum<D, NT>(tot, smem);
            if (threadIdx.x == 0) finish_message<D, false>(tot, ep, (unsigned long long)(i + 1), 0ull);
            for (int bb = threadIdx.x; bb < W; bb += NT) {  // re-arm this round's set behind the message
                uint64_t* row = rows + (size_t)bb * D * 2;
#pragma unroll
                for (int k = 0; k < 2 * D; k++) st_agent(row + k, MSG_INVALID);
            }
            if (threadIdx.x == 0) {
                if (i == i1 || !direct_poll) {
                    unsigned long long c0 = 0, c1 = 0;
                    const bool ok = poll_challenge(ep.mailbox, (unsigned long long)(i + 1), c0, c1, ep.poll_ticks);
                    s_chal[0] = c0;
                    s_chal[1] = c1;
                    s_chal[2] = ok ? 1ull : 0ull;
                    if (i == i1) {  // the next round belongs to the tail kernel: relay the way every launch does (read_challenge)
                        Bcast* bc = ep.bcast;
                        if (ok) {
                            __hip_atomic_store(&bc->chal[0], c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&bc->chal[1], c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        __hip_atomic_store(&bc->ready_seq, ok ? (unsigned)(i + 1) : ABORT_SEQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (i == i1) return;
            if (!direct_poll) {  // mailbox across PCIe: one poller, the challenge travels on through one line per workgroup
                __syncthreads();
                for (int t = threadIdx.x; t < W; t += NT) {
                    if (t == 0) continue;
                    MidRelay* line = relay + t;
                    __hip_atomic_store(&line->chal[0], s_chal[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&line->chal[1], s_chal[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&line->seq, (nonce << 8) | (s_chal[2] ? (unsigned long long)(i + 1) : 0xFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        } else if (i == i1) {
            return;
        }
        if (direct_poll) {
            // the mailbox lives in device memory the host writes through the BAR: every workgroup watches it itself
            if (threadIdx.x == 0) {
                unsigned long long c0 = 0, c1 = 0;
                const bool ok = poll_challenge(ep.mailbox, (unsigned long long)(i + 1), c0, c1, ep.poll_ticks);
                s_chal[0] = c0;
                s_chal[1] = c1;
                s_chal[2] = ok ? 1ull : 0ull;
            }
        } else if (b != 0) {
            if (threadIdx.x == 0) {
                const MidRelay* line = relay + b;
                const unsigned long long want = (nonce << 8) | (unsigned long long)(i + 1), dead = (nonce << 8) | 0xFFull;
                const unsigned long long t0 = wall_clock64();
                unsigned spins = 0;
                bool ok = false;
                for (;;) {
                    const unsigned long long v = __hip_atomic_load(&line->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v == want) { ok = true; break; }
                    if (v == dead) break;
                    if ((++spins & 63u) == 0 && wall_clock64() - t0 > 2 * ep.poll_ticks) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                s_chal[0] = ok ? __hip_atomic_load(&line->chal[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                s_chal[1] = ok ? __hip_atomic_load(&line->chal[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                s_chal[2] = ok ? 1ull : 0ull;
            }
        }
        __syncthreads();
        if (s_chal[2] == 0) return;  // aborted / timed out
        // fold in place: item j of a table reads elements 2j, 2j+1 and writes element j; a pass reads everything it needs
        // before the barrier, and later passes only touch higher indices
        const E2Pre rp = e2_pre(E2{s_chal[0], s_chal[1]});
        const int items = n_mles * pairs;
        for (int base = 0; base < items; base += NT) {
            const int idx = base + threadIdx.x;
            E2 v = e2_zero();
            int m = 0, j = 0;
            if (idx < items) {
                m = idx / pairs;
                j = idx - m * pairs;
                const E2 lo = tab[(size_t)m * stride + 2 * j], hi = tab[(size_t)m * stride + 2 * j + 1];
                v = lo + e2_mul_pre(rp, hi - lo);
            }
            __syncthreads();
            if (idx < items) tab[(size_t)m * stride + j] = v;
        }
        __syncthreads();
        pairs >>= 1;
    }
}


// tile geometry of k_tile: as many pairs per workgroup as keep the (term x pair) items within ONE pass of the 256 lanes
// (a second, nearly empty pass would double the evaluation time), bounded by the LDS stage
int tile_pairs(size_t n_flat, size_t n_mles, size_t pairs) {
    size_t tp = std::max<size_t>(1, NT / std::max<size_t>(1, n_flat));
    tp = std::min<size_t>(tp, 64);
    tp = std::min(tp, pairs);
    while (tp > 1 && n_mles * 2 * tp * sizeof(E2) > 48 * 1024) tp--;
    return (int)tp;
}
// persistent tail eligibility: tables of the round (2 * pairs per MLE) plus the ping-pong half must fit in LDS
static size_t tail_max_pairs() {
    static size_t v = [] {
        const char* e = getenv("CENO_HIP_TAIL_PAIRS");  // 0 disables the persistent tail (A/B measurements)
        return (size_t)(e ? atoi(e) : 128);
    }();
    return v;
}
static size_t tail_lds_bytes(size_t n_mles, size_t pairs, int d, size_t n_flat) {
    return (n_mles * 3 * pairs + (size_t)(NT / 64) * d) * sizeof(E2) + 64 + n_flat * 48;
}
bool tail_eligible(size_t n_mles, size_t pairs, int d, size_t n_flat) {
    return pairs >= 1 && pairs <= tail_max_pairs() && n_mles < 65536 && tail_lds_bytes(n_mles, pairs, d, n_flat) <= 60 * 1024;
}
template <int D>
static void launch_tail_d(const DevPlan& pl, const MleSlot* last_slots, int n_mles, int n_flat, size_t pairs, int i0, int n, const Epilogue& ep,
                          E2* out_evals, E2* export_host, hipStream_t st) {
    hipLaunchKernelGGL((k_tail<D>), dim3(1), dim3(NT), tail_lds_bytes((size_t)n_mles, pairs, D, (size_t)n_flat), st, pl, last_slots, n_mles, n_flat, (int)pairs, i0, n,
                       e2_zero(), ep, out_evals, export_host);
}
void launch_tail(int d, const DevPlan& pl, const MleSlot* last_slots, int n_mles, int n_flat, size_t pairs, int i0, int n, const Epilogue& ep,
                        E2* out_evals, E2* export_host, hipStream_t st) {
    switch (d) {
    case 1: launch_tail_d<1>(pl, last_slots, n_mles, n_flat, pairs, i0, n, ep, out_evals, export_host, st); break;
    case 2: launch_tail_d<2>(pl, last_slots, n_mles, n_flat, pairs, i0, n, ep, out_evals, export_host, st); break;
    case 3: launch_tail_d<3>(pl, last_slots, n_mles, n_flat, pairs, i0, n, ep, out_evals, export_host, st); break;
    case 4: launch_tail_d<4>(pl, last_slots, n_mles, n_flat, pairs, i0, n, ep, out_evals, export_host, st); break;
    case 5: launch_tail_d<5>(pl, last_slots, n_mles, n_flat, pairs, i0, n, ep, out_evals, export_host, st); break;
    case 6: launch_tail_d<6>(pl, last_slots, n_mles, n_flat, pairs, i0, n, ep, out_evals, export_host, st); break;
    case 7: launch_tail_d<7>(pl, last_slots, n_mles, n_flat, pairs, i0, n, ep, out_evals, export_host, st); break;
    default: launch_tail_d<8>(pl, last_slots, n_mles, n_flat, pairs, i0, n, ep, out_evals, export_host, st); break;
    }
}

// persistent mid rounds (k_mid): W workgroups of S0 pairs each; S0 = the largest power of two <= 128 whose slice fits
static size_t mid_lds_bytes(size_t n_mles, size_t s0, int d, size_t n_flat) {
    return (n_mles * 2 * s0 + (size_t)(NT / 64) * d) * sizeof(E2) + 64 + n_flat * 48;
}
// (read per call, not cached: the test-suite switches them between sumchecks)
static int mid_max_w() {
    const char* e = getenv("CENO_HIP_MID_W");  // 0 disables the persistent mid rounds (A/B measurements)
    int w = e ? atoi(e) : 256;
    while (w & (w - 1)) w &= w - 1;
    return std::min(w, 256);
}
// geometry for a round of `pairs` pairs, or W = 0 when the round is not (yet) one for k_mid: as many workgroups as allowed
// (CENO_HIP_MID_W, default 256, and what the context's residency budget still has), slices of at most CENO_HIP_MID_S0 (default 128) pairs that fit the LDS
void mid_geometry(size_t n_mles, size_t pairs, int d, size_t n_flat, int w_cap, int* W, int* S0) {
    *W = 0;
    *S0 = 0;
    w_cap = std::min(w_cap, mid_max_w());
    while (w_cap & (w_cap - 1)) w_cap &= w_cap - 1;
    if (w_cap < 4 || n_mles >= 65536) return;
    size_t s0 = 128;
    if (const char* e = getenv("CENO_HIP_MID_S0")) {
        s0 = (size_t)std::max(atoi(e), 2);
        while (s0 & (s0 - 1)) s0 &= s0 - 1;
    }
    while (s0 >= 2 && mid_lds_bytes(n_mles, s0, d, n_flat) > 60 * 1024) s0 >>= 1;
    if (s0 < 2 || pairs > (size_t)w_cap * s0 || pairs < 8) return;
    const size_t w = std::min<size_t>((size_t)w_cap, pairs / 2);  // at least two pairs per workgroup in the first round
    *W = (int)w;
    *S0 = (int)(pairs / w);
}
template <int D>
static int mid_occupancy_d(size_t lds) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_mid<D>, NT, lds) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nb;
}
int mid_blocks_per_cu(int d, size_t n_mles, size_t S0, size_t n_flat) {
    const size_t lds = mid_lds_bytes(n_mles, S0, std::min(std::max(d, 1), 8), n_flat);
    // cached by (degree, LDS rounded up to 2 KB): the query costs a few microseconds and sits in front of every tower layer
    static PoolMutex mu;
    static std::map<std::pair<int, size_t>, int> cache;
    const std::pair<int, size_t> key{std::min(std::max(d, 1), 8), (lds + 2047) / 2048};
    {
        std::lock_guard<PoolMutex> g(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    const size_t q = key.second * 2048;
    int nb = 0;
    switch (key.first) {
    case 1: nb = mid_occupancy_d<1>(q); break;
    case 2: nb = mid_occupancy_d<2>(q); break;
    case 3: nb = mid_occupancy_d<3>(q); break;
    case 4: nb = mid_occupancy_d<4>(q); break;
    case 5: nb = mid_occupancy_d<5>(q); break;
    case 6: nb = mid_occupancy_d<6>(q); break;
    case 7: nb = mid_occupancy_d<7>(q); break;
    default: nb = mid_occupancy_d<8>(q); break;
    }
    std::lock_guard<PoolMutex> g(mu);
    cache[key] = nb;
    return nb;
}
template <int D>
static void launch_mid_d(const DevPlan& pl, const MleSlot* out_slots, int n_mles, int n_flat, int W, int S0, int i0, int i1, const Epilogue& ep,
                         MidRelay* relay, unsigned long long nonce, int direct_poll, hipStream_t st) {
    hipLaunchKernelGGL((k_mid<D>), dim3((unsigned)W), dim3(NT), mid_lds_bytes((size_t)n_mles, (size_t)S0, D, (size_t)n_flat), st, pl, out_slots, n_mles, n_flat,
                       S0, i0, i1, e2_zero(), ep, relay, nonce, direct_poll);
}
void launch_mid(int d, const DevPlan& pl, const MleSlot* out_slots, int n_mles, int n_flat, int W, int S0, int i0, int i1, const Epilogue& ep,
                       MidRelay* relay, unsigned long long nonce, int direct_poll, hipStream_t st) {
    switch (d) {
    case 1: launch_mid_d<1>(pl, out_slots, n_mles, n_flat, W, S0, i0, i1, ep, relay, nonce, direct_poll, st); break;
    case 2: launch_mid_d<2>(pl, out_slots, n_mles, n_flat, W, S0, i0, i1, ep, relay, nonce, direct_poll, st); break;
    case 3: launch_mid_d<3>(pl, out_slots, n_mles, n_flat, W, S0, i0, i1, ep, relay, nonce, direct_poll, st); break;
    case 4: launch_mid_d<4>(pl, out_slots, n_mles, n_flat, W, S0, i0, i1, ep, relay, nonce, direct_poll, st); break;
    case 5: launch_mid_d<5>(pl, out_slots, n_mles, n_flat, W, S0, i0, i1, ep, relay, nonce, direct_poll, st); break;
    case 6: launch_mid_d<6>(pl, out_slots, n_mles, n_flat, W, S0, i0, i1, ep, relay, nonce, direct_poll, st); break;
    case 7: launch_mid_d<7>(pl, out_slots, n_mles, n_flat, W, S0, i0, i1, ep, relay, nonce, direct_poll, st); break;
    default: launch_mid_d<8>(pl, out_slots, n_mles, n_flat, W, S0, i0, i1, ep, relay, nonce, direct_poll, st); break;
    }
}

bool tile_eligible(size_t n_mles, size_t pairs) {
    static int v = [] {
        const char* e = getenv("CENO_HIP_TILE");  // 0 restores the one-lane-per-pair kernel (A/B measurements)
        return e ? atoi(e) : 1;
    }();
    static size_t max_pairs = [] {
        const char* e = getenv("CENO_HIP_TILE_MAX_LOG");
        return (size_t)1 << (e ? atoi(e) : 16);
    }();
    return v != 0 && pairs <= max_pairs && n_mles <= 1024;
}
template <int D>
static void launch_tile_d(const DevPlan& pl, int n_mles, int n_flat, size_t pairs, E2 r, const Epilogue& ep, hipStream_t st) {
    const int tp = tile_pairs((size_t)n_flat, (size_t)n_mles, pairs);
    size_t lds = ((size_t)n_mles * 2 * tp + (NT / 64) * D) * sizeof(E2) + 64;
    const int flat_in_lds = n_mles < 65536 && lds + (size_t)n_flat * 48 <= 60 * 1024;
    if (flat_in_lds) lds += (size_t)n_flat * 48;
    const size_t tiles = (pairs + tp - 1) / tp;
    hipLaunchKernelGGL((k_tile<D>), dim3((unsigned)std::min<size_t>(tiles, MAXB)), dim3(NT), lds, st, pl, n_mles, n_flat, tp, pairs, r, ep, flat_in_lds);
}
void launch_tile(int d, const DevPlan& pl, int n_mles, int n_flat, size_t pairs, E2 r, const Epilogue& ep, hipStream_t st) {
    switch (d) {
    case 1: launch_tile_d<1>(pl, n_mles, n_flat, pairs, r, ep, st); break;
    case 2: launch_tile_d<2>(pl, n_mles, n_flat, pairs, r, ep, st); break;
    case 3: launch_tile_d<3>(pl, n_mles, n_flat, pairs, r, ep, st); break;
    case 4: launch_tile_d<4>(pl, n_mles, n_flat, pairs, r, ep, st); break;
    case 5: launch_tile_d<5>(pl, n_mles, n_flat, pairs, r, ep, st); break;
    case 6: launch_tile_d<6>(pl, n_mles, n_flat, pairs, r, ep, st); break;
    case 7: launch_tile_d<7>(pl, n_mles, n_flat, pairs, r, ep, st); break;
    default: launch_tile_d<8>(pl, n_mles, n_flat, pairs, r, ep, st); break;
    }
}

