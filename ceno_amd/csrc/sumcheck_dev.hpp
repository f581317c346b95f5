// Device-side pieces shared by the sumcheck round kernels (sumcheck.hip, sumcheck_gen.hip): launch geometry, the in-kernel
// message reduction ("last block done"), the pipelined challenge relay, slot tables.
#pragma once
#include "common.hpp"
#include "reduce.hpp"

using namespace gl;

static constexpr int NT = 256;
static constexpr int MAXD = 8;
static constexpr int MAXK = 4;
// 4 workgroups per CU: every workgroup of a launch is resident at once (1024 <= 256 CUs x 6 at 75 VGPRs);
// measured on MI355X: 1024/1280/1536 are within 1 %, 2048 (a second dispatch wave) is 3-6 % slower
static constexpr unsigned MAXB = 1024;
static constexpr int MAX_CLASSES = 40;

// ------------------------------------------------------------------------------------------------
// In-kernel message reduction ("last block done").  Every block publishes its D partial sums, the last
// block to arrive adds all of them, applies the class coefficient, chains the running total of the
// round (several size classes = several launches on one stream) and — for the last class of the round —
// adds the host-computed front-load scalars and writes the message either to device memory or straight
// into pinned host memory whose words the host watches (no D2H copy, no second launch, no flag).
// Cross-workgroup visibility follows the agent-scope release/acquire recipe (per-XCD L2s are not
// coherent): write-through (sc1) partial stores, drained (vmcnt(0)) before the agent-scope counter add;
// acquire fence + agent-scope (sc1) loads in the last block.  No release fence: it would flush the L2.
// ------------------------------------------------------------------------------------------------
struct Epilogue {
    uint64_t* partials;            // gridDim.x * D * 2 words
    unsigned* counter;             // arrival counter, zero when the kernel starts; reset by the last block
    E2* round_acc;                 // running total of this round's message (device, MAXD)
    uint64_t* out_msg;             // destination of the finished message (device or host-mapped), d * 2 words
    unsigned long long* flag;      // non-null: out_msg is host-mapped; the message words themselves tell the host it has arrived
    unsigned long long seq;
    E2 coeff;                      // class coefficient
    E2 scalars[MAXD];              // front-loaded terms, added once by the last class
    int first_class;               // 1: start the running total, 0: add to it
    int last_class;                // 1: finish the message
    int d;                         // message length (>= the D the kernel accumulates)
    // pipelined mode: the kernel was enqueued before its challenge existed and fetches it itself
    const struct Mailbox* mailbox; // host-mapped, written by the host
    struct Bcast* bcast;           // device memory, challenge relay between workgroups
    unsigned long long wait_seq;   // 0: challenge is the kernel argument; else it was relayed as round `wait_seq`
    unsigned long long next_seq;   // != 0: after publishing, fetch challenge `next_seq` from the host for the next launch
    int dbg;                       // 1: record wall-clock stamps per round in bcast->dbg (CENO_HIP_DEBUG); each stamp costs a
                                   // realtime read and a store on the round's critical path
    unsigned long long poll_ticks; // pipelined: how long (100 MHz ticks) a queued round waits for its challenge before it gives up
};

// host -> device mailbox in pinned memory (one cache line)
struct Mailbox {
    unsigned long long chal_seq;   // round whose challenge is valid (written last, release)
    unsigned long long chal[2];
    unsigned long long abort;      // non-zero: every waiting kernel exits without touching memory
};
// device-side relay: the first workgroup to arrive polls the host mailbox, the others poll this
struct Bcast {
    unsigned ticket;               // (unused)
    unsigned ready_seq;            // round whose challenge has been relayed (ABORT_SEQ: give up)
    unsigned long long chal[2];
    unsigned long long dbg[64][4]; // wall-clock stamps per round: start, before publish, after flag, after poll
};
static constexpr unsigned ABORT_SEQ = 0xFFFFFFFFu;
// what the host writes into the pinned message / evaluation words before a kernel is to fill them: >= p, so never a value
static constexpr uint64_t MSG_INVALID = ~0ull;

__device__ __forceinline__ void st_agent(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ld_agent(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// thread 0 of the finishing block: class coefficient, running round total, front-load scalars, publish
// Pipelined launches.  The finishing workgroup of round i — alone on the chip at that point — publishes
// the message, then polls the host mailbox for challenge i (bounded: CENO_HIP_PIPE_TIMEOUT_S, 60 s by default, or the host's
// `abort`) and relays it through device memory; the already queued kernel of round i+1 picks it up with a
// single load at its start.  Exactly one lane ever polls PCIe, nothing spins inside the big kernels.
// one lane polls the host's mailbox for challenge `want_seq` (bounded: ep.poll_ticks or the host's `abort`)
__device__ __forceinline__ bool poll_challenge(const Mailbox* mb, unsigned long long want_seq, unsigned long long& c0, unsigned long long& c1,
                                               unsigned long long poll_ticks) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    unsigned spins = 0;
    for (;;) {
        // relaxed polls: an acquire per poll would invalidate the (large) L2 every iteration
        if (__hip_atomic_load(&mb->chal_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == want_seq) break;
        if ((++spins & 63u) == 0) {
            if (__hip_atomic_load(&mb->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || wall_clock64() - t0 > poll_ticks) return false;
        }
    }
    // the host stores chal[] before chal_seq (release); these loads are issued only after the seq load
    // has returned (control dependency + waitcnt) and bypass the caches, so they see the new words
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    c0 = __hip_atomic_load(&mb->chal[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    c1 = __hip_atomic_load(&mb->chal[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
}
__device__ __forceinline__ void fetch_next_challenge(const Epilogue& ep) {
    Bcast* bc = ep.bcast;
    unsigned long long c0 = 0, c1 = 0;
    const bool ok = poll_challenge(ep.mailbox, ep.next_seq, c0, c1, ep.poll_ticks);
    if (ok) {
        __hip_atomic_store(&bc->chal[0], c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&bc->chal[1], c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_store(&bc->ready_seq, ok ? (unsigned)ep.next_seq : ABORT_SEQ, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// start of a pipelined round kernel: the challenge was relayed by the previous launch (kernel boundary
// = visibility); anything else means the pipeline was aborted
__device__ __forceinline__ bool read_challenge(const Epilogue& ep, E2& r, unsigned long long* s_c /* 3 words of LDS */) {
    // ONE lane per workgroup reads the relay words, LDS broadcast to the rest
    if (threadIdx.x == 0) {
        // plain (cacheable) loads: the words were written by the PREVIOUS launch, the kernel boundary makes
        // them visible; cache-bypassing (sc1) loads of one line from 2048 workgroups serialise at ~90 per us
        const volatile Bcast* bc = ep.bcast;
        s_c[2] = bc->ready_seq == (unsigned)ep.wait_seq;
        s_c[0] = bc->chal[0];
        s_c[1] = bc->chal[1];
    }
    __syncthreads();
    r = E2{s_c[0], s_c[1]};
    return s_c[2] != 0;
}

// `seq` / `next_seq` are passed apart from `ep` so that the persistent tail kernel can publish round after round from the
// kernel-argument copy of the epilogue: a modified local copy of the struct would live in scratch memory (its arrays are
// indexed at run time) and every field read on this single-lane critical path would become a scratch load.
// SCALARS = false: the caller is a persistent single-class kernel (k_mid, k_tail): no front-load scalars exist there, and not
// reading them keeps 32 SGPRs of kernel arguments out of the persistent loop (those kernels spilled 60-280 SGPRs into VGPR lanes).
template <int D, bool SCALARS = true>
__device__ __forceinline__ void finish_message(const E2 (&tot)[D], const Epilogue& ep, unsigned long long seq, unsigned long long next_seq) {
    const bool unit = (ep.coeff.c0 == 1 && ep.coeff.c1 == 0);
    if (ep.dbg && ep.bcast) ep.bcast->dbg[seq & 63][1] = wall_clock64();
    // one lane runs this on the critical path of every round: the D accumulated points are handled with STATIC indices (a
    // run-time index into the register array goes through scratch memory) and their independent loads / multiplies overlap;
    // points beyond D (a class of lower degree than the message) only carry the running total and the scalars
    auto emit = [&](int t, E2 v) {
        if (!ep.first_class) v = v + ep.round_acc[t];
        if (ep.last_class) {
            if (SCALARS) v = v + ep.scalars[t];
            if (ep.flag) {
                // ONE 16-byte write-through system-scope store per point (each such store is its own fabric transaction)
                typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                const u4 w = {(unsigned)v.c0, (unsigned)(v.c0 >> 32), (unsigned)v.c1, (unsigned)(v.c1 >> 32)};
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(ep.out_msg + 2 * t), "v"(w) : "memory");
            } else {
                ep.out_msg[2 * t] = v.c0;
                ep.out_msg[2 * t + 1] = v.c1;
            }
        } else {
            ep.round_acc[t] = v;
        }
    };
#pragma unroll
    for (int t = 0; t < D; t++)
        if (t < ep.d) emit(t, unit ? tot[t] : tot[t] * ep.coeff);
    for (int t = D; t < ep.d; t++) emit(t, e2_zero());
    // No flag and no wait behind the message: the host pre-fills the message words with a pattern that is not a canonical
    // field element (MSG_INVALID) and takes the message once every word has changed.  Each 16-byte store above is its own
    // fabric transaction and lands atomically per 8-byte word, their order does not matter, and the lane goes straight on to
    // fetch the next challenge (draining the stores and then storing a sequence flag cost ~1.2 us per round).
    if (ep.dbg && ep.bcast) ep.bcast->dbg[seq & 63][2] = wall_clock64();
    if (next_seq != 0) fetch_next_challenge(ep);
    if (ep.dbg && ep.bcast) ep.bcast->dbg[seq & 63][3] = wall_clock64();
}
template <int D>
__device__ __forceinline__ void finish_message(const E2 (&tot)[D], const Epilogue& ep) {
    finish_message<D>(tot, ep, ep.seq, ep.next_seq);
}

template <int D, int TNT>
__device__ __forceinline__ void epilogue(E2 (&acc)[D], const Epilogue& ep, E2* smem, int* s_flag) {
    int& s_is_last = *s_flag;
    red::block_sum<D, TNT>(acc, smem);
    if (gridDim.x == 1) {  // latency-critical tail rounds: nothing to exchange between workgroups
        if (threadIdx.x == 0) {
            finish_message<D>(acc, ep);
        }
        return;
    }
    if (threadIdx.x == 0) {
        uint64_t* row = ep.partials + (size_t)blockIdx.x * D * 2;
#pragma unroll
        for (int t = 0; t < D; t++) {
            st_agent(row + 2 * t, acc[t].c0);
            st_agent(row + 2 * t + 1, acc[t].c1);
        }
        // the partials were stored write-through (sc1): draining this wave's stores is enough, and a
        // release fence here would write back the whole XCD L2 (GBs of freshly folded table data) per block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(ep.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_is_last = (prev == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!s_is_last) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    E2 tot[D];
#pragma unroll
    for (int t = 0; t < D; t++) tot[t] = e2_zero();
    for (unsigned b = threadIdx.x; b < gridDim.x; b += TNT) {
        const uint64_t* row = ep.partials + (size_t)b * D * 2;
#pragma unroll
        for (int t = 0; t < D; t++) tot[t] = tot[t] + E2{ld_agent(row + 2 * t), ld_agent(row + 2 * t + 1)};
    }
    __syncthreads();  // smem is reused
    red::block_sum<D, TNT>(tot, smem);
    if (threadIdx.x == 0) {
        __hip_atomic_store(ep.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        finish_message<D>(tot, ep);
    }
}

__device__ __forceinline__ E2 ld_e2(const uint64_t* p) { return *reinterpret_cast<const E2*>(p); }
__device__ __forceinline__ void st_e2(uint64_t* p, E2 v) { *reinterpret_cast<E2*>(p) = v; }


struct MleSlot {
    const uint64_t* in;  // table of the previous round
    uint64_t* out;       // table of this round (ext), written by the fold
    int in_ext;
    int pad;
};

// ---- the generic CSR plan as the round kernels see it, and the per-factor helpers they share ----
struct DevPlan {
    const MleSlot* slots;
    int use_out;                    // 1: read slot.out (ext), 0: read slot.in (round 0)
    int n_groups;
    const uint32_t* group_term_off; // n_groups + 1 -> range in group_terms
    const uint32_t* group_terms;    // term ids
    const uint32_t* common_off;     // n_groups + 1 -> range in common_idx
    const uint32_t* common_idx;     // local mle ids
    const E2* coeffs;               // per term
    const uint32_t* term_off;       // per term -> range in term_idx
    const uint32_t* term_idx;       // local mle ids
};

__device__ __forceinline__ void load_pair(const MleSlot& sl, int use_out, size_t p, E2& lo, E2& hi) {
    if (use_out) {
        const E2* q = reinterpret_cast<const E2*>(sl.out) + 2 * p;
        lo = q[0];
        hi = q[1];
    } else if (sl.in_ext) {
        const E2* q = reinterpret_cast<const E2*>(sl.in) + 2 * p;
        lo = q[0];
        hi = q[1];
    } else {
        ulonglong2 v = *reinterpret_cast<const ulonglong2*>(sl.in + 2 * p);
        lo = E2{v.x, 0};
        hi = E2{v.y, 0};
    }
}

// pr[t] *= f(t) for the factor f(X) = x + (X - 1) * delta at the points 1..D.  `pr` starts as the term's coefficient c; from
// 3 points on the FIRST factor takes c along as c*f(X) = c*x + (X - 1) * c*delta — two multiplications instead of D.
template <int D>
__device__ __forceinline__ void mul_points(E2 (&pr)[D], bool& seeded, const E2& c, E2 x, E2 delta) {
    if (D >= 3 && !seeded) {
        x = c * x;
        delta = c * delta;
#pragma unroll
        for (int t = 0; t < D; t++) {
            pr[t] = x;
            x = x + delta;
        }
        seeded = true;
    } else {
#pragma unroll
        for (int t = 0; t < D; t++) {
            pr[t] = pr[t] * x;
            x = x + delta;
        }
    }
}

