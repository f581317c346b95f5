// A COHORT of tower-layer sumchecks in ONE launch (round 6).
//
// A shard's chip-proof phase is ~54 independent chains of dependent device <-> host round trips (one per sumcheck round of every tower
// layer of every chip: CpuTowerProver::create_proof, ceno_zkvm/src/scheme/cpu/mod.rs:346-554).  The command processor runs FOUR queues at
// a time (profiles/r03_lane_launch_latency.json) and a persistent round kernel that waits for its host holds its queue: more lanes than
// queues gain little (profiles/r06_shard_wide_lane_cap_sweep.jsonl: 28 ms on 4 lanes, 25 ms on 8 - 12).  What scales instead is one launch
// in which MANY chains wait side by side: layer r of a tower is 2^r entries per limb whatever the chip's size, so all chips that have a
// layer r run it in lock-step — one workgroup per chain (or per SUB-CUBE of a chain, below), each with its own pair of mailboxes, all of
// them resident at once, the host serving them round-robin with each chain's own transcript.
//
// One workgroup = one sumcheck of  sum_x eq(x, rt) [ sum_i alpha_i a_i(x) b_i(x) + sum_k (an_k (p1 q2 + p2 q1) + ad_k q1 q2) ]  over n <= 13
// variables, LSB first, the fold of round i fused with the evaluation of round i + 1 (the schedule of every other kernel of this library),
// final evaluations of all tables at the end.  The eq factor is out of the tables and of the evaluation points: the device reports q_i(1) and
// q_i's leading coefficient, the library's host side makes the round message p_i(1), p_i(2), p_i(3) of them from the running claim
// (cohort_message).  Small rounds (<= 64 pairs) walk the polynomial's bilinear terms lane-parallel, 4 or 8 lanes per pair.
// Layers of more entries than one workgroup should fold are split by their TOP index bits into sub-cubes — the trick the multi-rank sumcheck
// uses between GPUs, here between workgroups: the first rounds fold low variables, which never cross a sub-cube.  The
// sub-cubes of a layer are a GROUP: every member stores its message scaled by eq over the top variables, the member that arrives last adds
// them up and publishes ONE line to the host (a store into host memory is a PCIe write: ~10 M per second get through), all members poll
// ONE mailbox; the caller finishes the remaining rounds on the sub-cubes' final evaluations itself (prover_host_tower_rounds).
//
// Field arithmetic is exact: messages, challenges and evaluations equal the per-chip path's (and the oracle's) bit for bit.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "sumcheck_dev.hpp"

namespace {

// lanes per workgroup: 4 waves, one per SIMD.  A sub-cube's rounds are VALU work on ONE compute unit (a 2^13-entry sub-cube: ~350 us), so
// the callers cut a layer into as many sub-cubes as the device holds at once (four workgroups per compute unit: ~1000) — small workgroups
// make that many, and a sub-cube of <= 2^9 entries has at most one pair per lane in every round (the rest is round-trip latency).
constexpr int CNT = 256;
constexpr int COHORT_MAX_TABS = 14; // tables besides eq: 2 per product tower (<= 3), 4 per LogUp tower (<= 2)
constexpr int COHORT_SUB = 13;      // variables of a sub-cube

struct CohortJob {
    const E2* in[COHORT_MAX_TABS];  // this sub-cube's slice of every table: 2^n entries each
    E2* eq;                         // scratch: 2^n
    E2* ping;                       // scratch: K x 2^(n-1)
    E2* pong;                       // scratch: K x 2^(n-2)
    uint64_t* h_msg;                // pinned host memory: n rounds x 8 words (6 used), MSG_INVALID until written
    uint64_t* h_fin;                // pinned host memory: K x 2 words
    const Mailbox* box;             // device-memory mailbox the host writes the challenges into (large BAR)
    E2 rt[COHORT_SUB];              // the low n coordinates of the layer's point
    E2 a_prod[3], a_num[2], a_den[2];
    int n, np, nl, pad_;
    unsigned long long poll_ticks;
    // a GROUP of jobs (the sub-cubes of one layer) publishes ONE message per round: the sum of the jobs' messages, each scaled by its `scale`
    E2 scale;
    uint64_t* part;     // the group's partial messages: 8 words per job
    unsigned* counter;  // the group's arrival counter (monotonic: G arrivals per round)
    int G, g;           // jobs in the group (1: no group), this job's place in it
};

struct CTerm {
    E2 coef;
    int a, b;            // table numbers (1 ..: J.in[m - 1])
    int keep_a, keep_b;  // this term stores the folded table (each table has exactly one keeper)
};

__device__ __forceinline__ void put16(uint64_t* dst, E2 v) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 w = {(unsigned)v.c0, (unsigned)(v.c0 >> 32), (unsigned)v.c1, (unsigned)(v.c1 >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(w) : "memory");
}

// poll_challenge (sumcheck_dev.hpp) with a pause between polls: hundreds of workgroups poll at once here (`gap` units of ~0.2 us)
__device__ __forceinline__ bool poll_challenge_paced(const Mailbox* mb, unsigned long long want_seq, unsigned long long& c0, unsigned long long& c1,
                                                     unsigned long long poll_ticks, int gap) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    unsigned spins = 0;
    for (;;) {
        if (__hip_atomic_load(&mb->chal_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == want_seq) break;
        for (int k = 0; k < gap; k++) __builtin_amdgcn_s_sleep(8);
        if ((++spins & 63u) == 0) {
            if (__hip_atomic_load(&mb->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 || wall_clock64() - t0 > poll_ticks) return false;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    c0 = __hip_atomic_load(&mb->chal[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    c1 = __hip_atomic_load(&mb->chal[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
}

__global__ void __launch_bounds__(CNT) k_tower_cohort(const CohortJob* __restrict__ jobs) {
    __shared__ E2 smem[(CNT / 64) * 3];  // (two sums per wave; sized for three)
    __shared__ unsigned long long s_c[3];
    __shared__ int s_last;
    __shared__ CohortJob J;
    __shared__ CTerm s_terms[3 + 3 * 2];
    __shared__ int s_nterms;
    {   // the job record: one cooperative copy into LDS (its pointer arrays are indexed at run time)
        const uint64_t* src = reinterpret_cast<const uint64_t*>(jobs + blockIdx.x);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&J);
        for (int i = threadIdx.x; i < (int)(sizeof(CohortJob) / 8); i += CNT) dst[i] = src[i];
    }
    __syncthreads();
    const int n = J.n, np = J.np, nl = J.nl, K = 1 + 2 * np + 4 * nl;
    if (threadIdx.x == 0) {  // F's bilinear terms (the small rounds walk them lane-parallel); every table is kept by exactly one of them
        int u = 0, m = 1;
        for (int t = 0; t < np; t++, m += 2) s_terms[u++] = CTerm{J.a_prod[t], m, m + 1, 1, 1};
        for (int k = 0; k < nl; k++, m += 4) {
            s_terms[u++] = CTerm{J.a_num[k], m, m + 3, 1, 1};      // an p1 q2
            s_terms[u++] = CTerm{J.a_num[k], m + 1, m + 2, 1, 1};  // an p2 q1
            s_terms[u++] = CTerm{J.a_den[k], m + 2, m + 3, 0, 0};  // ad q1 q2
        }
        s_nterms = u;
    }
    __syncthreads();
    // ---- the eq factor, taken out of the tables (as in k_tower, sumcheck_tower.hpp): the round polynomial is
    //   p_i(X) = e_i * eq(X, rt_i) * q_i(X),   q_i(X) = sum_{x'} E_i[x'] F(r_0 .. r_{i-1}, X, x'),   E_i[x'] = eq(x', rt_{i+1 ..}),
    // q_i of degree 2: the workgroup reports q_i(1) and q_i's leading coefficient (two values, not three evaluations of a cubic; no eq table to
    // fold); the host knows e_i and the running claim p_i(0) + p_i(1) and makes p_i(1), p_i(2), p_i(3) of them (cohort_message below).
    // E_{n-1} = [1]; E_{i-1}[2 x + b] = eq(b, rt_i) E_i[x]; E_i lives at eq + 2^(n-1-i) - 1.
    E2* eq = J.eq;
    if (threadIdx.x == 0) eq[0] = e2_one();
    __syncthreads();
    for (int lvl = n - 1; lvl >= 1; lvl--) {
        const E2 rj = J.rt[lvl];
        const size_t half = (size_t)1 << (n - 1 - lvl);
        const E2* src = eq + (half - 1);
        E2* dst = eq + (2 * half - 1);
        for (size_t x = threadIdx.x; x < half; x += CNT) {
            const E2 v = src[x], hi = v * rj;
            dst[2 * x + 1] = hi;
            dst[2 * x] = v - hi;
        }
        __syncthreads();
    }
    E2 r = e2_zero();
    unsigned long long t_chal = wall_clock64();  // (lane 0: the start of round 0 = the launch)
    E2* cur = J.ping;   // tables of the round being evaluated (rounds >= 1): K x len, table m at cur + m * len (slot 0 unused)
    E2* prev = nullptr; // tables of the round before
    size_t prev_len = 0;
    for (int i = 0; i < n; i++) {
        const size_t pairs = (size_t)1 << (n - 1 - i), len = 2 * pairs;
        const E2Pre rp = e2_pre(r);
        const E2* W = eq + (pairs - 1);  // E_i
        E2 acc[2] = {e2_zero(), e2_zero()};
        // SMALL rounds (<= CNT / 4 pairs): one pair per lane would be a chain of ~36 dependent extension multiplications (~10 us) on a few
        // lanes while the rest of the workgroup idles.  F is a sum of BILINEAR terms coef * A(X) * B(X) (a product tower: alpha a b; a LogUp
        // tower: an p1 q2, an p2 q1, ad q1 q2): R = 4 or 8 lanes share a pair, lane `role` takes the terms role, role + R, ... — the same
        // instructions on different tables (no divergence), 8 multiplications per term; a table two terms read is folded twice and stored once.
        const int R = pairs * 8 <= (size_t)CNT ? 8 : (pairs * 4 <= (size_t)CNT ? 4 : 1);
        if (R > 1) {
            const size_t p = threadIdx.x / (unsigned)R;
            const int role = (int)(threadIdx.x % (unsigned)R);
            if (p < pairs) {
                E2 s1 = e2_zero(), c2 = e2_zero();
                for (int u = role; u < s_nterms; u += R) {
                    const CTerm T = s_terms[u];
                    auto fold = [&](int m, bool keep, E2& lo, E2& hi) {
                        if (i == 0) {
                            const E2* t = J.in[m - 1];
                            lo = t[2 * p];
                            hi = t[2 * p + 1];
                        } else {
                            const E2* q = (i == 1 ? J.in[m - 1] : prev + (size_t)m * prev_len) + 4 * p;
                            const E2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
                            lo = a0 + e2_mul_pre(rp, a1 - a0);
                            hi = a2 + e2_mul_pre(rp, a3 - a2);
                            if (keep) {
                                E2* o = cur + (size_t)m * len + 2 * p;
                                o[0] = lo;
                                o[1] = hi;
                            }
                        }
                    };
                    E2 alo, ahi, blo, bhi;
                    fold(T.a, T.keep_a != 0, alo, ahi);
                    fold(T.b, T.keep_b != 0, blo, bhi);
                    s1 = s1 + (T.coef * ahi) * bhi;
                    c2 = c2 + (T.coef * (ahi - alo)) * (bhi - blo);
                }
                const E2 w = W[p];
                acc[0] = w * s1;
                acc[1] = w * c2;
            }
        } else
        for (size_t p = threadIdx.x; p < pairs; p += CNT) {
            // (lo, hi) of table m at this pair: round 0 reads the inputs, later rounds fold the previous round's tables and keep the result
            auto load = [&](int m, E2& lo, E2& hi) {
                if (i == 0) {
                    const E2* t = J.in[m - 1];
                    lo = t[2 * p];
                    hi = t[2 * p + 1];
                } else {
                    const E2* q = (i == 1 ? J.in[m - 1] : prev + (size_t)m * prev_len) + 4 * p;
                    const E2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
                    lo = a0 + e2_mul_pre(rp, a1 - a0);
                    hi = a2 + e2_mul_pre(rp, a3 - a2);
                    E2* o = cur + (size_t)m * len + 2 * p;
                    o[0] = lo;
                    o[1] = hi;
                }
            };
            E2 s1 = e2_zero(), c2 = e2_zero();  // this pair's F at X = 1 and the coefficient of X^2 of F (products of the slopes)
            int m = 1;
            for (int t = 0; t < np; t++, m += 2) {
                E2 alo, ahi, blo, bhi;
                load(m, alo, ahi);
                load(m + 1, blo, bhi);
                const E2 al = J.a_prod[t];
                s1 = s1 + (al * ahi) * bhi;
                c2 = c2 + (al * (ahi - alo)) * (bhi - blo);
            }
            for (int k = 0; k < nl; k++, m += 4) {
                E2 p1l, p1, p2l, p2, q1l, q1, q2l, q2;
                load(m, p1l, p1);
                load(m + 1, p2l, p2);
                load(m + 2, q1l, q1);
                load(m + 3, q2l, q2);
                const E2 dp1 = p1 - p1l, dp2 = p2 - p2l, dq1 = q1 - q1l, dq2 = q2 - q2l;
                const E2 an = J.a_num[k], ad = J.a_den[k];
                s1 = s1 + an * (p1 * q2 + p2 * q1) + ad * (q1 * q2);
                c2 = c2 + an * (dp1 * dq2 + dp2 * dq1) + ad * (dq1 * dq2);
            }
            const E2 w = W[p];
            acc[0] = acc[0] + w * s1;
            acc[1] = acc[1] + w * c2;
        }
        red::block_sum<2, CNT>(acc, smem);
        // Every store into host memory is a PCIe write, and the device gets ~10 M of them through per second: a layer cut into hundreds of
        // sub-cubes that each sent their own message spent its rounds queueing there (measured: 320 jobs x 4 stores = ~125 us per round).
        // A group therefore adds its messages up on the device — every job stores its scaled share write-through, the job that arrives
        // last sums them (the pattern of `epilogue`, sumcheck_dev.hpp) — and the message leaves as ONE 64-byte write.
        bool publish = true;
        if (J.G > 1) {
            if (threadIdx.x == 0) {
                uint64_t* row = J.part + 8 * (size_t)J.g;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const E2 v = J.scale * acc[e];
                    st_agent(row + 2 * e, v.c0);
                    st_agent(row + 2 * e + 1, v.c1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (write-through stores: draining them is enough, see `epilogue`)
                const unsigned prev = __hip_atomic_fetch_add(J.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = prev == (unsigned)(J.G * (i + 1) - 1) ? 1 : 0;
            }
            __syncthreads();
            publish = s_last != 0;
            if (publish) {
                if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
                E2 tot[2] = {e2_zero(), e2_zero()};
                for (int b = threadIdx.x; b < J.G; b += CNT) {
                    const uint64_t* row = J.part + 8 * (size_t)b;
#pragma unroll
                    for (int e = 0; e < 2; e++) tot[e] = tot[e] + E2{ld_agent(row + 2 * e), ld_agent(row + 2 * e + 1)};
                }
                __syncthreads();  // smem is reused
                red::block_sum<2, CNT>(tot, smem);
#pragma unroll
                for (int e = 0; e < 2; e++) acc[e] = tot[e];
            }
        }
        if (publish && threadIdx.x < 4) {  // lanes 0 .. 3 of wave 0: one coalesced 64-byte store [q(1), leading coefficient, -, clocks]
            E2 v[4] = {acc[0], acc[1], e2_zero(), E2{(uint64_t)t_chal, (uint64_t)wall_clock64()}};  // (clocks: diagnostics, 100 MHz)
            E2 mine = v[0];
#pragma unroll
            for (int e = 1; e < 4; e++) {
                const E2 from0 = E2{(uint64_t)__shfl((unsigned long long)v[e].c0, 0), (uint64_t)__shfl((unsigned long long)v[e].c1, 0)};
                if ((int)threadIdx.x == e) mine = from0;
            }
            put16(J.h_msg + 8 * (size_t)i + 2 * threadIdx.x, mine);
        }
        if (threadIdx.x == 0) {
            unsigned long long c0 = 0, c1 = 0;
            const bool ok = poll_challenge_paced(J.box, (unsigned long long)(i + 1), c0, c1, J.poll_ticks, J.pad_);
            s_c[0] = c0;
            s_c[1] = c1;
            s_c[2] = ok ? 1ull : 0ull;
            t_chal = wall_clock64();
        }
        __syncthreads();  // (also: every lane's stores into `cur` are done before the next round reads them as `prev`)
        if (s_c[2] == 0) return;  // aborted / timed out: leave the evaluations unwritten
        r = E2{s_c[0], s_c[1]};
        if (i >= 1) {
            prev = cur;
            prev_len = len;
            cur = (cur == J.ping) ? J.pong : J.ping;
        }
        __syncthreads();  // s_c is rewritten by the next round
    }
    // ---- final evaluations: the last round's tables have two entries each ----
    if ((int)threadIdx.x >= 1 && (int)threadIdx.x < K) {  // (slot 0, the eq factor's value, is the host's: a product of n numbers it knows)
        const int m = threadIdx.x;
        const E2* t = n == 1 ? J.in[m - 1] : prev + (size_t)m * prev_len;
        put16(J.h_fin + 2 * (size_t)m, t[0] + r * (t[1] - t[0]));
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------
struct ceno_hip_cohort {
    ceno_hip_ctx* ctx = nullptr;
    hipStream_t st = nullptr;
    int n_jobs = 0;
    std::vector<int> n, K;
    void* d_scratch = nullptr;       // pool block: eq + ping + pong of every job
    void* d_group = nullptr;         // pool block: the groups' partial messages and arrival counters
    std::vector<int> leader;         // the job whose try_message yields the group's message
    std::vector<int> np, nl, group_size;
    // what the host keeps per leader to turn the device's (q_i(1), leading coefficient) into the round message p_i(1), p_i(2), p_i(3)
    struct Lead {
        int n = 0;
        E2 claim{0, 0};
        std::vector<E2> rt, inv1m, chal, e;  // e[i] = prod_{j < i} eq(r_j, rt_j)
        E2 q0{0, 0}, c1{0, 0}, c2{0, 0};     // q_{i-1}'s coefficients (with e_{i-1})
        std::vector<uint64_t> msgs;          // 6 words per finished round
        std::vector<char> have, have_chal;
    };
    std::vector<Lead> lead;                  // indexed by job; only leaders' entries are used
    std::vector<size_t> scratch_off; // a job's scratch inside d_scratch (extension elements)
    size_t jobs_off = 0;             // the job records inside the pinned block (bytes)
    bool launched = false;
    uint64_t* h_area = nullptr;      // pinned: per job [COHORT_SUB rounds x 8 words][16 x 2 words of evaluations]
    uint64_t* d_area = nullptr;      // its device view
    Mailbox* boxes = nullptr;        // device memory the host writes (large BAR): one 64-byte line per job
    bool own_boxes = false;
    size_t box_first = 0;            // its first line in the arena
};
namespace {
constexpr size_t COHORT_H_WORDS = 8 * COHORT_SUB + 2 * 16;
// host-writable device memory for the challenge mailboxes: one arena per context, grown on demand, kept for the context's life
struct BoxArena {
    void* p = nullptr;
    size_t lines = 0;
    std::map<size_t, size_t> live;  // first line -> lines of every cohort that holds some (a handful: first fit over the gaps)
    bool take(size_t n, size_t* first) {
        size_t at = 0;
        for (const auto& r : live) {
            if (r.first - at >= n) break;
            at = r.first + r.second;
        }
        if (at + n > lines) return false;
        live[at] = n;
        *first = at;
        return true;
    }
};
std::mutex g_box_mu;
std::map<ceno_hip_ctx*, BoxArena> g_box;
inline void host_fence() {
#if defined(__x86_64__)
    __builtin_ia32_sfence();
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
}  // namespace

extern "C" {

int ceno_hip_tower_cohort_max_vars(void) { return COHORT_SUB; }

int ceno_hip_tower_cohort_capacity(ceno_hip_ctx* ctx) {
    if (!ctx) return 0;
    static std::mutex mu;
    static std::map<int, int> cap;
    std::lock_guard<std::mutex> g(mu);
    auto it = cap.find(ctx->device);
    if (it != cap.end()) return it->second;
    int per_cu = 0, cus = 0, large_bar = 0;
    // (no host-writable device memory for the challenge mailboxes: no cohorts on this device — callers take the per-chip path)
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, ctx->device) != hipSuccess || !large_bar) return cap[ctx->device] = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tower_cohort, CNT, 0) != hipSuccess) per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess) cus = 0;
    // every workgroup of a launch waits for its host: a launch of more workgroups than the device holds at once would leave some undispatched
    // behind workgroups that wait for THEIR messages (a layer split into sub-cubes needs all of them) — callers keep a launch below this
    return cap[ctx->device] = std::max(0, per_cu) * std::max(0, cus);
}

// begin in three steps, so that the job records — the bulk of the work for a launch of ~1000 jobs — can be written by several host threads:
// open (shapes and groups: allocations, the scratch layout), set_job (one job's record; any thread, distinct jobs concurrently), launch
int ceno_hip_tower_cohort_open(ceno_hip_ctx* ctx, const ceno_hip_cohort_shape* shapes, int n_jobs, ceno_hip_stream s, ceno_hip_cohort** out) {
    CENO_TIMED("tower_cohort_open");
    CHECK_ARG(ctx, ctx && shapes && out && n_jobs >= 1 && n_jobs <= 4096, "tower_cohort_open: bad arguments");
    int large_bar = 0;
    if (hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, ctx->device) != hipSuccess || !large_bar)
        return ctx_fail(ctx, CENO_HIP_ERR_UNSUPPORTED, "tower cohort: the challenge mailboxes need host-writable device memory (large BAR)");
    hipStream_t st = ctx_stream(ctx, s);
    auto* c = new ceno_hip_cohort();
    c->ctx = ctx;
    c->st = st;
    c->n_jobs = n_jobs;
    c->n.resize((size_t)n_jobs);
    c->K.resize((size_t)n_jobs);
    c->leader.resize((size_t)n_jobs);
    c->np.resize((size_t)n_jobs);
    c->nl.resize((size_t)n_jobs);
    c->scratch_off.resize((size_t)n_jobs);
    c->group_size.assign((size_t)n_jobs, 1);
    c->lead.resize((size_t)n_jobs);
    size_t scratch_e2 = 0;
    for (int j = 0; j < n_jobs; j++) {
        const ceno_hip_cohort_shape& G = shapes[j];
        const int K = 1 + 2 * G.n_prod + 4 * G.n_logup;
        if (G.share_mailbox_of < 0 || G.share_mailbox_of > n_jobs || G.n < 1 || G.n > COHORT_SUB || G.n_prod < 0 || G.n_prod > 3 || G.n_logup < 0 || G.n_logup > 2 || K < 3) {
            delete c;
            return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "tower cohort: job %d: 1 .. %d variables, <= 3 product and <= 2 LogUp towers, at least one", j, COHORT_SUB);
        }
        // groups: consecutive jobs naming the same leader, the leader first
        const int l = G.share_mailbox_of > 0 ? G.share_mailbox_of - 1 : j;
        const bool ok = l <= j && (l == j || (shapes[l].share_mailbox_of == l + 1 && shapes[j - 1].share_mailbox_of == l + 1 && G.n == shapes[l].n));
        if (!ok) {
            delete c;
            return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "tower cohort: job %d: a group is consecutive jobs of one size naming their first job", j);
        }
        if (l != j) c->group_size[(size_t)l]++;
        c->n[(size_t)j] = G.n;
        c->K[(size_t)j] = K;
        c->np[(size_t)j] = G.n_prod;
        c->nl[(size_t)j] = G.n_logup;
        c->leader[(size_t)j] = l;
        c->scratch_off[(size_t)j] = scratch_e2;
        const size_t len = (size_t)1 << G.n;
        scratch_e2 += len + (size_t)K * (len / 2) + (size_t)K * std::max<size_t>(len / 4, 1);
    }
    int rc = ctx_alloc(ctx, scratch_e2 * sizeof(E2), &c->d_scratch);
    if (!rc) rc = ctx_alloc(ctx, (size_t)128 * n_jobs, &c->d_group);  // per job: 8 words of partial message, 8 words holding the group counter
    void *hb = nullptr, *db = nullptr;
    // pinned host memory: the message / evaluation slots of every job, then the job records — the workgroups read their record straight
    // from there (one 432-byte read each at their start: no upload, no wait before the launch)
    c->jobs_off = COHORT_H_WORDS * 8 * (size_t)n_jobs;
    if (!rc) rc = ctx_pinned_alloc(ctx, c->jobs_off + sizeof(CohortJob) * (size_t)n_jobs, &hb, &db);
    if (!rc) {
        std::lock_guard<std::mutex> g(g_box_mu);
        BoxArena& A = g_box[ctx];
        if (!A.p) {
            void* p = nullptr;
            const size_t lines = 8192;  // 512 KB of host-writable device memory per context, allocated once
            if (hipExtMallocWithFlags(&p, lines * 64, hipDeviceMallocFinegrained) != hipSuccess || !p) rc = ctx_fail(ctx, CENO_HIP_ERR_OOM, "tower cohort: mailbox arena");
            else {
                A.p = p;
                A.lines = lines;
            }
        }
        size_t first = 0;
        if (!rc && !A.take((size_t)n_jobs, &first)) rc = ctx_fail(ctx, CENO_HIP_ERR_OOM, "tower cohort: more than %zu mailboxes in flight on one context", A.lines);
        if (!rc) {
            c->boxes = reinterpret_cast<Mailbox*>((char*)A.p + 64 * first);
            c->box_first = first;
            c->own_boxes = true;
        }
    }
    if (rc) {
        if (hb) ctx_pinned_free(ctx, hb);
        if (c->d_group) ctx_free(ctx, c->d_group);
        if (c->d_scratch) ctx_free(ctx, c->d_scratch);
        delete c;
        return rc;
    }
    c->h_area = (uint64_t*)hb;
    c->d_area = (uint64_t*)db;
    *out = c;
    return 0;
}

int ceno_hip_tower_cohort_set_job(ceno_hip_cohort* c, int j, const ceno_hip_cohort_job* job) {
    if (!c || !job || j < 0 || j >= c->n_jobs || c->launched) return CENO_HIP_ERR_INVALID;
    const ceno_hip_cohort_job& G = *job;
    const int K = c->K[(size_t)j], leader = c->leader[(size_t)j];
    if (G.n != c->n[(size_t)j] || G.n_prod != c->np[(size_t)j] || G.n_logup != c->nl[(size_t)j] || !G.rt || !G.tables ||
        (G.share_mailbox_of > 0 ? G.share_mailbox_of - 1 : j) != leader || (G.n_prod && !G.alpha_prod) || (G.n_logup && (!G.alpha_num || !G.alpha_den)))
        return CENO_HIP_ERR_INVALID;
    if (leader == j) {
        if (!G.claim) return CENO_HIP_ERR_INVALID;
        ceno_hip_cohort::Lead& S = c->lead[(size_t)j];
        S.n = G.n;
        S.claim = E2{G.claim[0], G.claim[1]};
        S.rt.resize((size_t)G.n);
        S.inv1m.assign((size_t)G.n, e2_zero());
        S.chal.assign((size_t)G.n, e2_zero());
        S.e.assign((size_t)G.n + 1, e2_one());
        S.msgs.assign((size_t)6 * G.n, 0);
        S.have.assign((size_t)G.n, 0);
        S.have_chal.assign((size_t)G.n, 0);
        // 1 / (1 - rt_i) for every round, one inversion (prefix products); rt_i = 1 leaves q_i(0) undetermined by the claim
        std::vector<E2> pre((size_t)G.n, e2_one());
        E2 run = e2_one();
        for (int i = 0; i < G.n; i++) {
            S.rt[(size_t)i] = E2{G.rt[2 * i], G.rt[2 * i + 1]};
            const E2 v = e2_one() - S.rt[(size_t)i];
            if (v.c0 == 0 && v.c1 == 0) return CENO_HIP_ERR_UNSUPPORTED;
            pre[(size_t)i] = run;
            run = run * v;
        }
        E2 inv = e2_inv(run);
        for (int i = G.n - 1; i >= 0; i--) {
            S.inv1m[(size_t)i] = inv * pre[(size_t)i];
            inv = inv * (e2_one() - S.rt[(size_t)i]);
        }
    }
    static const unsigned long long ticks = [] {
        const char* e = getenv("CENO_HIP_PIPE_TIMEOUT_S");
        const double sec = e && atof(e) > 0 ? atof(e) : 60.0;
        return (unsigned long long)(sec * 1e8);
    }();
    static const int poll_gap = [] {
        const char* e = getenv("CENO_HIP_COHORT_POLL_GAP");
        return e ? atoi(e) : 0;
    }();
    {   // the words the device side writes: a leader's messages, every job's evaluations
        uint64_t* w = c->h_area + COHORT_H_WORDS * (size_t)j;
        if (leader == j)
            for (int i = 0; i < 8 * G.n; i++) w[i] = MSG_INVALID;
        w[8 * COHORT_SUB] = w[8 * COHORT_SUB + 1] = 0;  // (the eq factor's slot: filled by try_final)
        for (int i = 2; i < 2 * K; i++) w[8 * COHORT_SUB + i] = MSG_INVALID;
    }
    CohortJob& J = reinterpret_cast<CohortJob*>((char*)c->h_area + c->jobs_off)[j];
    memset(&J, 0, sizeof(J));
    const size_t len = (size_t)1 << G.n;
    for (int m = 0; m < K - 1; m++) J.in[m] = reinterpret_cast<const E2*>(G.tables[m]);
    E2* sp = (E2*)c->d_scratch + c->scratch_off[(size_t)j];
    J.eq = sp;
    sp += len;
    J.ping = sp;
    sp += (size_t)K * (len / 2);
    J.pong = sp;
    // a group (share_mailbox_of = leader + 1 on every member, the leader first): one mailbox, one message slot, one counter
    J.h_msg = c->d_area + COHORT_H_WORDS * (size_t)leader;
    J.h_fin = c->d_area + COHORT_H_WORDS * (size_t)j + 8 * COHORT_SUB;
    J.box = c->boxes + 2 * (size_t)leader;  // (Mailbox is 32 bytes: every job gets a 64-byte line of its own)
    J.G = c->group_size[(size_t)leader];
    J.g = j - leader;
    J.scale = G.scale ? E2{G.scale[0], G.scale[1]} : e2_one();
    J.part = (uint64_t*)c->d_group + 8 * (size_t)leader;
    J.counter = (unsigned*)((uint64_t*)c->d_group + 8 * (size_t)c->n_jobs + 8 * (size_t)leader);
    for (int v = 0; v < G.n; v++) J.rt[v] = E2{G.rt[2 * v], G.rt[2 * v + 1]};
    for (int t = 0; t < G.n_prod; t++) J.a_prod[t] = E2{G.alpha_prod[2 * t], G.alpha_prod[2 * t + 1]};
    for (int t = 0; t < G.n_logup; t++) {
        J.a_num[t] = E2{G.alpha_num[2 * t], G.alpha_num[2 * t + 1]};
        J.a_den[t] = E2{G.alpha_den[2 * t], G.alpha_den[2 * t + 1]};
    }
    J.n = G.n;
    J.np = G.n_prod;
    J.nl = G.n_logup;
    J.poll_ticks = ticks;
    J.pad_ = poll_gap;
    // the mailbox line: no challenge yet (a write through the BAR each: only the lines that are read — the leaders')
    if (leader == j) {
        volatile Mailbox* mb = c->boxes + 2 * (size_t)j;
        mb->chal_seq = 0;
        mb->abort = 0;
    }
    return 0;
}

static void cohort_release(ceno_hip_ctx* ctx, ceno_hip_cohort* c) {
    if (c->own_boxes) {
        std::lock_guard<std::mutex> g(g_box_mu);
        g_box[ctx].live.erase(c->box_first);
        c->own_boxes = false;
    }
    ctx_pinned_free(ctx, c->h_area);
    ctx_free_on(ctx, c->d_group, c->st);
    ctx_free_on(ctx, c->d_scratch, c->st);
    delete c;
}

int ceno_hip_tower_cohort_launch(ceno_hip_ctx* ctx, ceno_hip_cohort* c) {
    CENO_TIMED("tower_cohort_launch");
    CHECK_ARG(ctx, ctx && c && !c->launched, "tower_cohort_launch: bad arguments");
    host_fence();
    hipError_t e = hipMemsetAsync(c->d_group, 0, (size_t)128 * c->n_jobs, c->st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_tower_cohort, dim3((unsigned)c->n_jobs), dim3(CNT), 0, c->st, reinterpret_cast<const CohortJob*>((char*)c->d_area + c->jobs_off));
        e = hipGetLastError();
    }
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "tower cohort: %s", hipGetErrorString(e));
    c->launched = true;
    return 0;
}

int ceno_hip_tower_cohort_begin(ceno_hip_ctx* ctx, const ceno_hip_cohort_job* jobs, int n_jobs, ceno_hip_stream s, ceno_hip_cohort** out) {
    CENO_TIMED("tower_cohort_begin");
    CHECK_ARG(ctx, ctx && jobs && out && n_jobs >= 1 && n_jobs <= 4096, "tower_cohort_begin: bad arguments");
    std::vector<ceno_hip_cohort_shape> shapes((size_t)n_jobs);
    for (int j = 0; j < n_jobs; j++) shapes[(size_t)j] = ceno_hip_cohort_shape{jobs[j].n_prod, jobs[j].n_logup, jobs[j].n, jobs[j].share_mailbox_of};
    ceno_hip_cohort* c = nullptr;
    TRY(ceno_hip_tower_cohort_open(ctx, shapes.data(), n_jobs, s, &c));
    for (int j = 0; j < n_jobs; j++)
        if (const int rj = ceno_hip_tower_cohort_set_job(c, j, &jobs[j])) {
            cohort_release(ctx, c);
            if (rj == CENO_HIP_ERR_UNSUPPORTED) return ctx_fail(ctx, rj, "tower cohort: job %d: a coordinate of its point is 1 (the claim does not determine the round polynomials)", j);
            return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "tower cohort: job %d: NULL tables / point / alpha powers / claim", j);
        }
    const int rc = ceno_hip_tower_cohort_launch(ctx, c);
    if (rc) {
        cohort_release(ctx, c);
        return rc;
    }
    *out = c;
    return 0;
}

// the device's (q_i(1), leading coefficient of q_i) of round i -> p_i(1), p_i(2), p_i(3) with p_i(X) = e_i eq(X, rt_i) q_i(X): q_i(0) from the
// running claim p_i(0) + p_i(1) = (1 - rt_i) q_i(0) + rt_i q_i(1) (round 0: the layer's claim; later: p_{i-1}(r_{i-1})), as sc_tower_message
// (sumcheck.hip) does for the fused tower kernel
static void cohort_message(ceno_hip_cohort::Lead& S, int i, E2 dq1, E2 dc2, uint64_t* h) {
    const E2 rt = S.rt[(size_t)i];
    E2 claim = S.claim;
    if (i > 0) {
        const E2 rp = S.rt[(size_t)i - 1], r = S.chal[(size_t)i - 1];
        const E2 eq_r = e2_one() - rp - r + e2_mul_base(rp * r, 2);
        S.e[(size_t)i] = S.e[(size_t)i - 1] * eq_r;
        claim = eq_r * (S.q0 + r * (S.c1 + r * S.c2));
    }
    const E2 q1 = S.e[(size_t)i] * dq1, c2 = S.e[(size_t)i] * dc2;
    const E2 q0 = (claim - rt * q1) * S.inv1m[(size_t)i];
    const E2 c1 = q1 - q0 - c2;
    const E2 qa = q0 + e2_mul_base(c1, 2) + e2_mul_base(c2, 4), qb = q0 + e2_mul_base(c1, 3) + e2_mul_base(c2, 9);  // q_i(2), q_i(3)
    const E2 p1 = rt * q1;
    const E2 p2 = (e2_mul_base(rt, 3) - e2_one()) * qa;  // eq(2, rt) = 3 rt - 1
    const E2 p3 = (e2_mul_base(rt, 5) - E2{2, 0}) * qb;  // eq(3, rt) = 5 rt - 2
    h[0] = p1.c0; h[1] = p1.c1; h[2] = p2.c0; h[3] = p2.c1; h[4] = p3.c0; h[5] = p3.c1;
    S.q0 = q0;
    S.c1 = c1;
    S.c2 = c2;
}

int ceno_hip_tower_cohort_try_message(ceno_hip_cohort* c, int job, int round, uint64_t* out6) {
    if (!c || job < 0 || job >= c->n_jobs || round < 0 || round >= c->n[(size_t)job] || !out6) return CENO_HIP_ERR_INVALID;
    if (c->leader[(size_t)job] != job) return CENO_HIP_ERR_INVALID;  // a group's message is its leader's
    ceno_hip_cohort::Lead& S = c->lead[(size_t)job];
    if (!S.have[(size_t)round]) {
        if (round > 0 && (!S.have[(size_t)round - 1] || !S.have_chal[(size_t)round - 1])) return CENO_HIP_ERR_STATE;  // rounds are taken in order
        const volatile uint64_t* w = c->h_area + COHORT_H_WORDS * (size_t)job + 8 * (size_t)round;
        uint64_t v[4];
        for (int i = 0; i < 4; i++) {
            v[i] = w[i];
            if (v[i] >= gl::P) return 0;  // a word the device has not written yet (MSG_INVALID is no field element)
        }
        cohort_message(S, round, E2{v[0], v[1]}, E2{v[2], v[3]}, S.msgs.data() + 6 * (size_t)round);
        S.have[(size_t)round] = 1;
    }
    memcpy(out6, S.msgs.data() + 6 * (size_t)round, 48);
    return 1;
}

int ceno_hip_tower_cohort_round_times(ceno_hip_cohort* c, int job, int round, uint64_t* out2) {
    if (!c || job < 0 || job >= c->n_jobs || round < 0 || round >= c->n[(size_t)job] || !out2) return CENO_HIP_ERR_INVALID;
    const volatile uint64_t* w = c->h_area + COHORT_H_WORDS * (size_t)job + 8 * (size_t)round + 6;
    out2[0] = w[0];
    out2[1] = w[1];
    return 0;
}

int ceno_hip_tower_cohort_send_challenge(ceno_hip_cohort* c, int job, int round, const uint64_t* chal2) {
    if (!c || job < 0 || job >= c->n_jobs || round < 0 || round >= c->n[(size_t)job] || !chal2 || c->leader[(size_t)job] != job) return CENO_HIP_ERR_INVALID;
    c->lead[(size_t)job].chal[(size_t)round] = E2{chal2[0], chal2[1]};
    c->lead[(size_t)job].have_chal[(size_t)round] = 1;
    volatile Mailbox* mb = c->boxes + 2 * (size_t)job;
    mb->chal[0] = chal2[0];
    mb->chal[1] = chal2[1];
    host_fence();
    __atomic_store_n(&mb->chal_seq, (unsigned long long)(round + 1), __ATOMIC_RELEASE);
    host_fence();
    return 0;
}

int ceno_hip_tower_cohort_try_final(ceno_hip_cohort* c, int job, uint64_t* out_evals) {
    if (!c || job < 0 || job >= c->n_jobs || !out_evals) return CENO_HIP_ERR_INVALID;
    const volatile uint64_t* w = c->h_area + COHORT_H_WORDS * (size_t)job + 8 * COHORT_SUB;
    const int K = c->K[(size_t)job];
    for (int i = 2; i < 2 * K; i++)
        if (w[i] >= gl::P) return 0;
    // the eq factor at the point: e_n = prod_j eq(r_j, rt_j), the group's (every member folds the same low variables)
    ceno_hip_cohort::Lead& S = c->lead[(size_t)c->leader[(size_t)job]];
    const int n = S.n;
    if (n < 1 || !S.have_chal[(size_t)n - 1] || !S.have[(size_t)n - 1]) return CENO_HIP_ERR_STATE;
    const E2 rp = S.rt[(size_t)n - 1], r = S.chal[(size_t)n - 1];
    const E2 en = S.e[(size_t)n - 1] * (e2_one() - rp - r + e2_mul_base(rp * r, 2));
    S.e[(size_t)n] = en;
    out_evals[0] = en.c0;
    out_evals[1] = en.c1;
    for (int i = 2; i < 2 * K; i++) out_evals[i] = w[i];
    return 1;
}

int ceno_hip_tower_cohort_abort(ceno_hip_cohort* c) {
    if (!c) return CENO_HIP_ERR_INVALID;
    for (int j = 0; j < c->n_jobs; j++) {
        volatile Mailbox* mb = c->boxes + 2 * (size_t)j;
        mb->abort = 1;
    }
    host_fence();
    return 0;
}

int ceno_hip_tower_cohort_end(ceno_hip_ctx* ctx, ceno_hip_cohort* c) {
    if (!c) return 0;
    const hipError_t e = c->launched ? hipStreamSynchronize(c->st) : hipSuccess;  // every workgroup has left (all challenges answered, or aborted)
    cohort_release(ctx, c);
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "tower cohort: %s", hipGetErrorString(e));
    return 0;
}

}  // extern "C"
