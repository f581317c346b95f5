// The chip-proof phase with the records, the towers and the MIDDLE tower layers of all chips proved together.
//
// A shard with the reference's population has ~54 circuits, and each of their tower proofs (CpuTowerProver::create_proof,
// ceno_zkvm/src/scheme/cpu/mod.rs:346-554) is a chain of ~20 layer sumchecks whose rounds each wait for a transcript challenge from the host.
// On scheduler lanes (lanes.cpp; the reference's ChipScheduler, scheme/scheduler.rs:231-336) every chip's chain occupies a stream, the device
// runs four queues at a time, and the layers of 2^6 .. 2^18 entries — too large for the host, far too small to fill the device — make up most
// of a chip's time: the phase plateaus at ~25 ms however many lanes there are (profiles/r06_shard_wide_lane_cap_sweep.jsonl).  Here, on ONE
// pool of host threads (worker_pool.hpp):
//   A1  every chip's checks and record plan                                                                     all threads
//   A2  the towers of all chips straight from their record expressions, level-synchronous launches, tops in one copy      thread 0
//   A3  per chip: out-evaluations into its transcript, the tower prover's state, the layers the host proves (1 .. 5)       all threads
//   B   layers 6 .. 19 in COHORTS (csrc/tower_cohort.hip): one launch per layer for all chips — a workgroup per chip, or per sub-cube of a
//       layer cut as finely as the device holds at once, the sub-cubes' messages added up on the device — on a schedule known up front:
//       thread 0 coordinates (opens launch k + 1 and closes k - 1 while k is served), the others each answer their chips' rounds as the
//       messages arrive, with the chip's own transcript; the last rounds of a cut layer run on the host over the sub-cubes' evaluations
//   (then, lanes.cpp)  per chip on the lanes: the larger layers on the device-wide kernels, the main point, the rotation argument
// Every chip keeps its own forked transcript (prover.rs:556-570), so the proofs are the words the per-chip path writes
// (tests/test_gpu_shard_wide.py::test_cohort_layers_write_the_same_proofs).  If phase B fails — a launch that cannot be opened, rounds that do
// not arrive in time — every chip goes back to where it stood before it (state and transcript) and the lanes prove the rest.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ceno_prover.h"
#include "../csrc/gl64.hpp"
#include "chip_run.hpp"
#include "worker_pool.hpp"

using gl::E2;

int prover_set_error(int code, const char* msg);  // prover.cpp
int prover_tower_host_layers();                   // prover.cpp
void prover_host_tower_rounds(int n, std::vector<std::vector<E2>>& tabs, int n_prod_active, int n_logup_active, const std::vector<E2>& alpha_prod,
                              const std::vector<E2>& alpha_num, const std::vector<E2>& alpha_den, ceno_transcript* tr, uint64_t* msgs, uint64_t* chal,
                              uint64_t* fin);  // prover.cpp

namespace {

double now_ms();
struct SpinBarrier {
    std::atomic<int> arrived{0}, gen{0};
    int n = 1;
    void wait(int who = -1, int site = 0) {
        const int g = gen.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            arrived.store(0, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_release);
            return;
        }
        int spins = 0;
        double t0 = 0;
        bool said = false;
        while (gen.load(std::memory_order_acquire) == g)
            if (++spins > 2000) {
                sched_yield();
                if ((spins & 0xFFFF) == 0) {  // (a thread that never arrives is a bug of this file: say where the others stand)
                    if (t0 == 0) t0 = now_ms();
                    else if (!said && now_ms() - t0 > 20000.0) {
                        fprintf(stderr, "[ceno_prover] cohort: thread %d has waited 20 s at barrier site %d (generation %d, %d of %d arrived)\n", who, site, g,
                                arrived.load(), n);
                        said = true;
                    }
                }
            }
    }
};

// one chip's share of one cohort launch
struct LayerChip {
    ChipProofRun* run = nullptr;
    int L = 0, n_lo = 0, G = 1, first_job = 0, np_act = 0, nl_act = 0, K = 0;
    std::vector<uint64_t> a_prod, a_num, a_den;           // the active towers' alpha powers (words)
    std::vector<const uint64_t*> tables;                  // [job][K - 1]
    std::vector<E2> eq_hi;                                // eq(g; rt[n_lo ..]) over the sub-cubes (G > 1)
    // serving state
    bool started = false, done = false;
    int round = 0, n_got = 0;
    std::vector<char> got;
    std::vector<uint64_t> chal, fin, finbuf;
};

// How a layer of 2^L entries is cut for n_chips chips: a sub-cube's rounds are VALU work on one compute unit, so the cut is as fine as the
// device can hold at once (every sub-cube of a layer must be resident: the chip's next challenge waits for all of them) — but not below
// 2^9 entries (a workgroup of 256 lanes has at most one pair per lane there: what remains is round-trip latency, which finer cuts only
// lengthen by host rounds), nor into more than 32 sub-cubes (the host adds their partial messages every round).
int sub_cube_vars(int L, int n_chips, int capacity, int sub_max) {
    const char* e = getenv("CENO_TOWER_COHORT_SUB");  // (A/B and tests: a fixed sub-cube size)
    const int fixed = e ? atoi(e) : 0;
    if (fixed > 0) return std::min(L, std::min(fixed, sub_max));
    int n = std::min(L, 9);
    while (n < std::min(L, sub_max) && ((L - n) > 5 || (long)n_chips << (L - n) > capacity)) n++;
    return n;
}

void prepare(LayerChip& c, ChipProofRun* run, int L, int n_lo) {
    c.run = run;
    c.L = L;
    c.n_lo = std::min(L, n_lo);
    c.G = 1 << (L - c.n_lo);
    TowerProveState& st = run->st;
    c.a_prod.clear();
    c.a_num.clear();
    c.a_den.clear();
    std::vector<const ceno_hip_tower*> act_p, act_l;
    for (int i = 0; i < st.n_prod; i++)
        if (st.nv_of(st.prod[i]) > L) {
            act_p.push_back(st.prod[i]);
            c.a_prod.insert(c.a_prod.end(), {st.alpha[2 * i], st.alpha[2 * i + 1]});
        }
    for (int i = 0; i < st.n_logup; i++)
        if (st.nv_of(st.logup[i]) > L) {
            act_l.push_back(st.logup[i]);
            const size_t k = (size_t)st.n_prod + 2 * i;
            c.a_num.insert(c.a_num.end(), {st.alpha[2 * k], st.alpha[2 * k + 1]});
            c.a_den.insert(c.a_den.end(), {st.alpha[2 * k + 2], st.alpha[2 * k + 3]});
        }
    c.np_act = (int)act_p.size();
    c.nl_act = (int)act_l.size();
    c.K = 1 + 2 * c.np_act + 4 * c.nl_act;
    std::vector<const uint64_t*> base;
    for (auto* t : act_p)
        for (int b = 0; b < 2; b++) base.push_back(ceno_hip_tower_layer_ptr(t, L, b));
    for (auto* t : act_l)
        for (int b = 0; b < 4; b++) base.push_back(ceno_hip_tower_layer_ptr(t, L, b));
    c.tables.resize((size_t)c.G * base.size());
    for (int g = 0; g < c.G; g++) {
        const size_t off = (size_t)2 * ((size_t)g << c.n_lo);  // words: sub-cube g = the entries whose top index bits are g
        for (size_t m = 0; m < base.size(); m++) c.tables[(size_t)g * base.size() + m] = base[m] + off;
    }
    c.eq_hi.assign((size_t)c.G, gl::e2_one());
    for (int j = 0; j < L - c.n_lo; j++) {  // variable n_lo + j of the layer is bit j of g
        const E2 rj{st.out_rt[2 * (size_t)(c.n_lo + j)], st.out_rt[2 * (size_t)(c.n_lo + j) + 1]};
        for (size_t x = 0; x < ((size_t)1 << j); x++) {
            const E2 hi = c.eq_hi[x] * rj;
            c.eq_hi[x + ((size_t)1 << j)] = hi;
            c.eq_hi[x] = c.eq_hi[x] - hi;
        }
    }
    c.started = c.done = false;
    c.round = c.n_got = 0;
    c.got.assign((size_t)c.G, 0);
    c.chal.assign((size_t)2 * L, 0);
    c.fin.assign((size_t)2 * c.K, 0);
    c.finbuf.assign((size_t)2 * c.K * c.G, 0);
}

// answer what has arrived for this chip; 0 = nothing new or progress, < 0 = error.  Never blocks.
int serve(ceno_hip_cohort* co, LayerChip& c) {
    TowerProveState& st = c.run->st;
    if (!c.started) {  // IOPProverState::prove's header: the number of variables and the degree
        prover_tr_usize(st.tr, (uint64_t)c.L);
        prover_tr_usize(st.tr, 3);
        c.started = true;
    }
    if (c.round < c.n_lo) {
        uint64_t* msg = st.out->msgs + st.msg_off + 6 * (size_t)c.round;
        const int got = ceno_hip_tower_cohort_try_message(co, c.first_job, c.round, msg);  // (the group's message: the sub-cubes' sum)
        if (got < 0) return got;
        if (got == 0) return 0;
        const E2 ch = prover_tr_round(st.tr, msg);
        const uint64_t w[2] = {ch.c0, ch.c1};
        c.chal[2 * (size_t)c.round] = w[0];
        c.chal[2 * (size_t)c.round + 1] = w[1];
        if (int r = ceno_hip_tower_cohort_send_challenge(co, c.first_job, c.round, w)) return r;
        c.round++;
        return 0;
    }
    // every challenge of the launch is out: the sub-cubes' evaluations at the point
    for (int g = 0; g < c.G; g++) {
        if (c.got[(size_t)g]) continue;
        const int r = ceno_hip_tower_cohort_try_final(co, c.first_job + g, c.finbuf.data() + (size_t)2 * c.K * g);
        if (r < 0) return r;
        if (r == 1) {
            c.got[(size_t)g] = 1;
            c.n_got++;
        }
    }
    if (c.n_got < c.G) return 0;
    if (c.G == 1) c.fin = c.finbuf;
    else {
        // the tables over the top variables: entry g = sub-cube g's evaluation (eq: times eq over the top variables); the last L - n_lo
        // rounds on the host, as the row-sharded tower prover finishes its layers (dist_gkr.cpp)
        std::vector<std::vector<E2>> tabs((size_t)c.K, std::vector<E2>((size_t)c.G));
        for (int g = 0; g < c.G; g++)
            for (int m = 0; m < c.K; m++) {
                const uint64_t* w = c.finbuf.data() + (size_t)2 * c.K * g + 2 * (size_t)m;
                tabs[(size_t)m][(size_t)g] = m == 0 ? c.eq_hi[(size_t)g] * E2{w[0], w[1]} : E2{w[0], w[1]};
            }
        auto as_e2 = [](const std::vector<uint64_t>& v) {
            std::vector<E2> o(v.size() / 2);
            for (size_t i = 0; i < o.size(); i++) o[i] = E2{v[2 * i], v[2 * i + 1]};
            return o;
        };
        prover_host_tower_rounds(c.L - c.n_lo, tabs, c.np_act, c.nl_act, as_e2(c.a_prod), as_e2(c.a_num), as_e2(c.a_den), st.tr,
                                 st.out->msgs + st.msg_off + 6 * (size_t)c.n_lo, c.chal.data() + 2 * (size_t)c.n_lo, c.fin.data());
    }
    c.done = true;
    return tower_state_layer_epilogue(st, c.chal.data(), c.fin.data());
}

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// Phases A and B for all chips on one pool of `n_threads` threads (thread t drives lane stream t mod 8):
//   A1  every chip's checks and record plan (host work)                                         all threads, chips dealt round-robin
//   A2  the records of ALL chips in one launch, their towers in level-synchronous launches, the tops to the host in one copy    thread 0
//   A3  per chip: out-evaluations into its transcript, the tower prover's state, the layers the host proves    all threads
//   B   layers host_layers + 1 .. last_layer in cohorts                                                  all threads serve
// A chip whose status is set is skipped from then on; a failure sets the status of the chips it touches.  Returns the first error of
// a step that concerns all chips (0 when only single chips failed).
int cohort_chip_proofs(ceno_hip_ctx* ctx, const ceno_chip_task* tasks, const uint64_t* challenges4, ceno_transcript* const* transcripts,
                       ceno_chip_proof* out_proofs, std::vector<ChipProofRun*>& runs, std::vector<int>& status, int host_layers, int last_layer,
                       int n_threads, int host_layers_here, bool more_threads_than_cpus) {
    // host_layers: what the tower prover's state fetches of every tower (CENO_TOWER_HOST_LAYERS: the tops' one copy covers it);
    // host_layers_here <= host_layers: the layers this phase really leaves to the host — a cohort layer of 2^6 .. 2^8 entries costs ~0.15 ms for
    // all chips, the host ~0.1 ms of every serving thread's time per such layer
    const int sub = ceno_hip_tower_cohort_max_vars();
    int capacity = ceno_hip_tower_cohort_capacity(ctx);
    if (const char* e = getenv("CENO_TOWER_COHORT_CAPACITY"))  // (tests: several launches per layer)
        if (atoi(e) > 0) capacity = std::min(capacity, atoi(e));
    if (capacity < 1) return prover_set_error(CENO_HIP_ERR_UNSUPPORTED, "cohort: the device holds no cohort workgroup");
    n_threads = std::max(1, std::min<int>(n_threads, (int)runs.size()));
    std::vector<ceno_hip_stream> streams((size_t)std::min(n_threads, 8), nullptr);
    for (size_t l = 0; l < streams.size(); l++)
        if (int rc = ceno_hip_lane_stream(ctx, (int)l, &streams[l])) return prover_set_error(rc, ceno_hip_last_error(ctx));
    ceno_hip_stream stream = streams[0];
    static const bool trace = getenv("CENO_COHORT_TRACE") != nullptr;
    static const bool times = getenv("CENO_COHORT_TIMES") != nullptr;
    const char* e_fail = getenv("CENO_TOWER_COHORT_FAIL_AT");  // (tests: the cohort phase fails while launch k is served)
    const int fail_at = e_fail ? atoi(e_fail) : -1;
    // how long a launch may take before the phase gives up on cohorts (a healthy one takes under a millisecond; after a failure the lanes take
    // over): CENO_TOWER_COHORT_TIMEOUT_S, 15 s, never more than the device side's own bound CENO_HIP_PIPE_TIMEOUT_S
    static const double timeout_ms = [] {
        const char *e = getenv("CENO_HIP_PIPE_TIMEOUT_S"), *c = getenv("CENO_TOWER_COHORT_TIMEOUT_S");
        const double pipe = e && atof(e) > 0 ? atof(e) : 60.0, mine = c && atof(c) > 0 ? atof(c) : 15.0;
        return 1e3 * std::min(pipe, mine);
    }();
    SpinBarrier bar;
    bar.n = n_threads;
    std::atomic<int> err{0};
    std::vector<double> busy((size_t)n_threads, 0.0);  // (trace: time spent answering, per thread)
    std::string err_msg;
    double t_prep = 0, t_end = 0;  // (trace: the coordinator's time opening launches / closing them)
    // The schedule of phase B is known before it starts — which chips have a layer L is a matter of their towers' heights — so everything a launch
    // needs that does not depend on the transcript (allocations, shapes: `open`) is done by a COORDINATOR (thread 0, which serves no chips)
    // while the launch before it is being served, and so is the closing of the launch before that: between two launches the device waits for
    // the epilogues, the job records (written by the serving threads, in parallel) and the launch itself.
    struct Launch {
        int L = 0, n_lo = 0, jobs = 0;
        std::vector<LayerChip> chips;
        ceno_hip_cohort* co = nullptr;
    };
    std::vector<Launch> launches;
    std::atomic<bool> go{true};
    bool phase_b_reached = false;  // (an error before the cohort layers — records, towers — is not the cohorts')
    auto build_schedule = [&]() {
        int L0 = 0;
        for (size_t i = 0; i < runs.size(); i++)
            if (!status[i] && !runs[i]->st.done() && runs[i]->st.round <= last_layer) L0 = L0 ? std::min(L0, runs[i]->st.round) : runs[i]->st.round;
        if (!L0) return;
        for (int L = L0; L <= last_layer; L++) {
            std::vector<size_t> at;
            for (size_t i = 0; i < runs.size(); i++)
                if (!status[i] && !runs[i]->st.done() && runs[i]->st.round <= L && L <= runs[i]->st.R) at.push_back(i);
            size_t next = 0;
            while (next < at.size()) {
                const int n_lo = sub_cube_vars(L, (int)(at.size() - next), capacity, sub);
                const int G = 1 << (L - n_lo);
                if (G > capacity) return;  // (a layer no launch can hold: this one and the ones above are left to the per-chip prover)
                Launch la;
                la.L = L;
                la.n_lo = n_lo;
                while (next < at.size() && la.jobs + G <= capacity) {
                    la.chips.emplace_back();
                    LayerChip& c = la.chips.back();
                    c.run = runs[at[next++]];
                    c.L = L;
                    c.n_lo = n_lo;
                    c.first_job = la.jobs;
                    la.jobs += G;
                }
                launches.push_back(std::move(la));
            }
        }
    };
    // Where every chip stands before the first cohort layer — its tower prover's state and a copy of its transcript: if the cohort phase fails
    // (a launch that cannot be opened, rounds that do not arrive in time on a device somebody else is using, ...) the chips go back there and
    // their remaining layers are proved chip by chip on the lanes, as if the cohorts had not been tried.  Only for transcripts that can be
    // cloned and whose clones are released like the original (the library's own; a foreign transcript without `fork`: the failure stands).
    struct Snap {
        bool have = false;
        int round = 0;
        size_t msg_off = 0;
        E2 claim{0, 0};
        bool have_claim = false;
        std::vector<uint64_t> alpha, out_rt;
        ceno_transcript* tr = nullptr;
    };
    std::vector<Snap> snaps(runs.size());
    bool can_restore = true;
    auto take_snapshots = [&]() {
        for (size_t i = 0; i < runs.size(); i++) {
            if (status[i] || runs[i]->st.done()) continue;
            TowerProveState& st = runs[i]->st;
            Snap& sn = snaps[i];
            ceno_transcript* t = st.tr;
            sn.tr = (t && t->fork && t->fork_free && t->destroy == t->fork_free) ? ceno_transcript_clone(t) : nullptr;
            if (!sn.tr) {
                can_restore = false;
                continue;
            }
            sn.have = true;
            sn.round = st.round;
            sn.msg_off = st.msg_off;
            sn.claim = st.claim;
            sn.have_claim = st.have_claim;
            sn.alpha = st.alpha;
            sn.out_rt = st.out_rt;
        }
    };
    auto drop_snapshots = [&]() {
        for (auto& sn : snaps)
            if (sn.tr) {
                ceno_transcript_free(sn.tr);
                sn.tr = nullptr;
            }
    };
    auto open_launch = [&](Launch& la) {
        const double t_in = trace ? now_ms() : 0;
        // the shapes (which towers still have this layer) and the allocations
        std::vector<ceno_hip_cohort_shape> shapes((size_t)la.jobs);
        for (auto& c : la.chips) {
            const TowerProveState& st = c.run->st;
            int np = 0, nl = 0;
            for (int k = 0; k < st.n_prod; k++) np += st.nv_of(st.prod[k]) > c.L;
            for (int k = 0; k < st.n_logup; k++) nl += st.nv_of(st.logup[k]) > c.L;
            const int G = 1 << (c.L - c.n_lo);
            for (int g = 0; g < G; g++) shapes[(size_t)(c.first_job + g)] = ceno_hip_cohort_shape{np, nl, c.n_lo, G > 1 ? c.first_job + 1 : 0};
        }
        if (const int rc = ceno_hip_tower_cohort_open(ctx, shapes.data(), la.jobs, stream, &la.co)) {
            int zero = 0;
            if (err.compare_exchange_strong(zero, rc)) err_msg = ceno_hip_last_error(ctx);
            la.co = nullptr;
        }
        if (trace) t_prep += now_ms() - t_in;
    };
    auto close_launch = [&](Launch& la, bool aborted) {
        if (!la.co) return;
        const double t_in = trace ? now_ms() : 0;
        if (aborted) (void)ceno_hip_tower_cohort_abort(la.co);
        const int rc = ceno_hip_tower_cohort_end(ctx, la.co);
        la.co = nullptr;
        if (rc && !aborted) {
            int zero = 0;
            if (err.compare_exchange_strong(zero, rc)) err_msg = ceno_hip_last_error(ctx);
        }
        if (trace) t_end += now_ms() - t_in;
    };
    // a chip's job records (its serving thread: the bulk of a launch's set-up, in parallel)
    auto set_jobs = [&](ceno_hip_cohort* co, LayerChip& c) -> int {
        for (int g = 0; g < c.G; g++) {
            ceno_hip_cohort_job J{};
            J.tables = c.tables.data() + (size_t)g * (size_t)(c.K - 1);
            J.n_prod = c.np_act;
            J.n_logup = c.nl_act;
            J.n = c.n_lo;
            J.rt = c.run->st.out_rt.data();
            J.alpha_prod = c.a_prod.data();
            J.alpha_num = c.a_num.data();
            J.alpha_den = c.a_den.data();
            // the sub-cubes of a layer are a group: one mailbox, one message per round (added up on the device, scaled by eq_hi)
            J.share_mailbox_of = c.G > 1 ? c.first_job + 1 : 0;
            J.scale = reinterpret_cast<const uint64_t*>(&c.eq_hi[(size_t)g]);
            J.claim = reinterpret_cast<const uint64_t*>(&c.run->st.claim);  // (the whole layer's: what its sumcheck proves)
            if (const int rc = ceno_hip_tower_cohort_set_job(co, c.first_job + g, &J)) return rc;
        }
        return 0;
    };
    double t_a1 = 0, t_a2 = 0, t_a3 = 0;
    std::vector<ceno_hip_wit_plan> plans(runs.size());
    const double t_start = now_ms();
    std::vector<double> t_started((size_t)n_threads, 0.0);
    auto worker = [&](int t) {
        if (trace) t_started[(size_t)t] = now_ms() - t_start;
        (void)ceno_hip_make_current(ctx);
        ceno_hip_stream mine = streams[(size_t)t % streams.size()];
        (void)ceno_hip_stream_bind(ctx, mine);  // (a pool thread may remember a stream of an earlier run)
        // ---- A1: the checks and the record plans (host work), then ONE launch for the records of all chips ----
        for (size_t i = (size_t)t; i < runs.size(); i += (size_t)n_threads)
            status[i] = chip_run_records_plan(*runs[i], ctx, &tasks[i], challenges4, transcripts[i], &out_proofs[i], &plans[i]);
        bar.wait(t, 1);
        // ---- A2 ----
        if (t == 0) {
            // the towers of all chips: straight from the record expressions (no record tables: the reference's ..._from_virtual_ext_batch), or —
            // a plan too large for that kernel's LDS stage, CENO_TOWER_VIRTUAL_RECORDS=0 — records first, then towers from them
            std::vector<ceno_hip_wit_plan> live;
            std::vector<int> first((size_t)runs.size() + 1, 0);
            std::vector<ceno_hip_virtual_tower_spec> vspecs;
            for (size_t i = 0; i < runs.size(); i++) {
                first[i] = (int)vspecs.size();
                if (status[i]) continue;
                ceno_hip_virtual_tower_spec s3[3];
                const int k = chip_run_virtual_tower_specs(*runs[i], (int)live.size(), s3);
                vspecs.insert(vspecs.end(), s3, s3 + k);
                live.push_back(plans[i]);
            }
            first[runs.size()] = (int)vspecs.size();
            std::vector<ceno_hip_tower*> towers(vspecs.size(), nullptr);
            int rc = 0;
            if (!vspecs.empty()) {
                const char* e_v = getenv("CENO_TOWER_VIRTUAL_RECORDS");
                rc = e_v && atoi(e_v) == 0 ? CENO_HIP_ERR_UNSUPPORTED : ceno_hip_tower_build_many_virtual(ctx, live.data(), (int)live.size(), vspecs.data(), (int)vspecs.size(), stream, towers.data());
                if (rc == CENO_HIP_ERR_UNSUPPORTED) {
                    rc = ceno_hip_wit_infer_many(ctx, live.data(), (int)live.size(), stream);
                    t_a1 = now_ms() - t_start;
                    std::vector<ceno_hip_tower_spec> specs;
                    if (!rc)
                        for (size_t i = 0; i < runs.size(); i++)
                            if (!status[i]) {
                                ceno_hip_tower_spec s3[3];
                                const int k = chip_run_tower_specs(*runs[i], s3);
                                specs.insert(specs.end(), s3, s3 + k);
                            }
                    if (!rc) rc = ceno_hip_tower_build_many(ctx, specs.data(), (int)specs.size(), stream, towers.data());
                } else
                    t_a1 = now_ms() - t_start;
                if (!rc) {
                    rc = ceno_hip_tower_prefetch_tops(ctx, towers.data(), (int)towers.size(), host_layers + 1, stream);
                    if (rc)
                        for (auto* tw : towers) ceno_hip_tower_free(ctx, tw);
                }
                if (rc) {
                    err_msg = ceno_hip_last_error(ctx);
                    err.store(rc);
                }
                for (size_t i = 0; i < runs.size(); i++) {
                    if (status[i]) continue;
                    if (rc) status[i] = rc;
                    else status[i] = chip_run_adopt_towers(*runs[i], towers.data() + first[i], first[i + 1] - first[i]);
                    chip_run_free_records(*runs[i]);
                }
            }
            t_a2 = now_ms() - t_start;
        }
        bar.wait(t, 2);
        if (err.load()) return;
        // ---- A3 ----
        for (size_t i = (size_t)t; i < runs.size(); i += (size_t)n_threads) {
            if (status[i]) {
                chip_run_abandon(*runs[i]);
                continue;
            }
            int rc = chip_run_after_towers(*runs[i], mine);
            // (the cohort kernel needs the claim of the layer it proves: known from the layer before — layer 1's is never formed, so with
            // CENO_TOWER_HOST_LAYERS=0 the first layer still goes the per-chip way)
            while (!rc && !runs[i]->st.done() && (runs[i]->st.round <= host_layers_here || !runs[i]->st.have_claim)) {
                rc = tower_state_step(runs[i]->st);
                if (rc) chip_run_abandon(*runs[i]);
            }
            status[i] = rc;
        }
        bar.wait(t, 3);
        if (t == 0) t_a3 = now_ms() - t_start;
        // ---- B ----
        if (t == 0) {
            build_schedule();
            phase_b_reached = true;
            if (!launches.empty()) {
                take_snapshots();
                open_launch(launches[0]);
            }
        }
        bar.wait(t, 4);
        const bool serves = n_threads == 1 || t > 0;
        const size_t n_serving = (size_t)std::max(1, n_threads - 1), me = n_threads == 1 ? 0 : (size_t)t - 1;
        auto fail_with = [&](int rc, const char* msg) {
            int zero = 0;
            if (err.compare_exchange_strong(zero, rc)) err_msg = msg;
        };
        for (size_t k = 0; k < launches.size(); k++) {
            Launch& la = launches[k];
            if (serves && !err.load())
                for (size_t i = me; i < la.chips.size(); i += n_serving) {
                    prepare(la.chips[i], la.chips[i].run, la.L, la.n_lo);
                    if (const int rc = set_jobs(la.co, la.chips[i]))
                        fail_with(rc, rc == CENO_HIP_ERR_UNSUPPORTED ? "cohort: a coordinate of a layer's point is 1" : "cohort: a job record was refused");
                }
            bar.wait(t, 5);
            const double t_begin = now_ms();
            if (t == 0) {
                if (!err.load())
                    if (const int rc = ceno_hip_tower_cohort_launch(ctx, la.co)) fail_with(rc, ceno_hip_last_error(ctx));
                go.store(err.load() == 0);  // (ONE decision per launch, taken between two barriers: every thread leaves the loop at the same place)
            }
            bar.wait(t, 6);
            if (!go.load()) break;
            auto serve_mine = [&]() {
                if (fail_at >= 0 && (size_t)fail_at == k && me == 0) fail_with(CENO_HIP_ERR_STATE, "CENO_TOWER_COHORT_FAIL_AT (a test's failure)");
                size_t open = 0;
                for (size_t i = me; i < la.chips.size(); i += n_serving) open++;
                unsigned spins = 0, idle = 0;
                while (open && !err.load(std::memory_order_relaxed)) {
                    bool moved = false;
                    for (size_t i = me; i < la.chips.size(); i += n_serving) {
                        LayerChip& c = la.chips[i];
                        if (c.done) continue;
                        const int round_before = c.round;
                        const double t_s = trace ? now_ms() : 0;
                        const int r = serve(la.co, c);
                        if (trace && (c.round != round_before || c.done)) busy[(size_t)t] += now_ms() - t_s;
                        if (r) {
                            fail_with(r, r == CENO_HIP_ERR_INVALID ? "cohort: a mailbox call was refused" : ceno_prover_last_error());
                            break;
                        }
                        if (c.done) open--;
                        moved = moved || c.done || c.round != round_before;
                    }
                    // (threads that poll without yielding starve each other where there are fewer CPUs than threads: a small box, a tight quota)
                    if (moved) idle = 0;
                    else if (more_threads_than_cpus && ++idle >= 64) {
                        sched_yield();
                        idle = 0;
                    }
                    if ((++spins & 1023) == 0 && now_ms() - t_begin > timeout_ms)
                        fail_with(CENO_HIP_ERR_STATE, "cohort: a tower layer's rounds did not arrive in time (CENO_TOWER_COHORT_TIMEOUT_S)");
                }
                // nobody answers this launch any more: release its workgroups now (the coordinator may be waiting for its stream)
                if (err.load()) (void)ceno_hip_tower_cohort_abort(la.co);
            };
            auto coordinate = [&]() {  // while launch k is served: open the next, close the one before (its wait covers launch k: same stream)
                if (k + 1 < launches.size() && !err.load()) open_launch(launches[k + 1]);
                if (k > 0) close_launch(launches[k - 1], false);
            };
            if (n_threads == 1) {
                serve_mine();
                coordinate();
            } else if (t == 0)
                coordinate();
            else
                serve_mine();
            bar.wait(t, 7);
            if (t == 0 && trace) {
                double mx = 0, sum = 0;
                for (double& b : busy) {
                    mx = std::max(mx, b);
                    sum += b;
                    b = 0;
                }
                fprintf(stderr, "[ceno_prover] cohort: layer %d, %zu chips, %d jobs of 2^%d (the device holds %d): %.3f ms from launch to the last epilogue (threads answering: busiest %.3f ms, all %.3f ms)\n",
                        la.L, la.chips.size(), la.jobs, la.n_lo, capacity, now_ms() - t_begin, mx, sum);
            }
            if (t == 0 && times && !err.load() && !la.chips.empty()) {
                // (CENO_COHORT_TIMES: the first and the last chip's first sub-cube, round by round: device time from challenge to message,
                // then how long the message waited for its answer)
                for (const LayerChip* c : {&la.chips.front(), &la.chips.back()}) {
                    std::string line;
                    uint64_t prev_sent = 0, first = 0;
                    for (int i = 0; i < c->n_lo; i++) {
                        uint64_t w[2] = {0, 0};
                        (void)ceno_hip_tower_cohort_round_times(la.co, c->first_job, i, w);
                        if (i == 0) first = w[0];
                        char buf[96];
                        snprintf(buf, sizeof buf, " [%d: wait %.1f, compute %.1f]", i, i ? (double)(w[0] - prev_sent) / 100.0 : 0.0, (double)(w[1] - w[0]) / 100.0);
                        line += buf;
                        prev_sent = w[1];
                    }
                    fprintf(stderr, "[ceno_prover] cohort times (us), layer %d job %d, %.1f us in all:%s\n", c->L, c->first_job, (double)(prev_sent - first) / 100.0, line.c_str());
                }
            }
        }
        bar.wait(t, 8);  // (every thread has left the loop: nobody reads a cohort any more)
        if (t == 0)
            for (auto& la : launches) close_launch(la, err.load() != 0);  // (the last one; after an error: whatever is open, its workgroups released)
    };
    WorkerPool::instance().run(n_threads, worker);
    if (trace) fprintf(stderr, "[ceno_prover] cohort: the last of %d threads started %.3f ms in\n", n_threads, *std::max_element(t_started.begin(), t_started.end()));
    if (trace) fprintf(stderr, "[ceno_prover] cohort: the coordinator spent %.3f ms opening launches, %.3f ms closing them (beside the serving threads)\n", t_prep, t_end);
    if (trace) fprintf(stderr, "[ceno_prover] chip proofs in cohorts: records %.3f ms, towers of all chips %.3f, to the cohort layers %.3f, cohort layers to %d %.3f\n", t_a1,
                       t_a2 - t_a1, t_a3 - t_a2, last_layer, now_ms() - t_start - t_a3);
    if (const int rc = err.load()) {
        if (phase_b_reached && can_restore) {
            // the cohort phase failed: every chip back to where it stood before it (state + transcript), the lanes prove the rest
            for (size_t i = 0; i < runs.size(); i++) {
                Snap& sn = snaps[i];
                if (!sn.have || status[i]) continue;
                TowerProveState& st = runs[i]->st;
                st.round = sn.round;
                st.msg_off = sn.msg_off;
                st.claim = sn.claim;
                st.have_claim = sn.have_claim;
                st.alpha = sn.alpha;
                st.out_rt = sn.out_rt;
                std::swap(st.tr->self, sn.tr->self);  // (the clone now holds the used-up state and is released with it)
            }
            drop_snapshots();
            fprintf(stderr, "[ceno_prover] WARNING: the cohort layers of the chip-proof phase failed (%s); the chips' tower proofs restart on the lanes\n", err_msg.c_str());
            return 0;
        }
        drop_snapshots();
        for (size_t i = 0; i < runs.size(); i++)
            if (!status[i] && !runs[i]->st.done()) status[i] = rc;  // (their transcripts may be mid-layer: these proofs are lost)
        return prover_set_error(rc, err_msg.c_str());
    }
    drop_snapshots();
    return 0;
}
