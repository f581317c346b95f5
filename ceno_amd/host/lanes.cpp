// Concurrent chip proving on lanes — a minimal C++ counterpart of the reference's chip scheduler
// (ceno_zkvm/src/scheme/scheduler.rs:231-336 worker loop, :342-347,:622-652 memory booking;
// docs/src/concurrent-chip-proving.md): tasks sorted by estimated device memory, one worker thread per lane, each lane
// bound to its own HIP stream (thread-bound streams: gkr_iop/src/gpu/mod.rs:87-154), greedy back-fill — a worker takes
// the LARGEST pending task whose estimate can be booked against the pool (ceno_hip_mem_book) and skips to smaller ones
// when it cannot; if nothing is in flight and nothing fits, the smallest pending task runs unbooked so that the batch
// always makes progress.  The task body is the caller's (a closure over the C ABI calls of one chip proof).
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ceno_prover.h"
#include "worker_pool.hpp"

int prover_set_error(int code, const char* msg);  // prover.cpp

// the lane streams belong to the CONTEXT: two runs on one context at the same time would interleave on the same four streams (the pool's
// tags and drain checks would see one owner, and one run's round kernels would queue behind the other's host-waiting kernels), so runs on
// one context take turns
static std::mutex g_runs_mu;
static std::map<ceno_hip_ctx*, std::shared_ptr<std::mutex>> g_runs;
extern "C" int ceno_prover_lanes_effective(int n_lanes) {
    const char* e = getenv("CENO_HIP_MAX_LANES");
    const int cap = e && atoi(e) > 0 ? atoi(e) : 4;
    return std::max(1, std::min(n_lanes, cap));
}
// the lanes a run of n_tasks tasks really gets (see the comment in ceno_prover_lanes_run)
int ceno_prover_lanes_effective_for(int n_lanes, int n_tasks) {
    const char* e = getenv("CENO_HIP_MAX_LANES");
    const int cap = e && atoi(e) > 0 ? atoi(e) : (n_tasks >= 24 ? 8 : 4);
    return std::max(1, std::min(n_lanes, cap));
}
static std::shared_ptr<std::mutex> run_mutex_of(ceno_hip_ctx* ctx) {
    std::lock_guard<std::mutex> g(g_runs_mu);
    auto& slot = g_runs[ctx];
    if (!slot) slot = std::make_shared<std::mutex>();
    return slot;
}
static int lanes_run_locked(ceno_hip_ctx* ctx, int n_lanes, const ceno_lane_task* tasks, int n_tasks, int* out_status, int* out_lane);
extern "C" int ceno_prover_lanes_run(ceno_hip_ctx* ctx, int n_lanes, const ceno_lane_task* tasks, int n_tasks, int* out_status, int* out_lane) {
    if (!ctx || !tasks || n_lanes < 1 || n_lanes > 64 || n_tasks < 0) return prover_set_error(CENO_HIP_ERR_INVALID, "lanes_run: bad arguments");
    std::shared_ptr<std::mutex> run_mu = run_mutex_of(ctx);
    std::lock_guard<std::mutex> one_run(*run_mu);
    return lanes_run_locked(ctx, n_lanes, tasks, n_tasks, out_status, out_lane);
}
static int lanes_run_locked(ceno_hip_ctx* ctx, int n_lanes, const ceno_lane_task* tasks, int n_tasks, int* out_status, int* out_lane) {
    {
        // The command processor dispatches FOUR queues concurrently; further streams are time-multiplexed onto them, and a lane whose
        // stream shares a queue with another lane's persistent round kernel waits behind it (tools/ubench_lanes.hip,
        // profiles/r03_lane_launch_latency.json: launch + wait 116 us for 16 launches on 4 streams, 219 us on 8; eight chip-proof
        // lanes measured 10-20 % slower than four on a shard of eight chips).  More lanes than that are therefore run as four —
        // unless the batch is MANY tasks (a shard with the reference's ~54 circuits): then the gaps one lane leaves on its queue
        // (host layers, set-up between launches) are worth a second lane per queue — 54 chip proofs: 29.4 ms on 4 lanes, 26.5 on 6,
        // 24.7-25.9 on 8-10, 25.5 on 12 (profiles/r06_shard_wide_lanes.jsonl).  CENO_HIP_MAX_LANES overrides.
        n_lanes = ceno_prover_lanes_effective_for(n_lanes, n_tasks);
    }
    std::vector<int> order(n_tasks);
    for (int i = 0; i < n_tasks; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tasks[a].estimated_bytes > tasks[b].estimated_bytes; });
    std::vector<char> taken(n_tasks, 0);
    std::mutex mu;
    std::condition_variable cv;
    int in_flight = 0, remaining = n_tasks, first_err = 0;
    // the lanes' streams belong to the context (created once, ~4 ms each; never per run)
    std::vector<ceno_hip_stream> streams(n_lanes, nullptr);
    for (int l = 0; l < n_lanes; l++) {
        int rc = ceno_hip_lane_stream(ctx, l, &streams[l]);
        if (rc) return prover_set_error(rc, ceno_hip_last_error(ctx));
    }
    const bool dbg = getenv("CENO_HIP_DEBUG") != nullptr;
    const bool trace = getenv("CENO_LANES_TRACE") != nullptr;  // one line per task: lane, estimate, start and end (ms since the run began)
    const auto t_run = std::chrono::steady_clock::now();
    auto ms_now = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_run).count(); };
    auto worker = [&](int lane) {
        // a fresh thread starts with device 0 current; allocations, stream creation and launches follow the current device
        (void)ceno_hip_make_current(ctx);
        if (dbg) {
            cpu_set_t set;
            CPU_ZERO(&set);
            sched_getaffinity(0, sizeof(set), &set);
            fprintf(stderr, "[ceno_prover] lane %d on cpu %d, affinity mask holds %d cpus\n", lane, sched_getcpu(), CPU_COUNT(&set));
        }
        for (;;) {
            int pick = -1;
            size_t booked = 0;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (remaining == 0) return;
                    // largest pending task that can be booked
                    for (int idx : order) {
                        if (taken[idx]) continue;
                        if (ceno_hip_mem_book(ctx, tasks[idx].estimated_bytes) == 0) {
                            pick = idx;
                            booked = tasks[idx].estimated_bytes;
                            break;
                        }
                    }
                    if (pick < 0 && in_flight == 0) {  // nothing fits and nobody will free memory: run the smallest anyway
                        for (auto it = order.rbegin(); it != order.rend(); ++it)
                            if (!taken[*it]) {
                                pick = *it;
                                break;
                            }
                    }
                    if (pick >= 0) break;
                    cv.wait(lk);  // a finishing task unbooks and wakes us
                }
                taken[pick] = 1;
                remaining--;
                in_flight++;
            }
            const double t_begin = trace ? ms_now() : 0.0;
            const int rc = tasks[pick].fn(tasks[pick].arg, lane, streams[lane]);
            (void)ceno_hip_stream_sync(ctx, streams[lane]);
            if (trace)
                fprintf(stderr, "[ceno_prover] lanes trace: task %d (%.1f MB booked) lane %d  %.3f -> %.3f ms\n", pick, (double)tasks[pick].estimated_bytes / 1e6,
                        lane, t_begin, ms_now());
            {
                std::lock_guard<std::mutex> lk(mu);
                if (booked) ceno_hip_mem_unbook(ctx, booked);
                if (out_status) out_status[pick] = rc;
                if (out_lane) out_lane[pick] = lane;
                if (rc && !first_err) first_err = rc;
                in_flight--;
            }
            cv.notify_all();
        }
    };
    WorkerPool::instance().run(n_lanes, worker);  // (threads kept between runs: starting eight costs ~0.2 ms)
    return first_err ? prover_set_error(first_err, "lanes_run: a task failed (see the per-task status)") : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The chip-proof phase of ZKVMProver::create_proof on the scheduler (ceno_zkvm/src/scheme/prover.rs:556-570: one forked
// transcript per chip, `ChipScheduler::execute` scheduler.rs:231-336, results in task order :303-304): every task is one
// ceno_prover_create_chip_proof on the lane that picked it.  The booking estimate of a task is what its proof allocates on top
// of its (borrowed) tables: records + three towers + sumcheck work buffers, all extension-field tables of its row count.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct ChipJob {
    ceno_hip_ctx* ctx;
    const ceno_chip_task* task;
    const uint64_t* challenges4;
    ceno_transcript* tr;
    ceno_chip_proof* out;
};
int chip_job_fn(void* arg, int lane, ceno_hip_stream stream) {
    (void)lane;
    auto* j = (ChipJob*)arg;
    return ceno_prover_create_chip_proof(j->ctx, j->task, j->challenges4, j->tr, stream, j->out);
}
}  // namespace

// Device bytes one chip proof allocates on top of its (borrowed) tables — what the scheduler books before it starts the task (the
// reference's estimator: ceno_zkvm/src/scheme/gpu/memory.rs:54-145, checked there against real usage; here
// tests/test_gpu_flows.py::test_chip_proof_booking_estimate_covers_the_pool_high_water_mark holds it between 1x and 2x the pool's
// high-water mark).  All tables are extension-field tables of `rows` entries: the records; per product tower the interleaved last
// layer of rows * next_pow2(k) entries and the layers above it (2x); the LogUp tower with numerators and denominators (4 limbs); the
// ping-pong buffers of the largest layer's sumcheck.
extern "C" size_t ceno_prover_chip_proof_estimate_bytes(const ceno_chip_task* t) {
    if (!t) return 0;
    auto np2 = [](size_t k) {
        size_t p = 1;
        while (p < k) p <<= 1;
        return p;
    };
    const double rows = (double)((size_t)1 << (t->log2_num_instances + t->rotation_vars));
    const size_t n_lk = t->num_lk_tables > 0 ? 2 * (size_t)t->num_lk_tables : (size_t)t->num_lk;
    const size_t n_rec = (size_t)t->num_reads + t->num_writes + n_lk;
    double bytes = 16.0 * rows * (double)(n_rec ? n_rec : 1);                      // records
    double last_layers = 0.0;
    if (t->num_reads) last_layers += 16.0 * rows * (double)np2(t->num_reads);       // product towers: last layer (2 limbs)
    if (t->num_writes) last_layers += 16.0 * rows * (double)np2(t->num_writes);
    if (n_lk) last_layers += 2.0 * 16.0 * rows * (double)np2(n_lk);                 // LogUp tower: 4 limbs
    bytes += 2.0 * last_layers;                                                     // all layers of the towers
    bytes += 0.75 * last_layers;                                                    // sumcheck ping (1/2) + pong (1/4) of the largest layer
    bytes += 32.0 * rows * (double)(t->n_rotation_pairs > 0 ? 2 * t->n_rotation_pairs + 2 : 0);  // rotated copies + selector of the rotation argument
    return (size_t)(bytes * 1.15) + ((size_t)4 << 20);                              // bucket rounding, small fixed blocks
}

// ---- the same phase with the middle tower layers of all chips proved together (cohort.cpp) ----
#include "chip_run.hpp"
int cohort_chip_proofs(ceno_hip_ctx* ctx, const ceno_chip_task* tasks, const uint64_t* challenges4, ceno_transcript* const* transcripts,
                       ceno_chip_proof* out_proofs, std::vector<ChipProofRun*>& runs, std::vector<int>& status, int host_layers, int last_layer,
                       int n_threads, int host_layers_here, bool more_threads_than_cpus);  // cohort.cpp
int prover_tower_host_layers();                                                                                                       // prover.cpp
namespace {
// CPUs this process may really use: the affinity mask, capped by the cgroup's CPU quota (a container on a 256-thread host with a quota of 16)
int cpu_budget() {
    static const int budget = [] {
        cpu_set_t set;
        CPU_ZERO(&set);
        int n = sched_getaffinity(0, sizeof(set), &set) == 0 ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            long period = 0;
            if (fscanf(f, "%31s %ld", q, &period) == 2 && period > 0 && q[0] != 'm') n = std::min<long>(n, std::max<long>(1, atol(q) / period));
            fclose(f);
        }
        return std::max(1, n);
    }();
    return budget;
}
struct PhaseJob {
    ChipProofRun* run;
    ceno_hip_ctx* ctx;
    const ceno_chip_task* task;
    const uint64_t* challenges4;
    ceno_transcript* tr;
    ceno_chip_proof* out;
    int* status;
    int host_layers;
};
int phase_c_fn(void* arg, int, ceno_hip_stream stream) {
    auto* j = (PhaseJob*)arg;
    if (*j->status) {  // lost in an earlier phase
        chip_run_abandon(*j->run);
        return *j->status;
    }
    return *j->status = chip_run_finish(*j->run, stream);
}
// the highest tower layer proved in cohorts: CENO_TOWER_COHORT_LAYERS (0 = the per-chip prover alone; default 19: past that the device-wide kernels of the per-chip prover do as well or better,
// profiles/r06_cohort_last_layer.txt)
int cohort_last_layer() {
    const char* e = getenv("CENO_TOWER_COHORT_LAYERS");
    const int v = e ? atoi(e) : 19;
    return std::max(0, std::min(v, 24));
}
}  // namespace

extern "C" int ceno_prover_create_chip_proofs(ceno_hip_ctx* ctx, const ceno_chip_task* tasks, int n_tasks, const uint64_t* challenges4,
                                              ceno_transcript* const* transcripts, int n_lanes, ceno_chip_proof* out_proofs, int* out_status) {
    if (!ctx || !tasks || !challenges4 || !transcripts || !out_proofs || n_tasks < 0 || n_lanes < 1)
        return prover_set_error(CENO_HIP_ERR_INVALID, "create_chip_proofs: bad arguments");
    for (int i = 0; i < n_tasks; i++)
        if (!transcripts[i]) return prover_set_error(CENO_HIP_ERR_INVALID, "create_chip_proofs: NULL transcript");
    const int lanes = std::min(n_lanes, std::max(n_tasks, 1));
    // Cohorts pay when there are many chips (fewer than the lanes: every chip has a stream to itself anyway), need every chip's towers
    // resident at once — booked here as one block, the per-chip path when that is refused — and a device whose memory the host can write.
    const int last_layer = cohort_last_layer(), host_layers = prover_tower_host_layers();
    size_t total = 0;
    for (int i = 0; i < n_tasks; i++) total += ceno_prover_chip_proof_estimate_bytes(&tasks[i]);
    // (CENO_TOWER_COHORT_MIN_TASKS: the fewest chips that go through cohorts — 8; a single chip measured in tools/dev/cohort_one_chip.py)
    const char* e_min = getenv("CENO_TOWER_COHORT_MIN_TASKS");
    const int min_tasks = e_min && atoi(e_min) > 0 ? atoi(e_min) : 8;
    const bool cohort = last_layer > host_layers && n_tasks >= min_tasks && ceno_hip_tower_cohort_capacity(ctx) >= 8 && ceno_hip_mem_book(ctx, total + ((size_t)1 << 30)) == 0;
    if (!cohort) {
        std::vector<ChipJob> jobs(n_tasks);
        std::vector<ceno_lane_task> lt(n_tasks);
        for (int i = 0; i < n_tasks; i++) {
            jobs[i] = ChipJob{ctx, &tasks[i], challenges4, transcripts[i], &out_proofs[i]};
            lt[i] = ceno_lane_task{chip_job_fn, &jobs[i], ceno_prover_chip_proof_estimate_bytes(&tasks[i])};
        }
        return ceno_prover_lanes_run(ctx, lanes, lt.data(), n_tasks, out_status, nullptr);
    }
    // (the lane streams are the context's: the whole phase is one run on it)
    std::shared_ptr<std::mutex> run_mu = run_mutex_of(ctx);
    std::lock_guard<std::mutex> one_run(*run_mu);
    std::vector<ChipProofRun> runs((size_t)n_tasks);
    std::vector<ChipProofRun*> run_ptrs((size_t)n_tasks);
    std::vector<int> status((size_t)n_tasks, 0);
    std::vector<PhaseJob> jobs((size_t)n_tasks);
    std::vector<ceno_lane_task> lt((size_t)n_tasks);
    for (int i = 0; i < n_tasks; i++) {
        run_ptrs[(size_t)i] = &runs[(size_t)i];
        jobs[(size_t)i] = PhaseJob{&runs[(size_t)i], ctx, &tasks[i], challenges4, transcripts[i], &out_proofs[i], &status[(size_t)i], host_layers};
        // (the block above is the booking; what a task "books" here only orders the tasks, largest first)
        lt[(size_t)i] = ceno_lane_task{phase_c_fn, &jobs[(size_t)i], ceno_prover_chip_proof_estimate_bytes(&tasks[i]) >> 20};
    }
    static const bool trace = getenv("CENO_COHORT_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    // (the serving threads wait for messages and hash transcripts — host work, not bounded by the device's four queues)
    const char* e_thr = getenv("CENO_COHORT_THREADS");
    const int n_threads = e_thr && atoi(e_thr) > 0 ? atoi(e_thr) : std::max(ceno_prover_lanes_effective_for(lanes, n_tasks), std::min(16, cpu_budget()));
    // (the layers the host proves inside the cohort phase: CENO_TOWER_COHORT_HOST_LAYERS, 5, at most the tower prover's own CENO_TOWER_HOST_LAYERS)
    const char* e_hl = getenv("CENO_TOWER_COHORT_HOST_LAYERS");
    const int host_layers_here = std::max(0, std::min(host_layers, e_hl ? atoi(e_hl) : 5));
    const int rc_b = cohort_chip_proofs(ctx, tasks, challenges4, transcripts, out_proofs, run_ptrs, status, host_layers, last_layer, n_threads, host_layers_here,
                                        n_threads > cpu_budget());
    const std::string msg_b = rc_b ? ceno_prover_last_error() : "";
    const double t_b = ms();
    (void)lanes_run_locked(ctx, lanes, lt.data(), n_tasks, nullptr, nullptr);
    if (trace) fprintf(stderr, "[ceno_prover] chip proofs in cohorts: to layer %d %.3f ms, the rest on lanes %.3f ms\n", last_layer, t_b, ms() - t_b);
    ceno_hip_mem_unbook(ctx, total + ((size_t)1 << 30));
    int first_err = 0;
    for (int i = 0; i < n_tasks; i++) {
        if (out_status) out_status[i] = status[(size_t)i];
        if (status[(size_t)i] && !first_err) first_err = status[(size_t)i];
    }
    if (rc_b) return prover_set_error(rc_b, msg_b.c_str());
    return first_err ? prover_set_error(first_err, "create_chip_proofs: a task failed (see the per-task status)") : 0;
}

// ZKVMProver::run_chip_proofs (ceno_zkvm/src/scheme/prover.rs:618-710) with the scheduler's forking (`ChipScheduler::execute`,
// scheduler.rs:231-336: one clone of the fork transcript per task): every task's transcript is a clone of `fork_parent` bound to the two
// global challenges and the task's words (task id, circuit index, instance counts: prover.rs:646-654) "in the same order as verifier";
// after the proofs one sample of every fork goes back to the caller, who merges them into the main transcript (prover.rs:567-570).
extern "C" int ceno_prover_run_chip_proofs(ceno_hip_ctx* ctx, const ceno_chip_task* tasks, int n_tasks, const uint64_t* challenges4, const ceno_transcript* fork_parent,
                                           const uint64_t* bind_words, const uint32_t* bind_offsets, int n_lanes, ceno_chip_proof* out_proofs,
                                           uint64_t* out_samples, int* out_status) {
    if (!ctx || !tasks || !challenges4 || !fork_parent || !bind_offsets || !out_proofs || !out_samples || n_tasks < 0 || n_lanes < 1)
        return prover_set_error(CENO_HIP_ERR_INVALID, "run_chip_proofs: bad arguments");
    std::vector<ceno_transcript*> forks((size_t)n_tasks, nullptr);
    auto free_forks = [&]() {
        for (auto* f : forks)
            if (f) ceno_transcript_free(f);
    };
    for (int i = 0; i < n_tasks; i++) {
        ceno_transcript* f = ceno_transcript_clone(fork_parent);
        if (!f) {
            free_forks();
            return prover_set_error(CENO_HIP_ERR_INVALID, "run_chip_proofs: the fork transcript cannot be cloned");
        }
        forks[(size_t)i] = f;
        f->append_ext(f->self, challenges4);
        f->append_ext(f->self, challenges4 + 2);
        for (uint32_t k = bind_offsets[i]; k < bind_offsets[i + 1]; k++) f->append_base(f->self, bind_words[k]);
    }
    const int rc = ceno_prover_create_chip_proofs(ctx, tasks, n_tasks, challenges4, forks.data(), n_lanes, out_proofs, out_status);
    for (int i = 0; i < n_tasks; i++) forks[(size_t)i]->sample_ext(forks[(size_t)i]->self, out_samples + 2 * (size_t)i);
    free_forks();
    return rc;
}
