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
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ceno_prover.h"

int prover_set_error(int code, const char* msg);  // prover.cpp

extern "C" int ceno_prover_lanes_run(ceno_hip_ctx* ctx, int n_lanes, const ceno_lane_task* tasks, int n_tasks, int* out_status, int* out_lane) {
    if (!ctx || !tasks || n_lanes < 1 || n_lanes > 64 || n_tasks < 0) return prover_set_error(CENO_HIP_ERR_INVALID, "lanes_run: bad arguments");
    {
        // The command processor dispatches FOUR queues concurrently; further streams are time-multiplexed onto them, and a lane whose
        // stream shares a queue with another lane's persistent round kernel waits behind it (tools/ubench_lanes.hip,
        // profiles/r03_lane_launch_latency.json: launch + wait 116 us for 16 launches on 4 streams, 219 us on 8; eight chip-proof
        // lanes measured 10-20 % slower than four).  More lanes than that are therefore run as four; CENO_HIP_MAX_LANES overrides.
        const char* e = getenv("CENO_HIP_MAX_LANES");
        const int cap = e && atoi(e) > 0 ? atoi(e) : 4;
        n_lanes = std::min(n_lanes, cap);
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
    std::vector<std::thread> th;
    for (int l = 0; l < n_lanes; l++) th.emplace_back(worker, l);
    for (auto& t : th) t.join();
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

extern "C" int ceno_prover_create_chip_proofs(ceno_hip_ctx* ctx, const ceno_chip_task* tasks, int n_tasks, const uint64_t* challenges4,
                                              ceno_transcript* const* transcripts, int n_lanes, ceno_chip_proof* out_proofs, int* out_status) {
    if (!ctx || !tasks || !challenges4 || !transcripts || !out_proofs || n_tasks < 0 || n_lanes < 1)
        return prover_set_error(CENO_HIP_ERR_INVALID, "create_chip_proofs: bad arguments");
    std::vector<ChipJob> jobs(n_tasks);
    std::vector<ceno_lane_task> lt(n_tasks);
    for (int i = 0; i < n_tasks; i++) {
        if (!transcripts[i]) return prover_set_error(CENO_HIP_ERR_INVALID, "create_chip_proofs: NULL transcript");
        jobs[i] = ChipJob{ctx, &tasks[i], challenges4, transcripts[i], &out_proofs[i]};
        const ceno_chip_task& t = tasks[i];
        const size_t rows = (size_t)1 << (t.log2_num_instances + t.rotation_vars);
        const size_t n_rec = (size_t)t.num_reads + t.num_writes + (t.num_lk_tables > 0 ? 2 * (size_t)t.num_lk_tables : (size_t)t.num_lk);
        // records (16 B x rows each), towers ~ 2 x the interleaved records, sumcheck ping-pong ~ 0.75 x the tower's last layer
        lt[i] = ceno_lane_task{chip_job_fn, &jobs[i], (size_t)(16.0 * (double)rows * (double)(n_rec ? n_rec : 1) * 4.0)};
    }
    return ceno_prover_lanes_run(ctx, std::min(n_lanes, std::max(n_tasks, 1)), lt.data(), n_tasks, out_status, nullptr);
}
