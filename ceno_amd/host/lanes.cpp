// Concurrent chip proving on lanes — a minimal C++ counterpart of the reference's chip scheduler
// (ceno_zkvm/src/scheme/scheduler.rs:231-336 worker loop, :342-347,:622-652 memory booking;
// docs/src/concurrent-chip-proving.md): tasks sorted by estimated device memory, one worker thread per lane, each lane
// bound to its own HIP stream (thread-bound streams: gkr_iop/src/gpu/mod.rs:87-154), greedy back-fill — a worker takes
// the LARGEST pending task whose estimate can be booked against the pool (ceno_hip_mem_book) and skips to smaller ones
// when it cannot; if nothing is in flight and nothing fits, the smallest pending task runs unbooked so that the batch
// always makes progress.  The task body is the caller's (a closure over the C ABI calls of one chip proof).
#include <sched.h>

#include <algorithm>
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
    std::vector<int> order(n_tasks);
    for (int i = 0; i < n_tasks; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tasks[a].estimated_bytes > tasks[b].estimated_bytes; });
    std::vector<char> taken(n_tasks, 0);
    std::mutex mu;
    std::condition_variable cv;
    int in_flight = 0, remaining = n_tasks, first_err = 0;
    std::vector<ceno_hip_stream> streams(n_lanes, nullptr);
    for (int l = 0; l < n_lanes; l++) {
        int rc = ceno_hip_stream_create_lane(ctx, l, &streams[l]);
        if (rc) {
            for (int k = 0; k < l; k++) ceno_hip_stream_destroy(ctx, streams[k]);
            return prover_set_error(rc, ceno_hip_last_error(ctx));
        }
    }
    const bool dbg = getenv("CENO_HIP_DEBUG") != nullptr;
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
            const int rc = tasks[pick].fn(tasks[pick].arg, lane, streams[lane]);
            (void)ceno_hip_stream_sync(ctx, streams[lane]);
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
    for (auto st : streams) ceno_hip_stream_destroy(ctx, st);
    return first_err ? prover_set_error(first_err, "lanes_run: a task failed (see the per-task status)") : 0;
}
