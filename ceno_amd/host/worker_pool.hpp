// Host threads kept between runs.  A chip-proof phase needs sixteen threads for a few milliseconds; starting them costs 0.4 ms each time
// (measured: the last of 16 std::threads runs 0.43 ms after the first was created).  The pool parks its threads on a condition variable and
// hands run(n, fn) the caller's thread as worker 0 and n - 1 parked ones as workers 1 .. n - 1; a run that finds the pool busy starts its own threads.
// The threads are detached and the pool is never destroyed: nothing to join at process exit.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>

class WorkerPool {
public:
    static WorkerPool& instance() {
        // a forked child inherits the pool's bookkeeping but none of its threads: its handler only forgets the pool (no allocation, no lock between
        // fork and exec), the child's first run makes a new one
        static const int registered = pthread_atfork(nullptr, nullptr, [] { slot().store(nullptr, std::memory_order_relaxed); });
        (void)registered;
        WorkerPool* p = slot().load(std::memory_order_acquire);
        if (!p) {
            WorkerPool* fresh = new WorkerPool();  // (leaked on purpose; has no threads until its first run)
            if (slot().compare_exchange_strong(p, fresh, std::memory_order_acq_rel)) p = fresh;
            else delete fresh;
        }
        return *p;
    }
    // fn(0) on the calling thread, fn(1) .. fn(n - 1) on pool threads; returns when all have returned
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 1) {
            fn(0);
            return;
        }
        // (the pool serves one run at a time; a second run at the same moment — another context on another thread — starts threads of its own
        // rather than wait: its tasks may be what the first run's tasks wait for)
        std::unique_lock<std::mutex> one_run(run_mu_, std::try_to_lock);
        if (!one_run.owns_lock()) {
            std::vector<std::thread> th;
            for (int t = 1; t < n; t++) th.emplace_back([&fn, t] { fn(t); });
            fn(0);
            for (auto& x : th) x.join();
            return;
        }
        {
            std::unique_lock<std::mutex> lk(mu_);
            while ((int)n_threads_ < n - 1) {
                const int id = (int)n_threads_++;
                std::thread([this, id] { loop(id); }).detach();
            }
            fn_ = &fn;
            want_ = n - 1;
            left_ = n - 1;
            gen_++;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu_);
        while (!done_cv_.wait_for(lk, std::chrono::seconds(30), [&] { return left_ == 0; }))
            fprintf(stderr, "[ceno_prover] worker pool: a run of %d has waited 30 s for %d of its workers\n", n, left_);
        fn_ = nullptr;
    }

private:
    static std::atomic<WorkerPool*>& slot() {
        static std::atomic<WorkerPool*> s{nullptr};
        return s;
    }
    void loop(int id) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (id < want_) fn = fn_;
            }
            if (!fn) continue;
            (*fn)(id + 1);
            std::lock_guard<std::mutex> lk(mu_);
            if (--left_ == 0) done_cv_.notify_all();
        }
    }
    std::mutex run_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)>* fn_ = nullptr;
    size_t n_threads_ = 0;
    int want_ = 0, left_ = 0;
    unsigned long long gen_ = 0;
};
