// The pool's spin lock (plain C++: also built into the sanitizer drivers, tools/sanitize/).
#pragma once
#include <sched.h>

#include <atomic>

// The pool's lock.  Its critical sections are a map lookup and a vector pop — tens of nanoseconds — and a sumcheck takes it ~20 times
// (begin: a buffer pair per table, partials, counters, plan; free: the same again).  A contended std::mutex parks the thread in the
// kernel: with four lanes proving small chips side by side that measured ~5 us per acquisition, 110 us per tower layer instead of 10, and
// the lanes ran at half speed (tools/dev/lanes_sc.cpp, CENO_HIP_HOST_TIMING=1).  The lane threads are busy-polling threads anyway:
// they spin here too (and yield now and then: the rare holder that calls into the runtime — a stream query — keeps it longer).
struct PoolMutex {
    std::atomic<int> held{0};
    void lock() {
        for (int spins = 0;;) {
            if (!held.exchange(1, std::memory_order_acquire)) return;
            while (held.load(std::memory_order_relaxed)) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if (++spins >= 2048) {
                    spins = 0;
                    sched_yield();
                }
            }
        }
    }
    bool try_lock() { return !held.exchange(1, std::memory_order_acquire); }
    void unlock() { held.store(0, std::memory_order_release); }
};
