// Sanitizer driver (CPU only; never linked into the product): the cross-rank shared-memory exchange of ceno_amd/host/dist.cpp and the pool's spin
// lock (ceno_amd/csrc/pool_mutex.hpp) under ThreadSanitizer / AddressSanitizer.
//   * `world` THREADS attach to one POSIX segment as ranks 0 .. world-1 (each with its own communicator: the segment does not care whether its
//     ranks are processes or threads) and run ceno_dist_shm_selftest — `iters` gathers of 1 .. 64 extension elements whose every word is checked —
//     concurrently: the release / acquire sequence words and the parity-indexed slots are what TSan looks at;
//   * `threads` threads take the PoolMutex around a plain (non-atomic) counter and a small vector: any hole in the lock is a reported race and a
//     wrong total.
// Built and run by `python -m ceno_amd.build --sanitize` (tests/test_sanitizers.py behind CENO_RUN_SANITIZERS=1).
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ceno_prover.h"
#include "../../ceno_amd/csrc/pool_mutex.hpp"

int main(int argc, char** argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 4, iters = argc > 2 ? atoi(argv[2]) : 20000, threads = argc > 3 ? atoi(argv[3]) : 8;
    int bad = 0;
    {
        const std::string name = "/ceno_san_" + std::to_string((long)getpid());
        std::vector<ceno_dist_comm*> comms((size_t)world, nullptr);
        if (ceno_dist_comm_attach_shm(&comms[0], world, 0, name.c_str(), 1) != 0) {
            fprintf(stderr, "attach (create) failed: %s\n", ceno_dist_last_error());
            return 2;
        }
        for (int r = 1; r < world; r++)
            if (ceno_dist_comm_attach_shm(&comms[(size_t)r], world, r, name.c_str(), 0) != 0) {
                fprintf(stderr, "attach rank %d failed: %s\n", r, ceno_dist_last_error());
                return 2;
            }
        ceno_dist_shm_unlink(name.c_str());
        std::vector<int> rcs((size_t)world, -1);
        std::vector<std::thread> ths;
        for (int r = 0; r < world; r++) ths.emplace_back([&, r] { rcs[(size_t)r] = ceno_dist_shm_selftest(comms[(size_t)r], iters); });
        for (auto& t : ths) t.join();
        for (int r = 0; r < world; r++) {
            if (rcs[(size_t)r] != 0) {
                fprintf(stderr, "rank %d: selftest failed (%d): %s\n", r, rcs[(size_t)r], ceno_dist_last_error());
                bad++;
            }
            ceno_dist_comm_destroy(comms[(size_t)r]);
        }
        printf("shm exchange: %d ranks x %d gathers: %s\n", world, iters, bad ? "FAILED" : "ok");
    }
    {
        PoolMutex mu;
        long counter = 0;
        std::vector<int> bag;
        const int per = 200000;
        std::vector<std::thread> ths;
        for (int t = 0; t < threads; t++)
            ths.emplace_back([&] {
                for (int i = 0; i < per; i++) {
                    mu.lock();
                    counter++;
                    if (bag.size() < 64) bag.push_back(i);
                    else bag.pop_back();
                    mu.unlock();
                    if ((i & 1023) == 0 && mu.try_lock()) {
                        counter += 0;
                        mu.unlock();
                    }
                }
            });
        for (auto& t : ths) t.join();
        const bool ok = counter == (long)threads * per;
        printf("PoolMutex: %d threads x %d sections: %s\n", threads, per, ok ? "ok" : "FAILED");
        if (!ok) bad++;
    }
    return bad ? 1 : 0;
}
