// Where do concurrent latency-bound sumchecks lose their time?  T threads, each on its own context lane stream, run M small generic
// sumchecks (4 ext tables, 2 cubic terms) through the round-granular C ABI with a SplitMix challenge (no transcript, no Python).
// Per thread: microseconds in begin / rounds / finish+free.   build: see tools/dev/lanes_sc.sh
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "ceno_hip.h"
static const uint64_t P = 0xFFFFFFFF00000001ULL;
static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return (z ^ (z >> 31)) % P; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int nv = argc > 1 ? atoi(argv[1]) : 12, M = argc > 2 ? atoi(argv[2]) : 32, maxT = argc > 3 ? atoi(argv[3]) : 4, minT = argc > 4 ? atoi(argv[4]) : 1;
    ceno_hip_ctx* ctx;
    if (ceno_hip_init(0, 0, &ctx)) return 1;
    std::vector<std::vector<ceno_hip_mle*>> tabs(maxT);
    for (int t = 0; t < maxT; t++)
        for (int j = 0; j < 4; j++) {
            ceno_hip_mle* m;
            ceno_hip_mle_alloc(ctx, nv, 1, &m);
            ceno_hip_mle_fill_splitmix(ctx, m, 100 * t + j, 0, nullptr);
            tabs[t].push_back(m);
        }
    ceno_hip_stream_sync(ctx, nullptr);
    const uint64_t coeffs[4] = {3, 1, 5, 2};
    const uint32_t toff[3] = {0, 3, 6}, tidx[6] = {0, 1, 2, 1, 2, 3}, z[1] = {0};
    for (int T = minT; T <= maxT; T *= 2) {
        std::vector<double> tb(T), tr(T), tf(T), wall(T);
        std::vector<int> rc(T, 0);
        for (int rep = 0; rep < 2; rep++) {
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    ceno_hip_stream s;
                    ceno_hip_lane_stream(ctx, t, &s);
                    ceno_hip_sumcheck_plan pl{4, 2, coeffs, toff, tidx, 0, z, z, z, z, nv, 3};
                    double b = 0, r = 0, f = 0;
                    const double w0 = now();
                    for (int k = 0; k < M; k++) {
                        uint64_t st = 77 + k, ch[2], msg[8], fin[8];
                        ceno_hip_sumcheck* sc;
                        double t0 = now();
                        if ((rc[t] = ceno_hip_sumcheck_begin(ctx, tabs[t].data(), &pl, s, &sc))) return;
                        ceno_hip_sumcheck_set_pipelined(ctx, sc, 1);
                        double t1 = now();
                        for (int i = 0; i < nv; i++) {
                            if ((rc[t] = ceno_hip_sumcheck_round(ctx, sc, i ? ch : nullptr, msg))) return;
                            ch[0] = sm(st), ch[1] = sm(st);
                        }
                        double t2 = now();
                        rc[t] = ceno_hip_sumcheck_finish(ctx, sc, ch, fin);
                        ceno_hip_sumcheck_free(ctx, sc);
                        double t3 = now();
                        b += t1 - t0, r += t2 - t1, f += t3 - t2;
                    }
                    wall[t] = now() - w0, tb[t] = b, tr[t] = r, tf[t] = f;
                });
            for (auto& x : th) x.join();
        }
        for (int t = 0; t < T; t++)
            if (rc[t]) { fprintf(stderr, "rc %d: %s\n", rc[t], ceno_hip_last_error(ctx)); return 2; }
        double b = 0, r = 0, f = 0, w = 0;
        for (int t = 0; t < T; t++) b += tb[t], r += tr[t], f += tf[t], w = w > wall[t] ? w : wall[t];
        printf("nv %2d  T %d: per sumcheck  begin %6.1f us  rounds %7.1f us (%5.2f / round)  finish+free %6.1f us   | wall for %d x %d: %.2f ms\n", nv, T,
               b / (T * M), r / (T * M), r / (T * M) / nv, f / (T * M), T, M, w / 1e3);
    }
    for (auto& v : tabs)
        for (auto* m : v) ceno_hip_mle_free(ctx, m);
    ceno_hip_destroy(ctx);
    return 0;
}
