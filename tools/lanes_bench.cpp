// Concurrent chip proving through the C ABI only (no Python in the loop): N ADD-shaped chips (record inference, two
// product towers + one LogUp tower, tower proof, main sumcheck with a Prefix selector) run by ceno_prover_lanes_run
// with 1, 2, 4, 8 lanes.  Build + run on the GPU box:
//   g++ -O2 -std=c++17 -I include tools/lanes_bench.cpp -L ceno_amd -lceno_prover -lceno_hip -Wl,-rpath,$PWD/ceno_amd -lpthread -o /tmp/lanes_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ceno_prover.h"

static const uint64_t P = 0xFFFFFFFF00000001ULL;
static ceno_hip_ctx* g_ctx;
static int g_nv = 18, g_w = 22;

struct Chip {
    std::vector<ceno_hip_mle*> cols;
    int seed;
};

#define CK(x)                                                                      \
    do {                                                                           \
        int _rc = (x);                                                             \
        if (_rc) {                                                                 \
            fprintf(stderr, "%s -> %d: %s / %s\n", #x, _rc, ceno_hip_last_error(g_ctx), ceno_prover_last_error()); \
            return _rc;                                                            \
        }                                                                          \
    } while (0)

static int prove_chip(void* arg, int lane, ceno_hip_stream s) {
    (void)lane;
    Chip* c = (Chip*)arg;
    const int n = g_nv, w = g_w;
    const size_t rows = (size_t)1 << n;
    const uint64_t alpha[2] = {0x1234567, 0x89abcde}, beta[2] = {0x13579b, 0x2468ac};
    const uint64_t b2[2] = {(uint64_t)(((unsigned __int128)beta[0] * beta[0] + (unsigned __int128)7 * beta[1] * beta[1]) % P),
                            (uint64_t)(((unsigned __int128)2 * beta[0] * beta[1]) % P)};
    // record inference: 16 records r_k = beta w_{2k} + beta^2 w_{2k+1} + alpha w_a w_b
    std::vector<uint64_t> coeffs;
    std::vector<uint32_t> toff{0}, tidx, ooff{0};
    for (int k = 0; k < 16; k++) {
        const uint64_t* cs[3] = {beta, b2, alpha};
        const std::vector<std::vector<uint32_t>> ts = {{(uint32_t)((2 * k) % w)}, {(uint32_t)((2 * k + 1) % w)},
                                                       {(uint32_t)((3 * k + 5) % w), (uint32_t)((k + 7) % w)}};
        for (int t = 0; t < 3; t++) {
            coeffs.push_back(cs[t][0]);
            coeffs.push_back(cs[t][1]);
            for (uint32_t i : ts[t]) tidx.push_back(i);
            toff.push_back((uint32_t)tidx.size());
        }
        ooff.push_back((uint32_t)toff.size() - 1);
    }
    std::vector<ceno_hip_mle*> recs(16, nullptr);
    CK(ceno_hip_wit_infer(g_ctx, c->cols.data(), w, coeffs.data(), toff.data(), tidx.data(), (int)toff.size() - 1, ooff.data(), 16, n, s, recs.data()));
    const uint64_t one[2] = {1, 0};
    ceno_hip_tower *pt[2] = {nullptr, nullptr}, *lt[1] = {nullptr};
    CK(ceno_hip_tower_build_prod(g_ctx, recs.data(), 4, rows, one, s, &pt[0]));
    CK(ceno_hip_tower_build_prod(g_ctx, recs.data() + 4, 4, rows, one, s, &pt[1]));
    CK(ceno_hip_tower_build_logup(g_ctx, nullptr, recs.data() + 8, 8, rows, alpha, s, &lt[0]));
    ceno_transcript* tr = ceno_transcript_stub_new((uint64_t)c->seed);
    const int max_nv = ceno_hip_tower_num_vars(lt[0]);
    std::vector<uint64_t> out_evals(2 * (2 * 2 + 4)), msgs(ceno_tower_msgs_words(max_nv) + 8);
    std::vector<uint64_t> pe(2 * 2 * 2 * (size_t)max_nv), le(2 * 4 * (size_t)max_nv), point(2 * (size_t)max_nv + 2);
    ceno_tower_proof proof{0, msgs.data(), pe.data(), le.data(), point.data()};
    CK(ceno_prover_prove_tower_relation(g_ctx, pt, 2, lt, 1, tr, s, out_evals.data(), &proof));
    // main sumcheck: sel * (pairs and triples of witness columns)
    std::vector<uint64_t> pt_(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        pt_[2 * i] = ((uint64_t)i * 7919 + 13) % P;
        pt_[2 * i + 1] = ((uint64_t)i * 104729 + 17) % P;
    }
    ceno_hip_mle* sel = nullptr;
    CK(ceno_hip_selector_build(g_ctx, CENO_HIP_SEL_PREFIX, pt_.data(), n, 0, rows - 3, nullptr, 0, 0, s, &sel));
    std::vector<ceno_hip_mle*> mles = c->cols;
    mles.push_back(sel);
    std::vector<uint64_t> mc;
    std::vector<uint32_t> mo{0}, mi, gto{0}, gti, co{0}, ci;
    int nt = 0;
    for (int j = 0; j < w; j++, nt++) {
        mi.push_back(j); mi.push_back((j + 1) % w); mo.push_back((uint32_t)mi.size());
        mc.push_back((3 + 5 * nt) % P); mc.push_back((11 * nt + 1) % P);
    }
    for (int j = 0; j < w; j += 2, nt++) {
        mi.push_back(j); mi.push_back((j + 3) % w); mi.push_back((j + 5) % w); mo.push_back((uint32_t)mi.size());
        mc.push_back((3 + 5 * nt) % P); mc.push_back((11 * nt + 1) % P);
    }
    for (int t = 0; t < nt; t++) gti.push_back(t);
    gto.push_back((uint32_t)gti.size());
    ci.push_back((uint32_t)w);
    co.push_back(1);
    ceno_hip_sumcheck_plan plan;
    memset(&plan, 0, sizeof(plan));
    plan.num_mles = w + 1; plan.num_terms = nt; plan.term_coeffs = mc.data(); plan.term_offsets = mo.data(); plan.term_mle_idx = mi.data();
    plan.num_groups = 1; plan.group_term_offsets = gto.data(); plan.group_term_idx = gti.data(); plan.common_offsets = co.data(); plan.common_mle_idx = ci.data();
    plan.max_num_vars = n; plan.max_degree = 4;
    std::vector<uint64_t> m2(2 * 4 * (size_t)n), ch2(2 * (size_t)n), fin(2 * (size_t)(w + 1));
    CK(ceno_prover_sumcheck_prove(g_ctx, mles.data(), &plan, tr, s, m2.data(), ch2.data(), fin.data()));
    ceno_transcript_free(tr);
    ceno_hip_mle_free(g_ctx, sel);
    for (auto* t : pt) ceno_hip_tower_free(g_ctx, t);
    ceno_hip_tower_free(g_ctx, lt[0]);
    for (auto* m : recs) ceno_hip_mle_free(g_ctx, m);
    return 0;
}

int main(int argc, char** argv) {
    int n_chips = argc > 1 ? atoi(argv[1]) : 8;
    if (argc > 2) g_nv = atoi(argv[2]);
    if (ceno_hip_init(0, 0, &g_ctx)) { fprintf(stderr, "init failed\n"); return 1; }
    std::vector<Chip> chips(n_chips);
    for (int c = 0; c < n_chips; c++) {
        chips[c].seed = 10 * c;
        chips[c].cols.resize(g_w);
        for (int j = 0; j < g_w; j++) {
            if (ceno_hip_mle_alloc(g_ctx, g_nv, 0, &chips[c].cols[j]) || ceno_hip_mle_fill_splitmix(g_ctx, chips[c].cols[j], 7000 + 100 * c + j, 0, nullptr)) return 2;
        }
    }
    ceno_hip_stream_sync(g_ctx, nullptr);
    std::vector<ceno_lane_task> tasks(n_chips);
    const size_t est = ((size_t)1 << g_nv) * 16 * 80;  // records + towers + sumcheck work buffers, generous
    for (int c = 0; c < n_chips; c++) tasks[c] = ceno_lane_task{prove_chip, &chips[c], est};
    double base = 0;
    for (int lanes : {1, 2, 3, 4, 6, 8}) {
        double best = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            std::vector<int> st(n_chips, -1);
            auto t0 = std::chrono::steady_clock::now();
            int rc = ceno_prover_lanes_run(g_ctx, lanes, tasks.data(), n_chips, st.data(), nullptr);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rc) { fprintf(stderr, "lanes_run failed: %s\n", ceno_prover_last_error()); return 3; }
            best = ms < best ? ms : best;
        }
        if (lanes == 1) base = best;
        printf("lanes %d: %8.2f ms for %d chips of 2^%d rows  (x%.2f)\n", lanes, best, n_chips, g_nv, base / best);
    }
    return 0;
}
