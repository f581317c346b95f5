// Host layer above the device C ABI: the reference's prover control flow for the hot path,
// restated in C++ (the reference is Rust; see include/ceno_prover.h).
//
//   ceno_prover_sumcheck_prove       IOPProverState::prove (EXT sumcheck); transcript script per
//                                    ceno_recursion_v2/src/main/mod.rs:3503-3529
//   ceno_prover_tower_create_proof   CpuTowerProver::create_proof, ceno_zkvm/src/scheme/cpu/mod.rs:346-554
//   ceno_prover_prove_tower_relation TowerProver::prove_tower_relation, scheme/cpu/mod.rs:765-797
#include "../../include/ceno_prover.h"

#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include <string>
#include <vector>

#include "../csrc/gl64.hpp"
#include "../csrc/e2_host_avx512.hpp"
#include "transcript.hpp"
#include "tower_state.hpp"

using gl::E2;

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
static int fail_from_ctx(ceno_hip_ctx* ctx, int code) {
    const char* m = ceno_hip_last_error(ctx);
    g_err = m ? m : "";
    return code;
}

extern "C" const char* ceno_prover_last_error(void) { return g_err.c_str(); }
int prover_set_error(int code, const char* msg) {
    g_err = msg ? msg : "";
    return code;
}

// ------------------------------------------------------------------------------------------------
// transcript helpers (same call shapes as the reference's Transcript trait)
// ------------------------------------------------------------------------------------------------
static void tr_label(ceno_transcript* t, const char* s) { t->append_label(t->self, (const uint8_t*)s, strlen(s)); }
static void tr_usize(ceno_transcript* t, uint64_t v) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));  // usize::to_le_bytes
    t->append_label(t->self, b, 8);
}
static void tr_ext(ceno_transcript* t, const uint64_t* e) { t->append_ext(t->self, e); }
static E2 tr_sample(ceno_transcript* t) {
    uint64_t o[2];
    t->sample_ext(t->self, o);
    return E2{o[0], o[1]};
}
// get_challenge_pows(size, transcript): label b"combine subset evals", one sample, [1, a, a^2, ...]
static void tr_challenge_pows(ceno_transcript* t, int n, std::vector<uint64_t>& out) {
    tr_label(t, "combine subset evals");
    E2 a = tr_sample(t), acc = gl::e2_one();
    out.resize((size_t)2 * (n > 0 ? n : 0));
    for (int i = 0; i < n; i++) {
        out[2 * i] = acc.c0;
        out[2 * i + 1] = acc.c1;
        acc = acc * a;
    }
}

extern "C" {

void ceno_transcript_append_label(ceno_transcript* t, const uint8_t* bytes, size_t n) { t->append_label(t->self, bytes, n); }
void ceno_transcript_append_ext(ceno_transcript* t, const uint64_t* e2) { t->append_ext(t->self, e2); }
void ceno_transcript_sample_ext(ceno_transcript* t, uint64_t* out2) { t->sample_ext(t->self, out2); }
void ceno_transcript_append_base(ceno_transcript* t, uint64_t v) {
    if (t->append_base) t->append_base(t->self, v);
    else {  // a transcript without the entry: a base element is the extension element (v, 0)?  No — it is ONE observed element.
        const uint8_t* b = reinterpret_cast<const uint8_t*>(&v);
        t->append_label(t->self, b, 8);  // 8 little-endian bytes pack into exactly one field element
    }
}
void ceno_transcript_free(ceno_transcript* t) {
    if (!t) return;
    if (t->destroy) t->destroy(t->self);
    delete t;
}
// sample_bits: low bits of the canonical value of ONE base sample (ceno_recursion_v2/src/pcs/mod.rs:8164-8204)
uint64_t ceno_transcript_sample_bits(ceno_transcript* t, int bits) { return t->sample_bits(t->self, bits); }
// check_witness: observe(witness); sample_bits(bits) == 0 (pcs/mod.rs:8125-8155)
int ceno_transcript_check_witness(ceno_transcript* t, int bits, uint64_t witness) {
    ceno_transcript_append_base(t, witness);
    return ceno_transcript_sample_bits(t, bits) == 0 ? 1 : 0;
}
ceno_transcript* ceno_transcript_clone(const ceno_transcript* t) {
    if (!t || !t->fork || !t->fork_free) return nullptr;
    auto* c = new ceno_transcript(*t);
    c->self = t->fork(t->self);
    c->destroy = t->fork_free;
    if (!c->self) {
        delete c;
        return nullptr;
    }
    return c;
}
int ceno_transcript_export_state(ceno_transcript* t, uint64_t* out16) { return (t && t->export_state) ? t->export_state(t->self, out16) : 0; }
int ceno_transcript_import_state(ceno_transcript* t, const uint64_t* in16) {
    return (t && t->import_state) ? t->import_state(t->self, in16) : CENO_HIP_ERR_INVALID;
}

int ceno_prover_transcript_grind(ceno_hip_ctx* ctx, ceno_transcript* t, int bits, ceno_hip_stream s, uint64_t* out_witness) {
    if (!t || !out_witness || bits < 0 || bits > 40) return fail(CENO_HIP_ERR_INVALID, "bad transcript_grind arguments");
    if (!t->sample_bits || !t->append_base) return fail(CENO_HIP_ERR_INVALID, "transcript_grind: the transcript has no base-field operations (sample_bits / append_base)");
    uint64_t st16[16];
    uint64_t w = 0;
    if (bits > 0 && ceno_transcript_export_state(t, st16) == CENO_TRANSCRIPT_DUPLEX8) {
        if (!ctx) return fail(CENO_HIP_ERR_INVALID, "transcript_grind: NULL context");
        int rc = ceno_hip_pow_grind_duplex(ctx, st16, bits, &w, s);  // one permutation per candidate, on the device
        if (rc) return fail_from_ctx(ctx, rc);
    } else if (bits > 0 && t->grind) {  // the transcript's own GrindingChallenger::grind: it also runs check_witness on itself
        *out_witness = t->grind(t->self, bits);
        return 0;
    } else if (bits > 0) {
        if (!t->fork || !t->fork_free) return fail(CENO_HIP_ERR_INVALID, "transcript_grind: the transcript can neither export its state, grind, nor fork");
        for (;; w++) {
            ceno_transcript c = *t;
            c.self = t->fork(t->self);
            const int ok = ceno_transcript_check_witness(&c, bits, w);
            t->fork_free(c.self);
            if (ok) break;
            if (w > ((uint64_t)1 << 44)) return fail(CENO_HIP_ERR_STATE, "transcript_grind: no witness found");
        }
    }
    if (!ceno_transcript_check_witness(t, bits, w)) return fail(CENO_HIP_ERR_STATE, "transcript_grind: the witness found does not pass check_witness");
    *out_witness = w;
    return 0;
}

int ceno_prover_sumcheck_run(ceno_hip_ctx* ctx, ceno_hip_sumcheck* sc, int n, int d, int num_mles, ceno_transcript* tr,
                             uint64_t* out_msgs, uint64_t* out_challenges, uint64_t* out_final_evals) {
    (void)num_mles;
    if (!ctx || !sc || !tr || !out_msgs || !out_challenges || !out_final_evals) return fail(CENO_HIP_ERR_INVALID, "NULL argument");
    tr_usize(tr, (uint64_t)n);
    tr_usize(tr, (uint64_t)d);
    uint64_t ch[2] = {0, 0};
    static const bool trace_rounds = getenv("CENO_PROVER_ROUND_TRACE") != nullptr;  // wall time of every round of every sumcheck driven here
    for (int round = 0; round < n; round++) {
        uint64_t* msg = out_msgs + (size_t)2 * d * round;
        const auto t_round = std::chrono::steady_clock::now();
        int rc = ceno_hip_sumcheck_round(ctx, sc, round == 0 ? nullptr : ch, msg);
        if (trace_rounds)
            fprintf(stderr, "[ceno_prover] sumcheck of %d variables, round %d: %.1f us\n", n, round,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_round).count());
        if (rc) return fail_from_ctx(ctx, rc);
        for (int t = 0; t < d; t++) tr_ext(tr, msg + 2 * t);
        tr_label(tr, "Internal round");
        E2 r = tr_sample(tr);
        ch[0] = r.c0;
        ch[1] = r.c1;
        out_challenges[2 * round] = r.c0;
        out_challenges[2 * round + 1] = r.c1;
    }
    int rc = ceno_hip_sumcheck_finish(ctx, sc, n > 0 ? ch : nullptr, out_final_evals);
    if (rc) return fail_from_ctx(ctx, rc);
    return 0;
}

int ceno_prover_sumcheck_prove(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan, ceno_transcript* tr,
                               ceno_hip_stream s, uint64_t* out_msgs, uint64_t* out_challenges, uint64_t* out_final_evals) {
    return ceno_prover_sumcheck_prove_eq(ctx, mles, plan, 0, nullptr, nullptr, nullptr, nullptr, tr, s, out_msgs, out_challenges, out_final_evals);
}

int ceno_prover_sumcheck_prove_eq(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan, int num_eq, const int* eq_mle_idx,
                                  const uint64_t* const* eq_points, const size_t* eq_lo, const size_t* eq_hi, ceno_transcript* tr, ceno_hip_stream s,
                                  uint64_t* out_msgs, uint64_t* out_challenges, uint64_t* out_final_evals) {
    if (!ctx || !plan) return fail(CENO_HIP_ERR_INVALID, "NULL argument");
    ceno_hip_sumcheck* sc = nullptr;
    static const bool dbg = getenv("CENO_HIP_DEBUG") != nullptr;
    auto now_us = []() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e6 + ts.tv_nsec / 1e3;
    };
    const double t0 = dbg ? now_us() : 0;
    int rc = ceno_hip_sumcheck_begin_eq(ctx, mles, plan, num_eq, eq_mle_idx, eq_points, eq_lo, eq_hi, s, &sc);
    if (rc) return fail_from_ctx(ctx, rc);
    const double t1 = dbg ? now_us() : 0;
    ceno_hip_sumcheck_set_pipelined(ctx, sc, 1);  // this loop drives the rounds back to back
    rc = ceno_prover_sumcheck_run(ctx, sc, plan->max_num_vars, plan->max_degree, plan->num_mles, tr, out_msgs, out_challenges,
                                  out_final_evals);
    const double t2 = dbg ? now_us() : 0;
    ceno_hip_sumcheck_free(ctx, sc);
    if (dbg) fprintf(stderr, "[ceno_prover] sumcheck_prove (%d mles, %d terms, n=%d): begin %.0f us, rounds %.0f us, free %.0f us\n", plan->num_mles,
                     plan->num_terms, plan->max_num_vars, t1 - t0, t2 - t1, now_us() - t2);
    return rc;
}

size_t ceno_tower_msgs_words(int max_nv) {
    size_t tot = 0;
    for (int r = 1; r < max_nv; r++) tot += (size_t)r * 3 * 2;
    return tot;
}

// ------------------------------------------------------------------------------------------------------------------
// Small tower layers are proved ON THE HOST.  A layer of 2^l entries per limb is l sumcheck rounds of a few hundred to a few
// thousand field multiplications each; on the device it costs ~45 us of per-layer set-up (eq table, plan, launches) plus
// ~9 us of latency per round whatever its size.  The top layers of every tower arrive with one copy each
// (ceno_hip_tower_download_top) and the layer sumcheck of scheme/cpu/mod.rs:417-494 runs right here:
//   sum_x eq(x, rt) * [ sum_i alpha_i a_i b_i + sum_k (alpha_n (p1 q2 + p2 q1) + alpha_d q1 q2) ],  messages at X = 1, 2, 3.
// Field arithmetic is exact, so the messages, challenges and evaluations equal the device path's bit for bit.
// ------------------------------------------------------------------------------------------------------------------
namespace {
int tower_host_layers() {
    const char* e = getenv("CENO_TOWER_HOST_LAYERS");  // layers 1 .. this are proved on the host (0: none)
    return e ? atoi(e) : 8;
}
// one layer sumcheck on the host; tabs[0] = eq, then per active product spec (a, b), per active logup spec (p1, p2, q1, q2)
#if defined(__x86_64__)
// eight pairs [p, p + 8) of one round of host_tower_layer, one pair per AVX-512 lane (csrc/e2_host_avx512.hpp); adds to acc[0..3)
__attribute__((target("avx512f,avx512dq"))) void host_tower_eval8(const std::vector<std::vector<E2>>& tabs, int n_prod_active, int n_logup_active,
                                                                   const std::vector<E2>& alpha_prod, const std::vector<E2>& alpha_num,
                                                                   const std::vector<E2>& alpha_den, size_t p, E2* acc) {
    using namespace e2v;
#define hi_of(t) load(tabs[(t)].data(), 2 * p + 1, 2)
#define lo_of(t) load(tabs[(t)].data(), 2 * p, 2)
    VE2 inner[3] = {bcast(gl::e2_zero()), bcast(gl::e2_zero()), bcast(gl::e2_zero())};
    size_t t = 1;
    for (int i = 0; i < n_prod_active; i++, t += 2) {
        const VE2 a1 = hi_of(t), da = sub(a1, lo_of(t)), b1 = hi_of(t + 1), db = sub(b1, lo_of(t + 1));
        const VE2 al = bcast(alpha_prod[i]);
        VE2 ca = mul(al, a1);
        const VE2 cda = mul(al, da);
        VE2 b = b1;
        for (int e = 0; e < 3; e++) {
            inner[e] = add(inner[e], mul(ca, b));
            ca = add(ca, cda);
            b = add(b, db);
        }
    }
    for (int k = 0; k < n_logup_active; k++, t += 4) {
        VE2 p1 = hi_of(t), p2 = hi_of(t + 1), q1 = hi_of(t + 2), q2 = hi_of(t + 3);
        const VE2 dp1 = sub(p1, lo_of(t)), dp2 = sub(p2, lo_of(t + 1)), dq1 = sub(q1, lo_of(t + 2)), dq2 = sub(q2, lo_of(t + 3));
        const VE2 an = bcast(alpha_num[k]), ad = bcast(alpha_den[k]);
        for (int e = 0; e < 3; e++) {
            inner[e] = add(inner[e], add(mul(an, add(mul(p1, q2), mul(p2, q1))), mul(ad, mul(q1, q2))));
            p1 = add(p1, dp1);
            p2 = add(p2, dp2);
            q1 = add(q1, dq1);
            q2 = add(q2, dq2);
        }
    }
    VE2 ev = hi_of(0);
    const VE2 de = sub(ev, lo_of(0));
    for (int e = 0; e < 3; e++) {
        acc[e] = acc[e] + hsum(mul(ev, inner[e]));
        ev = add(ev, de);
    }
#undef hi_of
#undef lo_of
}
__attribute__((target("avx512f,avx512dq"))) void host_tower_fold8(E2* t, size_t first, E2 r) {
    const e2v::VE2 lo = e2v::load(t, 2 * first, 2), hi = e2v::load(t, 2 * first + 1, 2);
    e2v::store(t, first, e2v::add(lo, e2v::mul(e2v::bcast(r), e2v::sub(hi, lo))));
}
#endif
void host_tower_layer(int n, std::vector<std::vector<E2>>& tabs, int n_prod_active, int n_logup_active, const std::vector<E2>& alpha_prod,
                      const std::vector<E2>& alpha_num, const std::vector<E2>& alpha_den, ceno_transcript* tr, uint64_t* msgs, uint64_t* chal,
                      uint64_t* fin, bool prologue = true) {
    if (prologue) {  // (false: the last n rounds of a layer whose first rounds ran elsewhere — the row-sharded tower prover, dist_gkr.cpp)
        tr_usize(tr, (uint64_t)n);
        tr_usize(tr, 3);
    }
    size_t len = (size_t)1 << n;
    for (int round = 0; round < n; round++) {
        const size_t pairs = len / 2;
        E2 acc[3] = {gl::e2_zero(), gl::e2_zero(), gl::e2_zero()};
        size_t p0 = 0;
#if defined(__x86_64__)
        if (p2host::have_avx512())
            for (; p0 + 8 <= pairs; p0 += 8) host_tower_eval8(tabs, n_prod_active, n_logup_active, alpha_prod, alpha_num, alpha_den, p0, acc);
#endif
        for (size_t p = p0; p < pairs; p++) {
            E2 inner[3] = {gl::e2_zero(), gl::e2_zero(), gl::e2_zero()};
            size_t t = 1;
            for (int i = 0; i < n_prod_active; i++, t += 2) {
                const E2 a1 = tabs[t][2 * p + 1], da = a1 - tabs[t][2 * p];
                const E2 b1 = tabs[t + 1][2 * p + 1], db = b1 - tabs[t + 1][2 * p];
                E2 ca = alpha_prod[i] * a1;
                const E2 cda = alpha_prod[i] * da;  // the coefficient rides on the first factor
                E2 b = b1;
                for (int e = 0; e < 3; e++) {
                    inner[e] = inner[e] + ca * b;
                    ca = ca + cda;
                    b = b + db;
                }
            }
            for (int k = 0; k < n_logup_active; k++, t += 4) {
                E2 p1 = tabs[t][2 * p + 1], p2 = tabs[t + 1][2 * p + 1], q1 = tabs[t + 2][2 * p + 1], q2 = tabs[t + 3][2 * p + 1];
                const E2 dp1 = p1 - tabs[t][2 * p], dp2 = p2 - tabs[t + 1][2 * p], dq1 = q1 - tabs[t + 2][2 * p], dq2 = q2 - tabs[t + 3][2 * p];
                for (int e = 0; e < 3; e++) {
                    inner[e] = inner[e] + alpha_num[k] * (p1 * q2 + p2 * q1) + alpha_den[k] * (q1 * q2);
                    p1 = p1 + dp1;
                    p2 = p2 + dp2;
                    q1 = q1 + dq1;
                    q2 = q2 + dq2;
                }
            }
            E2 ev = tabs[0][2 * p + 1];
            const E2 de = ev - tabs[0][2 * p];
            for (int e = 0; e < 3; e++) {
                acc[e] = acc[e] + ev * inner[e];
                ev = ev + de;
            }
        }
        uint64_t* msg = msgs + (size_t)6 * round;
        for (int e = 0; e < 3; e++) {
            msg[2 * e] = acc[e].c0;
            msg[2 * e + 1] = acc[e].c1;
            tr_ext(tr, msg + 2 * e);
        }
        tr_label(tr, "Internal round");
        const E2 r = tr_sample(tr);
        chal[2 * round] = r.c0;
        chal[2 * round + 1] = r.c1;
        for (auto& T : tabs) {
            size_t q0 = 0;
#if defined(__x86_64__)
            if (p2host::have_avx512())
                for (; q0 + 8 <= pairs; q0 += 8) host_tower_fold8(T.data(), q0, r);  // in place: pass q0 reads [2 q0, 2 q0 + 16), at or beyond what it writes
#endif
            for (size_t p = q0; p < pairs; p++) T[p] = T[2 * p] + r * (T[2 * p + 1] - T[2 * p]);
        }
        len = pairs;
    }
    for (size_t t = 0; t < tabs.size(); t++) {
        fin[2 * t] = tabs[t][0].c0;
        fin[2 * t + 1] = tabs[t][0].c1;
    }
}
}  // namespace

}  // extern "C"
int prover_tower_host_layers() { return std::max(tower_host_layers(), 0); }
// the last `n` rounds of a tower layer's sumcheck on host tables (tabs[0] = eq, then (a, b) per product tower, (p1, p2, q1, q2) per LogUp
// tower), no prologue: what the row-sharded tower prover runs on the gathered tables after its local rounds (dist_gkr.cpp)
void prover_host_tower_rounds(int n, std::vector<std::vector<E2>>& tabs, int n_prod_active, int n_logup_active, const std::vector<E2>& alpha_prod,
                              const std::vector<E2>& alpha_num, const std::vector<E2>& alpha_den, ceno_transcript* tr, uint64_t* msgs, uint64_t* chal,
                              uint64_t* fin) {
    host_tower_layer(n, tabs, n_prod_active, n_logup_active, alpha_prod, alpha_num, alpha_den, tr, msgs, chal, fin, false);
}
#include "tower_hook.hpp"
int prover_tower_create_proof_hooked(ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup, int n_logup,
                                     ceno_transcript* tr, ceno_hip_stream s, ceno_tower_proof* out, const TowerDistHook* hook);
extern "C" {
int ceno_prover_tower_create_proof(ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup, int n_logup,
                                   ceno_transcript* tr, ceno_hip_stream s, ceno_tower_proof* out) {
    return prover_tower_create_proof_hooked(ctx, prod, n_prod, logup, n_logup, tr, s, out, nullptr);
}
}  // extern "C"
// CpuTowerProver::create_proof.  hook == NULL: the towers hold every layer.  hook != NULL (row-sharded chip proof): the towers passed in are
// the REPLICATED tops of towers whose large layers live sharded across ranks — the numbers of variables come from the hook, rounds above
// hook->r_rep are proved by hook->layer (same outputs: the round's messages, challenges and final evaluations).
// Written as a resumable state (tower_state.hpp): init, one step per layer, finish — the cohort driver interleaves the steps of many chips.
int TowerProveState::nv_of(const ceno_hip_tower* t) const {
    if (hook) {
        for (int i = 0; i < n_prod; i++) if (prod[i] == t) return hook->nv_global[i];
        for (int i = 0; i < n_logup; i++) if (logup[i] == t) return hook->nv_global[n_prod + i];
    }
    return ::ceno_hip_tower_num_vars(t);
}
void prover_tr_usize(ceno_transcript* t, uint64_t v) { tr_usize(t, v); }
void prover_tr_ext_words(ceno_transcript* t, const uint64_t* ext, int n_ext) {
    for (int i = 0; i < n_ext; i++) tr_ext(t, ext + 2 * i);
}
E2 prover_tr_round(ceno_transcript* t, const uint64_t* msg6) {
    for (int e = 0; e < 3; e++) tr_ext(t, msg6 + 2 * e);
    tr_label(t, "Internal round");
    return tr_sample(t);
}

int tower_state_init(TowerProveState& st, ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup, int n_logup,
                     ceno_transcript* tr, ceno_hip_stream s, ceno_tower_proof* out, const TowerDistHook* hook) {
    if (!ctx || !tr || !out) return fail(CENO_HIP_ERR_INVALID, "NULL argument");
    st.ctx = ctx; st.prod = prod; st.n_prod = n_prod; st.logup = logup; st.n_logup = n_logup; st.tr = tr; st.s = s; st.out = out; st.hook = hook;
    st.max_nv = 0;
    for (int i = 0; i < n_prod; i++) st.max_nv = std::max(st.max_nv, st.nv_of(prod[i]));
    for (int i = 0; i < n_logup; i++) st.max_nv = std::max(st.max_nv, st.nv_of(logup[i]));
    if (st.max_nv < 1) return fail(CENO_HIP_ERR_INVALID, "tower: no specs");
    st.n_alpha = n_prod + 2 * n_logup;
    tr_challenge_pows(tr, st.n_alpha, st.alpha);    // cpu/mod.rs:375-380
    tr_label(tr, "product_sum");                    // cpu/mod.rs:381  (sample_and_append_vec, log2(fanin) = 1)
    st.out_rt.assign(2 * (size_t)(st.max_nv + 1), 0);
    {
        E2 r0 = tr_sample(tr);
        st.out_rt[0] = r0.c0;
        st.out_rt[1] = r0.c1;
    }
    st.R = st.max_nv - 1;
    out->num_rounds = st.R;
    st.msg_off = 0;
    st.round = 1;
    st.claim = gl::e2_zero();  // round 1: from the out-evals, not formed here
    st.have_claim = false;
    // the small layers of every tower, one copy per tower
    st.host_layers = std::min(tower_host_layers(), st.R);
    st.top_prod.assign((size_t)n_prod, HostTowerTop());
    st.top_logup.assign((size_t)n_logup, HostTowerTop());
    if (st.host_layers >= 1) {
        auto fetch = [&](ceno_hip_tower* t, HostTowerTop& h) -> int {
            // layers 0 .. min(host_layers, num_vars - 1) of this tower (layer `round` exists when num_vars > round)
            h.n_limbs = ceno_hip_tower_num_limbs(t);
            h.n_layers = std::min(std::min(st.host_layers + 1, ::ceno_hip_tower_num_vars(t)), ceno_hip_tower_top_layers(t));
            if (h.n_layers < 1) return 0;
            h.words.resize((size_t)2 * h.n_limbs * (((size_t)1 << h.n_layers) - 1));
            return ceno_hip_tower_download_top(ctx, t, h.n_layers, h.words.data(), s);
        };
        for (int i = 0; i < n_prod; i++)
            if (int rc = fetch(prod[i], st.top_prod[i])) return fail_from_ctx(ctx, rc);
        for (int i = 0; i < n_logup; i++)
            if (int rc = fetch(logup[i], st.top_logup[i])) return fail_from_ctx(ctx, rc);
    }
    return 0;
}

int tower_state_step(TowerProveState& st) {
    if (st.done()) return 0;
    ceno_hip_ctx* ctx = st.ctx;
    ceno_transcript* tr = st.tr;
    ceno_hip_stream s = st.s;
    ceno_tower_proof* out = st.out;
    const TowerDistHook* hook = st.hook;
    ceno_hip_tower* const* prod = st.prod;
    ceno_hip_tower* const* logup = st.logup;
    const int n_prod = st.n_prod, n_logup = st.n_logup, round = st.round;
    std::vector<uint64_t>& alpha = st.alpha;
    std::vector<uint64_t>& out_rt = st.out_rt;
    auto ceno_hip_tower_num_vars = [&](const ceno_hip_tower* t) { return st.nv_of(t); };  // (shadows the C entry: the GLOBAL height of a tower)
    std::vector<uint64_t> chal, fin;
    static const bool dbg = getenv("CENO_HIP_DEBUG") != nullptr || getenv("CENO_PROVER_LAYER_TRACE") != nullptr;
    auto now_us = []() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e6 + ts.tv_nsec / 1e3;
    };
    {                                               // cpu/mod.rs:409: skip(1) for the output layer
        bool on_host = round <= st.host_layers;
        for (int i = 0; i < n_prod && on_host; i++)
            if (ceno_hip_tower_num_vars(prod[i]) > round && st.top_prod[i].n_layers <= round) on_host = false;
        for (int i = 0; i < n_logup && on_host; i++)
            if (ceno_hip_tower_num_vars(logup[i]) > round && st.top_logup[i].n_layers <= round) on_host = false;
        if (hook && round > hook->r_rep) {
            int n_mles = 1;
            for (int i = 0; i < n_prod; i++) if (ceno_hip_tower_num_vars(prod[i]) > round) n_mles += 2;
            for (int i = 0; i < n_logup; i++) if (ceno_hip_tower_num_vars(logup[i]) > round) n_mles += 4;
            chal.assign((size_t)2 * round, 0);
            fin.assign((size_t)2 * n_mles, 0);
            if (int rc = hook->layer(hook->self, round, out_rt.data(), alpha.data(), tr, out->msgs + st.msg_off, chal.data(), fin.data())) return rc;
        } else if (on_host) {
            std::vector<std::vector<E2>> tabs;
            std::vector<E2> a_prod, a_num, a_den;
            const size_t len = (size_t)1 << round;
            {   // eq(x, out_rt): variable j of x is bit j of the index (LSB first)
                std::vector<E2> eq(len);
                eq[0] = gl::e2_one();
                for (int j = 0; j < round; j++) {
                    const E2 rj{out_rt[2 * j], out_rt[2 * j + 1]};
                    for (size_t x = 0; x < ((size_t)1 << j); x++) {
                        const E2 hi = eq[x] * rj;
                        eq[x + ((size_t)1 << j)] = hi;
                        eq[x] = eq[x] - hi;
                    }
                }
                tabs.push_back(std::move(eq));
            }
            int np_act = 0, nl_act = 0;
            for (int i = 0; i < n_prod; i++) {
                if (ceno_hip_tower_num_vars(prod[i]) <= round) continue;
                for (int b = 0; b < 2; b++) tabs.emplace_back(st.top_prod[i].limb(round, b), st.top_prod[i].limb(round, b) + len);
                a_prod.push_back(E2{alpha[2 * i], alpha[2 * i + 1]});
                np_act++;
            }
            for (int i = 0; i < n_logup; i++) {
                if (ceno_hip_tower_num_vars(logup[i]) <= round) continue;
                for (int b = 0; b < 4; b++) tabs.emplace_back(st.top_logup[i].limb(round, b), st.top_logup[i].limb(round, b) + len);
                a_num.push_back(E2{alpha[2 * (n_prod + 2 * i)], alpha[2 * (n_prod + 2 * i) + 1]});
                a_den.push_back(E2{alpha[2 * (n_prod + 2 * i + 1)], alpha[2 * (n_prod + 2 * i + 1) + 1]});
                nl_act++;
            }
            if (np_act + nl_act == 0) return fail(CENO_HIP_ERR_INVALID, "tower: no spec has this layer");
            chal.assign((size_t)2 * round, 0);
            fin.assign((size_t)2 * tabs.size(), 0);
            host_tower_layer(round, tabs, np_act, nl_act, a_prod, a_num, a_den, tr, out->msgs + st.msg_off, chal.data(), fin.data());
        } else {
        ceno_hip_sumcheck* sc = nullptr;
        const double t_a = dbg ? now_us() : 0;
        int rc = ceno_hip_tower_layer_sumcheck_begin(ctx, prod, n_prod, logup, n_logup, round, out_rt.data(), alpha.data(), s, &sc);
        if (rc) return fail_from_ctx(ctx, rc);
        const double t_b = dbg ? now_us() : 0;
        // MLE order of the handle: [eq, active prod (a,b)..., active logup (p1,p2,q1,q2)...]
        int n_mles = 1;
        for (int i = 0; i < n_prod; i++) if (ceno_hip_tower_num_vars(prod[i]) > round) n_mles += 2;
        for (int i = 0; i < n_logup; i++) if (ceno_hip_tower_num_vars(logup[i]) > round) n_mles += 4;
        chal.assign((size_t)2 * round, 0);
        fin.assign((size_t)2 * n_mles, 0);
        ceno_hip_sumcheck_set_pipelined(ctx, sc, 1);  // the loop below drives the rounds back to back
        if (st.have_claim) {                          // the fused tower rounds then need two values in round 0 instead of three
            const uint64_t c2[2] = {st.claim.c0, st.claim.c1};
            rc = ceno_hip_sumcheck_set_claim(ctx, sc, c2);
            if (rc) { ceno_hip_sumcheck_free(ctx, sc); return fail_from_ctx(ctx, rc); }
        }
        rc = ceno_prover_sumcheck_run(ctx, sc, round, 3, n_mles, tr, out->msgs + st.msg_off, chal.data(), fin.data());
        const double t_c = dbg ? now_us() : 0;
        ceno_hip_sumcheck_free(ctx, sc);
        if (dbg) fprintf(stderr, "[ceno_prover] tower layer %d: begin %.0f us, rounds %.0f us, free %.0f us\n", round, t_b - t_a, t_c - t_b, now_us() - t_c);
        if (rc) return rc;
        }
    }
    return tower_state_layer_epilogue(st, chal.data(), fin.data());
}

int tower_state_layer_epilogue(TowerProveState& st, const uint64_t* chal, const uint64_t* fin) {
    ceno_transcript* tr = st.tr;
    ceno_tower_proof* out = st.out;
    ceno_hip_tower* const* prod = st.prod;
    ceno_hip_tower* const* logup = st.logup;
    const int n_prod = st.n_prod, n_logup = st.n_logup, round = st.round, R = st.R;
    std::vector<uint64_t>& alpha = st.alpha;
    std::vector<uint64_t>& out_rt = st.out_rt;
    auto ceno_hip_tower_num_vars = [&](const ceno_hip_tower* t) { return st.nv_of(t); };
    {
        st.msg_off += (size_t)round * 3 * 2;
        // evaluations are bound into the transcript before r_merge is sampled (cpu/mod.rs:498-531)
        int cursor = 1;
        for (int i = 0; i < n_prod; i++) {
            uint64_t* dst = out->prod_evals + 2 * ((size_t)(i * R + (round - 1)) * 2);
            if (ceno_hip_tower_num_vars(prod[i]) <= round) { memset(dst, 0, 32); continue; }
            for (int k = 0; k < 2; k++) {
                memcpy(dst + 2 * k, fin + 2 * (cursor + k), 16);
                tr_ext(tr, dst + 2 * k);
            }
            cursor += 2;
        }
        for (int i = 0; i < n_logup; i++) {
            uint64_t* dst = out->logup_evals + 2 * ((size_t)(i * R + (round - 1)) * 4);
            if (ceno_hip_tower_num_vars(logup[i]) <= round) { memset(dst, 0, 64); continue; }
            for (int k = 0; k < 4; k++) {
                memcpy(dst + 2 * k, fin + 2 * (cursor + k), 16);
                tr_ext(tr, dst + 2 * k);
            }
            cursor += 4;
        }
        tr_label(tr, "merge");                       // cpu/mod.rs:534
        E2 r_merge = tr_sample(tr);
        memcpy(out_rt.data(), chal, (size_t)16 * round);          // rt' = challenges || r_merge (cpu/mod.rs:535)
        out_rt[2 * round] = r_merge.c0;
        out_rt[2 * round + 1] = r_merge.c1;
        tr_challenge_pows(tr, st.n_alpha, alpha);    // cpu/mod.rs:538-541
        // What layer round + 1 will prove: the alpha-combination (new powers) of this layer's tables merged at r_merge — a tower's layer as one
        // vector is [first half | second half], r_merge binds the top variable (the sum TowerVerify forms, scheme/verifier.rs:1587-1680).
        E2 claim = gl::e2_zero();
        cursor = 1;
        for (int i = 0; i < n_prod; i++) {
            const int nv = ceno_hip_tower_num_vars(prod[i]);
            if (nv <= round) continue;
            const E2 a{fin[2 * cursor], fin[2 * cursor + 1]}, b{fin[2 * cursor + 2], fin[2 * cursor + 3]};
            if (nv > round + 1) claim = claim + E2{alpha[2 * i], alpha[2 * i + 1]} * (a + r_merge * (b - a));
            cursor += 2;
        }
        for (int i = 0; i < n_logup; i++) {
            const int nv = ceno_hip_tower_num_vars(logup[i]);
            if (nv <= round) continue;
            auto f = [&](int k) { return E2{fin[2 * (cursor + k)], fin[2 * (cursor + k) + 1]}; };
            if (nv > round + 1) {
                const E2 an{alpha[2 * (n_prod + 2 * i)], alpha[2 * (n_prod + 2 * i) + 1]}, ad{alpha[2 * (n_prod + 2 * i + 1)], alpha[2 * (n_prod + 2 * i + 1) + 1]};
                claim = claim + an * (f(0) + r_merge * (f(1) - f(0))) + ad * (f(2) + r_merge * (f(3) - f(2)));
            }
            cursor += 4;
        }
        st.claim = claim;
        st.have_claim = true;
    }
    st.round++;
    return 0;
}

void tower_state_finish(TowerProveState& st) { memcpy(st.out->point, st.out_rt.data(), (size_t)16 * st.max_nv); }

int prover_tower_create_proof_hooked(ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup, int n_logup,
                                     ceno_transcript* tr, ceno_hip_stream s, ceno_tower_proof* out, const TowerDistHook* hook) {
    TowerProveState st;
    if (int rc = tower_state_init(st, ctx, prod, n_prod, logup, n_logup, tr, s, out, hook)) return rc;
    while (!st.done())
        if (int rc = tower_state_step(st)) return rc;
    tower_state_finish(st);
    return 0;
}

extern "C" {
int ceno_prover_prove_tower_relation(ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup, int n_logup,
                                     ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_evals, ceno_tower_proof* out) {
    if (!ctx || !tr || !out || !out_evals) return fail(CENO_HIP_ERR_INVALID, "NULL argument");
    // bind read/write/lookup out-evals into the transcript first (cpu/mod.rs:783-786)
    uint64_t* cur = out_evals;
    for (int i = 0; i < n_prod; i++) {
        int rc = ceno_hip_tower_out_evals(ctx, prod[i], cur, s);
        if (rc) return fail_from_ctx(ctx, rc);
        cur += 4;
    }
    for (int i = 0; i < n_logup; i++) {
        int rc = ceno_hip_tower_out_evals(ctx, logup[i], cur, s);
        if (rc) return fail_from_ctx(ctx, rc);
        cur += 8;
    }
    for (uint64_t* e = out_evals; e < cur; e += 2) tr_ext(tr, e);
    return ceno_prover_tower_create_proof(ctx, prod, n_prod, logup, n_logup, tr, s, out);
}

// BooleanHypercube::get_rotation_points (gkr_iop/src/gkr/booleanhypercube.rs:117-168)
static void rotation_points(const uint64_t* point, int n, int log2, uint64_t* left, uint64_t* right) {
    auto get = [&](int i) { return E2{point[2 * i], point[2 * i + 1]}; };
    auto put = [](uint64_t* dst, int i, E2 v) { dst[2 * i] = v.c0; dst[2 * i + 1] = v.c1; };
    put(left, 0, gl::e2_zero());
    put(right, 0, gl::e2_one());
    for (int i = 1; i < n; i++) {
        E2 v = (i <= log2 - 1) ? get(i - 1) : get(i);
        put(left, i, v);
        put(right, i, v);
    }
    if (log2 == 5 && n > 2) put(right, 2, gl::e2_one() - get(1));  // (1, r0, 1-r1, r2, r3, r5, ...)
    if (log2 == 6 && n > 1) put(right, 1, gl::e2_one() - get(0));  // (1, 1-r0, r1, ..., r4, r6, ...)
}

int ceno_prover_prove_rotation(ceno_hip_ctx* ctx, ceno_hip_mle* const* wit, const int* source_idx, const int* target_idx, int n_pairs,
                               int cyclic_subgroup_size, int cyclic_group_log2, const uint64_t* rt, int n, ceno_transcript* tr,
                               ceno_hip_stream s, uint64_t* out_msgs, uint64_t* out_evals, uint64_t* out_origin, uint64_t* out_left,
                               uint64_t* out_right) {
    if (!ctx || !wit || !source_idx || !target_idx || !rt || !tr || n_pairs < 1) return fail(CENO_HIP_ERR_INVALID, "bad rotation arguments");
    if (cyclic_group_log2 != 5 && cyclic_group_log2 != 6) return fail(CENO_HIP_ERR_INVALID, "cyclic group log2 must be 5 or 6");
    std::vector<ceno_hip_mle*> rotated(n_pairs, nullptr);
    ceno_hip_mle* sel = nullptr;
    auto cleanup = [&]() {
        for (auto* m : rotated) if (m) ceno_hip_mle_free(ctx, m);
        if (sel) ceno_hip_mle_free(ctx, sel);
    };
    int rc = ceno_hip_rotation_selector_build(ctx, rt, n, cyclic_subgroup_size, cyclic_group_log2, s, &sel);   // cpu/mod.rs:266-283
    for (int j = 0; j < n_pairs && !rc; j++) rc = ceno_hip_rotation_next_base_mle(ctx, wit[source_idx[j]], cyclic_group_log2, s, &rotated[j]);
    if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
    std::vector<uint64_t> alpha;
    tr_challenge_pows(tr, n_pairs, alpha);                                                                     // cpu/mod.rs:288-292
    // mles [rot_0, tgt_0, rot_1, tgt_1, ..., selector]; sel * sum_j alpha^j (rot_j - tgt_j)
    std::vector<ceno_hip_mle*> mles;
    std::vector<uint64_t> coeffs;
    std::vector<uint32_t> toff{0}, tidx, gterms;
    for (int j = 0; j < n_pairs; j++) {
        mles.push_back(rotated[j]);
        mles.push_back(wit[target_idx[j]]);
        E2 a{alpha[2 * j], alpha[2 * j + 1]}, na = gl::e2_neg(a);
        coeffs.insert(coeffs.end(), {a.c0, a.c1, na.c0, na.c1});
        tidx.push_back(2 * j);     toff.push_back((uint32_t)tidx.size()); gterms.push_back(2 * j);
        tidx.push_back(2 * j + 1); toff.push_back((uint32_t)tidx.size()); gterms.push_back(2 * j + 1);
    }
    mles.push_back(sel);
    std::vector<uint32_t> goff{0, (uint32_t)gterms.size()}, coff{0, 1}, cidx{(uint32_t)(2 * n_pairs)};
    ceno_hip_sumcheck_plan plan{};
    plan.num_mles = (int)mles.size();
    plan.num_terms = 2 * n_pairs;
    plan.term_coeffs = coeffs.data();
    plan.term_offsets = toff.data();
    plan.term_mle_idx = tidx.data();
    plan.num_groups = 1;
    plan.group_term_offsets = goff.data();
    plan.group_term_idx = gterms.data();
    plan.common_offsets = coff.data();
    plan.common_mle_idx = cidx.data();
    plan.max_num_vars = n;
    plan.max_degree = 2;
    std::vector<uint64_t> fin(2 * mles.size());
    rc = ceno_prover_sumcheck_prove(ctx, mles.data(), &plan, tr, s, out_msgs, out_origin, fin.data());
    if (rc) { cleanup(); return rc; }
    rotation_points(out_origin, n, cyclic_group_log2, out_left, out_right);                                   // cpu/mod.rs:341-342
    const int kk = cyclic_group_log2 - 1;
    const E2 rk{out_origin[2 * kk], out_origin[2 * kk + 1]};
    const E2 rk_inv = gl::e2_inv(rk);
    for (int j = 0; j < n_pairs; j++) {
        uint64_t le[2];
        rc = ceno_hip_mle_evaluate(ctx, wit[source_idx[j]], out_left, le, s);                                 // cpu/mod.rs:350-355
        if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
        const E2 left{le[0], le[1]}, rot{fin[4 * j], fin[4 * j + 1]}, target{fin[4 * j + 2], fin[4 * j + 3]};
        const E2 right = (rot - (gl::e2_one() - rk) * left) * rk_inv;                                          // booleanhypercube.rs:170-186
        uint64_t* e = out_evals + 6 * j;
        e[0] = left.c0; e[1] = left.c1; e[2] = right.c0; e[3] = right.c1; e[4] = target.c0; e[5] = target.c1;
    }
    for (int j = 0; j < 3 * n_pairs; j++) tr_ext(tr, out_evals + 2 * j);                                       // cpu/mod.rs:377
    cleanup();
    return 0;
}

}  // extern "C"

// The same argument over row-sharded columns (ceno_dist_create_chip_proof; layout: tower_hook.hpp RotationShard, dist_gkr.cpp).  The rotation
// pairs rows inside blocks of 2^log2 <= 2^q rows, which the block layout keeps on one rank: every rank rotates its own tables.  The selector is
// eq(x, rt) masked on the low log2 bits of x, so a rank's selector is the same construction at the point without the rank coordinates, times
// the scalar eq(rank, rt[q .. q + k)) — carried by the coefficients.  The first q rounds run on the local tables (two partial evaluations per
// round, summed over the ranks), then the folded tables — 2^(n - k - q) entries per rank — are gathered into the global tables (rank bits
// lowest) and the remaining n - q rounds run replicated; the left evaluations are sums of per-rank evaluations weighted by eq over the rank
// coordinates of the left point.  Messages, challenges, points and evaluations are those of ceno_prover_prove_rotation on the whole columns.
int prover_prove_rotation_sharded(ceno_hip_ctx* ctx, ceno_hip_mle* const* wit, const int* source_idx, const int* target_idx, int n_pairs,
                                  int cyclic_subgroup_size, int cyclic_group_log2, const uint64_t* rt, int n, ceno_transcript* tr, ceno_hip_stream s,
                                  uint64_t* out_msgs, uint64_t* out_evals, uint64_t* out_origin, uint64_t* out_left, uint64_t* out_right,
                                  const RotationShard* sh) {
    if (!ctx || !wit || !source_idx || !target_idx || !rt || !tr || n_pairs < 1 || !sh || !sh->allgather) return fail(CENO_HIP_ERR_INVALID, "bad rotation arguments");
    if (cyclic_group_log2 != 5 && cyclic_group_log2 != 6) return fail(CENO_HIP_ERR_INVALID, "cyclic group log2 must be 5 or 6");
    const int W = sh->world, k = sh->k, q = sh->q, n_loc = n - k;
    if (q < cyclic_group_log2 || n_loc < q + 1) return fail(CENO_HIP_ERR_INVALID, "sharded rotation: the row blocks must hold whole cyclic groups and every rank at least two blocks");
    auto point = [&](const uint64_t* p, int j) { return E2{p[2 * j], p[2 * j + 1]}; };
    auto eq_rank = [&](const uint64_t* p) {  // eq(rank, p[q .. q + k))
        E2 v = gl::e2_one();
        for (int j = 0; j < k; j++) v = v * (((sh->rank >> j) & 1) ? point(p, q + j) : gl::e2_one() - point(p, q + j));
        return v;
    };
    auto local_point = [&](const uint64_t* p, std::vector<uint64_t>& out) {
        out.clear();
        for (int j = 0; j < n; j++)
            if (j < q || j >= q + k) {
                out.push_back(p[2 * j]);
                out.push_back(p[2 * j + 1]);
            }
    };
    std::vector<ceno_hip_mle*> rotated(n_pairs, nullptr), glob;
    ceno_hip_mle* sel = nullptr;
    ceno_hip_sumcheck *sc = nullptr, *sc2 = nullptr;
    auto cleanup = [&]() {
        if (sc) ceno_hip_sumcheck_free(ctx, sc);
        if (sc2) ceno_hip_sumcheck_free(ctx, sc2);
        for (auto* m : rotated) if (m) ceno_hip_mle_free(ctx, m);
        for (auto* m : glob) if (m) ceno_hip_mle_free(ctx, m);
        if (sel) ceno_hip_mle_free(ctx, sel);
    };
    std::vector<uint64_t> rt_loc;
    local_point(rt, rt_loc);
    const E2 eq_g = eq_rank(rt);
    int rc = ceno_hip_rotation_selector_build(ctx, rt_loc.data(), n_loc, cyclic_subgroup_size, cyclic_group_log2, s, &sel);
    for (int j = 0; j < n_pairs && !rc; j++) rc = ceno_hip_rotation_next_base_mle(ctx, wit[source_idx[j]], cyclic_group_log2, s, &rotated[j]);
    if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
    std::vector<uint64_t> alpha;
    tr_challenge_pows(tr, n_pairs, alpha);
    // mles [rot_0, tgt_0, ..., selector]; sel * sum_j alpha^j (rot_j - tgt_j); the LOCAL plan carries the rank's eq factor in its coefficients
    const int n_mles = 2 * n_pairs + 1;
    std::vector<ceno_hip_mle*> mles;
    std::vector<uint64_t> coeffs, coeffs_loc;
    std::vector<uint32_t> toff{0}, tidx, gterms;
    for (int j = 0; j < n_pairs; j++) {
        mles.push_back(rotated[j]);
        mles.push_back(wit[target_idx[j]]);
        const E2 a{alpha[2 * j], alpha[2 * j + 1]}, na = gl::e2_neg(a), al = a * eq_g, nal = gl::e2_neg(al);
        coeffs.insert(coeffs.end(), {a.c0, a.c1, na.c0, na.c1});
        coeffs_loc.insert(coeffs_loc.end(), {al.c0, al.c1, nal.c0, nal.c1});
        tidx.push_back(2 * j);     toff.push_back((uint32_t)tidx.size()); gterms.push_back(2 * j);
        tidx.push_back(2 * j + 1); toff.push_back((uint32_t)tidx.size()); gterms.push_back(2 * j + 1);
    }
    mles.push_back(sel);
    std::vector<uint32_t> goff{0, (uint32_t)gterms.size()}, coff{0, 1}, cidx{(uint32_t)(2 * n_pairs)};
    ceno_hip_sumcheck_plan plan{};
    plan.num_mles = n_mles;
    plan.num_terms = 2 * n_pairs;
    plan.term_coeffs = coeffs_loc.data();
    plan.term_offsets = toff.data();
    plan.term_mle_idx = tidx.data();
    plan.num_groups = 1;
    plan.group_term_offsets = goff.data();
    plan.group_term_idx = gterms.data();
    plan.common_offsets = coff.data();
    plan.common_mle_idx = cidx.data();
    plan.max_num_vars = n_loc;
    plan.max_degree = 2;
    rc = ceno_hip_sumcheck_begin(ctx, mles.data(), &plan, s, &sc);
    if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
    tr_usize(tr, (uint64_t)n);
    tr_usize(tr, 2);
    uint64_t ch[2] = {0, 0};
    std::vector<uint64_t> all((size_t)W * 4);
    auto publish = [&](int round, const E2* p) {  // the round's message into the proof and the transcript, its challenge out
        uint64_t* msg = out_msgs + (size_t)4 * round;
        for (int e = 0; e < 2; e++) {
            msg[2 * e] = p[e].c0;
            msg[2 * e + 1] = p[e].c1;
            tr_ext(tr, msg + 2 * e);
        }
        tr_label(tr, "Internal round");
        const E2 r = tr_sample(tr);
        ch[0] = r.c0;
        ch[1] = r.c1;
        out_origin[2 * round] = r.c0;
        out_origin[2 * round + 1] = r.c1;
    };
    for (int i = 0; i < q; i++) {  // local rounds: the partial evaluations of every rank, summed
        uint64_t m[4];
        rc = ceno_hip_sumcheck_round(ctx, sc, i == 0 ? nullptr : ch, m);
        if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
        if (int rc2 = sh->allgather(sh->self, m, 4, all.data())) { cleanup(); return rc2; }
        E2 p[2] = {gl::e2_zero(), gl::e2_zero()};
        for (int g = 0; g < W; g++)
            for (int e = 0; e < 2; e++) p[e] = p[e] + E2{all[(size_t)g * 4 + 2 * e], all[(size_t)g * 4 + 2 * e + 1]};
        publish(i, p);
    }
    // the folded tables (as the next round would read them: folded q - 1 times; once more here), gathered with the rank bits lowest
    const size_t len_loc = (size_t)1 << (n_loc - q);
    const int nv_rem = n - q;
    const E2 r_last{ch[0], ch[1]};
    std::vector<E2> t(2 * len_loc), g_tab(len_loc * (size_t)W);
    std::vector<uint64_t> mine(2 * len_loc), gathered((size_t)W * 2 * len_loc);
    glob.assign((size_t)n_mles, nullptr);
    for (int mi = 0; mi < n_mles; mi++) {
        int nv = 0;
        rc = ceno_hip_sumcheck_table_host(ctx, sc, mi, reinterpret_cast<uint64_t*>(t.data()), t.size(), &nv);
        if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
        if (nv != n_loc - q + 1) { cleanup(); return fail(CENO_HIP_ERR_STATE, "sharded rotation: unexpected table shape after the local rounds"); }
        const bool is_sel = mi == n_mles - 1;  // (its rank factor went into the coefficients: the global selector carries it)
        for (size_t j = 0; j < len_loc; j++) {
            E2 v = t[2 * j] + r_last * (t[2 * j + 1] - t[2 * j]);
            if (is_sel) v = v * eq_g;
            mine[2 * j] = v.c0;
            mine[2 * j + 1] = v.c1;
        }
        if (int rc2 = sh->allgather(sh->self, mine.data(), 2 * len_loc, gathered.data())) { cleanup(); return rc2; }
        for (int g = 0; g < W; g++) {
            const E2* src = reinterpret_cast<const E2*>(gathered.data()) + (size_t)g * len_loc;
            for (size_t j = 0; j < len_loc; j++) g_tab[(j << k) | (size_t)g] = src[j];
        }
        rc = ceno_hip_mle_upload(ctx, reinterpret_cast<const uint64_t*>(g_tab.data()), nv_rem, 1, s, &glob[(size_t)mi]);
        if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
    }
    ceno_hip_sumcheck_free(ctx, sc);
    sc = nullptr;
    // the remaining rounds, replicated
    plan.term_coeffs = coeffs.data();
    plan.max_num_vars = nv_rem;
    rc = ceno_hip_sumcheck_begin(ctx, glob.data(), &plan, s, &sc2);
    if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
    for (int i = 0; i < nv_rem; i++) {
        uint64_t m[4];
        rc = ceno_hip_sumcheck_round(ctx, sc2, i == 0 ? nullptr : ch, m);
        if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
        const E2 p[2] = {E2{m[0], m[1]}, E2{m[2], m[3]}};
        publish(q + i, p);
    }
    std::vector<uint64_t> fin(2 * (size_t)n_mles);
    rc = ceno_hip_sumcheck_finish(ctx, sc2, nv_rem > 0 ? ch : nullptr, fin.data());
    if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
    rotation_points(out_origin, n, cyclic_group_log2, out_left, out_right);
    const int kk = cyclic_group_log2 - 1;
    const E2 rk{out_origin[2 * kk], out_origin[2 * kk + 1]};
    const E2 rk_inv = gl::e2_inv(rk);
    // left evaluations: every rank evaluates its rows at the left point without the rank coordinates, weighted by eq over them
    std::vector<uint64_t> left_loc;
    local_point(out_left, left_loc);
    const E2 eq_left = eq_rank(out_left);
    std::vector<uint64_t> part(2 * (size_t)n_pairs), parts((size_t)W * 2 * n_pairs);
    for (int j = 0; j < n_pairs; j++) {
        uint64_t le[2];
        rc = ceno_hip_mle_evaluate(ctx, wit[source_idx[j]], left_loc.data(), le, s);
        if (rc) { cleanup(); return fail_from_ctx(ctx, rc); }
        const E2 v = E2{le[0], le[1]} * eq_left;
        part[2 * j] = v.c0;
        part[2 * j + 1] = v.c1;
    }
    if (int rc2 = sh->allgather(sh->self, part.data(), part.size(), parts.data())) { cleanup(); return rc2; }
    for (int j = 0; j < n_pairs; j++) {
        E2 left = gl::e2_zero();
        for (int g = 0; g < W; g++) left = left + E2{parts[(size_t)g * 2 * n_pairs + 2 * j], parts[(size_t)g * 2 * n_pairs + 2 * j + 1]};
        const E2 rot{fin[4 * j], fin[4 * j + 1]}, target{fin[4 * j + 2], fin[4 * j + 3]};
        const E2 right = (rot - (gl::e2_one() - rk) * left) * rk_inv;
        uint64_t* e = out_evals + 6 * j;
        e[0] = left.c0; e[1] = left.c1; e[2] = right.c0; e[3] = right.c1; e[4] = target.c0; e[5] = target.c1;
    }
    for (int j = 0; j < 3 * n_pairs; j++) tr_ext(tr, out_evals + 2 * j);
    cleanup();
    return 0;
}

extern "C" {
// ---- the eight-lane host arithmetic (csrc/e2_host_avx512.hpp) against the scalar operators: a, b = eight extension elements each (c0, c1
// interleaved); out = a + b | a - b | a * b | fold(a as four pairs... ) — returns 0 when the CPU has no AVX-512 ----
#if defined(__x86_64__)
__attribute__((target("avx512f,avx512dq"))) static void test_e2v_ops(const E2* a, const E2* b, E2* out) {
    using namespace e2v;
    const VE2 va = load(a, 0, 1), vb = load(b, 0, 1);
    store(out, 0, add(va, vb));
    store(out, 8, sub(va, vb));
    store(out, 16, mul(va, vb));
    out[24] = hsum(va);
}
#endif
extern "C" int ceno_prover_test_e2v(const uint64_t* a16, const uint64_t* b16, uint64_t* out50, uint64_t* fold_io32, const uint64_t* r2) {
#if defined(__x86_64__)
    if (!p2host::have_avx512()) return 0;
    test_e2v_ops(reinterpret_cast<const E2*>(a16), reinterpret_cast<const E2*>(b16), reinterpret_cast<E2*>(out50));
    host_tower_fold8(reinterpret_cast<E2*>(fold_io32), 0, E2{r2[0], r2[1]});  // sixteen entries in, the eight folded ones in front
    return 1;
#else
    (void)a16; (void)b16; (void)out50; (void)fold_io32; (void)r2;
    return 0;
#endif
}

// ---- host-side field arithmetic exposed for CPU tests of the shared gl64.hpp code ----
uint64_t ceno_prover_test_gl_mul(uint64_t a, uint64_t b) { return gl::mul(a, b); }
uint64_t ceno_prover_test_gl_add(uint64_t a, uint64_t b) { return gl::add(a, b); }
uint64_t ceno_prover_test_gl_sub(uint64_t a, uint64_t b) { return gl::sub(a, b); }
uint64_t ceno_prover_test_gl_mul_small(uint64_t a, uint32_t c) { return gl::mul_small(a, c); }
void ceno_prover_test_e2_mul(const uint64_t* a, const uint64_t* b, uint64_t* o) {
    E2 r = E2{a[0], a[1]} * E2{b[0], b[1]};
    o[0] = r.c0;
    o[1] = r.c1;
}
uint64_t ceno_prover_test_gl_mul_ref(uint64_t a, uint64_t b) { return gl::mul_ref(a, b); }
uint64_t ceno_prover_test_gl_mul_nc(uint64_t a, uint64_t b) { return gl::canon(gl::mul_nc(a, b)); }
uint64_t ceno_prover_test_gl_mul_add(uint64_t a, uint64_t b, uint64_t c) { return gl::mul_add(a, b, c); }
uint64_t ceno_prover_test_gl_mul_add2(uint64_t a, uint64_t b, uint64_t c, uint64_t d) { return gl::mul_add2(a, b, c, d); }
void ceno_prover_test_e2_mul_ref(const uint64_t* a, const uint64_t* b, uint64_t* o) {
    E2 r = gl::e2_mul_ref(E2{a[0], a[1]}, E2{b[0], b[1]});
    o[0] = r.c0;
    o[1] = r.c1;
}
void ceno_prover_test_e2_mul_pre(const uint64_t* a, const uint64_t* b, uint64_t* o) {
    E2 r = gl::e2_mul_pre(gl::e2_pre(E2{a[0], a[1]}), E2{b[0], b[1]});
    o[0] = r.c0;
    o[1] = r.c1;
}
// non-canonical product (any 64-bit inputs) brought to canonical form for comparison
void ceno_prover_test_e2_mul_nc(const uint64_t* a, const uint64_t* b, uint64_t* o) {
    E2 r = gl::e2_mul_nc(E2{a[0], a[1]}, E2{b[0], b[1]});
    o[0] = gl::canon(r.c0);
    o[1] = gl::canon(r.c1);
}
// acc + r * b, the fold form
void ceno_prover_test_e2_fma_pre(const uint64_t* r, const uint64_t* b, const uint64_t* a, uint64_t* o) {
    E2 v = gl::e2_fma_pre(gl::e2_pre(E2{r[0], r[1]}), E2{b[0], b[1]}, E2{a[0], a[1]});
    o[0] = v.c0;
    o[1] = v.c1;
}
// sum_i a_i * b_i through the unreduced 160-bit accumulators; `reps` repeats the list to push the top limb
void ceno_prover_test_e2_acc(const uint64_t* a, const uint64_t* b, int n, int reps, uint64_t* o) {
    gl::E2Acc acc = gl::e2acc_zero();
    for (int k = 0; k < reps; k++)
        for (int i = 0; i < n; i++) gl::e2acc_mac(acc, E2{a[2 * i], a[2 * i + 1]}, E2{b[2 * i], b[2 * i + 1]});
    E2 v = gl::e2acc_reduce(acc);
    o[0] = v.c0;
    o[1] = v.c1;
}
void ceno_prover_test_e2_inv(const uint64_t* a, uint64_t* o) {
    E2 r = gl::e2_inv(E2{a[0], a[1]});
    o[0] = r.c0;
    o[1] = r.c1;
}

}  // extern "C"
