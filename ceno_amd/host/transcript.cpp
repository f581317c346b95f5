// Transcripts for the host prover loops.
//  - stub: SplitMix64-chained state, data dependent (every absorbed word changes all later
//    challenges) — a deterministic stand-in so that proofs can be compared end to end without the
//    reference's Poseidon2 parameters.
//  - poseidon2: duplex challenger over Goldilocks (poseidon2_host.cpp); PARITY UNPINNED.
#include "transcript.hpp"

#include <cstring>

#include "../csrc/gl64.hpp"

namespace {

inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct Stub {
    uint64_t s;
    void absorb(uint64_t w) { s = mix64(s ^ w); }
};

void stub_label(void* self, const uint8_t* bytes, size_t n) {
    auto* st = (Stub*)self;
    st->absorb(0x4c41424c00000000ULL | (uint64_t)n);  // "LABL" | len
    for (size_t i = 0; i < n; i += 8) {
        uint64_t w = 0;
        for (size_t k = 0; k < 8 && i + k < n; k++) w |= (uint64_t)bytes[i + k] << (8 * k);
        st->absorb(w);
    }
}
void stub_ext(void* self, const uint64_t* e) {
    auto* st = (Stub*)self;
    st->absorb(e[0]);
    st->absorb(e[1]);
}
void stub_sample(void* self, uint64_t* o) {
    auto* st = (Stub*)self;
    st->s = mix64(st->s);
    o[0] = st->s >= gl::P ? st->s - gl::P : st->s;
    st->s = mix64(st->s);
    o[1] = st->s >= gl::P ? st->s - gl::P : st->s;
}
void stub_destroy(void* self) { delete (Stub*)self; }
void stub_base(void* self, uint64_t v) {
    auto* st = (Stub*)self;
    st->absorb(0x4241534500000000ULL);  // "BASE"
    st->absorb(v);
}
uint64_t stub_sample_bits(void* self, int bits) {
    auto* st = (Stub*)self;
    st->s = mix64(st->s);
    const uint64_t v = st->s >= gl::P ? st->s - gl::P : st->s;
    return bits >= 64 ? v : v & (((uint64_t)1 << bits) - 1);
}
void* stub_fork(void* self) { return new Stub(*(Stub*)self); }

}  // namespace

extern "C" ceno_transcript* ceno_transcript_stub_new(uint64_t seed) {
    auto* t = new ceno_transcript();
    t->self = new Stub{mix64(seed)};
    t->append_label = stub_label;
    t->append_ext = stub_ext;
    t->sample_ext = stub_sample;
    t->destroy = stub_destroy;
    t->append_base = stub_base;
    t->sample_bits = stub_sample_bits;
    t->fork = stub_fork;
    t->fork_free = stub_destroy;
    t->export_state = nullptr;  // not a duplex sponge: grinding runs through fork / append_base / sample_bits
    t->import_state = nullptr;
    t->grind = nullptr;
    return t;
}

// ------------------------------------------------------------------------------------------------
// Poseidon2 duplex challenger over Goldilocks (width 8, rate 4) — the shape of p3-challenger's
// DuplexChallenger that the reference's EXT `transcript::BasicTranscript` wraps.  PARITY UNPINNED
// (SURVEY.md §8c(i)): round constants are placeholders (csrc/poseidon2.hpp) and the byte -> field packing
// of labels (`bytes_to_field_elements`, EXT ff_ext) is ASSUMED to be 8 little-endian bytes per element,
// by analogy with the 4-byte packing the in-tree BabyBear restatement uses
// (ceno_recursion_v2/src/utils.rs:44-67).
// ------------------------------------------------------------------------------------------------
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../csrc/poseidon2.hpp"
#include "../csrc/poseidon2_host.hpp"

namespace {

struct Duplex {
    p2::Params params;
    uint64_t state[p2::WIDTH] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint64_t> in, out;

    void duplexing() {
        for (size_t i = 0; i < in.size(); i++) state[i] = in[i];
        in.clear();
        p2host::permute(state, params);
        out.assign(state, state + p2::RATE);
    }
    void observe(uint64_t v) {
        out.clear();
        in.push_back(v);
        if ((int)in.size() == p2::RATE) duplexing();
    }
    uint64_t sample() {
        if (!in.empty() || out.empty()) duplexing();
        uint64_t v = out.back();
        out.pop_back();
        return v;
    }
};

void dx_label(void* self, const uint8_t* bytes, size_t n) {
    auto* d = (Duplex*)self;
    for (size_t i = 0; i < n; i += 8) {
        uint64_t w = 0;
        for (size_t k = 0; k < 8 && i + k < n; k++) w |= (uint64_t)bytes[i + k] << (8 * k);
        d->observe(w % gl::P);
    }
}
void dx_ext(void* self, const uint64_t* e) {
    auto* d = (Duplex*)self;
    d->observe(e[0]);
    d->observe(e[1]);
}
void dx_sample(void* self, uint64_t* o) {
    auto* d = (Duplex*)self;
    o[0] = d->sample();
    o[1] = d->sample();
}
void dx_destroy(void* self) { delete (Duplex*)self; }
void dx_base(void* self, uint64_t v) { ((Duplex*)self)->observe(v % gl::P); }
uint64_t dx_sample_bits(void* self, int bits) {
    const uint64_t v = ((Duplex*)self)->sample();  // canonical
    return bits >= 64 ? v : v & (((uint64_t)1 << bits) - 1);
}
void* dx_fork(void* self) { return new Duplex(*(Duplex*)self); }
// [sponge state 8][n pending inputs][pending inputs 4][n outputs left][0][0] (include/ceno_prover.h)
int dx_export(void* self, uint64_t* o) {
    auto* d = (Duplex*)self;
    memset(o, 0, 16 * 8);
    memcpy(o, d->state, 64);
    o[8] = d->in.size();
    for (size_t i = 0; i < d->in.size(); i++) o[9 + i] = d->in[i];
    o[13] = d->out.size();
    return CENO_TRANSCRIPT_DUPLEX8;
}
int dx_import(void* self, const uint64_t* in16) {
    auto* d = (Duplex*)self;
    if (in16[8] >= (uint64_t)p2::RATE || in16[13] > (uint64_t)p2::RATE) return CENO_HIP_ERR_INVALID;
    for (int i = 0; i < 13; i++)
        if (i != 8 && in16[i] >= gl::P) return CENO_HIP_ERR_INVALID;
    if (in16[8] != 0 && in16[13] != 0) return CENO_HIP_ERR_INVALID;  // observing clears the output buffer
    memcpy(d->state, in16, 64);
    d->in.assign(in16 + 9, in16 + 9 + in16[8]);
    d->out.assign(d->state, d->state + in16[13]);  // the output buffer is always a prefix of the state
    return 0;
}

}  // namespace

// process-wide parameter table of the host challenger (mirrors ceno_hip_poseidon2_set_constants on the device side)
namespace {
std::mutex g_params_mu;
p2::Params g_params;
bool g_params_set = false, g_params_pinned = false;
const p2::Params& host_params(bool warn) {
    std::lock_guard<std::mutex> g(g_params_mu);
    if (!g_params_set) {
        p2::default_params(g_params);
        g_params_set = true;
    }
    static bool warned = false;
    if (warn && !g_params_pinned && !warned && !getenv("CENO_HIP_QUIET_PLACEHOLDER")) {
        warned = true;
        fprintf(stderr, "[ceno_prover] WARNING: Poseidon2 transcript runs on placeholder round constants - challenges are NOT those of the "
                        "reference's BasicTranscript (PARITY UNPINNED); load the real table with ceno_transcript_poseidon2_set_constants\n");
    }
    return g_params;
}
}  // namespace

extern "C" int ceno_transcript_poseidon2_set_constants(const uint64_t* external_rc, const uint64_t* internal_rc, const uint64_t* internal_diag) {
    p2::Params p;
    p2::default_params(p);
    if (external_rc) memcpy(p.ext_rc, external_rc, sizeof(p.ext_rc));
    if (internal_rc) memcpy(p.int_rc, internal_rc, sizeof(p.int_rc));
    if (internal_diag) memcpy(p.int_diag, internal_diag, sizeof(p.int_diag));
    for (size_t i = 0; i < sizeof(p) / 8; i++)
        if (reinterpret_cast<const uint64_t*>(&p)[i] >= gl::P) return CENO_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> g(g_params_mu);
    g_params = p;
    g_params_set = true;
    g_params_pinned = external_rc && internal_rc && internal_diag;
    return 0;
}
extern "C" int ceno_transcript_poseidon2_is_pinned(void) {
    std::lock_guard<std::mutex> g(g_params_mu);
    return g_params_pinned ? 1 : 0;
}

extern "C" ceno_transcript* ceno_transcript_poseidon2_new(const uint8_t* label, size_t n) {
    static const bool strict = [] { const char* e = getenv("CENO_HIP_REQUIRE_PINNED_POSEIDON2"); return e && atoi(e) != 0; }();
    if (strict && !ceno_transcript_poseidon2_is_pinned()) return nullptr;  // production callers: no placeholder challenges
    auto* d = new Duplex();
    d->params = host_params(true);
    auto* t = new ceno_transcript();
    t->self = d;
    t->append_label = dx_label;
    t->append_ext = dx_ext;
    t->sample_ext = dx_sample;
    t->destroy = dx_destroy;
    t->append_base = dx_base;
    t->sample_bits = dx_sample_bits;
    t->fork = dx_fork;
    t->fork_free = dx_destroy;
    t->export_state = dx_export;
    t->import_state = dx_import;
    t->grind = nullptr;  // the device search (ceno_prover_transcript_grind) serves this challenger
    if (label && n) dx_label(d, label, n);  // BasicTranscript::new(label) absorbs the label
    return t;
}

// label packing as the host challenger does it (tests/test_ref_goldens.py compares it with the reference's bytes_to_field_elements)
extern "C" int ceno_prover_test_label_to_field(const uint8_t* bytes, size_t n, uint64_t* out, int max_out) {
    int k = 0;
    for (size_t i = 0; i < n; i += 8) {
        uint64_t w = 0;
        for (size_t j = 0; j < 8 && i + j < n; j++) w |= (uint64_t)bytes[i + j] << (8 * j);
        if (k < max_out) out[k] = w % gl::P;
        k++;
    }
    return k;
}

// host permutation for tests of the shared poseidon2.hpp source
extern "C" void ceno_prover_test_poseidon2_permute(uint64_t* state8) {
    p2::permute(state8, host_params(false));
}
extern "C" void ceno_prover_test_poseidon2_permute_fast(uint64_t* state8) { p2host::permute(state8, host_params(false)); }
// n independent states through p2host::permute_many (eight at a time on AVX-512 CPUs): returns 1 when the vector path exists on this CPU
extern "C" int ceno_prover_test_poseidon2_permute_many(uint64_t* states, size_t n) {
    p2host::permute_many(states, n, host_params(false));
    return p2host::have_avx512() ? 1 : 0;
}
// n chained permutations (timing of the host challenger's critical path without binding overhead)
extern "C" void ceno_prover_test_poseidon2_chain(uint64_t* state8, int n, int fast) {
    const p2::Params& p = host_params(false);
    for (int i = 0; i < n; i++) {
        if (fast) p2host::permute(state8, p);
        else p2::permute(state8, p);
    }
}
