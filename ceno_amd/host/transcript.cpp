// Transcripts for the host prover loops.
//  - stub: SplitMix64-chained state, data dependent (every absorbed word changes all later
//    challenges) — a deterministic stand-in so that proofs can be compared end to end without the
//    reference's Poseidon2 parameters.
//  - poseidon2: duplex challenger over Goldilocks (poseidon2_host.cpp); PARITY UNPINNED.
#include "transcript.hpp"

#include "../csrc/gl64.cuh"

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

}  // namespace

extern "C" ceno_transcript* ceno_transcript_stub_new(uint64_t seed) {
    auto* t = new ceno_transcript();
    t->self = new Stub{mix64(seed)};
    t->append_label = stub_label;
    t->append_ext = stub_ext;
    t->sample_ext = stub_sample;
    t->destroy = stub_destroy;
    return t;
}

// ------------------------------------------------------------------------------------------------
// Poseidon2 duplex challenger over Goldilocks (width 8, rate 4) — the shape of p3-challenger's
// DuplexChallenger that the reference's EXT `transcript::BasicTranscript` wraps.  PARITY UNPINNED
// (SURVEY.md §8c(i)): round constants are placeholders (csrc/poseidon2.cuh) and the byte -> field packing
// of labels (`bytes_to_field_elements`, EXT ff_ext) is ASSUMED to be 8 little-endian bytes per element,
// by analogy with the 4-byte packing the in-tree BabyBear restatement uses
// (ceno_recursion_v2/src/utils.rs:44-67).
// ------------------------------------------------------------------------------------------------
#include <vector>

#include "../csrc/poseidon2.cuh"

namespace {

struct Duplex {
    p2::Params params;
    uint64_t state[p2::WIDTH] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint64_t> in, out;

    void duplexing() {
        for (size_t i = 0; i < in.size(); i++) state[i] = in[i];
        in.clear();
        p2::permute(state, params);
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

}  // namespace

extern "C" ceno_transcript* ceno_transcript_poseidon2_new(const uint8_t* label, size_t n) {
    auto* d = new Duplex();
    p2::default_params(d->params);
    auto* t = new ceno_transcript();
    t->self = d;
    t->append_label = dx_label;
    t->append_ext = dx_ext;
    t->sample_ext = dx_sample;
    t->destroy = dx_destroy;
    if (label && n) dx_label(d, label, n);  // BasicTranscript::new(label) absorbs the label
    return t;
}

// host permutation for tests of the shared poseidon2.cuh source
extern "C" void ceno_prover_test_poseidon2_permute(uint64_t* state8) {
    p2::Params p;
    p2::default_params(p);
    p2::permute(state8, p);
}
