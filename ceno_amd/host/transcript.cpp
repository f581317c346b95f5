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
