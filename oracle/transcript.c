/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  Base-field side of the Fiat-Shamir challenger and the Poseidon2
 * duplex challenger itself.
 *
 * PARITY UNPINNED: the reference's `BasicTranscript<GoldilocksExt2>` is in an EXT crate
 * (scroll-tech/gkr-backend v1.0.0-alpha.35 `transcript`, over p3-challenger 0.4.3 `DuplexChallenger`); Poseidon2 round
 * constants and the byte -> field packing of labels are placeholders / assumptions until goldens exist.  What IS stated in
 * tree and followed here line by line:
 *   ceno_recursion_v2/src/pcs/mod.rs:8164-8204  sample_bits = low bits of the canonical value of ONE base sample
 *   ceno_recursion_v2/src/pcs/mod.rs:8125-8155  check_witness = observe(witness); sample_bits(bits) == 0
 * and the published p3-challenger 0.4.3 duplex rules:
 *   observe(v): output buffer cleared, v pushed to the input buffer, duplexing when RATE inputs are pending
 *   sample():   duplexing if inputs are pending or no output is left; pop from the BACK of the output buffer
 *   duplexing:  the pending inputs OVERWRITE state[0..k), permute, output buffer = state[0..RATE)
 *   sample_ext: D base samples, coefficient 0 first
 *   grind:      any witness for which a CLONE of the challenger passes check_witness; then check_witness on self
 */
#include "oracle.h"
#include "gl64.h"
#include <stdlib.h>
#include <string.h>

uint64_t orc_tr_sample_bits(orc_transcript* t, int bits) { return t->sample_bits(t->self, bits); }
int orc_tr_check_witness(orc_transcript* t, int bits, uint64_t witness) {
    t->append_base(t->self, witness);
    return orc_tr_sample_bits(t, bits) == 0;
}
uint64_t orc_tr_grind(orc_transcript* t, int bits) {
    uint64_t w = 0;
    for (;; w++) {
        orc_transcript c = *t;
        c.self = t->fork(t->self);
        int ok = orc_tr_check_witness(&c, bits, w);
        t->fork_free(c.self);
        if (ok) break;
    }
    (void)orc_tr_check_witness(t, bits, w);
    return w;
}

static void duplexing(orc_duplex_state* d) {
    for (int i = 0; i < d->n_in; i++) d->state[i] = d->in[i];
    d->n_in = 0;
    orc_poseidon2_permute(d->state, d->params);
    d->n_out = 4;
}
void orc_duplex_observe(orc_duplex_state* d, uint64_t v) {
    d->n_out = 0;
    d->in[d->n_in++] = v;
    if (d->n_in == 4) duplexing(d);
}
uint64_t orc_duplex_sample(orc_duplex_state* d) {
    if (d->n_in != 0 || d->n_out == 0) duplexing(d);
    return d->state[--d->n_out];
}
/* label packing: 8 little-endian bytes per element, reduced (ASSUMED by analogy with the 4-byte BabyBear packing of
 * ceno_recursion_v2/src/utils.rs:44-67) */
static void dx_label(void* s, const uint8_t* b, size_t n) {
    for (size_t i = 0; i < n; i += 8) {
        uint64_t w = 0;
        for (size_t k = 0; k < 8 && i + k < n; k++) w |= (uint64_t)b[i + k] << (8 * k);
        orc_duplex_observe((orc_duplex_state*)s, gl_reduce(w));
    }
}
static void dx_ext(void* s, const uint64_t* e) {
    orc_duplex_observe((orc_duplex_state*)s, e[0]);
    orc_duplex_observe((orc_duplex_state*)s, e[1]);
}
static void dx_sample_ext(void* s, uint64_t* o) {
    o[0] = orc_duplex_sample((orc_duplex_state*)s);
    o[1] = orc_duplex_sample((orc_duplex_state*)s);
}
static void dx_base(void* s, uint64_t v) { orc_duplex_observe((orc_duplex_state*)s, gl_reduce(v)); }
static uint64_t dx_sample_bits(void* s, int bits) {
    const uint64_t v = orc_duplex_sample((orc_duplex_state*)s); /* canonical */
    return bits >= 64 ? v : v & (((uint64_t)1 << bits) - 1);
}
static void* dx_fork(void* s) {
    orc_duplex_state* c = malloc(sizeof(*c));
    memcpy(c, s, sizeof(*c));
    return c;
}
void orc_duplex_init(orc_duplex_state* d, const uint64_t* params138, const uint8_t* label, size_t n) {
    memset(d, 0, sizeof(*d));
    memcpy(d->params, params138, sizeof(d->params));
    if (label && n) dx_label(d, label, n); /* BasicTranscript::new(label) absorbs the label */
}
void orc_duplex_bind(orc_transcript* t, orc_duplex_state* d) {
    t->append_label = dx_label; t->append_ext = dx_ext; t->sample_ext = dx_sample_ext; t->self = d;
    t->append_base = dx_base; t->sample_bits = dx_sample_bits; t->fork = dx_fork; t->fork_free = free;
}
