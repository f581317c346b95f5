// Plan tables of the LDS-blocked generic round kernel (sumcheck_gen.hip), built by the host side in sumcheck.hip.
#pragma once
#include "sumcheck_dev.hpp"

static constexpr unsigned GEN_PAD = 2;            // row padding of the stage, in 16-byte units (rows start on different banks)
static constexpr unsigned GEN_MAX_UNITS = 256;   // stage rows are addressed by one byte

// one monomial term: coefficient and the stage rows ("units", < 256) of its own factors, one byte each
struct alignas(16) GenTerm {
    E2 c;
    uint32_t nf;
    uint32_t pad;
    uint64_t idx8;
};
static_assert(sizeof(GenTerm) == 32, "GenTerm layout");
// a group: consecutive terms + the stage rows of the factors common to all of them (base_mask: first-round layout, bit k =
// common factor k sits in a base-field row)
struct alignas(16) GenGroup {
    uint32_t term_begin, term_end, n_common, base_mask;
    uint64_t common8;
    uint64_t pad;
};
static_assert(sizeof(GenGroup) == 32, "GenGroup layout");
// one connected component of a size class's plan in one round
struct alignas(16) GenComp {
    const MleSlot* slots;      // this round's tables of the component's MLEs
    const GenTerm* terms;
    const GenGroup* groups;
    const uint16_t* unit;      // stage row of every MLE (extension rows take two units, first-round base rows one)
    unsigned long long pairs;
    uint32_t n_mles, n_groups;
    uint32_t tile_begin, n_tiles;
    uint32_t tp_log;           // pairs per tile (16 .. 256)
    uint32_t wt_log;           // waves that share the terms of a group (1, 2 or 4); 256 / wt pairs are evaluated per pass
    uint32_t fold;             // 1: fold `in` with the challenge into `out` first (every round but the first)
    uint32_t pad;
};
static_assert(sizeof(GenComp) == 80, "GenComp layout");

// LDS of a launch: fixed block + staged rows + the cross-wave exchange of partial group sums (4 waves x D points x 64 lanes)
size_t gen_lds_bytes(int d, size_t stage_bytes);
void launch_gen(ceno_hip_ctx* ctx, int d, bool base0, const GenComp* comps, int n_comps, unsigned total_tiles, E2 r, const Epilogue& ep, size_t stage_bytes,
                hipStream_t st);
