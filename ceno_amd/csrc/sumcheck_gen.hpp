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
// common factor k sits in a base-field row).  eq = 1: the group's ONE common factor was declared as eq(., point) restricted to the rows
// [lo, hi) (ceno_hip_sumcheck_begin_eq) and the component runs in eq-factored form (below): `brow` = the group's pair of rows in the
// launch's boundary block.
struct alignas(16) GenGroup {
    uint32_t term_begin, term_end, n_common, base_mask;
    uint64_t common8;
    uint32_t eq, brow;
    uint64_t lo, hi;
    uint64_t pad[2];
};
static_assert(sizeof(GenGroup) == 64, "GenGroup layout");
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
    // ---- eq-factored form (sumcheck.hip "eq-factored main-constraint rounds") ----
    uint32_t eqf;              // bit 0: every group of the component is an eq group; bit 1: all D slots are wanted (the first round)
    uint32_t wg_begin, wg_count;  // component-aligned launch: the workgroups [wg_begin, wg_begin + wg_count) own this component's tiles
    uint32_t eq_slot;          // the component's index among the eq components of the sumcheck
    uint32_t shift;            // round index i: an entry of this round's tables stands for 2^i rows of the input tables
    uint32_t p2_tile_begin, p2_tile_end;  // eq components: only these tiles hold pairs inside a group's row range (the rest is folded only)
    E2 rt;                     // coordinate i of the component's eq point
    E2 inv1m;                  // 1 / (1 - rt)
};
static_assert(sizeof(GenComp) == 128, "GenComp layout");
// what an eq-factored launch needs besides the component list
struct GenEqArgs {
    int aligned;               // component-aligned workgroup mapping (wg_begin / wg_count valid)
    E2* q_out;                 // host-mapped: [eq_slot][D] per-component quotient sums (armed by the host)
    E2* b_out;                 // host-mapped: [brow + side][D] scaled values of a boundary pair (armed by the host where one exists)
    unsigned* counters;        // device: arrival counter per eq_slot (components with several workgroups), zero between launches
    const uint16_t* wg_comp;   // device: the component of every workgroup of the launch (NULL: walk the list)
    unsigned stride;           // entries per row of q_out / b_out (the sumcheck's message length: a launch of LOWER degree — the components whose
                               // own polynomial is shorter, round 5 — fills the first D' entries of the same rows)
};

// LDS of a launch: fixed block + staged rows + the cross-wave exchange of partial group sums (4 waves x D points x 64 lanes)
size_t gen_lds_bytes(int d, size_t stage_bytes);
// max_grid: most workgroups the launch may use (the component-aligned mapping was laid out for exactly that many); 0: one per tile up to
// what is resident at once.  gen_resident_cap reports that number for a (degree, layout, LDS) combination.
unsigned gen_resident_cap(ceno_hip_ctx* ctx, int d, bool base0, size_t stage_bytes);
// the first round of an eq-factored batch over base-field columns without the LDS stage (k_eq_base0): component list of the third layout
// (factor bytes = MLE indices), component-aligned grid
unsigned eq_base0_resident_cap(ceno_hip_ctx* ctx, int d);
void launch_eq_base0(ceno_hip_ctx* ctx, int d, const GenComp* comps, int n_comps, const Epilogue& ep, const GenEqArgs& eq, unsigned grid, hipStream_t st);
// the small rounds of an eq-factored batch (every component one tile, d >= 3): one workgroup per component and slot (sumcheck_gen.hip k_gen_eq_slots)
void launch_gen_eq_slots(int d, const GenComp* comps, int n_comps, E2 r, const Epilogue& ep, size_t stage_bytes, hipStream_t st, const GenEqArgs& eq,
                          unsigned grid);
void launch_gen(ceno_hip_ctx* ctx, int d, bool base0, const GenComp* comps, int n_comps, unsigned total_tiles, E2 r, const Epilogue& ep, size_t stage_bytes,
                hipStream_t st, const GenEqArgs* eq = nullptr, unsigned aligned_grid = 0);
