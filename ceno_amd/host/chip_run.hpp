// ZKVMProver::create_chip_proof (ceno_zkvm/src/scheme/prover.rs:717-833) in two halves around the tower prover's layers, so that the layers of
// MANY chips can be proved together (cohort.cpp): begin = records, towers, out-evaluations into the transcript, the tower prover's state before
// layer 1; finish = the layers still open, the main point, the rotation argument, the towers released.
#pragma once
#include "../../include/ceno_prover.h"
#include "tower_state.hpp"

struct ChipProofRun {
    ceno_hip_ctx* ctx = nullptr;
    const ceno_chip_task* task = nullptr;
    ceno_transcript* tr = nullptr;
    ceno_chip_proof* out = nullptr;
    ceno_tower_witness tw{};
    TowerProveState st;
    int num_var_with_rotation = 0;
    bool live = false;  // the towers exist
    const uint64_t* challenges4 = nullptr;
    std::vector<ceno_hip_mle*> records;  // between chip_run_records and chip_run_free_records
    std::vector<ceno_hip_mle*> present;  // what the record plan points into (chip_run_records_plan)
    std::vector<uint32_t> ridx;
};
// begin in pieces, for callers that build the towers of MANY chips in one go (cohort.cpp): records -> tower specs -> [ceno_hip_tower_build_many,
// ceno_hip_tower_prefetch_tops] -> adopt -> free records -> after_towers
int chip_run_records(ChipProofRun& run, ceno_hip_ctx* ctx, const ceno_chip_task* task, const uint64_t* challenges4, ceno_transcript* tr,
                     ceno_hip_stream s, ceno_chip_proof* out);
// (the checks and the plan alone: the caller runs it — ceno_hip_wit_infer_many over all chips — before anything else touches the run)
int chip_run_records_plan(ChipProofRun& run, ceno_hip_ctx* ctx, const ceno_chip_task* task, const uint64_t* challenges4, ceno_transcript* tr,
                          ceno_chip_proof* out, ceno_hip_wit_plan* plan);
int chip_run_tower_specs(ChipProofRun& run, ceno_hip_tower_spec* specs3);  // returns how many (<= 3)
int chip_run_virtual_tower_specs(ChipProofRun& run, int plan_index, ceno_hip_virtual_tower_spec* specs3);
int chip_run_adopt_towers(ChipProofRun& run, ceno_hip_tower* const* towers, int n);
void chip_run_free_records(ChipProofRun& run);
int chip_run_after_towers(ChipProofRun& run, ceno_hip_stream s);
int chip_run_begin(ChipProofRun& run, ceno_hip_ctx* ctx, const ceno_chip_task* task, const uint64_t* challenges4, ceno_transcript* tr,
                   ceno_hip_stream s, ceno_chip_proof* out);
int chip_run_finish(ChipProofRun& run, ceno_hip_stream s);  // on failure the proof and the towers are released
void chip_run_abandon(ChipProofRun& run);
void prover_tr_ext_words(ceno_transcript* t, const uint64_t* ext, int n_ext);  // prover.cpp
