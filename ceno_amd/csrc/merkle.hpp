// Merkle tree handle and helpers shared by poseidon2.hip (trace commitment) and basefold.hip (commit phase).
#pragma once
#include "common.hpp"
#include "poseidon2.hpp"

struct PoseidonParams {
    p2::Params p;
};

struct ceno_hip_merkle {
    int log_rows = 0;
    hipStream_t st = nullptr;        // the stream the tree was built on: its blocks go back to the pool tagged with it (merkle_release)
    std::vector<uint64_t*> levels;   // levels[0] = 2^log_rows leaf digests (4 words each) ... levels[log_rows] = root
    uint64_t* h_root = nullptr;      // pinned host copy of the root, written by the kernel that computes it (no D2H blit)
    uint64_t* d_root_view = nullptr; // device view of h_root
    bool root_on_host = false;       // the tree-top kernel has been told to write h_root
    // Host-finished top: levels[host_from .. log_rows] live in ONE pinned block (`levels` holds their device views, `h_levels` the
    // host pointers).  The device builds the tree up to level host_from — whose kernel therefore writes its digests straight into
    // host memory — and the HOST computes the levels above it (merkle_finish_host): a level of few nodes costs the device the
    // latency of one permutation (~11.5 us on eight lanes) whatever its size, the host 0.7 us per node.
    int host_from = -1;              // -1: the whole tree is built on the device
    std::vector<uint64_t*> h_levels; // [log_rows + 1], NULL below host_from
    void* h_top = nullptr;           // the pinned block
    bool host_top_pending = false;   // levels above host_from are not computed yet (needs the owner stream drained first)
    // mixed-height commitment (ceno_hip_mmcs_commit): the matrices in the CALLER's order (borrowed) and their device table
    struct Mat {
        const uint64_t* p;
        int log_rows, width;
    };
    std::vector<Mat> mats;
    size_t total_width = 0;
    void* d_table = nullptr;         // MmcsMat[n] in device memory (row gathers of the openings)
    void* h_table = nullptr;         // its pinned source
    size_t mat_table_off = 0;        // byte offset of the MmcsMat[n] table inside d_table
};

// device-side description of one committed matrix (openings) / of one input segment of the leaf hashing
struct MmcsMat {
    const uint64_t* p;   // column-major, column stride = 2^log_rows
    uint32_t log_rows, width;
    uint32_t out_off;    // word offset of this matrix's row inside one opening
    uint32_t pad;
};

int get_params(ceno_hip_ctx* ctx, const p2::Params** out);
// owner stream = the stream the calling thread resolved last.  `max_host_levels` caps the number of top levels left to the host
// (0 = none; a mixed-height tree passes the distance between the root and its highest injection level: injections stay on the device)
int merkle_alloc(ceno_hip_ctx* ctx, int log_rows, ceno_hip_merkle** out, int max_host_levels = 64);
// computes the host half of the tree if it is still pending (drains the owner stream first); called by every reader of the upper levels
int merkle_ensure_top(ceno_hip_ctx* ctx, ceno_hip_merkle* t);
void merkle_drop_host_params(ceno_hip_ctx* ctx);
// levels 1.. from the leaf digests in levels[0]; inject[l] (may be NULL / absent) = digests of the rows of the matrices whose
// height equals level l's node count: parent = compress(compress(left, right), inject[l][i])  (p3 MerkleTreeMmcs)
int merkle_build_upper(ceno_hip_ctx* ctx, ceno_hip_merkle* t, hipStream_t st, const uint64_t* const* inject = nullptr);
void merkle_release(ceno_hip_ctx* ctx, ceno_hip_merkle* t);
// authentication paths of leaves (idx[q] >> shift): out[q * out_stride + 4 * level + k] (basefold.hip)
int merkle_gather_paths(ceno_hip_ctx* ctx, ceno_hip_merkle* t, const uint64_t* dev_indices, size_t n, int shift, uint64_t* dev_out,
                        size_t out_stride_words, hipStream_t st);
