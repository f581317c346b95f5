// Poseidon2 permutation, row sponge and Merkle tree on gfx950 (Basefold commit path).
//
// Reference: `cuda_hal.basefold.batch_commit` (ceno_zkvm/src/scheme/gpu/mod.rs:1642-1646) — per trace
// matrix: hash every codeword ROW with a Poseidon2 sponge, build the 2-to-1 compression tree, root =
// commitment (shape restated in ceno_recursion_v2/src/pcs/mod.rs:1111-1316).
// Sponge = overwrite-mode padding-free sponge (rate 4, width 8, 4-word digest); compression =
// truncated permutation of left || right.  PARITY UNPINNED — constants, see poseidon2.hpp.
//
// One lane runs one permutation with the 8-word state in registers (16 VGPRs); the matrix is column-
// major so the lanes of a wave read consecutive rows of a column: 8 B per lane coalesced.  The kernel
// is ALU bound (118 S-boxes x 4 mults + linear layers per permutation); reported separately from the
// HBM roofline.
#include "common.hpp"
#include "poseidon2.hpp"

using namespace gl;

static constexpr int NT = 256;
static constexpr unsigned MAXB = 4096;

#include "merkle.hpp"

// The built-in round constants are PLACEHOLDERS (poseidon2.hpp): roots, challenges and proofs made with them are
// self-consistent but can never verify against the reference.  Until a complete table has been supplied through
// ceno_hip_poseidon2_set_constants every first use says so on stderr, and CENO_HIP_REQUIRE_PINNED_POSEIDON2=1 turns the
// commit / open / transcript entry points into errors instead (what a production caller should set).
static int placeholder_gate(ceno_hip_ctx* ctx) {
    if (ctx->poseidon_pinned) return 0;
    static const bool strict = [] { const char* e = getenv("CENO_HIP_REQUIRE_PINNED_POSEIDON2"); return e && atoi(e) != 0; }();
    if (strict)
        return ctx_fail(ctx, CENO_HIP_ERR_STATE, "Poseidon2 constants are placeholders: call ceno_hip_poseidon2_set_constants with the reference's "
                        "table first (CENO_HIP_REQUIRE_PINNED_POSEIDON2 is set)");
    static bool warned = false;
    if (!warned && !getenv("CENO_HIP_QUIET_PLACEHOLDER")) {
        warned = true;
        fprintf(stderr, "[ceno_hip] WARNING: Poseidon2 placeholder round constants in use - Merkle roots, proofs of work and openings are NOT "
                        "interoperable with the reference (PARITY UNPINNED); load the real table with ceno_hip_poseidon2_set_constants\n");
    }
    return 0;
}

int get_params(ceno_hip_ctx* ctx, const p2::Params** out) {
    TRY(placeholder_gate(ctx));
    std::lock_guard<std::mutex> g(ctx->tw_mu);  // lanes may arrive here together on first use
    if (!ctx->poseidon_dev) {
        PoseidonParams h;
        p2::default_params(h.p);
        void* d = nullptr;
        HIP_TRY(ctx, hipMalloc(&d, sizeof(PoseidonParams)));
        HIP_TRY(ctx, hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice));
        ctx->poseidon_dev = (PoseidonParams*)d;
    }
    *out = &ctx->poseidon_dev->p;
    return 0;
}

__global__ void __launch_bounds__(NT) k_permute(uint64_t* states, size_t n, const p2::Params* __restrict__ pp) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        uint64_t s[8];
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = states[i * 8 + k];
        p2::permute(s, sp);
#pragma unroll
        for (int k = 0; k < 8; k++) states[i * 8 + k] = s[k];
    }
}

// leaf digests: sponge over row `r` of a column-major matrix (col stride = rows)
template <bool CANON>
__global__ void __launch_bounds__(NT) k_leaf_hash(const uint64_t* __restrict__ m, size_t rows, int width, uint64_t* __restrict__ digests,
                                                  const p2::Params* __restrict__ pp) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        uint64_t s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < width; c += p2::RATE) {
#pragma unroll
            for (int k = 0; k < p2::RATE; k++)
                if (c + k < width) s[k] = m[(size_t)(c + k) * rows + r];  // overwrite mode; a short last chunk keeps the old tail
            if (CANON) p2::permute_canonical(s, sp);
            else p2::permute(s, sp);
        }
        *reinterpret_cast<ulonglong2*>(digests + 4 * r) = make_ulonglong2(s[0], s[1]);
        *reinterpret_cast<ulonglong2*>(digests + 4 * r + 2) = make_ulonglong2(s[2], s[3]);
    }
}

// parent[i] = perm(child[2i] || child[2i+1])[0..4)
__global__ void __launch_bounds__(NT) k_compress(const uint64_t* __restrict__ child, size_t n_parent, uint64_t* __restrict__ parent,
                                                 const p2::Params* __restrict__ pp) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n_parent; i += stride) {
        uint64_t s[8];
        const ulonglong2* c = reinterpret_cast<const ulonglong2*>(child + 8 * i);
        ulonglong2 a = c[0], b = c[1], d = c[2], e = c[3];
        s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y; s[4] = d.x; s[5] = d.y; s[6] = e.x; s[7] = e.y;
        p2::permute(s, sp);
        *reinterpret_cast<ulonglong2*>(parent + 4 * i) = make_ulonglong2(s[0], s[1]);
        *reinterpret_cast<ulonglong2*>(parent + 4 * i + 2) = make_ulonglong2(s[2], s[3]);
    }
}

void merkle_release(ceno_hip_ctx* ctx, ceno_hip_merkle* t) {
    if (!t) return;
    if (t->h_root) ctx_pinned_free(ctx, t->h_root);
    for (auto* p : t->levels) ctx_free(ctx, p);
    delete t;
}

// levels 1..log_rows over leaf digests already in levels[0]
int merkle_alloc(ceno_hip_ctx* ctx, int log_rows, ceno_hip_merkle** out) {
    auto* t = new ceno_hip_merkle();
    t->log_rows = log_rows;
    t->levels.assign(log_rows + 1, nullptr);
    for (int l = 0; l <= log_rows; l++) {
        void* p = nullptr;
        int rc = ctx_alloc(ctx, ((size_t)1 << (log_rows - l)) * 32, &p);
        if (rc) {
            merkle_release(ctx, t);
            return rc;
        }
        t->levels[l] = (uint64_t*)p;
    }
    void *h = nullptr, *d = nullptr;
    if (ctx_pinned_alloc(ctx, 64, &h, &d) == 0) {  // optional: without it the root is fetched with a copy
        t->h_root = (uint64_t*)h;
        t->d_root_view = (uint64_t*)d;
    }
    *out = t;
    return 0;
}

// up to 6 levels (<= 64 digests in) per workgroup of 256 lanes = 32 nodes per pass, one wave per SIMD: every level costs the
// latency of one 8-lane permutation, measured ~12.5 us (rocprofv3 kernel trace of the opening: 13 / 25 / 48 / 75 us for
// 1 / 2 / 4 / 6 levels).  A tree of h small levels is therefore h x 12.5 us however it is tiled (1024 lanes over 8 levels
// measured the same per level); the opening hides half of it by building the tree of round r+1 while round r is answered.
static constexpr int TOP_NT = 256, TOP_LEVELS = 6;
struct TopPtrs {
    uint64_t* p[TOP_LEVELS];  // by value in the kernel arguments: no host-to-device copy (a pageable one would block the host on the stream)
    uint64_t* root_host;      // != NULL in the launch that produces the root: it is also written to pinned host memory
};
__global__ void __launch_bounds__(TOP_NT) k_compress_top(const uint64_t* __restrict__ child, int levels, TopPtrs outs,
                                                         const p2::Params* __restrict__ pp) {
    __shared__ p2::Params sp;
    __shared__ uint64_t buf[2][4 << TOP_LEVELS];
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += TOP_NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    int n = 1 << levels;  // child digests of this workgroup's subtree
    const size_t b = blockIdx.x;
    child += 4 * (b << levels);
    for (int i = threadIdx.x; i < 4 * n; i += TOP_NT) buf[0][i] = child[i];
    __syncthreads();
    int cur = 0;
    const int g = threadIdx.x & 7, slot = threadIdx.x >> 3;
    for (int l = 0; l < levels; l++) {
        const int np = n >> 1;
        if (slot < np) {  // np <= 32 = one pass; a wave is entirely inside or outside (8 nodes per wave)
            uint64_t x = buf[cur][8 * slot + g];
            x = p2::permute_lanes8(x, sp);
            if (g < 4) {
                buf[cur ^ 1][4 * slot + g] = x;
                outs.p[l][4 * ((b << (levels - 1 - l)) + slot) + g] = x;
                if (outs.root_host && l == levels - 1) outs.root_host[g] = x;  // one node, one workgroup
            }
        }
        __syncthreads();
        cur ^= 1;
        n = np;
    }
}

int merkle_build_upper(ceno_hip_ctx* ctx, ceno_hip_merkle* t, hipStream_t st) {
    const p2::Params* pp;
    TRY(get_params(ctx, &pp));
    const int log_rows = t->log_rows;
    int l = 1;
    // levels with more than 2^14 nodes: one lane per node (throughput bound)
    for (; l <= log_rows && ((size_t)1 << (log_rows - l)) > ((size_t)1 << 14); l++) {
        size_t np = (size_t)1 << (log_rows - l);
        hipLaunchKernelGGL(k_compress, dim3(grid_for(np, NT, MAXB)), dim3(NT), 0, st, t->levels[l - 1], np, t->levels[l], pp);
    }
    // the rest is a chain of dependent permutations (one per level): 8 lanes per permutation, and every launch takes up to
    // TOP_LEVELS levels at once — each workgroup reduces its own 2^TOP_LEVELS-digest subtree in LDS — so 15 small levels
    // cost three kernel boundaries instead of fifteen
    int rem = log_rows - l + 1;  // the child level l-1 holds 2^rem digests
    while (rem > 0) {
        const int lv = rem > TOP_LEVELS ? TOP_LEVELS : rem;
        TopPtrs tp{};
        for (int i = 0; i < lv; i++) tp.p[i] = t->levels[l + i];
        if (rem == lv && t->d_root_view) {  // this launch ends at the root
            tp.root_host = t->d_root_view;
            t->root_on_host = true;
        }
        hipLaunchKernelGGL(k_compress_top, dim3(1u << (rem - lv)), dim3(TOP_NT), 0, st, t->levels[l - 1], lv, tp, pp);
        l += lv;
        rem -= lv;
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

extern "C" {

int ceno_hip_poseidon2_is_pinned(const ceno_hip_ctx* ctx) { return ctx && ctx->poseidon_pinned ? 1 : 0; }

int ceno_hip_poseidon2_set_constants(ceno_hip_ctx* ctx, const uint64_t* external_rc, const uint64_t* internal_rc, const uint64_t* internal_diag) {
    ctx_make_current(ctx);
    PoseidonParams h;
    p2::default_params(h.p);
    if (external_rc) memcpy(h.p.ext_rc, external_rc, sizeof(h.p.ext_rc));
    if (internal_rc) memcpy(h.p.int_rc, internal_rc, sizeof(h.p.int_rc));
    if (internal_diag) memcpy(h.p.int_diag, internal_diag, sizeof(h.p.int_diag));
    for (size_t i = 0; i < sizeof(h) / 8; i++)
        CHECK_ARG(ctx, reinterpret_cast<uint64_t*>(&h)[i] < gl::P, "poseidon2 constant %zu is not canonical", i);
    HIP_TRY(ctx, hipDeviceSynchronize());
    // kernels read the table through the scalar cache: publish the new table at a FRESH address instead
    // of overwriting the old one in place (no stale K$/L2 lines can exist for it)
    void* d = nullptr;
    HIP_TRY(ctx, hipMalloc(&d, sizeof(PoseidonParams)));
    HIP_TRY(ctx, hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice));
    if (ctx->poseidon_dev) (void)hipFree(ctx->poseidon_dev);
    ctx->poseidon_dev = (PoseidonParams*)d;
    ctx->poseidon_pinned = external_rc && internal_rc && internal_diag;  // NULLs restore (parts of) the placeholder table
    return 0;
}

int ceno_hip_poseidon2_permute(ceno_hip_ctx* ctx, uint64_t* dev_states, size_t n, ceno_hip_stream s) {
    CHECK_ARG(ctx, dev_states, "NULL argument");
    const p2::Params* pp;
    TRY(get_params(ctx, &pp));
    hipStream_t st = ctx_stream(ctx, s);
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_permute, dim3(grid_for(n, NT, MAXB)), dim3(NT), 0, st, dev_states, n, pp);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int ceno_hip_merkle_commit(ceno_hip_ctx* ctx, const uint64_t* dev_col_major, int log_rows, int width, ceno_hip_stream s, ceno_hip_merkle** out) {
    CHECK_ARG(ctx, dev_col_major && out && log_rows >= 0 && log_rows < 40 && width >= 1, "bad merkle arguments");
    const p2::Params* pp;
    TRY(get_params(ctx, &pp));
    hipStream_t st = ctx_stream(ctx, s);
    ceno_hip_merkle* t = nullptr;
    TRY(merkle_alloc(ctx, log_rows, &t));
    size_t rows = (size_t)1 << log_rows;
    // CENO_HIP_P2_CANONICAL=1: the straightforward all-canonical permutation (A/B timing and cross-check of the lazy form)
    static const bool canon = [] { const char* e = getenv("CENO_HIP_P2_CANONICAL"); return e && atoi(e) != 0; }();
    if (canon) hipLaunchKernelGGL(k_leaf_hash<true>, dim3(grid_for(rows, NT, MAXB)), dim3(NT), 0, st, dev_col_major, rows, width, t->levels[0], pp);
    else hipLaunchKernelGGL(k_leaf_hash<false>, dim3(grid_for(rows, NT, MAXB)), dim3(NT), 0, st, dev_col_major, rows, width, t->levels[0], pp);
    int rc = merkle_build_upper(ctx, t, st);
    if (rc) {
        merkle_release(ctx, t);
        return rc;
    }
    *out = t;
    return 0;
}

int ceno_hip_merkle_root(ceno_hip_ctx* ctx, ceno_hip_merkle* t, uint64_t* root4, ceno_hip_stream s) {
    CHECK_ARG(ctx, t && root4, "NULL argument");
    hipStream_t st = ctx_stream(ctx, s);
    if (t->root_on_host) {  // the tree-top kernel wrote it to pinned memory: wait for the stream, no copy engine round trip
        HIP_TRY(ctx, hipStreamSynchronize(st));
        memcpy(root4, t->h_root, 32);
        return 0;
    }
    HIP_TRY(ctx, hipMemcpyAsync(root4, t->levels[t->log_rows], 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return 0;
}

int ceno_hip_merkle_open(ceno_hip_ctx* ctx, ceno_hip_merkle* t, size_t index, uint64_t* path, ceno_hip_stream s) {
    CHECK_ARG(ctx, t && path, "NULL argument");
    CHECK_ARG(ctx, index < ((size_t)1 << t->log_rows), "leaf index out of range");
    hipStream_t st = ctx_stream(ctx, s);
    size_t idx = index;
    for (int l = 0; l < t->log_rows; l++) {
        HIP_TRY(ctx, hipMemcpyAsync(path + 4 * l, t->levels[l] + 4 * (idx ^ 1), 32, hipMemcpyDeviceToHost, st));
        idx >>= 1;
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return 0;
}

int ceno_hip_merkle_free(ceno_hip_ctx* ctx, ceno_hip_merkle* t) {
    // no wait: the pool hands a freed block to another stream only once the stream that used it last has drained (ctx_alloc)
    merkle_release(ctx, t);
    return 0;
}

}  // extern "C"
