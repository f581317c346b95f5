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
#include <algorithm>

#include "common.hpp"
#include "poseidon2.hpp"

using namespace gl;

static constexpr int NT = 256;
static constexpr unsigned MAXB = 4096;

#include "merkle.hpp"
#include "poseidon2_host.hpp"

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
        if (!ctx->poseidon_host) ctx->poseidon_host = new PoseidonParams();
        *ctx->poseidon_host = h;
    }
    *out = &ctx->poseidon_dev->p;
    return 0;
}
void merkle_drop_host_params(ceno_hip_ctx* ctx) {
    delete ctx->poseidon_host;
    ctx->poseidon_host = nullptr;
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

// parent[i] = perm(child[2i] || child[2i+1])[0..4); with `inj`: parent[i] = perm(that || inj[i])[0..4) (matrices of this height join)
__global__ void __launch_bounds__(NT) k_compress(const uint64_t* __restrict__ child, size_t n_parent, uint64_t* __restrict__ parent,
                                                 const uint64_t* __restrict__ inj, const p2::Params* __restrict__ pp) {
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
        if (inj) {  // uniform
            const ulonglong2* q = reinterpret_cast<const ulonglong2*>(inj + 4 * i);
            const ulonglong2 u = q[0], v = q[1];
            s[4] = u.x; s[5] = u.y; s[6] = v.x; s[7] = v.y;
            p2::permute(s, sp);
        }
        *reinterpret_cast<ulonglong2*>(parent + 4 * i) = make_ulonglong2(s[0], s[1]);
        *reinterpret_cast<ulonglong2*>(parent + 4 * i + 2) = make_ulonglong2(s[2], s[3]);
    }
}

// ---- mixed-height commitment: row digests of EVERY height class in one launch -------------------------------------------
// A class = the matrices of one height, in commitment order; its row r is the concatenation of their rows r, hashed by the
// overwrite-mode sponge (p3 `hash_iter_slices` over the rows of the equal-height matrices).  Workgroups [block0, block0 +
// nblocks) belong to a class, so the many small classes of a shard's commitment run beside the tall ones instead of one
// latency-bound launch each.  The tables are read through the scalar cache (wave-uniform).
struct MmcsClass {
    uint64_t* out;          // 2^log_rows digests: levels[0] for the tallest class, the inject buffer otherwise
    uint32_t log_rows;
    uint32_t seg0, nseg;    // segments [seg0, seg0 + nseg) of the segment table
    uint32_t block0, nblocks;
    uint32_t total_w;
};
template <bool CANON>
__global__ void __launch_bounds__(NT) k_leaf_hash_classes(const MmcsClass* __restrict__ cls, int n_cls, const MmcsMat* __restrict__ segs,
                                                          const p2::Params* __restrict__ pp) {
    __shared__ p2::Params sp;
    for (int i = threadIdx.x; i < (int)(sizeof(p2::Params) / 8); i += NT) reinterpret_cast<uint64_t*>(&sp)[i] = reinterpret_cast<const uint64_t*>(pp)[i];
    __syncthreads();
    int ci = 0;
    while (ci + 1 < n_cls && blockIdx.x >= cls[ci + 1].block0) ci++;
    const MmcsClass c = cls[ci];
    const size_t rows = (size_t)1 << c.log_rows, stride = (size_t)c.nblocks * NT;
    for (size_t r = (size_t)(blockIdx.x - c.block0) * NT + threadIdx.x; r < rows; r += stride) {
        uint64_t s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t seg = c.seg0, col = 0;
        const uint64_t* base = segs[seg].p;
        uint32_t w = segs[seg].width;
        for (uint32_t done = 0; done < c.total_w; done += p2::RATE) {
#pragma unroll
            for (int k = 0; k < p2::RATE; k++)
                if (done + k < c.total_w) {  // overwrite mode; a short last chunk keeps the old tail
                    s[k] = base[(size_t)col * rows + r];
                    if (++col == w && done + k + 1 < c.total_w) {
                        seg++;
                        col = 0;
                        base = segs[seg].p;
                        w = segs[seg].width;
                    }
                }
            if (CANON) p2::permute_canonical(s, sp);
            else p2::permute(s, sp);
        }
        *reinterpret_cast<ulonglong2*>(c.out + 4 * r) = make_ulonglong2(s[0], s[1]);
        *reinterpret_cast<ulonglong2*>(c.out + 4 * r + 2) = make_ulonglong2(s[2], s[3]);
    }
}

// openings of a mixed-height commitment: out[q] = [row (idx_q >> (H - log_rows_m)) of every matrix m, caller's order][path]
__global__ void __launch_bounds__(NT) k_mmcs_gather_rows(const MmcsMat* __restrict__ mats, int n_mats, int log_max, const uint64_t* __restrict__ idx,
                                                         size_t n_q, int shift, size_t q_stride, uint64_t* __restrict__ out) {
    // one workgroup column per matrix (blockIdx.y), lanes over (query, column)
    const MmcsMat m = mats[blockIdx.y];
    const size_t total = n_q * m.width, rows = (size_t)1 << m.log_rows;
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t t = (size_t)blockIdx.x * NT + threadIdx.x; t < total; t += stride) {
        const size_t q = t / m.width, c = t % m.width;
        const size_t r = (idx[q] >> shift) >> (log_max - (int)m.log_rows);
        out[q * q_stride + m.out_off + c] = m.p[c * rows + r];
    }
}

void merkle_release(ceno_hip_ctx* ctx, ceno_hip_merkle* t) {
    if (!t) return;
    if (t->h_top) ctx_pinned_free(ctx, t->h_top);       // holds the root too (h_root points into it)
    else if (t->h_root) ctx_pinned_free(ctx, t->h_root);
    if (t->h_table) ctx_pinned_free(ctx, t->h_table);
    if (t->d_table) ctx_free_on(ctx, t->d_table, t->st);
    // freed with the OWNER's stream, whatever stream the calling thread resolved last (a thread that drives several streams:
    // the opening, commit helpers) — another stream gets these blocks only once the owner has drained
    for (int l = 0; l < (int)t->levels.size(); l++)
        if (t->host_from < 0 || l < t->host_from) ctx_free_on(ctx, t->levels[l], t->st);
    delete t;
}

// number of top levels a tree leaves to the host (CENO_HIP_HOST_TOP, 0 = build everything on the device)
static int host_top_levels() {
    static const int v = [] {
        const char* e = getenv("CENO_HIP_HOST_TOP");
        const int x = e ? atoi(e) : 6;
        return x < 0 ? 0 : (x > 10 ? 10 : x);
    }();
    return v;
}

// levels 1..log_rows over leaf digests already in levels[0]
int merkle_alloc(ceno_hip_ctx* ctx, int log_rows, ceno_hip_merkle** out, int max_host_levels) {
    auto* t = new ceno_hip_merkle();
    t->log_rows = log_rows;
    t->st = ceno_tls_stream ? ceno_tls_stream : ctx->default_stream;  // every caller resolves its stream before it allocates
    t->levels.assign(log_rows + 1, nullptr);
    t->h_levels.assign(log_rows + 1, nullptr);
    const int host_lv = std::min(std::min(host_top_levels(), max_host_levels), log_rows);
    if (host_lv > 0) {  // optional: without the pinned block the device builds the whole tree
        void *h = nullptr, *d = nullptr;
        if (ctx_pinned_alloc(ctx, (size_t)64 << host_lv, &h, &d) == 0) {
            t->h_top = h;
            t->host_from = log_rows - host_lv;
            size_t off = 0;
            for (int l = t->host_from; l <= log_rows; l++) {
                t->h_levels[l] = (uint64_t*)((char*)h + off);
                t->levels[l] = (uint64_t*)((char*)d + off);
                off += ((size_t)1 << (log_rows - l)) * 32;
            }
            t->h_root = t->h_levels[log_rows];
        }
    }
    for (int l = 0; l <= log_rows; l++) {
        if (t->host_from >= 0 && l >= t->host_from) break;
        void* p = nullptr;
        int rc = ctx_alloc(ctx, ((size_t)1 << (log_rows - l)) * 32, &p);
        if (rc) {
            merkle_release(ctx, t);
            return rc;
        }
        t->levels[l] = (uint64_t*)p;
    }
    void *h = nullptr, *d = nullptr;
    if (t->host_from < 0 && ctx_pinned_alloc(ctx, 64, &h, &d) == 0) {  // optional: without it the root is fetched with a copy
        t->h_root = (uint64_t*)h;
        t->d_root_view = (uint64_t*)d;
    }
    *out = t;
    return 0;
}

// the host half of a tree: levels host_from + 1 .. log_rows from the digests the device wrote into the pinned level host_from
static void merkle_finish_host(ceno_hip_ctx* ctx, ceno_hip_merkle* t) {
    const p2::Params& hp = ctx->poseidon_host->p;
    for (int l = t->host_from + 1; l <= t->log_rows; l++) {
        const uint64_t* child = t->h_levels[l - 1];
        uint64_t* parent = t->h_levels[l];
        const size_t np = (size_t)1 << (t->log_rows - l);
        // a level's nodes are independent: eight permutations per pass where the CPU has AVX-512 (p2host::permute_many)
        uint64_t st[64 * 8];
        for (size_t i0 = 0; i0 < np; i0 += 64) {
            const size_t m = std::min<size_t>(64, np - i0);
            memcpy(st, child + 8 * i0, 64 * m);  // a node's state is its two children, adjacent in the child level
            p2host::permute_many(st, m, hp);
            for (size_t i = 0; i < m; i++) memcpy(parent + 4 * (i0 + i), st + 8 * i, 32);
        }
    }
    t->host_top_pending = false;
    t->root_on_host = true;
}
int merkle_ensure_top(ceno_hip_ctx* ctx, ceno_hip_merkle* t) {
    if (!t->host_top_pending) return 0;
    HIP_TRY(ctx, hipStreamSynchronize(t->st));
    merkle_finish_host(ctx, t);
    return 0;
}

// up to 6 levels (<= 64 digests in) per workgroup of 256 lanes = 32 nodes per pass, one wave per SIMD: every level costs the
// latency of one 8-lane permutation, measured ~12.5 us (rocprofv3 kernel trace of the opening: 13 / 25 / 48 / 75 us for
// 1 / 2 / 4 / 6 levels).  A tree of h small levels is therefore h x 12.5 us however it is tiled (1024 lanes over 8 levels
// measured the same per level); the opening hides half of it by building the tree of round r+1 while round r is answered.
static constexpr int TOP_NT = 256, TOP_LEVELS = 6;
struct TopPtrs {
    uint64_t* p[TOP_LEVELS];  // by value in the kernel arguments: no host-to-device copy (a pageable one would block the host on the stream)
    const uint64_t* inj[TOP_LEVELS];  // digests that join at that level (mixed-height commitment), or NULL
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
            if (outs.inj[l]) {  // uniform: lanes 0..3 keep the compressed pair, lanes 4..7 take the joining digest
                const size_t node = (b << (levels - 1 - l)) + slot;
                if (g >= 4) x = outs.inj[l][4 * node + (g - 4)];
                x = p2::permute_lanes8(x, sp);
            }
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

int merkle_build_upper(ceno_hip_ctx* ctx, ceno_hip_merkle* t, hipStream_t st, const uint64_t* const* inject) {
    const p2::Params* pp;
    TRY(get_params(ctx, &pp));
    const int log_rows = t->log_rows;
    // the device stops at level host_from when the host finishes the tree (its digests land in pinned memory: merkle_alloc)
    const int top = t->host_from >= 0 ? t->host_from : log_rows;
    if (t->host_from >= 0) {
        for (int lv = t->host_from + 1; inject && lv <= log_rows; lv++)
            if (inject[lv]) return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "merkle: an injection level above the host split (merkle_alloc max_host_levels)");
        t->host_top_pending = true;
    }
    int l = 1;
    // levels with more than 2^top_from nodes: one lane per node (throughput bound)
    static const int top_from = [] {
        const char* e = getenv("CENO_HIP_MERKLE_TOP_FROM_LOG");
        return e ? std::max(0, std::min(atoi(e), 30)) : 14;
    }();
    for (; l <= top && ((size_t)1 << (log_rows - l)) > ((size_t)1 << top_from); l++) {
        size_t np = (size_t)1 << (log_rows - l);
        hipLaunchKernelGGL(k_compress, dim3(grid_for(np, NT, MAXB)), dim3(NT), 0, st, t->levels[l - 1], np, t->levels[l],
                           inject ? inject[l] : (const uint64_t*)nullptr, pp);
    }
    // the rest is a chain of dependent permutations (one per level): 8 lanes per permutation, and every launch takes up to
    // TOP_LEVELS levels at once — each workgroup reduces its own 2^TOP_LEVELS-digest subtree in LDS — so 15 small levels
    // cost three kernel boundaries instead of fifteen
    int rem = top - l + 1;  // levels left to the device; the child level l-1 holds 2^(log_rows - l + 1) digests
    while (rem > 0) {
        const int lv = rem > TOP_LEVELS ? TOP_LEVELS : rem;
        TopPtrs tp{};
        for (int i = 0; i < lv; i++) {
            tp.p[i] = t->levels[l + i];
            tp.inj[i] = inject ? inject[l + i] : nullptr;
        }
        if (rem == lv && top == log_rows && t->d_root_view) {  // this launch ends at the root
            tp.root_host = t->d_root_view;
            t->root_on_host = true;
        }
        hipLaunchKernelGGL(k_compress_top, dim3(1u << (log_rows - l + 1 - lv)), dim3(TOP_NT), 0, st, t->levels[l - 1], lv, tp, pp);
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
    {
        std::lock_guard<std::mutex> g(ctx->tw_mu);
        if (!ctx->poseidon_host) ctx->poseidon_host = new PoseidonParams();
        *ctx->poseidon_host = h;
    }
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

// `leaf_digests` != NULL: the tree's layer 0 is GIVEN (2^log_leaves digests, e.g. the sub-tree roots of the ranks of a sharded
// commitment) and every matrix is shorter than it: they are injected at their levels above, exactly as in the full tree.
static int mmcs_commit_impl(ceno_hip_ctx* ctx, const uint64_t* leaf_digests, int log_leaves, const uint64_t* const* dev_col_major, const int* log_rows,
                            const int* widths, int n_mats, ceno_hip_stream s, ceno_hip_merkle** out) {
    int H = leaf_digests ? log_leaves : 0;
    size_t total_w = 0;
    for (int m = 0; m < n_mats; m++) {
        CHECK_ARG(ctx, dev_col_major[m] && log_rows[m] >= 0 && log_rows[m] < 40 && widths[m] >= 1, "bad matrix %d", m);
        if (leaf_digests) CHECK_ARG(ctx, log_rows[m] < log_leaves, "matrix %d is not shorter than the given digest layer", m);
        else H = std::max(H, log_rows[m]);
        total_w += (size_t)widths[m];
    }
    const p2::Params* pp;
    TRY(get_params(ctx, &pp));
    hipStream_t st = ctx_stream(ctx, s);
    ceno_hip_merkle* t = nullptr;
    // the host may finish the levels strictly above the highest injection level (the row digests of the short matrices stay on the
    // device); a tree over a GIVEN digest layer (the few sub-tree roots of a sharded commitment) is built on the device as before
    int max_host = leaf_digests ? 0 : 64;
    for (int m = 0; m < n_mats; m++)
        if (log_rows[m] < H) max_host = std::min(max_host, log_rows[m]);  // injected at level H - log_rows[m]: log_rows[m] levels above it
    TRY(merkle_alloc(ctx, H, &t, max_host));
    t->total_width = total_w;
    if (leaf_digests && hipMemcpyAsync(t->levels[0], leaf_digests, ((size_t)32) << H, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        merkle_release(ctx, t);
        return ctx_fail(ctx, CENO_HIP_ERR_HIP, "mmcs_commit: copy of the digest layer failed");
    }
    for (int m = 0; m < n_mats; m++) t->mats.push_back({dev_col_major[m], log_rows[m], widths[m]});
    // tallest first, STABLE: equal heights keep the caller's order (p3 MerkleTree::new sorts by Reverse(height))
    std::vector<int> order(n_mats);
    for (int i = 0; i < n_mats; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return log_rows[a] > log_rows[b]; });
    // classes and their segments (adjacent matrices that are contiguous in memory merge into one segment)
    std::vector<MmcsClass> cls;
    std::vector<MmcsMat> segs;
    std::vector<void*> inj_bufs;                        // released below (stream-ordered pool)
    std::vector<const uint64_t*> inject(H + 1, nullptr);  // by LEVEL: level l has 2^(H - l) nodes
    int rc = 0;
    unsigned nblocks_total = 0;
    for (int i = 0; i < n_mats && !rc;) {
        const int h = log_rows[order[i]];
        MmcsClass c{};
        c.log_rows = (uint32_t)h;
        c.seg0 = (uint32_t)segs.size();
        if (h == H && !leaf_digests) c.out = t->levels[0];
        else {
            void* p = nullptr;
            rc = ctx_alloc(ctx, ((size_t)1 << h) * 32, &p);
            if (rc) break;
            inj_bufs.push_back(p);
            c.out = (uint64_t*)p;
            inject[H - h] = (const uint64_t*)p;
        }
        for (; i < n_mats && log_rows[order[i]] == h; i++) {
            const int m = order[i];
            const size_t rows = (size_t)1 << h;
            if (segs.size() > c.seg0 && segs.back().p + (size_t)segs.back().width * rows == dev_col_major[m]) segs.back().width += (uint32_t)widths[m];
            else segs.push_back(MmcsMat{dev_col_major[m], (uint32_t)h, (uint32_t)widths[m], 0, 0});
            c.total_w += (uint32_t)widths[m];
        }
        c.nseg = (uint32_t)segs.size() - c.seg0;
        c.block0 = nblocks_total;
        c.nblocks = grid_for((size_t)1 << h, NT, MAXB);
        nblocks_total += c.nblocks;
        cls.push_back(c);
    }
    // tables: [classes][segments][matrices in the caller's order (openings)] in one pinned block -> one async copy
    const size_t b_cls = cls.size() * sizeof(MmcsClass), b_seg = segs.size() * sizeof(MmcsMat), b_mat = (size_t)n_mats * sizeof(MmcsMat);
    const size_t off_seg = (b_cls + 15) & ~(size_t)15, off_mat = off_seg + b_seg, bytes = off_mat + b_mat;
    void *h_tab = nullptr, *h_view = nullptr, *d_tab = nullptr;
    if (!rc) rc = ctx_pinned_alloc(ctx, bytes, &h_tab, &h_view);
    if (!rc) {
        t->h_table = h_tab;
        rc = ctx_alloc(ctx, bytes, &d_tab);
    }
    if (!rc) {
        t->d_table = d_tab;
        t->mat_table_off = off_mat;
        memcpy(h_tab, cls.data(), b_cls);
        memcpy((char*)h_tab + off_seg, segs.data(), b_seg);
        auto* hm = reinterpret_cast<MmcsMat*>((char*)h_tab + off_mat);
        uint32_t off = 0;
        for (int m = 0; m < n_mats; m++) {
            hm[m] = MmcsMat{dev_col_major[m], (uint32_t)log_rows[m], (uint32_t)widths[m], off, 0};
            off += (uint32_t)widths[m];
        }
        if (hipMemcpyAsync(d_tab, h_tab, bytes, hipMemcpyHostToDevice, st) != hipSuccess) rc = ctx_fail(ctx, CENO_HIP_ERR_HIP, "mmcs_commit: table upload failed");
    }
    if (!rc) {
        const auto* d_cls = reinterpret_cast<const MmcsClass*>(d_tab);
        const auto* d_seg = reinterpret_cast<const MmcsMat*>((char*)d_tab + off_seg);
        static const bool canon = [] { const char* e = getenv("CENO_HIP_P2_CANONICAL"); return e && atoi(e) != 0; }();
        if (cls.empty()) {
        } else if (canon) hipLaunchKernelGGL(k_leaf_hash_classes<true>, dim3(nblocks_total), dim3(NT), 0, st, d_cls, (int)cls.size(), d_seg, pp);
        else hipLaunchKernelGGL(k_leaf_hash_classes<false>, dim3(nblocks_total), dim3(NT), 0, st, d_cls, (int)cls.size(), d_seg, pp);
        rc = merkle_build_upper(ctx, t, st, inject.data());
    }
    for (void* p : inj_bufs) ctx_free(ctx, p);  // tagged with this stream: reused behind the kernels just queued
    if (rc) {
        merkle_release(ctx, t);
        return rc;
    }
    *out = t;
    return 0;
}

int ceno_hip_mmcs_commit(ceno_hip_ctx* ctx, const uint64_t* const* dev_col_major, const int* log_rows, const int* widths, int n_mats,
                         ceno_hip_stream s, ceno_hip_merkle** out) {
    CHECK_ARG(ctx, dev_col_major && log_rows && widths && out && n_mats >= 1 && n_mats <= 65535, "bad mmcs_commit arguments");
    return mmcs_commit_impl(ctx, nullptr, 0, dev_col_major, log_rows, widths, n_mats, s, out);
}

int ceno_hip_mmcs_commit_over(ceno_hip_ctx* ctx, const uint64_t* dev_leaf_digests, int log_leaves, const uint64_t* const* dev_col_major,
                              const int* log_rows, const int* widths, int n_mats, ceno_hip_stream s, ceno_hip_merkle** out) {
    CHECK_ARG(ctx, dev_leaf_digests && out && log_leaves >= 0 && log_leaves < 40 && n_mats >= 0 && n_mats <= 65535, "bad mmcs_commit_over arguments");
    CHECK_ARG(ctx, n_mats == 0 || (dev_col_major && log_rows && widths), "bad mmcs_commit_over arguments");
    return mmcs_commit_impl(ctx, dev_leaf_digests, log_leaves, dev_col_major, log_rows, widths, n_mats, s, out);
}

size_t ceno_hip_mmcs_opening_words(const ceno_hip_merkle* t) { return t ? t->total_width + 4 * (size_t)t->log_rows : 0; }

int ceno_hip_mmcs_open_batch(ceno_hip_ctx* ctx, ceno_hip_merkle* t, const uint64_t* dev_indices, size_t n, int shift, uint64_t* dev_out,
                             size_t out_stride_words, ceno_hip_stream s) {
    CHECK_ARG(ctx, t && dev_indices && dev_out && shift >= 0 && shift < 64, "bad mmcs_open_batch arguments");
    CHECK_ARG(ctx, t->d_table && !t->mats.empty(), "the tree is not a mixed-height commitment (ceno_hip_mmcs_commit)");
    const size_t per_q = t->total_width + 4 * (size_t)t->log_rows;
    CHECK_ARG(ctx, out_stride_words >= per_q, "output stride %zu is smaller than one opening (%zu words)", out_stride_words, per_q);
    if (n == 0) return 0;
    hipStream_t st = ctx_stream(ctx, s);
    // the matrix table sits behind the class and segment tables of the commit
    const MmcsMat* d_mats = reinterpret_cast<const MmcsMat*>((char*)t->d_table + t->mat_table_off);
    size_t widest = 1;
    for (auto& m : t->mats) widest = std::max(widest, (size_t)m.width);
    hipLaunchKernelGGL(k_mmcs_gather_rows, dim3(grid_for(n * widest, NT, 64), (unsigned)t->mats.size()), dim3(NT), 0, st, d_mats, (int)t->mats.size(),
                       t->log_rows, dev_indices, n, shift, out_stride_words, dev_out);
    HIP_TRY(ctx, hipGetLastError());
    if (t->log_rows > 0) TRY(merkle_gather_paths(ctx, t, dev_indices, n, shift, dev_out + t->total_width, out_stride_words, st));
    return 0;
}

int ceno_hip_merkle_root(ceno_hip_ctx* ctx, ceno_hip_merkle* t, uint64_t* root4, ceno_hip_stream s) {
    CHECK_ARG(ctx, t && root4, "NULL argument");
    hipStream_t st = ctx_stream(ctx, s);
    if (t->host_top_pending) {  // the device half sits in pinned memory once the stream has drained: the host finishes the tree
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (st != t->st) HIP_TRY(ctx, hipStreamSynchronize(t->st));
        merkle_finish_host(ctx, t);
        memcpy(root4, t->h_root, 32);
        return 0;
    }
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
    TRY(merkle_ensure_top(ctx, t));
    size_t idx = index;
    for (int l = 0; l < t->log_rows; l++) {
        if (t->h_levels[l]) {  // a host-resident level (after a drain of the stream: the device half may still be in flight)
            HIP_TRY(ctx, hipStreamSynchronize(st));
            memcpy(path + 4 * l, t->h_levels[l] + 4 * (idx ^ 1), 32);
        } else {
            HIP_TRY(ctx, hipMemcpyAsync(path + 4 * l, t->levels[l] + 4 * (idx ^ 1), 32, hipMemcpyDeviceToHost, st));
        }
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
