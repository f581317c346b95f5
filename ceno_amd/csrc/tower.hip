// Tower (grand-product / LogUp) witness construction on gfx950.
//
// Reference semantics: `interleaving_mles_to_mles` (ceno_zkvm/src/scheme/utils.rs:402-462),
// `infer_tower_product_witness` (:588-659), `infer_tower_logup_witness` (:488-582, tower_mle_4 :464-479);
// GPU call sites `build_prod_tower_from_virtual_ext_batch` / `build_logup_tower_from_virtual_ext_batch`
// and `GpuProverSpec::get_output_evals` (ceno_zkvm/src/scheme/gpu/mod.rs:2365-2402,379-410).
// Layer l of a tower holds 2 (product) or 4 (p1,p2,q1,q2) limbs of 2^l extension elements; limb s of
// layer l is computed from the s-th half of the limbs of layer l+1.  All kernels are element-wise
// streams: 48 B of traffic per product -> HBM bound.
#include "common.hpp"
#include "witinfer_dev.hpp"

#include <algorithm>
#include <functional>

using namespace gl;

static constexpr int NT = 256;
static constexpr unsigned MAXB = 2048;
static constexpr int MAX_REC = 64;

struct ceno_hip_tower {
    int num_vars = 0;   // number of layers
    int n_limbs = 2;    // 2 = product, 4 = logup
    std::vector<E2*> layers;  // layers[l] -> n_limbs * 2^l elements, limb-major
    // layers 0 .. top_layers-1 live back to back in ONE block (layer l at element offset n_limbs * (2^l - 1)): the host proves
    // the small layers itself from a single copy of that block (ceno_hip_tower_download_top)
    int top_layers = 0;
    // host copy of the layers 0 .. host_top_layers-1 (ceno_hip_tower_prefetch_tops: ONE synchronisation for all towers of a chip);
    // ceno_hip_tower_out_evals / ceno_hip_tower_download_top are served from it.  A tower is immutable once built.
    std::vector<uint64_t> host_top;
    int host_top_layers = 0;
};
static constexpr int TOWER_TOP_LAYERS = 11;  // layers of up to 2^10 entries per limb

struct RecArg {
    const uint64_t* ptr[MAX_REC];
    uint32_t cnt0[MAX_REC];  // rows available for limb 0
    uint32_t cnt1[MAX_REC];  // rows available for limb 1
    uint8_t is_ext[MAX_REC];
    int k;
    int log_s;              // per-instance slot count = 2^log_s
    uint64_t start1;        // first source row of limb 1 (= per_fanin_len)
    E2 dflt;
    int ones;               // 1: no records, fill both limbs with `dflt`
};

// out[limb][i * S + j] = rec_j[start_limb + i]  (or default)
__global__ void __launch_bounds__(NT) k_interleave(RecArg ra, E2* __restrict__ out0, E2* __restrict__ out1, size_t out_len) {
    const size_t stride = (size_t)gridDim.x * NT;
    const size_t smask = ((size_t)1 << ra.log_s) - 1;
    for (size_t o = (size_t)blockIdx.x * NT + threadIdx.x; o < 2 * out_len; o += stride) {
        const int limb = o >= out_len;
        const size_t x = limb ? o - out_len : o;
        const size_t i = x >> ra.log_s;
        const int j = (int)(x & smask);
        E2 v = ra.dflt;
        if (!ra.ones && j < ra.k) {
            const uint32_t cnt = limb ? ra.cnt1[j] : ra.cnt0[j];
            if (i < cnt) {
                const size_t src = (limb ? ra.start1 : 0) + i;
                if (ra.is_ext[j]) v = reinterpret_cast<const E2*>(ra.ptr[j])[src];
                else v = E2{ra.ptr[j][src], 0};
            }
        }
        (limb ? out1 : out0)[x] = v;
    }
}

// product layer: out[s*half + j] = a[s*half + j] * b[s*half + j], a/b = limbs of the layer below (len 2*half)
__global__ void __launch_bounds__(NT) k_prod_layer(const E2* __restrict__ below, E2* __restrict__ out, size_t len_below) {
    const size_t stride = (size_t)gridDim.x * NT;
    const E2* a = below;
    const E2* b = below + len_below;
    // out limbs are contiguous: out[0..half) = limb 0, out[half..2*half) = limb 1 ; index x = s*half + j
    for (size_t x = (size_t)blockIdx.x * NT + threadIdx.x; x < len_below; x += stride) out[x] = a[x] * b[x];
}

// logup layer: (p, q) <- (q1 p2 + q2 p1, q1 q2); `below` = [p1|p2|q1|q2] each len_below; out = [p1|p2|q1|q2] each len_below/2
__global__ void __launch_bounds__(NT) k_logup_layer(const E2* __restrict__ below, E2* __restrict__ out, size_t len_below) {
    const size_t stride = (size_t)gridDim.x * NT;
    const E2* p1 = below;
    const E2* p2 = below + len_below;
    const E2* q1 = below + 2 * len_below;
    const E2* q2 = below + 3 * len_below;
    // out p limbs: indices [0, len_below) cover p1|p2 (index x = s*half + j) ; q limbs follow
    for (size_t x = (size_t)blockIdx.x * NT + threadIdx.x; x < len_below; x += stride) {
        E2 a = q1[x], b = q2[x];
        out[x] = a * p2[x] + b * p1[x];
        out[len_below + x] = a * b;
    }
}

static int ceil_log2_sz(size_t x) {
    int l = 0;
    while (((size_t)1 << l) < x) l++;
    return l;
}
static size_t next_pow2_instance_padding(size_t n) {  // ceno_zkvm/src/scheme/hal.rs:127-128
    size_t p = 1;
    while (p < n) p <<= 1;
    return p < 2 ? 2 : p;
}

static void tower_release(ceno_hip_ctx* ctx, ceno_hip_tower* t) {
    if (!t) return;
    // (one call for all blocks of the tower: one lock, one stream query instead of a dozen — 42 us per tower, 160 towers per shard)
    std::vector<void*> blocks;
    for (size_t l = 0; l < t->layers.size(); l++)
        if (l == 0 || (int)l >= t->top_layers) blocks.push_back(t->layers[l]);  // layers 1 .. top_layers-1 point into layer 0's block
    ctx_free_many_on(ctx, blocks.data(), blocks.size(), ceno_tls_stream ? ceno_tls_stream : ctx->default_stream);
    delete t;
}

static int tower_alloc(ceno_hip_ctx* ctx, int num_vars, int n_limbs, ceno_hip_tower** out) {
    auto* t = new ceno_hip_tower();
    t->num_vars = num_vars;
    t->n_limbs = n_limbs;
    t->layers.assign(num_vars, nullptr);
    const int top = std::min(num_vars, TOWER_TOP_LAYERS);
    {
        void* p = nullptr;
        int rc = ctx_alloc(ctx, (size_t)n_limbs * (((size_t)1 << top) - 1) * sizeof(E2), &p);
        if (rc) {
            delete t;
            return rc;
        }
        for (int l = 0; l < top; l++) t->layers[l] = (E2*)p + (size_t)n_limbs * (((size_t)1 << l) - 1);
        t->top_layers = top;
    }
    for (int l = top; l < num_vars; l++) {
        void* p = nullptr;
        int rc = ctx_alloc(ctx, ((size_t)n_limbs << l) * sizeof(E2), &p);
        if (rc) {
            tower_release(ctx, t);
            return rc;
        }
        t->layers[l] = (E2*)p;
    }
    *out = t;
    return 0;
}

// the contiguous top of a tower in ONE launch: layers from-1 .. 0 out of layer `from` (<= 2^10 entries per limb), one workgroup,
// a barrier between layers — eleven launches of a few microseconds each were mostly kernel boundary
template <int LIMBS>
__global__ void __launch_bounds__(1024) k_tower_top(E2* __restrict__ block, int from) {
    for (int l = from - 1; l >= 0; l--) {
        const E2* below = block + (size_t)LIMBS * (((size_t)1 << (l + 1)) - 1);
        E2* out = block + (size_t)LIMBS * (((size_t)1 << l) - 1);
        const size_t len_below = (size_t)1 << (l + 1);
        for (size_t x = threadIdx.x; x < len_below; x += 1024) {
            if (LIMBS == 2) {
                out[x] = below[x] * below[len_below + x];
            } else {
                const E2 a = below[2 * len_below + x], b = below[3 * len_below + x];
                out[x] = a * below[len_below + x] + b * below[x];
                out[len_below + x] = a * b;
            }
        }
        __syncthreads();
    }
}

// ---- MANY towers per launch (ceno_hip_tower_build_many): the towers of all chips of a shard, level by level ---------------------------------
// A shard's ~54 chips have ~160 towers of ~11 dependent launches each; chip by chip on scheduler lanes that is ~1800 dependent launches over
// four hardware queues (~1.7 us apiece however many lanes there are).  Level-synchronous over ALL towers it is one interleave launch, one
// launch per layer size and one for the contiguous tops: ~15 launches, each wide enough to fill the device.
struct BlkRef {
    uint32_t job, blk, nblk;  // workgroup b of the launch is block `blk` of `nblk` of job `job`
};
struct IlvJob {
    RecArg ra;
    E2 *out0, *out1;
    size_t out_len;
};
struct LayerJob {
    const E2* below;
    E2* out;
    size_t len_below;
    int limbs, pad_;
};
struct TopJob {
    E2* block;
    int from, limbs;
};
struct CopyJob {
    const E2* src;
    E2* dst;
    size_t n;
};
__global__ void __launch_bounds__(NT) k_interleave_many(const IlvJob* __restrict__ jobs, const BlkRef* __restrict__ blks) {
    const BlkRef b = blks[blockIdx.x];
    const IlvJob& J = jobs[b.job];
    const RecArg& ra = J.ra;
    const size_t out_len = J.out_len, stride = (size_t)b.nblk * NT;
    const int log_s = ra.log_s, k = ra.k, ones = ra.ones;
    const size_t smask = ((size_t)1 << log_s) - 1, start1 = ra.start1;
    const E2 dflt = ra.dflt;
    E2 *out0 = J.out0, *out1 = J.out1;
    for (size_t o = (size_t)b.blk * NT + threadIdx.x; o < 2 * out_len; o += stride) {
        const int limb = o >= out_len;
        const size_t x = limb ? o - out_len : o;
        const size_t i = x >> log_s;
        const int j = (int)(x & smask);
        E2 v = dflt;
        if (!ones && j < k) {
            const uint32_t cnt = limb ? ra.cnt1[j] : ra.cnt0[j];
            if (i < cnt) {
                const size_t src = (limb ? start1 : 0) + i;
                if (ra.is_ext[j]) v = reinterpret_cast<const E2*>(ra.ptr[j])[src];
                else v = E2{ra.ptr[j][src], 0};
            }
        }
        (limb ? out1 : out0)[x] = v;
    }
}
// the last layer of a tower straight from the record EXPRESSIONS (the reference's build_prod_tower_from_virtual_ext_batch /
// build_logup_tower_from_virtual_ext_batch, ceno_zkvm/src/scheme/gpu/mod.rs:2365-2402): out[limb][i * S + j] = record (rec0 + j) of the plan at
// row limb * half + i — what k_wit_infer writes into a record table and k_interleave copies from it, without the table.  Records of a chip
// proof have one row per slot of the padded trace (num_instances = 2^num_vars: every limb is full, no default inside a record's range).
struct VtJob {
    WiPlan pl;
    int num_mles, num_terms, num_factors;
    int log_s, k, rec0, ones;
    int pad_plan, pad_;  // (host: which plan the job's pointers go to)
    size_t half, out_len;
    E2 dflt;
    E2 *out0, *out1;
};
__global__ void __launch_bounds__(NT) k_virtual_last_layer(const VtJob* __restrict__ jobs, const BlkRef* __restrict__ blks) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    const BlkRef b = blks[blockIdx.x];
    const VtJob J = jobs[b.job];
    WiLds L{};
    if (!J.ones) L = wi_stage<NT>(dyn, J.pl, J.num_mles, J.num_terms, J.num_factors, false);
    // one work item = one element of the last layer, consecutive lanes = consecutive slots of a row (coalesced writes; the lanes of a row read the
    // same witness words).  Measured against a wave walking the same records over 64 consecutive rows (coalesced reads, one control flow, a
    // 64-byte line written per lane): 0.49 against 0.62 ms per shard — the writes are what matters (profiles/r06_shard_wide_kernel_stats.csv)
    const size_t out_len = J.out_len, stride = (size_t)b.nblk * NT, smask = ((size_t)1 << J.log_s) - 1;
    for (size_t o = (size_t)b.blk * NT + threadIdx.x; o < 2 * out_len; o += stride) {
        const int limb = o >= out_len;
        const size_t x = limb ? o - out_len : o;
        const size_t i = x >> J.log_s;
        const int j = (int)(x & smask);
        E2 v = J.dflt;
        if (!J.ones && j < J.k) v = wi_eval(L, J.rec0 + j, (limb ? J.half : 0) + i);
        (limb ? J.out1 : J.out0)[x] = v;
    }
}
__global__ void __launch_bounds__(NT) k_layer_many(const LayerJob* __restrict__ jobs, const BlkRef* __restrict__ blks) {
    const BlkRef b = blks[blockIdx.x];
    const LayerJob J = jobs[b.job];
    const size_t stride = (size_t)b.nblk * NT, len_below = J.len_below;
    const E2* below = J.below;
    E2* out = J.out;
    if (J.limbs == 2) {
        for (size_t x = (size_t)b.blk * NT + threadIdx.x; x < len_below; x += stride) out[x] = below[x] * below[len_below + x];
    } else {
        for (size_t x = (size_t)b.blk * NT + threadIdx.x; x < len_below; x += stride) {
            const E2 a = below[2 * len_below + x], c = below[3 * len_below + x];
            out[x] = a * below[len_below + x] + c * below[x];
            out[len_below + x] = a * c;
        }
    }
}
__global__ void __launch_bounds__(1024) k_tower_top_many(const TopJob* __restrict__ jobs) {
    const TopJob J = jobs[blockIdx.x];
    E2* block = J.block;
    const size_t limbs = (size_t)J.limbs;
    for (int l = J.from - 1; l >= 0; l--) {
        const E2* below = block + limbs * (((size_t)1 << (l + 1)) - 1);
        E2* out = block + limbs * (((size_t)1 << l) - 1);
        const size_t len_below = (size_t)1 << (l + 1);
        for (size_t x = threadIdx.x; x < len_below; x += 1024) {
            if (J.limbs == 2) {
                out[x] = below[x] * below[len_below + x];
            } else {
                const E2 a = below[2 * len_below + x], b = below[3 * len_below + x];
                out[x] = a * below[len_below + x] + b * below[x];
                out[len_below + x] = a * b;
            }
        }
        __syncthreads();
    }
}
// the top blocks of many towers into one staging block (pinned host memory: the device writes it through PCIe), one workgroup per tower
__global__ void __launch_bounds__(256) k_copy_many(const CopyJob* __restrict__ jobs) {
    const CopyJob J = jobs[blockIdx.x];
    for (size_t i = threadIdx.x; i < J.n; i += 256) J.dst[i] = J.src[i];
}

static int tower_build_upper(ceno_hip_ctx* ctx, ceno_hip_tower* t, hipStream_t st) {
    static const bool fuse = !(getenv("CENO_HIP_TOWER_TOP_FUSED") && atoi(getenv("CENO_HIP_TOWER_TOP_FUSED")) == 0);  // A/B switch
    const int from = fuse ? std::min(t->num_vars - 1, t->top_layers - 1) : 0;  // layers below `from` come from the fused kernel
    for (int l = t->num_vars - 2; l >= from; l--) {
        size_t len_below = (size_t)1 << (l + 1);
        unsigned g = grid_for(len_below, NT, MAXB);
        if (t->n_limbs == 2) hipLaunchKernelGGL(k_prod_layer, dim3(g), dim3(NT), 0, st, t->layers[l + 1], t->layers[l], len_below);
        else hipLaunchKernelGGL(k_logup_layer, dim3(g), dim3(NT), 0, st, t->layers[l + 1], t->layers[l], len_below);
    }
    if (from >= 1) {
        if (t->n_limbs == 2) hipLaunchKernelGGL(k_tower_top<2>, dim3(1), dim3(1024), 0, st, t->layers[0], from);
        else hipLaunchKernelGGL(k_tower_top<4>, dim3(1), dim3(1024), 0, st, t->layers[0], from);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// fill RecArg following interleaving_mles_to_mles (utils.rs:410-457) for num_limbs = 2
static int make_rec_arg(ceno_hip_ctx* ctx, ceno_hip_mle* const* recs, int k, size_t num_instances, E2 dflt, RecArg& ra, size_t& out_len) {
    CHECK_ARG(ctx, k >= 1 && k <= MAX_REC, "tower: %d records unsupported (1..%d)", k, MAX_REC);
    const size_t np2 = next_pow2_instance_padding(num_instances);
    for (int j = 0; j < k; j++) {
        CHECK_ARG(ctx, recs[j], "tower: record %d is NULL", j);
        CHECK_ARG(ctx, recs[j]->len() <= np2, "tower: record %d longer than padded instance count", j);
    }
    const int log2_num_instances = ceil_log2_sz(np2);
    const size_t mle0_len = recs[0]->len();
    const size_t per_fanin_len = std::max<size_t>(mle0_len / 2, 1);
    const int log_s = ceil_log2_sz((size_t)k);
    out_len = (size_t)1 << (log_s + std::max(log2_num_instances - 1, 0));
    const size_t n_chunks = out_len >> log_s;
    ra.k = k;
    ra.log_s = log_s;
    ra.start1 = per_fanin_len;
    ra.dflt = dflt;
    ra.ones = 0;
    for (int limb = 0; limb < 2; limb++) {
        const size_t start = per_fanin_len * (size_t)limb;
        for (int j = 0; j < k; j++) {
            size_t cnt = 0;
            if (start < num_instances) {
                const size_t valid = std::min(per_fanin_len, num_instances - start);
                // Ext arm slices start..start+valid, Base arm start..start+per_fanin_len; `.get(range)` is
                // empty when the range exceeds the vector (utils.rs:436-455)
                cnt = recs[j]->is_ext ? valid : per_fanin_len;
                if (start + cnt > recs[j]->len()) cnt = 0;
                cnt = std::min(cnt, n_chunks);
            }
            (limb ? ra.cnt1 : ra.cnt0)[j] = (uint32_t)cnt;
        }
    }
    for (int j = 0; j < k; j++) {
        ra.ptr[j] = recs[j]->d;
        ra.is_ext[j] = (uint8_t)recs[j]->is_ext;
    }
    return 0;
}

// the rest of a many-tower build once the jobs of the LAST layers are known: one blob [head: the first kernel's jobs (and plans) | its block
// table | per layer size: jobs, blocks | the tops], one copy, then the first kernel (k_interleave_many over records that exist, or
// k_virtual_last_layer over record expressions), one k_layer_many per layer size and k_tower_top_many
static int towers_build_upper_many(ceno_hip_ctx* ctx, const std::vector<ceno_hip_tower*>& towers, hipStream_t st, const void* head, size_t head_bytes,
                                   const std::vector<BlkRef>& first_blks, size_t first_lds, bool virtual_first, size_t jobs_off = 0,
                                   const std::function<void(char*, const char*)>& fixup = {}) {
    auto add_blocks = [](std::vector<BlkRef>& v, uint32_t job, size_t work) {
        // a few elements per lane, at most 1024 workgroups per job: the launch as a whole is what fills the device
        const uint32_t nblk = (uint32_t)std::min<size_t>(std::max<size_t>((work + (size_t)NT * 4 - 1) / ((size_t)NT * 4), 1), 1024);
        for (uint32_t b = 0; b < nblk; b++) v.push_back(BlkRef{job, b, nblk});
    };
    int max_nv = 0;
    for (auto* t : towers) max_nv = std::max(max_nv, t->num_vars);
    // levels: layer l of tower t comes from a layer kernel for from_t <= l <= nv_t - 2, the layers below from_t from the top kernel
    struct Level {
        std::vector<LayerJob> jobs;
        std::vector<BlkRef> blks;
    };
    std::vector<Level> levels((size_t)std::max(max_nv, 1));
    std::vector<TopJob> tops;
    for (auto* t : towers) {
        const int from = std::min(t->num_vars - 1, t->top_layers - 1);
        for (int l = t->num_vars - 2; l >= from; l--) {
            Level& L = levels[(size_t)l];
            const size_t len_below = (size_t)1 << (l + 1);
            add_blocks(L.blks, (uint32_t)L.jobs.size(), len_below);
            L.jobs.push_back(LayerJob{t->layers[l + 1], t->layers[l], len_below, t->n_limbs, 0});
        }
        if (from >= 1) tops.push_back(TopJob{t->layers[0], from, t->n_limbs});
    }
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    size_t total = al(head_bytes);
    const size_t o_first_blk = total;
    total = al(total + first_blks.size() * sizeof(BlkRef));
    std::vector<size_t> o_jobs(levels.size(), 0), o_blks(levels.size(), 0);
    for (size_t l = 0; l < levels.size(); l++) {
        o_jobs[l] = total;
        total = al(total + levels[l].jobs.size() * sizeof(LayerJob));
        o_blks[l] = total;
        total = al(total + levels[l].blks.size() * sizeof(BlkRef));
    }
    const size_t o_tops = total;
    total = al(total + tops.size() * sizeof(TopJob));
    void* d_blob = nullptr;
    TRY(ctx_alloc(ctx, std::max<size_t>(total, 16), &d_blob));
    std::vector<char> blob(total, 0);
    memcpy(blob.data(), head, head_bytes);
    if (fixup) fixup(blob.data(), (const char*)d_blob);
    if (!first_blks.empty()) memcpy(blob.data() + o_first_blk, first_blks.data(), first_blks.size() * sizeof(BlkRef));
    for (size_t l = 0; l < levels.size(); l++) {
        if (!levels[l].jobs.empty()) memcpy(blob.data() + o_jobs[l], levels[l].jobs.data(), levels[l].jobs.size() * sizeof(LayerJob));
        if (!levels[l].blks.empty()) memcpy(blob.data() + o_blks[l], levels[l].blks.data(), levels[l].blks.size() * sizeof(BlkRef));
    }
    if (!tops.empty()) memcpy(blob.data() + o_tops, tops.data(), tops.size() * sizeof(TopJob));
    // (pageable source: the runtime has captured it when hipMemcpyAsync returns)
    hipError_t e = hipMemcpyAsync(d_blob, blob.data(), total, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        char* d = (char*)d_blob;
        if (virtual_first)
            hipLaunchKernelGGL(k_virtual_last_layer, dim3((unsigned)first_blks.size()), dim3(NT), first_lds, st, (const VtJob*)(d + jobs_off), (const BlkRef*)(d + o_first_blk));
        else
            hipLaunchKernelGGL(k_interleave_many, dim3((unsigned)first_blks.size()), dim3(NT), 0, st, (const IlvJob*)d, (const BlkRef*)(d + o_first_blk));
        for (int l = (int)levels.size() - 1; l >= 0; l--)
            if (!levels[(size_t)l].blks.empty())
                hipLaunchKernelGGL(k_layer_many, dim3((unsigned)levels[(size_t)l].blks.size()), dim3(NT), 0, st, (const LayerJob*)(d + o_jobs[(size_t)l]),
                                   (const BlkRef*)(d + o_blks[(size_t)l]));
        if (!tops.empty()) hipLaunchKernelGGL(k_tower_top_many, dim3((unsigned)tops.size()), dim3(1024), 0, st, (const TopJob*)(d + o_tops));
        e = hipGetLastError();
    }
    ctx_free(ctx, d_blob);  // (tagged with this stream: handed to another one only after the launches above have drained)
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "tower_build_many: %s", hipGetErrorString(e));
    return 0;
}

extern "C" {

int ceno_hip_tower_build_prod(ceno_hip_ctx* ctx, ceno_hip_mle* const* records, int k, size_t num_instances, const uint64_t* default2,
                              ceno_hip_stream s, ceno_hip_tower** out) {
    CENO_TIMED("tower_build_prod");
    CHECK_ARG(ctx, records && default2 && out, "NULL argument");
    hipStream_t st = ctx_stream(ctx, s);
    RecArg ra{};
    size_t out_len = 0;
    TRY(make_rec_arg(ctx, records, k, num_instances, E2{default2[0], default2[1]}, ra, out_len));
    const int num_vars = ceil_log2_sz(out_len) + 1;
    ceno_hip_tower* t = nullptr;
    TRY(tower_alloc(ctx, num_vars, 2, &t));
    E2* last = t->layers[num_vars - 1];
    hipLaunchKernelGGL(k_interleave, dim3(grid_for(2 * out_len, NT, MAXB)), dim3(NT), 0, st, ra, last, last + out_len, out_len);
    int rc = tower_build_upper(ctx, t, st);
    if (rc) {
        tower_release(ctx, t);
        return rc;
    }
    *out = t;
    return 0;
}

int ceno_hip_tower_build_logup(ceno_hip_ctx* ctx, ceno_hip_mle* const* p_records, ceno_hip_mle* const* q_records, int k,
                               size_t num_instances, const uint64_t* default2, ceno_hip_stream s, ceno_hip_tower** out) {
    CENO_TIMED("tower_build_logup");
    CHECK_ARG(ctx, q_records && default2 && out, "NULL argument");
    hipStream_t st = ctx_stream(ctx, s);
    RecArg rq{};
    size_t out_len = 0;
    const E2 dflt{default2[0], default2[1]};
    TRY(make_rec_arg(ctx, q_records, k, num_instances, dflt, rq, out_len));
    const int num_vars = ceil_log2_sz(out_len) + 1;
    ceno_hip_tower* t = nullptr;
    TRY(tower_alloc(ctx, num_vars, 4, &t));
    E2* last = t->layers[num_vars - 1];
    unsigned g = grid_for(2 * out_len, NT, MAXB);
    if (p_records) {
        RecArg rp{};
        size_t pl = 0;
        int rc = make_rec_arg(ctx, p_records, k, num_instances, dflt, rp, pl);
        if (rc || pl != out_len) {
            tower_release(ctx, t);
            return rc ? rc : ctx_fail(ctx, CENO_HIP_ERR_INVALID, "logup numerator / denominator shapes differ");
        }
        hipLaunchKernelGGL(k_interleave, dim3(g), dim3(NT), 0, st, rp, last, last + out_len, out_len);
    } else {
        // numerators absent: the input layer's p limbs are all ONE (utils.rs:558-579)
        RecArg rp{};
        rp.ones = 1;
        rp.dflt = e2_one();
        rp.log_s = 0;
        hipLaunchKernelGGL(k_interleave, dim3(g), dim3(NT), 0, st, rp, last, last + out_len, out_len);
    }
    hipLaunchKernelGGL(k_interleave, dim3(g), dim3(NT), 0, st, rq, last + 2 * out_len, last + 3 * out_len, out_len);
    int rc = tower_build_upper(ctx, t, st);
    if (rc) {
        tower_release(ctx, t);
        return rc;
    }
    *out = t;
    return 0;
}

int ceno_hip_tower_from_last_layer(ceno_hip_ctx* ctx, ceno_hip_mle* const* limbs, int n_limbs, ceno_hip_stream s, ceno_hip_tower** out) {
    CHECK_ARG(ctx, limbs && out && (n_limbs == 2 || n_limbs == 4), "tower: n_limbs must be 2 or 4");
    hipStream_t st = ctx_stream(ctx, s);
    // logup with absent numerators: limbs[0], limbs[1] may be NULL
    const ceno_hip_mle* ref = limbs[n_limbs - 1];
    CHECK_ARG(ctx, ref, "tower: last limb is NULL");
    for (int i = 0; i < n_limbs; i++) {
        if (!limbs[i]) {
            CHECK_ARG(ctx, n_limbs == 4 && i < 2, "tower: limb %d is NULL", i);
            continue;
        }
        CHECK_ARG(ctx, limbs[i]->is_ext && limbs[i]->num_vars == ref->num_vars, "tower: limbs must be ext tables of one size");
    }
    const int num_vars = ref->num_vars + 1;
    ceno_hip_tower* t = nullptr;
    TRY(tower_alloc(ctx, num_vars, n_limbs, &t));
    const size_t len = ref->len();
    E2* last = t->layers[num_vars - 1];
    for (int i = 0; i < n_limbs; i++) {
        if (limbs[i]) {
            hipError_t e = hipMemcpyAsync(last + (size_t)i * len, limbs[i]->d, len * sizeof(E2), hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) {
                tower_release(ctx, t);
                return ctx_fail(ctx, CENO_HIP_ERR_HIP, "tower copy: %s", hipGetErrorString(e));
            }
        }
    }
    if (n_limbs == 4 && (!limbs[0] || !limbs[1])) {
        RecArg rp{};
        rp.ones = 1;
        rp.dflt = e2_one();
        hipLaunchKernelGGL(k_interleave, dim3(grid_for(2 * len, NT, MAXB)), dim3(NT), 0, st, rp, last, last + len, len);
    }
    int rc = tower_build_upper(ctx, t, st);
    if (rc) {
        tower_release(ctx, t);
        return rc;
    }
    *out = t;
    return 0;
}

int ceno_hip_tower_build_many(ceno_hip_ctx* ctx, const ceno_hip_tower_spec* specs, int n, ceno_hip_stream s, ceno_hip_tower** out) {
    CENO_TIMED("tower_build_many");
    CHECK_ARG(ctx, specs && out && n >= 1 && n <= 4096, "tower_build_many: bad arguments");
    hipStream_t st = ctx_stream(ctx, s);
    std::vector<ceno_hip_tower*> towers((size_t)n, nullptr);
    auto release_all = [&]() {
        for (auto* t : towers) tower_release(ctx, t);
    };
    std::vector<IlvJob> ilv;
    std::vector<BlkRef> ilv_blk;
    auto add_blocks = [](std::vector<BlkRef>& v, uint32_t job, size_t work) {
        // a few elements per lane, at most 1024 workgroups per job: the launch as a whole is what fills the device
        const uint32_t nblk = (uint32_t)std::min<size_t>(std::max<size_t>((work + (size_t)NT * 4 - 1) / ((size_t)NT * 4), 1), 1024);
        for (uint32_t b = 0; b < nblk; b++) v.push_back(BlkRef{job, b, nblk});
    };
    int max_nv = 0;
    for (int i = 0; i < n; i++) {
        const ceno_hip_tower_spec& S = specs[i];
        int rc = 0;
        if (!S.records || S.k < 1) rc = ctx_fail(ctx, CENO_HIP_ERR_INVALID, "tower_build_many: tower %d has no records", i);
        const E2 dflt{S.default2[0], S.default2[1]};
        IlvJob q{};
        size_t out_len = 0;
        if (!rc) rc = make_rec_arg(ctx, S.records, S.k, S.num_instances, dflt, q.ra, out_len);
        const int num_vars = ceil_log2_sz(out_len) + 1;
        if (!rc) rc = tower_alloc(ctx, num_vars, S.logup ? 4 : 2, &towers[(size_t)i]);
        if (!rc) {
            E2* last = towers[(size_t)i]->layers[num_vars - 1];
            if (S.logup) {
                IlvJob pj{};
                if (S.numerators) {
                    size_t pl = 0;
                    rc = make_rec_arg(ctx, S.numerators, S.k, S.num_instances, dflt, pj.ra, pl);
                    if (!rc && pl != out_len) rc = ctx_fail(ctx, CENO_HIP_ERR_INVALID, "logup numerator / denominator shapes differ");
                } else {  // numerators absent: the input layer's p limbs are all ONE (utils.rs:558-579)
                    pj.ra.ones = 1;
                    pj.ra.dflt = e2_one();
                    pj.ra.log_s = 0;
                }
                pj.out0 = last;
                pj.out1 = last + out_len;
                pj.out_len = out_len;
                q.out0 = last + 2 * out_len;
                q.out1 = last + 3 * out_len;
                q.out_len = out_len;
                if (!rc) {
                    add_blocks(ilv_blk, (uint32_t)ilv.size(), 2 * out_len);
                    ilv.push_back(pj);
                }
            } else {
                q.out0 = last;
                q.out1 = last + out_len;
                q.out_len = out_len;
            }
            if (!rc) {
                add_blocks(ilv_blk, (uint32_t)ilv.size(), 2 * out_len);
                ilv.push_back(q);
            }
            max_nv = std::max(max_nv, num_vars);
        }
        if (rc) {
            release_all();
            return rc;
        }
    }
    const int rc_up = towers_build_upper_many(ctx, towers, st, ilv.data(), ilv.size() * sizeof(IlvJob), ilv_blk, 0, false);
    if (rc_up) {
        release_all();
        return rc_up;
    }
    for (int i = 0; i < n; i++) out[i] = towers[(size_t)i];
    return 0;
}

int ceno_hip_tower_build_many_virtual(ceno_hip_ctx* ctx, const ceno_hip_wit_plan* plans, int n_plans, const ceno_hip_virtual_tower_spec* specs, int n,
                                      ceno_hip_stream s, ceno_hip_tower** out) {
    CENO_TIMED("tower_build_many_virtual");
    CHECK_ARG(ctx, plans && specs && out && n_plans >= 1 && n >= 1 && n <= 4096, "tower_build_many_virtual: bad arguments");
    hipStream_t st = ctx_stream(ctx, s);
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    // the plans, each once, in the head of the blob (the jobs point into it); a plan too large for the LDS stage: not this path
    struct Lay {
        size_t o_slots, o_coeffs, o_toff, o_tidx, o_ooff, n_fac;
    };
    std::vector<Lay> lay((size_t)n_plans);
    size_t plan_bytes = 0, max_lds = 0;
    for (int i = 0; i < n_plans; i++) {
        const ceno_hip_wit_plan& P = plans[i];
        CHECK_ARG(ctx, P.mles && P.term_coeffs && P.term_offsets && P.term_mle_idx && P.out_term_offsets, "tower_build_many_virtual: plan %d: NULL argument", i);
        CHECK_ARG(ctx, P.num_mles >= 1 && P.num_terms >= 0 && P.num_outs >= 1 && P.num_vars >= 1 && P.num_vars < 40, "tower_build_many_virtual: plan %d is empty", i);
        CHECK_ARG(ctx, P.out_term_offsets[0] == 0 && (int)P.out_term_offsets[P.num_outs] == P.num_terms, "tower_build_many_virtual: plan %d: out_term_offsets must cover all terms", i);
        for (int j = 0; j < P.num_mles; j++)
            CHECK_ARG(ctx, P.mles[j] && P.mles[j]->num_vars == P.num_vars, "tower_build_many_virtual: plan %d: mle %d must have %d variables", i, j, P.num_vars);
        Lay& L = lay[(size_t)i];
        L.n_fac = P.term_offsets[P.num_terms];
        for (uint32_t k = 0; k < L.n_fac; k++) CHECK_ARG(ctx, (int)P.term_mle_idx[k] < P.num_mles, "tower_build_many_virtual: plan %d: term factor %u out of range", i, P.term_mle_idx[k]);
        const size_t lds = wit_infer_lds(P.num_mles, P.num_terms, (int)L.n_fac, P.num_outs);
        if (lds > 60 * 1024) return ctx_fail(ctx, CENO_HIP_ERR_UNSUPPORTED, "tower_build_many_virtual: plan %d does not fit the LDS stage", i);
        max_lds = std::max(max_lds, lds);
        L.o_slots = plan_bytes;
        plan_bytes = al(plan_bytes + (size_t)P.num_mles * sizeof(WiSlot));
        L.o_coeffs = plan_bytes;
        plan_bytes = al(plan_bytes + (size_t)P.num_terms * sizeof(E2));
        L.o_toff = plan_bytes;
        plan_bytes = al(plan_bytes + ((size_t)P.num_terms + 1) * 4);
        L.o_tidx = plan_bytes;
        plan_bytes = al(plan_bytes + L.n_fac * 4);
        L.o_ooff = plan_bytes;
        plan_bytes = al(plan_bytes + ((size_t)P.num_outs + 1) * 4);
    }
    std::vector<ceno_hip_tower*> towers((size_t)n, nullptr);
    auto release_all = [&]() {
        for (auto* t : towers) tower_release(ctx, t);
    };
    std::vector<VtJob> jobs;
    std::vector<BlkRef> blks;
    auto add_blocks = [](std::vector<BlkRef>& v, uint32_t job, size_t work) {
        const uint32_t nblk = (uint32_t)std::min<size_t>(std::max<size_t>((work + (size_t)NT * 2 - 1) / ((size_t)NT * 2), 1), 2048);
        for (uint32_t b = 0; b < nblk; b++) v.push_back(BlkRef{job, b, nblk});
    };
    for (int i = 0; i < n; i++) {
        const ceno_hip_virtual_tower_spec& S = specs[i];
        int rc = 0;
        if (S.plan < 0 || S.plan >= n_plans || S.k < 1 || S.k > MAX_REC || S.first_record < 0 || S.first_record + S.k > plans[S.plan].num_outs ||
            (S.first_numerator >= 0 && (!S.logup || S.first_numerator + S.k > plans[S.plan].num_outs)))
            rc = ctx_fail(ctx, CENO_HIP_ERR_INVALID, "tower_build_many_virtual: tower %d names records its plan does not have", i);
        if (rc) {
            release_all();
            return rc;
        }
        const ceno_hip_wit_plan& P = plans[S.plan];
        const size_t len = (size_t)1 << P.num_vars, half = len / 2;
        const int log_s = ceil_log2_sz((size_t)S.k);
        const size_t out_len = half << log_s;
        const int num_vars = ceil_log2_sz(out_len) + 1;
        rc = tower_alloc(ctx, num_vars, S.logup ? 4 : 2, &towers[(size_t)i]);
        if (rc) {
            release_all();
            return rc;
        }
        E2* last = towers[(size_t)i]->layers[num_vars - 1];
        VtJob q{};
        q.pl.num_outs = P.num_outs;  // (the pointers are set once the blob has its address)
        q.num_mles = P.num_mles;
        q.num_terms = P.num_terms;
        q.num_factors = (int)lay[(size_t)S.plan].n_fac;
        q.log_s = log_s;
        q.k = S.k;
        q.rec0 = S.first_record;
        q.half = half;
        q.out_len = out_len;
        q.dflt = E2{S.default2[0], S.default2[1]};
        q.pad_plan = S.plan;
        if (S.logup) {
            VtJob pj = q;
            if (S.first_numerator >= 0) pj.rec0 = S.first_numerator;
            else {  // numerators absent: the input layer's p limbs are all ONE (utils.rs:558-579)
                pj.ones = 1;
                pj.dflt = e2_one();
            }
            pj.out0 = last;
            pj.out1 = last + out_len;
            q.out0 = last + 2 * out_len;
            q.out1 = last + 3 * out_len;
            add_blocks(blks, (uint32_t)jobs.size(), 2 * out_len);
            jobs.push_back(pj);
        } else {
            q.out0 = last;
            q.out1 = last + out_len;
        }
        add_blocks(blks, (uint32_t)jobs.size(), 2 * out_len);
        jobs.push_back(q);
    }
    // the head of the blob: plans, then the jobs; towers_build_upper_many appends the levels and launches everything
    std::vector<char> head(al(plan_bytes) + jobs.size() * sizeof(VtJob), 0);
    for (int i = 0; i < n_plans; i++) {
        const Lay& L = lay[(size_t)i];
        const ceno_hip_wit_plan& P = plans[i];
        for (int j = 0; j < P.num_mles; j++) reinterpret_cast<WiSlot*>(head.data() + L.o_slots)[j] = WiSlot{P.mles[j]->d, P.mles[j]->is_ext, 0};
        for (int t = 0; t < P.num_terms; t++) reinterpret_cast<E2*>(head.data() + L.o_coeffs)[t] = E2{P.term_coeffs[2 * t], P.term_coeffs[2 * t + 1]};
        memcpy(head.data() + L.o_toff, P.term_offsets, ((size_t)P.num_terms + 1) * 4);
        if (L.n_fac) memcpy(head.data() + L.o_tidx, P.term_mle_idx, L.n_fac * 4);
        memcpy(head.data() + L.o_ooff, P.out_term_offsets, ((size_t)P.num_outs + 1) * 4);
    }
    // (the jobs' plan pointers are offsets into the blob until it has an address: fixed up by the helper through `fix`)
    struct Fix {
        const std::vector<Lay>* lay;
        std::vector<VtJob>* jobs;
        size_t jobs_off;
    } fix{&lay, &jobs, al(plan_bytes)};
    auto fixup = [&](char* host, const char* dev) {
        for (size_t j = 0; j < jobs.size(); j++) {
            VtJob& J = reinterpret_cast<VtJob*>(host + fix.jobs_off)[j];
            J = jobs[j];
            const Lay& L = lay[(size_t)jobs[j].pad_plan];
            J.pl.mles = reinterpret_cast<const WiSlot*>(dev + L.o_slots);
            J.pl.coeffs = reinterpret_cast<const E2*>(dev + L.o_coeffs);
            J.pl.term_off = reinterpret_cast<const uint32_t*>(dev + L.o_toff);
            J.pl.term_idx = reinterpret_cast<const uint32_t*>(dev + L.o_tidx);
            J.pl.out_term_off = reinterpret_cast<const uint32_t*>(dev + L.o_ooff);
            J.pl.outs = nullptr;
        }
    };
    const int rc_up = towers_build_upper_many(ctx, towers, st, head.data(), head.size(), blks, max_lds, true, fix.jobs_off, fixup);
    if (rc_up) {
        release_all();
        return rc_up;
    }
    for (int i = 0; i < n; i++) out[i] = towers[(size_t)i];
    return 0;
}

int ceno_hip_tower_num_vars(const ceno_hip_tower* t) { return t ? t->num_vars : -1; }
int ceno_hip_tower_num_limbs(const ceno_hip_tower* t) { return t ? t->n_limbs : -1; }

int ceno_hip_tower_layer(ceno_hip_ctx* ctx, ceno_hip_tower* t, int layer, int limb, ceno_hip_mle** out) {
    CHECK_ARG(ctx, t && out, "NULL argument");
    CHECK_ARG(ctx, layer >= 0 && layer < t->num_vars && limb >= 0 && limb < t->n_limbs, "tower layer/limb out of range");
    return ceno_hip_mle_wrap(ctx, reinterpret_cast<uint64_t*>(t->layers[layer] + ((size_t)limb << layer)), layer, 1, out);
}

const uint64_t* ceno_hip_tower_layer_ptr(const ceno_hip_tower* t, int layer, int limb) {
    if (!t || layer < 0 || layer >= t->num_vars || limb < 0 || limb >= t->n_limbs) return nullptr;
    return reinterpret_cast<const uint64_t*>(t->layers[layer] + ((size_t)limb << layer));
}

int ceno_hip_tower_out_evals(ceno_hip_ctx* ctx, ceno_hip_tower* t, uint64_t* out, ceno_hip_stream s) {
    CHECK_ARG(ctx, t && out, "NULL argument");
    if (t->host_top_layers >= 1) {  // prefetched (ceno_hip_tower_prefetch_tops): layer 0 heads the block
        memcpy(out, t->host_top.data(), (size_t)t->n_limbs * sizeof(E2));
        return 0;
    }
    hipStream_t st = ctx_stream(ctx, s);
    HIP_TRY(ctx, hipMemcpyAsync(out, t->layers[0], (size_t)t->n_limbs * sizeof(E2), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return 0;
}

int ceno_hip_tower_prefetch_tops(ceno_hip_ctx* ctx, ceno_hip_tower* const* towers, int n_towers, int n_layers, ceno_hip_stream s) {
    CENO_TIMED("tower_prefetch_tops");
    CHECK_ARG(ctx, towers && n_towers >= 0 && n_layers >= 1, "tower prefetch: bad arguments");
    hipStream_t st = ctx_stream(ctx, s);
    // every tower's top block into ONE pinned staging block, one wait for all of them (a copy into pageable memory per tower — and per
    // out-evaluation — cost a staged blit and a synchronisation each: nine per chip proof, ~30 us apiece)
    std::vector<size_t> off((size_t)n_towers + 1, 0);
    std::vector<int> nl((size_t)n_towers, 0);
    for (int i = 0; i < n_towers; i++) {
        CHECK_ARG(ctx, towers[i], "tower prefetch: tower %d is NULL", i);
        nl[(size_t)i] = std::min(n_layers, towers[i]->top_layers);
        off[(size_t)i + 1] = off[(size_t)i] + (size_t)towers[i]->n_limbs * (((size_t)1 << nl[(size_t)i]) - 1) * sizeof(E2);
    }
    if (off[(size_t)n_towers] == 0) return 0;
    void *hb = nullptr, *db = nullptr;
    TRY(ctx_pinned_alloc(ctx, off[(size_t)n_towers], &hb, &db));
    hipError_t e = hipSuccess;
    void* d_jobs = nullptr;
    if (n_towers > 6) {
        // many towers (a whole shard's): one kernel writes every top block into the staging block — a copy command per tower costs ~5 us
        std::vector<CopyJob> cj((size_t)n_towers);
        for (int i = 0; i < n_towers; i++)
            cj[(size_t)i] = CopyJob{towers[i]->layers[0], reinterpret_cast<E2*>((char*)db + off[(size_t)i]), (off[(size_t)i + 1] - off[(size_t)i]) / sizeof(E2)};
        int rc = ctx_alloc(ctx, cj.size() * sizeof(CopyJob), &d_jobs);
        if (rc) {
            ctx_pinned_free(ctx, hb);
            return rc;
        }
        e = hipMemcpyAsync(d_jobs, cj.data(), cj.size() * sizeof(CopyJob), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_copy_many, dim3((unsigned)n_towers), dim3(256), 0, st, (const CopyJob*)d_jobs);
            e = hipGetLastError();
        }
    } else {
        for (int i = 0; i < n_towers && e == hipSuccess; i++)
            if (off[(size_t)i + 1] > off[(size_t)i])
                e = hipMemcpyAsync((char*)hb + off[(size_t)i], towers[i]->layers[0], off[(size_t)i + 1] - off[(size_t)i], hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d_jobs) ctx_free(ctx, d_jobs);
    if (e == hipSuccess) {
        for (int i = 0; i < n_towers; i++) {
            const uint64_t* src = reinterpret_cast<const uint64_t*>((char*)hb + off[(size_t)i]);
            towers[i]->host_top.assign(src, src + (off[(size_t)i + 1] - off[(size_t)i]) / 8);
            towers[i]->host_top_layers = nl[(size_t)i];
        }
    }
    ctx_pinned_free(ctx, hb);
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "tower prefetch: %s", hipGetErrorString(e));
    return 0;
}

int ceno_hip_tower_download_top(ceno_hip_ctx* ctx, ceno_hip_tower* t, int n_layers, uint64_t* host_out, ceno_hip_stream s) {
    CHECK_ARG(ctx, t && host_out, "NULL argument");
    CHECK_ARG(ctx, n_layers >= 1 && n_layers <= t->top_layers, "tower: %d top layers requested, %d are contiguous", n_layers, t->top_layers);
    if (n_layers <= t->host_top_layers) {  // prefetched (ceno_hip_tower_prefetch_tops)
        memcpy(host_out, t->host_top.data(), (size_t)t->n_limbs * (((size_t)1 << n_layers) - 1) * sizeof(E2));
        return 0;
    }
    hipStream_t st = ctx_stream(ctx, s);
    HIP_TRY(ctx, hipMemcpyAsync(host_out, t->layers[0], (size_t)t->n_limbs * (((size_t)1 << n_layers) - 1) * sizeof(E2), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return 0;
}
int ceno_hip_tower_top_layers(const ceno_hip_tower* t) { return t ? t->top_layers : 0; }

int ceno_hip_tower_free(ceno_hip_ctx* ctx, ceno_hip_tower* t) {
    CENO_TIMED("tower_free");
    tower_release(ctx, t);
    return 0;
}

// defined in sumcheck.hip
}  // extern "C"

void sumcheck_adopt_mle(ceno_hip_sumcheck* sc, ceno_hip_mle* m);
void sumcheck_enable_tower_fast(ceno_hip_sumcheck* sc, const uint64_t* rt, int n_prod, int n_logup);
void sumcheck_adopt_alloc(ceno_hip_sumcheck* sc, void* p);

extern "C" int ceno_hip_tower_layer_sumcheck_begin(ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup,
                                                   int n_logup, int layer, const uint64_t* out_rt, const uint64_t* alpha_pows,
                                                   ceno_hip_stream s, ceno_hip_sumcheck** out) {
    CHECK_ARG(ctx, out && out_rt && alpha_pows && layer >= 1, "tower layer sumcheck: bad arguments");
    CENO_TIMED("tower_layer_sumcheck_begin");
    CHECK_ARG(ctx, (n_prod == 0 || prod) && (n_logup == 0 || logup), "NULL tower list");
    (void)ctx_stream(ctx, s);  // allocations below belong to work on `s`: bind the thread first (pool tags, include/ceno_hip.h "Memory")
    // sum_x eq(x, out_rt) * [ sum_i alpha_i a_i b_i + sum_k (alpha_n (p1 q2 + p2 q1) + alpha_d q1 q2) ]
    // -> one common-factor group (eq) over all residual terms  (scheme/cpu/mod.rs:417-494)
    ceno_hip_mle* eq = nullptr;
    void* eq_tmp = nullptr;  // scratch of the eq build; lives (like eq) until the sumcheck handle is freed
    static HostTimeSlot* const hs_eq = host_time_slot("tower_layer_sumcheck_begin: eq table allocation");
    static HostTimeSlot* const hs_views = host_time_slot("tower_layer_sumcheck_begin: views + terms");
    static HostTimeSlot* const hs_build = host_time_slot("tower_layer_sumcheck_begin: sumcheck handle");
    static HostTimeSlot* const hs_launch = host_time_slot("tower_layer_sumcheck_begin: eq build + set-up launch");
    timespec ht = host_time_mark();
    TRY(ceno_hip_mle_alloc(ctx, layer, 1, &eq));
    ht = host_time_add(hs_eq, ht);
    // The eq table is built AFTER the handle below exists: a small table's kernel then also does the handle's set-up work
    // (launch_eq_build_with_setup), one dependent launch instead of two in front of every layer.  CENO_HIP_TOWER_EQ_SETUP=0 keeps them apart.
    std::vector<ceno_hip_mle*> mles{eq};
    std::vector<ceno_hip_mle*> views;
    std::vector<uint64_t> coeffs;
    std::vector<uint32_t> toff{0}, tidx, gterms;
    auto cleanup = [&]() {
        for (auto* v : views) ceno_hip_mle_free(ctx, v);
    };
    auto add_term = [&](const uint64_t* c, std::initializer_list<uint32_t> f) {
        coeffs.push_back(c[0]);
        coeffs.push_back(c[1]);
        for (uint32_t x : f) tidx.push_back(x);
        gterms.push_back((uint32_t)toff.size() - 1);
        toff.push_back((uint32_t)tidx.size());
    };
    int rc = 0, n_prod_live = 0, n_logup_live = 0;  // towers that have this layer
    for (int i = 0; i < n_prod && !rc; i++) {
        if (!prod[i] || prod[i]->n_limbs != 2) { rc = ctx_fail(ctx, CENO_HIP_ERR_INVALID, "prod tower %d invalid", i); break; }
        if (prod[i]->num_vars <= layer) continue;  // spec has no layer `layer`
        uint32_t base = (uint32_t)mles.size();
        for (int l = 0; l < 2 && !rc; l++) {
            ceno_hip_mle* v = nullptr;
            rc = ceno_hip_tower_layer(ctx, prod[i], layer, l, &v);
            if (!rc) { views.push_back(v); mles.push_back(v); }
        }
        if (!rc) add_term(alpha_pows + 2 * i, {base, base + 1});
        n_prod_live++;
    }
    for (int i = 0; i < n_logup && !rc; i++) {
        if (!logup[i] || logup[i]->n_limbs != 4) { rc = ctx_fail(ctx, CENO_HIP_ERR_INVALID, "logup tower %d invalid", i); break; }
        if (logup[i]->num_vars <= layer) continue;
        uint32_t base = (uint32_t)mles.size();
        for (int l = 0; l < 4 && !rc; l++) {
            ceno_hip_mle* v = nullptr;
            rc = ceno_hip_tower_layer(ctx, logup[i], layer, l, &v);
            if (!rc) { views.push_back(v); mles.push_back(v); }
        }
        if (rc) break;
        const uint64_t* an = alpha_pows + 2 * (n_prod + 2 * i);
        const uint64_t* ad = alpha_pows + 2 * (n_prod + 2 * i + 1);
        uint32_t p1 = base, p2 = base + 1, q1 = base + 2, q2 = base + 3;
        add_term(an, {p1, q2});
        add_term(an, {p2, q1});
        add_term(ad, {q1, q2});
        n_logup_live++;
    }
    if (!rc && toff.size() == 1) rc = ctx_fail(ctx, CENO_HIP_ERR_INVALID, "no tower has layer %d", layer);
    if (rc) {
        cleanup();
        (void)hipStreamSynchronize(ctx_stream(ctx, s));
        ctx_free(ctx, eq_tmp);
        ceno_hip_mle_free(ctx, eq);
        return rc;
    }
    ht = host_time_add(hs_views, ht);
    std::vector<uint32_t> goff{0, (uint32_t)gterms.size()}, coff{0, 1}, cidx{0};
    ceno_hip_sumcheck_plan plan{};
    plan.num_mles = (int)mles.size();
    plan.num_terms = (int)toff.size() - 1;
    plan.term_coeffs = coeffs.data();
    plan.term_offsets = toff.data();
    plan.term_mle_idx = tidx.data();
    plan.num_groups = 1;
    plan.group_term_offsets = goff.data();
    plan.group_term_idx = gterms.data();
    plan.common_offsets = coff.data();
    plan.common_mle_idx = cidx.data();
    plan.max_num_vars = layer;
    plan.max_degree = 3;
    static const bool fuse = !(getenv("CENO_HIP_TOWER_EQ_SETUP") && atoi(getenv("CENO_HIP_TOWER_EQ_SETUP")) == 0);
    SetupJob job;
    hipStream_t st = ctx_stream(ctx, s);
    rc = fuse ? sumcheck_begin_deferred(ctx, mles.data(), &plan, st, out, &job) : ceno_hip_sumcheck_begin(ctx, mles.data(), &plan, s, out);
    cleanup();  // views are borrowed wrappers; the sumcheck copied the pointers
    ht = host_time_add(hs_build, ht);
    if (!rc) {
        int fused = 1;
        if (job.dst) fused = launch_eq_build_with_setup(ctx, out_rt, layer, gl::e2_one(), eq->d, st, job);
        if (fused < 0 || fused > 1) rc = fused;
        else if (fused == 1) {  // a large table (or nothing deferred): the two kernels separately
            if (job.dst) launch_setup_job(job, st);
            rc = launch_eq_build(ctx, out_rt, layer, gl::e2_one(), eq->d, st, &eq_tmp);
        }
        if (rc) {
            ceno_hip_sumcheck_free(ctx, *out);
            *out = nullptr;
        }
    }
    if (rc) {
        (void)hipStreamSynchronize(ctx_stream(ctx, s));
        ctx_free(ctx, eq_tmp);
        ceno_hip_mle_free(ctx, eq);
        return rc;
    }
    ht = host_time_add(hs_launch, ht);
    const bool fast = !(getenv("CENO_HIP_TOWER_FAST") && atoi(getenv("CENO_HIP_TOWER_FAST")) == 0);  // A/B switch (read per call: tests toggle it)
    if (fast) sumcheck_enable_tower_fast(*out, out_rt, n_prod_live, n_logup_live);
    sumcheck_adopt_mle(*out, eq);      // eq lives as long as the sumcheck
    sumcheck_adopt_alloc(*out, eq_tmp);
    return 0;
}
