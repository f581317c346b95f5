// MLE kernels: synthetic fill, fold (fix_variable), eq / selector tables, evaluate.
// Reference semantics: EXT multilinear_extensions `build_eq_x_r_vec`, `MultilinearExtension::
// {evaluate, fix_variables}` (call sites gkr_iop/src/selector.rs:140-194, layer/cpu/mod.rs:266);
// selectors gkr_iop/src/selector.rs:131-245; LSB-first variable order gkr_iop/src/utils.rs:215-232.
// All kernels are HBM-streaming: 16 B per lane coalesced loads/stores, grid-stride.
#include <algorithm>

#include "common.hpp"
#include "reduce.hpp"

using namespace gl;

static constexpr int NT = 256;
static constexpr unsigned MAXB = 2048;  // 256 CUs x 8 blocks

struct PointArg {
    E2 r[40];
};

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) k_fill_splitmix(uint64_t* out, size_t n_words, uint64_t seed, uint64_t off) {
    size_t stride = (size_t)gridDim.x * NT;
    // two words per thread-iteration -> 16 B stores
    for (size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 2; i < n_words; i += stride * 2) {
        uint64_t a = splitmix_gl(seed, off + i);
        if (i + 1 < n_words) {
            uint64_t b = splitmix_gl(seed, off + i + 1);
            *reinterpret_cast<ulonglong2*>(out + i) = make_ulonglong2(a, b);
        } else {
            out[i] = a;
        }
    }
}

// out[j] = in[2j] + r (in[2j+1] - in[2j])
template <bool IN_EXT>
__global__ void __launch_bounds__(NT) k_fold(const uint64_t* __restrict__ in, E2* __restrict__ out, size_t half, E2 r) {
    size_t stride = (size_t)gridDim.x * NT;
    const E2Pre rp = e2_pre(r);
    for (size_t j = (size_t)blockIdx.x * NT + threadIdx.x; j < half; j += stride) {
        E2 lo, hi;
        if (IN_EXT) {
            const E2* p = reinterpret_cast<const E2*>(in) + 2 * j;
            lo = p[0];
            hi = p[1];
            out[j] = lo + e2_mul_pre(rp, hi - lo);
        } else {
            ulonglong2 v = *reinterpret_cast<const ulonglong2*>(in + 2 * j);
            uint64_t d = sub(v.y, v.x);
            E2 t = e2_mul_base(r, d);
            out[j] = E2{add(t.c0, v.x), t.c1};
        }
    }
}

int launch_fold(ceno_hip_ctx* ctx, const uint64_t* in, int in_is_ext, uint64_t* out, size_t half, E2 r, hipStream_t st) {
    unsigned g = grid_for(half, NT, MAXB);
    if (in_is_ext)
        hipLaunchKernelGGL(k_fold<true>, dim3(g), dim3(NT), 0, st, in, reinterpret_cast<E2*>(out), half, r);
    else
        hipLaunchKernelGGL(k_fold<false>, dim3(g), dim3(NT), 0, st, in, reinterpret_cast<E2*>(out), half, r);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// eq(x, r) = prod_k (x_k r_k + (1-x_k)(1-r_k)), LSB-first (k_eq_fused below), optionally masked by a selector
// ------------------------------------------------------------------------------------------------
// one half table by direct products (large tables: the outer-product form below)
__global__ void __launch_bounds__(NT) k_eq_half(E2* out, int first_var, int n_vars, PointArg pt, E2 scalar) {
    size_t len = (size_t)1 << n_vars;
    size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < len; i += stride) {
        E2 acc = scalar;
        for (int k = 0; k < n_vars; k++) {
            E2 r = pt.r[first_var + k];
            E2 f = ((i >> k) & 1) ? r : (e2_one() - r);
            acc = acc * f;
        }
        out[i] = acc;
    }
}


struct SelArg {
    int kind;
    int num_vars;
    uint64_t start, end;       // PREFIX: keep [start,end);  ORDERED_SPARSE/QUARK: end = num_instances
    int sparse_num_vars;
    uint64_t sparse_mask[4];   // up to 2^8 positions per chunk
    uint32_t quark_seq[40];
};

__device__ __forceinline__ bool sel_keep(const SelArg& sa, size_t x) {
    switch (sa.kind) {
    case CENO_HIP_SEL_WHOLE: return true;
    case CENO_HIP_SEL_PREFIX: return x >= sa.start && x < sa.end;
    case CENO_HIP_SEL_ORDERED_SPARSE: {
        size_t chunk = x >> sa.sparse_num_vars;
        if (chunk >= sa.end) return false;
        unsigned pos = (unsigned)(x & (((size_t)1 << sa.sparse_num_vars) - 1));
        return (sa.sparse_mask[pos >> 6] >> (pos & 63)) & 1;
    }
    case CENO_HIP_SEL_QUARK_LT: {
        // region i = indices whose top i bits are 1 and next bit is 0 (selector.rs:217-238)
        int nv = sa.num_vars;
        int i = 0;
        while (i < nv && ((x >> (nv - 1 - i)) & 1)) i++;
        if (i >= nv) return false;  // last hypercube entry is zeroed
        size_t len = (size_t)1 << (nv - 1 - i);
        size_t pos = x & (len - 1);
        return pos < sa.quark_seq[i];
    }
    }
    return false;
}

// large tables: eq[i] = lo[i & (2^a - 1)] * hi[i >> a] from two half tables in global memory (L2-resident): one multiplication and
// one 16-byte store per entry — write-bandwidth bound (0.42 of the HBM roofline at nv = 24)
__global__ void __launch_bounds__(NT) k_eq_outer(E2* __restrict__ out, const E2* __restrict__ lo, const E2* __restrict__ hi, int a,
                                                 size_t len, SelArg sa) {
    size_t stride = (size_t)gridDim.x * NT;
    size_t mask = ((size_t)1 << a) - 1;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < len; i += stride) {
        E2 v = e2_zero();
        if (sel_keep(sa, i)) v = lo[i & mask] * hi[i >> a];
        out[i] = v;
    }
}


// SMALL tables (n <= 16: the eq tables of tower layers and small sumchecks, ~60 per chip proof) in ONE launch instead of three
// with a scratch allocation in between.  Every workgroup builds eq over the LOW min(n, 11) variables in LDS by
// the doubling construction (one multiplication per entry: new[j] = old[j] (1 - r_k), new[j + 2^k] = old[j] r_k), then serves
// tiles of 2^11 consecutive outputs: out[tile * 2^11 + j] = P(tile) * low[j], P = scalar * prod over the high variables of the
// tile index's bits.  Two multiplications and one 16-byte store per entry; the selector masks apply on the way out.
static constexpr int EQ_LB = 11;
// largest table built by the one-launch form (CENO_HIP_EQ_FUSED_MAX overrides: A/B measurements)
static int eq_fused_max() {
    static const int v = [] {
        const char* e = getenv("CENO_HIP_EQ_FUSED_MAX");
        return e ? atoi(e) : 16;
    }();
    return v;
}
__global__ void __launch_bounds__(NT) k_eq_fused(E2* __restrict__ out, int n, PointArg pt, E2 scalar, SelArg sa, SetupJob job) {
    __shared__ E2 tab[1 << EQ_LB];
    const unsigned eq_blocks = job.dst ? gridDim.x - 1 : gridDim.x;
    if (blockIdx.x == eq_blocks) {  // (only with a job) the extra workgroup does the sumcheck handle's set-up work: k_setup's three loops
        for (size_t i = threadIdx.x; i < job.zero_words; i += NT) job.zero[i] = 0;
        for (size_t i = threadIdx.x; i < job.words; i += NT) job.dst[i] = job.src_host_view[i];
        for (size_t i = threadIdx.x; i < job.ones_words; i += NT) job.ones[i] = ~0ull;  // MSG_INVALID
        return;
    }
    const int lb = n < EQ_LB ? n : EQ_LB;
    if (threadIdx.x == 0) tab[0] = e2_one();
    __syncthreads();
    for (int k = 0; k < lb; k++) {
        const int half = 1 << k;
        const E2 r = pt.r[k];
        for (int j = threadIdx.x; j < half; j += NT) {
            const E2 v = tab[j], hi = v * r;
            tab[j + half] = hi;
            tab[j] = v - hi;
        }
        __syncthreads();
    }
    const size_t tiles = (size_t)1 << (n - lb), tile_len = (size_t)1 << lb;
    for (size_t t = blockIdx.x; t < tiles; t += eq_blocks) {
        E2 p = scalar;
        for (int k = lb; k < n; k++) {  // uniform over the workgroup
            const E2 r = pt.r[k];
            p = p * (((t >> (k - lb)) & 1) ? r : (e2_one() - r));
        }
        const size_t base = t << lb;
        for (size_t j = threadIdx.x; j < tile_len; j += NT) {
            E2 v = e2_zero();
            if (sel_keep(sa, base + j)) v = p * tab[j];
            out[base + j] = v;
        }
    }
}

// `keep_tmp` != nullptr: the scratch halves of a LARGE table are returned to the caller (who frees them once the stream has passed
// this point) and the call does not synchronise; otherwise the call synchronises and frees them.  Small tables need no scratch.
static int eq_build_impl(ceno_hip_ctx* ctx, const uint64_t* point, int n, E2 scalar, const SelArg& sa, uint64_t* dev_out, hipStream_t st,
                         void** keep_tmp = nullptr) {
    CHECK_ARG(ctx, n >= 0 && n <= 40, "eq: num_vars %d out of range", n);
    PointArg pt;
    for (int k = 0; k < n; k++) pt.r[k] = E2{point[2 * k], point[2 * k + 1]};
    if (keep_tmp) *keep_tmp = nullptr;
    if (n <= eq_fused_max()) {
        // latency-bound sizes: per block the LDS table costs as much as a tile's worth of products, which only pays when the
        // alternative is two more launches (at nv = 24 the fused form measured 0.143 ms against 0.080 ms for the outer product)
        const size_t tiles = (size_t)1 << (n > EQ_LB ? n - EQ_LB : 0);
        hipLaunchKernelGGL(k_eq_fused, dim3((unsigned)std::min<size_t>(tiles, 1024)), dim3(NT), 0, st, (E2*)dev_out, n, pt, scalar, sa, SetupJob{});
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    int a = (n + 1) / 2, b = n - a;
    void* tmp = nullptr;
    TRY(ctx_alloc(ctx, (((size_t)1 << a) + ((size_t)1 << b)) * sizeof(E2), &tmp));
    E2* lo = (E2*)tmp;
    E2* hi = lo + ((size_t)1 << a);
    hipLaunchKernelGGL(k_eq_half, dim3(grid_for((size_t)1 << a, NT, MAXB)), dim3(NT), 0, st, lo, 0, a, pt, e2_one());
    hipLaunchKernelGGL(k_eq_half, dim3(grid_for((size_t)1 << b, NT, MAXB)), dim3(NT), 0, st, hi, a, b, pt, scalar);
    size_t len = (size_t)1 << n;
    hipLaunchKernelGGL(k_eq_outer, dim3(grid_for(len, NT, MAXB)), dim3(NT), 0, st, (E2*)dev_out, lo, hi, a, len, sa);
    hipError_t e = hipGetLastError();
    if (keep_tmp && e == hipSuccess) {
        *keep_tmp = tmp;
        return 0;
    }
    // the scratch halves are read by the queued kernel: return them to the pool only after it ran
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    ctx_free(ctx, tmp);
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "eq build: %s", hipGetErrorString(e));
    return 0;
}

int launch_eq_build_with_setup(ceno_hip_ctx* ctx, const uint64_t* host_point, int n, E2 scalar, uint64_t* dev_out, hipStream_t st, const SetupJob& job) {
    if (n < 0 || n > eq_fused_max() || !job.dst) return 1;
    SelArg sa{};
    sa.kind = CENO_HIP_SEL_WHOLE;
    sa.num_vars = n;
    PointArg pt;
    for (int k = 0; k < n; k++) pt.r[k] = E2{host_point[2 * k], host_point[2 * k + 1]};
    const size_t tiles = (size_t)1 << (n > EQ_LB ? n - EQ_LB : 0);
    hipLaunchKernelGGL(k_eq_fused, dim3((unsigned)std::min<size_t>(tiles, 1024) + 1), dim3(NT), 0, st, (E2*)dev_out, n, pt, scalar, sa, job);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
int launch_eq_build(ceno_hip_ctx* ctx, const uint64_t* host_point, int n, E2 scalar, uint64_t* dev_out, hipStream_t st, void** keep_tmp) {
    SelArg sa{};
    sa.kind = CENO_HIP_SEL_WHOLE;
    sa.num_vars = n;
    return eq_build_impl(ctx, host_point, n, scalar, sa, dev_out, st, keep_tmp);
}

// ------------------------------------------------------------------------------------------------
// evaluate: sum_i f[i] * lo[i & mask] * hi[i >> a]   (one read-only pass over the table)
// ------------------------------------------------------------------------------------------------
// both half tables of an evaluation in ONE launch (lo over the first a variables, hi over the remaining b); also clears the
// arrival counter of the reduction that follows on the same stream
__global__ void __launch_bounds__(NT) k_eq_halves(E2* __restrict__ lo, int a, E2* __restrict__ hi, int b, PointArg pt, unsigned* __restrict__ counter) {
    const size_t na = (size_t)1 << a, total = na + ((size_t)1 << b);
    const size_t stride = (size_t)gridDim.x * NT;
    if (blockIdx.x == 0 && threadIdx.x == 0) *counter = 0;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += stride) {
        const bool second = i >= na;
        const size_t x = second ? i - na : i;
        const int first_var = second ? a : 0, n_vars = second ? b : a;
        E2 acc = e2_one();
        for (int k = 0; k < n_vars; k++) {
            const E2 r = pt.r[first_var + k];
            acc = acc * (((x >> k) & 1) ? r : (e2_one() - r));
        }
        (second ? hi : lo)[x] = acc;
    }
}
// sum f[i] lo[i & mask] hi[i >> a] in one read-only pass; the last workgroup to arrive adds the per-workgroup partials and
// writes the evaluation straight into pinned host words the caller watches (no second launch, no copy, no stream wait)
template <bool IN_EXT>
__global__ void __launch_bounds__(NT) k_eval_dot_host(const uint64_t* __restrict__ f, const E2* __restrict__ lo, const E2* __restrict__ hi, int a, size_t len,
                                                      uint64_t* __restrict__ partials, unsigned* __restrict__ counter, E2* __restrict__ out_host) {
    __shared__ E2 smem[NT / 64];
    __shared__ int s_last;
    const size_t stride = (size_t)gridDim.x * NT;
    const size_t mask = ((size_t)1 << a) - 1;
    E2 acc[1] = {e2_zero()};
    if (a >= 8) {
        // Row form: a workgroup owns whole rows of 2^a consecutive entries (one hi index each).  Inside a row every entry costs ONE
        // unreduced multiply-accumulate with lo (four wide products into 160-bit accumulators; two for a base-field table), the
        // row's sum is reduced once and multiplied by hi[row] once per lane — the element-wise form below spends two full
        // extension multiplications per entry and was VALU-bound at a quarter of the HBM roofline.
        const size_t row_len = (size_t)1 << a, rows = len >> a;
        for (size_t h = blockIdx.x; h < rows; h += gridDim.x) {
            E2 row;
            if (IN_EXT) {
                E2Acc w = e2acc_zero();
                const E2* fr = reinterpret_cast<const E2*>(f) + (h << a);
                for (size_t j = threadIdx.x; j < row_len; j += NT) e2acc_mac(w, fr[j], lo[j]);
                row = e2acc_reduce(w);
            } else {
                Acc5 w0{0, 0, 0, 0, 0}, w1{0, 0, 0, 0, 0};
                const uint64_t* fr = f + (h << a);
                for (size_t j = threadIdx.x; j < row_len; j += NT) {
                    const E2 l = lo[j];
                    const uint64_t v = fr[j];
                    acc5_add(w0, mul_wide(l.c0, v));
                    acc5_add(w1, mul_wide(l.c1, v));
                }
                row = E2{acc5_reduce(w0), acc5_reduce(w1)};
            }
            acc[0] = acc[0] + row * hi[h];
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < len; i += stride) {
            const E2 w = lo[i & mask] * hi[i >> a];
            if (IN_EXT) acc[0] = acc[0] + reinterpret_cast<const E2*>(f)[i] * w;
            else acc[0] = acc[0] + e2_mul_base(w, f[i]);
        }
    }
    red::block_sum<1, NT>(acc, smem);
    if (threadIdx.x == 0) {
        // write-through partial, drained before the agent-scope arrival count (per-XCD L2s are not coherent)
        __hip_atomic_store(partials + 2 * blockIdx.x, acc[0].c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(partials + 2 * blockIdx.x + 1, acc[0].c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    E2 tot[1] = {e2_zero()};
    for (unsigned bb = threadIdx.x; bb < gridDim.x; bb += NT)
        tot[0] = tot[0] + E2{__hip_atomic_load(partials + 2 * bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             __hip_atomic_load(partials + 2 * bb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
    __syncthreads();
    red::block_sum<1, NT>(tot, smem);
    if (threadIdx.x == 0) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 w = {(unsigned)tot[0].c0, (unsigned)(tot[0].c0 >> 32), (unsigned)tot[0].c1, (unsigned)(tot[0].c1 >> 32)};
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(out_host), "v"(w) : "memory");
    }
}

// ------------------------------------------------------------------------------------------------
// rotation (keccak-style chips): cyclic sequence x^i in GF(2)[X]/(X^5+X^2+1) resp. /(X^6+X+1)
// (gkr_iop/src/gkr/booleanhypercube.rs:10-113)
// ------------------------------------------------------------------------------------------------
struct RotArg {
    uint8_t next[64];  // next[x^i] = x^(i+1), next[0] = 0
};
static int cyclic_table(int log2, uint32_t* out) {
    uint32_t modulus = log2 == 5 ? 0x25u : log2 == 6 ? 0x43u : 0u;
    if (!modulus) return -1;
    uint32_t cur = 1;
    for (int i = 0; i < (1 << log2); i++) {
        out[i] = cur;
        cur <<= 1;
        if (cur & (1u << log2)) cur ^= modulus;
    }
    return 0;
}
__global__ void __launch_bounds__(NT) k_rotate_base(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, size_t len, int log2, RotArg ra) {
    const size_t stride = (size_t)gridDim.x * NT;
    const size_t mask = ((size_t)1 << log2) - 1;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < len; i += stride) out[i] = in[(i & ~mask) | ra.next[i & mask]];
}

__global__ void __launch_bounds__(NT) k_take_stride2(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, size_t n, int elem_words, int odd) {
    const size_t total = n * elem_words, stride = (size_t)gridDim.x * NT;
    for (size_t t = (size_t)blockIdx.x * NT + threadIdx.x; t < total; t += stride) {
        const size_t i = t / elem_words, w = t % elem_words;
        out[t] = in[(2 * i + odd) * elem_words + w];
    }
}

extern "C" {

int ceno_hip_rotation_next_base_mle(ceno_hip_ctx* ctx, const ceno_hip_mle* in, int cyclic_group_log2, ceno_hip_stream s, ceno_hip_mle** out) {
    CHECK_ARG(ctx, in && out, "NULL argument");
    CHECK_ARG(ctx, !in->is_ext, "rotation source must be a base-field table (layer/gpu/utils.rs:250-254)");
    CHECK_ARG(ctx, in->num_vars >= cyclic_group_log2, "table smaller than one cyclic group");
    (void)ctx_stream(ctx, s);  // allocations below belong to work on `s`: bind the thread first (pool tags, include/ceno_hip.h "Memory")
    uint32_t r[64];
    CHECK_ARG(ctx, cyclic_table(cyclic_group_log2, r) == 0, "cyclic group log2 must be 5 or 6");
    RotArg ra{};
    const int g = 1 << cyclic_group_log2;
    for (int i = 0; i < g; i++) ra.next[i] = (uint8_t)i;   // positions outside the cycle (only 0) keep their value
    for (int i = 0; i + 1 < g; i++) ra.next[r[i]] = (uint8_t)r[i + 1];  // utils.rs:45-49
    ceno_hip_mle* m = nullptr;
    TRY(ceno_hip_mle_alloc(ctx, in->num_vars, 0, &m));
    hipLaunchKernelGGL(k_rotate_base, dim3(grid_for(in->len(), NT, MAXB)), dim3(NT), 0, ctx_stream(ctx, s), in->d, m->d, in->len(), cyclic_group_log2, ra);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ceno_hip_mle_free(ctx, m);
        return ctx_fail(ctx, CENO_HIP_ERR_HIP, "rotation: %s", hipGetErrorString(e));
    }
    *out = m;
    return 0;
}

int ceno_hip_rotation_selector_build(ceno_hip_ctx* ctx, const uint64_t* point, int num_vars, int cyclic_subgroup_size, int cyclic_group_log2,
                                     ceno_hip_stream s, ceno_hip_mle** out) {
    CHECK_ARG(ctx, out && point, "NULL argument");
    (void)ctx_stream(ctx, s);  // allocations below belong to work on `s`: bind the thread first (pool tags, include/ceno_hip.h "Memory")
    uint32_t r[64];
    CHECK_ARG(ctx, cyclic_table(cyclic_group_log2, r) == 0, "cyclic group log2 must be 5 or 6");
    CHECK_ARG(ctx, cyclic_subgroup_size >= 0 && cyclic_subgroup_size <= (1 << cyclic_group_log2), "cyclic subgroup larger than the group");
    CHECK_ARG(ctx, num_vars >= cyclic_group_log2 && num_vars <= 40, "num_vars out of range");
    SelArg sa{};
    sa.kind = CENO_HIP_SEL_ORDERED_SPARSE;
    sa.num_vars = num_vars;
    sa.sparse_num_vars = cyclic_group_log2;
    sa.end = (uint64_t)1 << (num_vars - cyclic_group_log2);  // every chunk is active (utils.rs:66-74)
    for (int i = 0; i < cyclic_subgroup_size; i++) sa.sparse_mask[r[i] >> 6] |= (uint64_t)1 << (r[i] & 63);
    ceno_hip_mle* m = nullptr;
    TRY(ceno_hip_mle_alloc(ctx, num_vars, 1, &m));
    int rc = eq_build_impl(ctx, point, num_vars, e2_one(), sa, m->d, ctx_stream(ctx, s));
    if (rc) {
        ceno_hip_mle_free(ctx, m);
        return rc;
    }
    *out = m;
    return 0;
}

int ceno_hip_mle_fill_splitmix(ceno_hip_ctx* ctx, ceno_hip_mle* m, uint64_t seed, uint64_t word_offset, ceno_hip_stream s) {
    CHECK_ARG(ctx, m, "NULL mle");
    size_t n_words = m->len() * (m->is_ext ? 2 : 1);
    hipStream_t st = ctx_stream(ctx, s);
    hipLaunchKernelGGL(k_fill_splitmix, dim3(grid_for((n_words + 1) / 2, NT, MAXB)), dim3(NT), 0, st, m->d, n_words, seed, word_offset);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int ceno_hip_mle_fill_zero(ceno_hip_ctx* ctx, ceno_hip_mle* m, ceno_hip_stream s) {
    CHECK_ARG(ctx, m, "NULL mle");
    HIP_TRY(ctx, hipMemsetAsync(m->d, 0, m->len() * (m->is_ext ? 16 : 8), ctx_stream(ctx, s)));
    return 0;
}

int ceno_hip_eq_build(ceno_hip_ctx* ctx, const uint64_t* point, int num_vars, const uint64_t* scalar2, ceno_hip_stream s, ceno_hip_mle** out) {
    CHECK_ARG(ctx, out && (point || num_vars == 0), "NULL argument");
    (void)ctx_stream(ctx, s);  // allocations below belong to work on `s`: bind the thread first (pool tags, include/ceno_hip.h "Memory")
    ceno_hip_mle* m = nullptr;
    TRY(ceno_hip_mle_alloc(ctx, num_vars, 1, &m));
    E2 sc = scalar2 ? E2{scalar2[0], scalar2[1]} : e2_one();
    int rc = launch_eq_build(ctx, point, num_vars, sc, m->d, ctx_stream(ctx, s), &m->aux);  // half tables live with the handle: no sync
    if (rc) {
        ceno_hip_mle_free(ctx, m);
        return rc;
    }
    *out = m;
    return 0;
}

int ceno_hip_selector_build(ceno_hip_ctx* ctx, int kind, const uint64_t* point, int num_vars, size_t offset, size_t num_instances,
                            const uint32_t* sparse_indices, int n_sparse, int sparse_num_vars, ceno_hip_stream s, ceno_hip_mle** out) {
    CHECK_ARG(ctx, out && (point || num_vars == 0), "NULL argument");
    CHECK_ARG(ctx, num_vars >= 0 && num_vars <= 40, "num_vars out of range");
    (void)ctx_stream(ctx, s);  // allocations below belong to work on `s`: bind the thread first (pool tags, include/ceno_hip.h "Memory")
    SelArg sa{};
    sa.kind = kind;
    sa.num_vars = num_vars;
    size_t len = (size_t)1 << num_vars;
    switch (kind) {
    case CENO_HIP_SEL_WHOLE: break;
    case CENO_HIP_SEL_PREFIX:
        // selector.rs:144-150: end <= 2^num_vars
        CHECK_ARG(ctx, offset + num_instances <= len, "prefix selector: offset %zu + num_instances %zu > 2^%d", offset, num_instances, num_vars);
        sa.start = offset;
        sa.end = offset + num_instances;
        break;
    case CENO_HIP_SEL_ORDERED_SPARSE: {
        CHECK_ARG(ctx, sparse_num_vars >= 0 && sparse_num_vars <= 8 && sparse_num_vars <= num_vars, "sparse_num_vars %d unsupported", sparse_num_vars);
        CHECK_ARG(ctx, offset == 0, "ordered-sparse selector requires offset 0 (layer/gpu/utils.rs:145)");
        sa.end = num_instances;
        sa.sparse_num_vars = sparse_num_vars;
        // indices are assumed ascending (selector.rs:50-52): anything out of order is skipped by the reference's merge walk
        uint32_t prev = 0;
        bool first = true;
        for (int i = 0; i < n_sparse; i++) {
            uint32_t ix = sparse_indices[i];
            if (ix >= (1u << sparse_num_vars)) break;
            if (!first && ix <= prev) break;
            sa.sparse_mask[ix >> 6] |= (uint64_t)1 << (ix & 63);
            prev = ix;
            first = false;
        }
        break;
    }
    case CENO_HIP_SEL_QUARK_LT: {
        CHECK_ARG(ctx, offset == 0, "quark selector requires offset 0 (selector.rs:192)");
        size_t n_inst = num_instances;
        for (int i = 0; i < num_vars; i++) {
            sa.quark_seq[i] = (uint32_t)(n_inst / 2);
            n_inst = (n_inst + 1) / 2;
        }
        break;
    }
    default: return ctx_fail(ctx, CENO_HIP_ERR_INVALID, "unknown selector kind %d", kind);
    }
    ceno_hip_mle* m = nullptr;
    TRY(ceno_hip_mle_alloc(ctx, num_vars, 1, &m));
    // no synchronisation: the half tables stay attached to the handle (a stream sync per selector was 0.7 ms for the 24
    // chips of a batched main sumcheck)
    int rc = eq_build_impl(ctx, point, num_vars, e2_one(), sa, m->d, ctx_stream(ctx, s), &m->aux);
    if (rc) {
        ceno_hip_mle_free(ctx, m);
        return rc;
    }
    *out = m;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// all Whole / Prefix selector tables of a batch (the chips of a main-constraint sumcheck) in TWO launches: every half table, then every
// outer product (a tile list over all tables).  One selector at a time was three launches with 640-byte kernel arguments and two
// allocations each: ~45 us of host work per chip, 0.57 ms in front of a 24-chip batch whose tables take 0.13 ms to write.
// ------------------------------------------------------------------------------------------------
struct SelDesc {
    E2* out;
    const E2* lo;          // half table over the variables [0, a)
    const E2* hi;          // half table over the variables [a, n)
    unsigned long long start, end;  // rows kept
    unsigned tile_begin;   // first tile (of 2^SEL_TILE_LOG entries) of this table in the launch's tile list
    int a, n;
    unsigned half_begin;   // first entry of this selector in the flat list of half-table entries
    unsigned pad;
    E2 r[40];
};
static constexpr int SEL_TILE_LOG = 14;

__global__ void __launch_bounds__(NT) k_sel_halves_batch(const SelDesc* __restrict__ desc, int n_sel, unsigned total) {
    for (unsigned i = blockIdx.x * NT + threadIdx.x; i < total; i += gridDim.x * NT) {
        int k = 0;
        while (k + 1 < n_sel && desc[k + 1].half_begin <= i) k++;
        const SelDesc& D = desc[k];
        unsigned x = i - D.half_begin;
        const unsigned na = 1u << D.a;
        const bool second = x >= na;
        if (second) x -= na;
        const int first_var = second ? D.a : 0, n_vars = second ? D.n - D.a : D.a;
        E2 acc = e2_one();
        for (int v = 0; v < n_vars; v++) {
            const E2 r = D.r[first_var + v];
            acc = acc * (((x >> v) & 1) ? r : (e2_one() - r));
        }
        (second ? const_cast<E2*>(D.hi) : const_cast<E2*>(D.lo))[x] = acc;
    }
}
__global__ void __launch_bounds__(NT) k_sel_outer_batch(const SelDesc* __restrict__ desc, int n_sel, unsigned total_tiles) {
    int k = 0;
    for (unsigned t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        while (k + 1 < n_sel && desc[k + 1].tile_begin <= t) k++;  // tiles are visited in increasing order
        const SelDesc& D = desc[k];
        const size_t len = (size_t)1 << D.n, base = (size_t)(t - D.tile_begin) << SEL_TILE_LOG;
        const size_t mask = ((size_t)1 << D.a) - 1;
        for (size_t j = threadIdx.x; j < ((size_t)1 << SEL_TILE_LOG) && base + j < len; j += NT) {
            const size_t i = base + j;
            E2 v = e2_zero();
            if (i >= D.start && i < D.end) v = D.lo[i & mask] * D.hi[i >> D.a];
            D.out[i] = v;
        }
    }
}

extern "C" int ceno_hip_selector_build_batch(ceno_hip_ctx* ctx, int n, const int* kinds, const uint64_t* const* points, const int* num_vars,
                                             const size_t* offsets, const size_t* num_instances, ceno_hip_stream s, ceno_hip_mle** outs) {
    CHECK_ARG(ctx, n >= 0 && (n == 0 || (kinds && points && num_vars && offsets && num_instances && outs)), "selector batch: NULL argument");
    if (n == 0) return 0;
    hipStream_t st = ctx_stream(ctx, s);
    std::vector<SelDesc> desc((size_t)n);
    size_t half_total = 0, tiles = 0;
    for (int k = 0; k < n; k++) {
        outs[k] = nullptr;
        CHECK_ARG(ctx, kinds[k] == CENO_HIP_SEL_WHOLE || kinds[k] == CENO_HIP_SEL_PREFIX, "selector batch: kind %d is built one at a time (ceno_hip_selector_build)", kinds[k]);
        CHECK_ARG(ctx, num_vars[k] >= 0 && num_vars[k] < 40 && (points[k] || num_vars[k] == 0), "selector batch: bad table %d", k);
        const size_t len = (size_t)1 << num_vars[k];
        if (kinds[k] == CENO_HIP_SEL_PREFIX) CHECK_ARG(ctx, offsets[k] + num_instances[k] <= len, "prefix selector: offset %zu + num_instances %zu > 2^%d", offsets[k], num_instances[k], num_vars[k]);
        SelDesc& D = desc[(size_t)k];
        D.n = num_vars[k];
        D.a = (D.n + 1) / 2;
        D.start = kinds[k] == CENO_HIP_SEL_WHOLE ? 0 : offsets[k];
        D.end = kinds[k] == CENO_HIP_SEL_WHOLE ? len : offsets[k] + num_instances[k];
        D.half_begin = (unsigned)half_total;
        D.tile_begin = (unsigned)tiles;
        for (int v = 0; v < D.n; v++) D.r[v] = E2{points[k][2 * v], points[k][2 * v + 1]};
        half_total += ((size_t)1 << D.a) + ((size_t)1 << (D.n - D.a));
        tiles += (len + (((size_t)1 << SEL_TILE_LOG) - 1)) >> SEL_TILE_LOG;
    }
    CHECK_ARG(ctx, half_total < ((size_t)1 << 31) && tiles < ((size_t)1 << 31), "selector batch too large");
    int rc = 0;
    for (int k = 0; k < n && !rc; k++) rc = ceno_hip_mle_alloc(ctx, num_vars[k], 1, &outs[k]);
    void* scratch = nullptr;
    void *hb = nullptr, *db = nullptr;
    if (!rc) rc = ctx_alloc(ctx, half_total * sizeof(E2) + (size_t)n * sizeof(SelDesc), &scratch);
    if (!rc) rc = ctx_pinned_alloc(ctx, (size_t)n * sizeof(SelDesc), &hb, &db);
    if (rc) {
        for (int k = 0; k < n; k++)
            if (outs[k]) { ceno_hip_mle_free(ctx, outs[k]); outs[k] = nullptr; }
        if (scratch) ctx_free(ctx, scratch);
        return rc;
    }
    E2* halves = (E2*)scratch;
    SelDesc* d_desc = reinterpret_cast<SelDesc*>(halves + half_total);
    for (int k = 0; k < n; k++) {
        SelDesc& D = desc[(size_t)k];
        D.out = (E2*)outs[k]->d;
        D.lo = halves + D.half_begin;
        D.hi = D.lo + ((size_t)1 << D.a);
    }
    memcpy(hb, desc.data(), (size_t)n * sizeof(SelDesc));
    // the descriptors travel by ONE copy from pinned memory; the wait covers that copy only (the two kernels are queued after it and
    // run while the caller goes on — building the sumcheck handle, for the main constraints)
    hipError_t e = hipMemcpyAsync(d_desc, hb, (size_t)n * sizeof(SelDesc), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    ctx_pinned_free(ctx, hb);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sel_halves_batch, dim3(grid_for(half_total, NT, MAXB)), dim3(NT), 0, st, d_desc, n, (unsigned)half_total);
        hipLaunchKernelGGL(k_sel_outer_batch, dim3((unsigned)std::min<size_t>(tiles, 8192)), dim3(NT), 0, st, d_desc, n, (unsigned)tiles);
        e = hipGetLastError();
    }
    // the scratch (half tables + descriptors) is read by the queued kernels: the pool hands a block tagged with this stream to the same
    // stream in stream order and to another stream only once this one has drained
    ctx_free_on(ctx, scratch, st);
    if (e != hipSuccess) {
        for (int k = 0; k < n; k++) { ceno_hip_mle_free(ctx, outs[k]); outs[k] = nullptr; }
        return ctx_fail(ctx, CENO_HIP_ERR_HIP, "selector batch: %s", hipGetErrorString(e));
    }
    return 0;
}

int ceno_hip_mle_evaluate(ceno_hip_ctx* ctx, const ceno_hip_mle* m, const uint64_t* point, uint64_t* out2, ceno_hip_stream s) {
    CHECK_ARG(ctx, m && out2 && (point || m->num_vars == 0), "NULL argument");
    hipStream_t st = ctx_stream(ctx, s);
    int n = m->num_vars;
    PointArg pt;
    for (int k = 0; k < n; k++) pt.r[k] = E2{point[2 * k], point[2 * k + 1]};
    int a = (n + 1) / 2, b = n - a;
    size_t len = m->len();
    unsigned g = grid_for(len, NT, MAXB);
    // two launches and no copy: half tables (+ counter reset), then the dot product whose last workgroup writes the value
    // into pinned words armed with a non-canonical pattern — the caller's thread watches them instead of waiting for the stream
    void* tmp = nullptr;
    TRY(ctx_alloc(ctx, (((size_t)1 << a) + ((size_t)1 << b) + g + 1) * sizeof(E2), &tmp));
    E2* lo = (E2*)tmp;
    E2* hi = lo + ((size_t)1 << a);
    uint64_t* partials = reinterpret_cast<uint64_t*>(hi + ((size_t)1 << b));
    unsigned* counter = reinterpret_cast<unsigned*>(partials + 2 * (size_t)g);
    void *h_res = nullptr, *d_res = nullptr;
    int rc = ctx_pinned_alloc(ctx, 64, &h_res, &d_res);
    if (rc) {
        ctx_free(ctx, tmp);
        return rc;
    }
    volatile uint64_t* hw = (volatile uint64_t*)h_res;
    hw[0] = hw[1] = ~0ull;
    const size_t total = ((size_t)1 << a) + ((size_t)1 << b);
    hipLaunchKernelGGL(k_eq_halves, dim3(grid_for(total, NT, MAXB)), dim3(NT), 0, st, lo, a, hi, b, pt, counter);
    if (m->is_ext)
        hipLaunchKernelGGL(k_eval_dot_host<true>, dim3(g), dim3(NT), 0, st, m->d, lo, hi, a, len, partials, counter, (E2*)d_res);
    else
        hipLaunchKernelGGL(k_eval_dot_host<false>, dim3(g), dim3(NT), 0, st, m->d, lo, hi, a, len, partials, counter, (E2*)d_res);
    hipError_t e = hipGetLastError();
    unsigned long long spins = 0;
    while (e == hipSuccess && (__atomic_load_n(&hw[0], __ATOMIC_ACQUIRE) == ~0ull || __atomic_load_n(&hw[1], __ATOMIC_ACQUIRE) == ~0ull)) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        __builtin_ia32_pause();
#endif
        if ((++spins & 0xFFFFF) == 0) {  // a faulted kernel never writes: make sure the stream is still alive
            const hipError_t q = hipStreamQuery(st);
            if (q != hipSuccess && q != hipErrorNotReady) e = q;
            else if (q == hipSuccess && (hw[0] == ~0ull || hw[1] == ~0ull)) e = hipErrorUnknown;
        }
    }
    out2[0] = hw[0];
    out2[1] = hw[1];
    ctx_pinned_free(ctx, h_res);
    ctx_free(ctx, tmp);  // tagged with this stream; the last kernel has written its result, nothing of it is still queued
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "evaluate: %s", hipGetErrorString(e));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Columns that a constraint system only reads LINEARLY (the record RLCs of a chip: selector x column for three quarters of its columns,
// gkr_iop/src/gkr/layer/zerocheck_layer.rs:118-140) need not go through the rounds of the main-constraint sumcheck one by one:
//     sum_j c_j sel(x) col_j(x) = sel(x) (A(x) + X B(x)),   A = sum_j c_j.c0 col_j,  B = sum_j c_j.c1 col_j   (two BASE-field tables)
// and their evaluations at the sumcheck's point follow from one read-only pass afterwards (host/main_constraints.cpp).  Two kernels:
// the combinations of many groups in one launch, and the evaluations of many base-field tables at prefixes of one point in one launch.
// Both stream every column once: HBM-bound (18 VALU instructions per 8 bytes).
// ------------------------------------------------------------------------------------------------
struct LinGroup {
    const uint64_t* const* cols;  // device array of the group's column tables
    const uint64_t* coeffs;       // two words per column (device)
    uint64_t *out0, *out1;
    uint32_t n_cols, log_rows, wg_begin, pad;
};
constexpr int LIN_L = 2;                      // 16-byte loads per lane and column
constexpr unsigned LIN_TILE = 2 * LIN_L * NT;  // rows per workgroup: load l of lane t = rows 2 NT l + 2 t, + 1
__global__ void __launch_bounds__(NT) k_lincomb_base(const LinGroup* __restrict__ groups, int n_groups) {
    int g = 0;
    {   // the group of this workgroup (groups are sorted by wg_begin): binary search, wave-uniform
        int lo = 0, hi = n_groups - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (groups[mid].wg_begin <= blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        g = lo;
    }
    const LinGroup G = groups[g];
    const size_t rows = (size_t)1 << G.log_rows;
    const size_t r0 = (size_t)(blockIdx.x - G.wg_begin) * LIN_TILE + 2 * threadIdx.x;
    if (r0 >= rows) return;
    if (rows == 1) {
        Acc5 a0{0, 0, 0, 0, 0}, a1{0, 0, 0, 0, 0};
        for (uint32_t j = 0; j < G.n_cols; j++) {
            const uint64_t v = G.cols[j][0];
            acc5_add(a0, mul_wide(G.coeffs[2 * j], v));
            acc5_add(a1, mul_wide(G.coeffs[2 * j + 1], v));
        }
        G.out0[0] = acc5_reduce(a0);
        G.out1[0] = acc5_reduce(a1);
        return;
    }
    Acc5 a[LIN_L][4];  // per load: (c0, row 0), (c1, row 0), (c0, row 1), (c1, row 1)
    bool on[LIN_L];
#pragma unroll
    for (int l = 0; l < LIN_L; l++) {
        on[l] = r0 + (size_t)l * 2 * NT < rows;  // (a table of fewer rows than a tile)
#pragma unroll
        for (int k = 0; k < 4; k++) a[l][k] = Acc5{0, 0, 0, 0, 0};
    }
    auto mac = [&](int l, uint64_t c0, uint64_t c1, const ulonglong2& v) {
        acc5_add(a[l][0], mul_wide(c0, v.x));
        acc5_add(a[l][1], mul_wide(c1, v.x));
        acc5_add(a[l][2], mul_wide(c0, v.y));
        acc5_add(a[l][3], mul_wide(c1, v.y));
    };
    uint32_t j = 0;
    for (; j + 4 <= G.n_cols; j += 4) {  // four columns in flight
        ulonglong2 v[4][LIN_L];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int l = 0; l < LIN_L; l++) v[u][l] = on[l] ? *reinterpret_cast<const ulonglong2*>(G.cols[j + u] + r0 + (size_t)l * 2 * NT) : ulonglong2{0, 0};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t c0 = G.coeffs[2 * (j + u)], c1 = G.coeffs[2 * (j + u) + 1];
#pragma unroll
            for (int l = 0; l < LIN_L; l++) mac(l, c0, c1, v[u][l]);
        }
    }
    for (; j < G.n_cols; j++) {
        const uint64_t c0 = G.coeffs[2 * j], c1 = G.coeffs[2 * j + 1];
#pragma unroll
        for (int l = 0; l < LIN_L; l++)
            if (on[l]) mac(l, c0, c1, *reinterpret_cast<const ulonglong2*>(G.cols[j] + r0 + (size_t)l * 2 * NT));
    }
#pragma unroll
    for (int l = 0; l < LIN_L; l++) {
        if (!on[l]) continue;
        *reinterpret_cast<ulonglong2*>(G.out0 + r0 + (size_t)l * 2 * NT) = ulonglong2{acc5_reduce(a[l][0]), acc5_reduce(a[l][2])};
        *reinterpret_cast<ulonglong2*>(G.out1 + r0 + (size_t)l * 2 * NT) = ulonglong2{acc5_reduce(a[l][1]), acc5_reduce(a[l][3])};
    }
}

extern "C" int ceno_hip_lincomb_base_batch(ceno_hip_ctx* ctx, int n_groups, const uint32_t* group_offsets, ceno_hip_mle* const* cols, const uint64_t* coeffs,
                                           ceno_hip_stream s, ceno_hip_mle** out0, ceno_hip_mle** out1) {
    CHECK_ARG(ctx, n_groups >= 0 && (n_groups == 0 || (group_offsets && cols && coeffs && out0 && out1)), "lincomb batch: NULL argument");
    if (n_groups == 0) return 0;
    hipStream_t st = ctx_stream(ctx, s);
    const size_t n_cols = group_offsets[n_groups];
    std::vector<LinGroup> desc((size_t)n_groups);
    size_t wgs = 0;
    for (int g = 0; g < n_groups; g++) {
        out0[g] = out1[g] = nullptr;
        const uint32_t b = group_offsets[g], e = group_offsets[g + 1];
        CHECK_ARG(ctx, e > b, "lincomb batch: group %d is empty", g);
        const int nv = cols[b] ? cols[b]->num_vars : -1;
        for (uint32_t j = b; j < e; j++)
            CHECK_ARG(ctx, cols[j] && !cols[j]->is_ext && cols[j]->num_vars == nv, "lincomb batch: the columns of group %d are base-field tables of one size", g);
        desc[(size_t)g].n_cols = e - b;
        desc[(size_t)g].log_rows = (uint32_t)nv;
        desc[(size_t)g].wg_begin = (uint32_t)wgs;
        wgs += (((size_t)1 << nv) + LIN_TILE - 1) / LIN_TILE;
    }
    CHECK_ARG(ctx, wgs < ((size_t)1 << 31), "lincomb batch too large");
    int rc = 0;
    for (int g = 0; g < n_groups && !rc; g++) {
        rc = ceno_hip_mle_alloc(ctx, (int)desc[(size_t)g].log_rows, 0, &out0[g]);
        if (!rc) rc = ceno_hip_mle_alloc(ctx, (int)desc[(size_t)g].log_rows, 0, &out1[g]);
    }
    const size_t bytes = (size_t)n_groups * sizeof(LinGroup) + n_cols * (sizeof(uint64_t*) + 2 * sizeof(uint64_t));
    void *scratch = nullptr, *hb = nullptr, *db = nullptr;
    if (!rc) rc = ctx_alloc(ctx, bytes, &scratch);
    if (!rc) rc = ctx_pinned_alloc(ctx, bytes, &hb, &db);
    auto fail = [&](int code) {
        for (int g = 0; g < n_groups; g++) {
            if (out0[g]) { ceno_hip_mle_free(ctx, out0[g]); out0[g] = nullptr; }
            if (out1[g]) { ceno_hip_mle_free(ctx, out1[g]); out1[g] = nullptr; }
        }
        if (scratch) ctx_free_on(ctx, scratch, st);
        if (hb) ctx_pinned_free(ctx, hb);
        return code;
    };
    if (rc) return fail(rc);
    // one block: descriptors | column pointers | coefficients
    LinGroup* d_desc = (LinGroup*)scratch;
    const uint64_t** d_cols = reinterpret_cast<const uint64_t**>(d_desc + n_groups);
    uint64_t* d_coeffs = reinterpret_cast<uint64_t*>(d_cols + n_cols);
    char* h = (char*)hb;
    const uint64_t** h_cols = reinterpret_cast<const uint64_t**>(h + (size_t)n_groups * sizeof(LinGroup));
    for (size_t j = 0; j < n_cols; j++) h_cols[j] = cols[j]->d;
    memcpy(h_cols + n_cols, coeffs, n_cols * 2 * sizeof(uint64_t));
    for (int g = 0; g < n_groups; g++) {
        LinGroup& G = desc[(size_t)g];
        G.cols = d_cols + group_offsets[g];
        G.coeffs = d_coeffs + 2 * (size_t)group_offsets[g];
        G.out0 = out0[g]->d;
        G.out1 = out1[g]->d;
    }
    memcpy(h, desc.data(), (size_t)n_groups * sizeof(LinGroup));
    hipError_t e = hipMemcpyAsync(scratch, hb, bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // (covers the copy only: the staging block goes back to the pool)
    ctx_pinned_free(ctx, hb);
    hb = nullptr;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_lincomb_base, dim3((unsigned)wgs), dim3(NT), 0, st, d_desc, n_groups);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        fail(0);
        return ctx_fail(ctx, CENO_HIP_ERR_HIP, "lincomb batch: %s", hipGetErrorString(e));
    }
    ctx_free_on(ctx, scratch, st);  // read by the queued kernel: stream-ordered reuse
    return 0;
}

struct EvalCol {
    const uint64_t* col;
    const E2 *lo, *hi;  // eq over the first a variables / over the rest (of THIS table's prefix of the point)
    uint32_t a, n_hi, wg_begin, n_chunks;
};
constexpr unsigned EVAL_A = 10, EVAL_HC = 64;  // entries per row of the split (four per lane), rows per workgroup
__global__ void __launch_bounds__(NT) k_eval_cols(const EvalCol* __restrict__ cols, int n_cols, E2* __restrict__ partials) {
    __shared__ E2 smem[NT / 64];
    int c = 0;
    {
        int lo = 0, hi = n_cols - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cols[mid].wg_begin <= blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        c = lo;
    }
    const EvalCol C = cols[c];
    const unsigned h0 = (blockIdx.x - C.wg_begin) * EVAL_HC, h1 = min(h0 + EVAL_HC, C.n_hi);
    E2 acc[1] = {e2_zero()};
    if (C.a == EVAL_A) {
        // sum_h hi[h] col[h, e] per owned entry e, unreduced (hi[h] is wave-uniform: scalar registers), then once  x lo[e]
        Acc5 w[4][2];
#pragma unroll
        for (int m = 0; m < 4; m++) w[m][0] = w[m][1] = Acc5{0, 0, 0, 0, 0};
        const uint64_t* base = C.col + 2 * threadIdx.x;
        unsigned h = h0;
        for (; h + 2 <= h1; h += 2) {
            const ulonglong2 v0 = *reinterpret_cast<const ulonglong2*>(base + ((size_t)h << EVAL_A));
            const ulonglong2 v1 = *reinterpret_cast<const ulonglong2*>(base + ((size_t)h << EVAL_A) + 512);
            const ulonglong2 v2 = *reinterpret_cast<const ulonglong2*>(base + ((size_t)(h + 1) << EVAL_A));
            const ulonglong2 v3 = *reinterpret_cast<const ulonglong2*>(base + ((size_t)(h + 1) << EVAL_A) + 512);
            const E2 ha = C.hi[h], hb = C.hi[h + 1];
            acc5_add(w[0][0], mul_wide(ha.c0, v0.x)); acc5_add(w[0][1], mul_wide(ha.c1, v0.x));
            acc5_add(w[1][0], mul_wide(ha.c0, v0.y)); acc5_add(w[1][1], mul_wide(ha.c1, v0.y));
            acc5_add(w[2][0], mul_wide(ha.c0, v1.x)); acc5_add(w[2][1], mul_wide(ha.c1, v1.x));
            acc5_add(w[3][0], mul_wide(ha.c0, v1.y)); acc5_add(w[3][1], mul_wide(ha.c1, v1.y));
            acc5_add(w[0][0], mul_wide(hb.c0, v2.x)); acc5_add(w[0][1], mul_wide(hb.c1, v2.x));
            acc5_add(w[1][0], mul_wide(hb.c0, v2.y)); acc5_add(w[1][1], mul_wide(hb.c1, v2.y));
            acc5_add(w[2][0], mul_wide(hb.c0, v3.x)); acc5_add(w[2][1], mul_wide(hb.c1, v3.x));
            acc5_add(w[3][0], mul_wide(hb.c0, v3.y)); acc5_add(w[3][1], mul_wide(hb.c1, v3.y));
        }
        for (; h < h1; h++) {
            const ulonglong2 v0 = *reinterpret_cast<const ulonglong2*>(base + ((size_t)h << EVAL_A));
            const ulonglong2 v1 = *reinterpret_cast<const ulonglong2*>(base + ((size_t)h << EVAL_A) + 512);
            const E2 ha = C.hi[h];
            acc5_add(w[0][0], mul_wide(ha.c0, v0.x)); acc5_add(w[0][1], mul_wide(ha.c1, v0.x));
            acc5_add(w[1][0], mul_wide(ha.c0, v0.y)); acc5_add(w[1][1], mul_wide(ha.c1, v0.y));
            acc5_add(w[2][0], mul_wide(ha.c0, v1.x)); acc5_add(w[2][1], mul_wide(ha.c1, v1.x));
            acc5_add(w[3][0], mul_wide(ha.c0, v1.y)); acc5_add(w[3][1], mul_wide(ha.c1, v1.y));
        }
        const E2* lo = C.lo + 2 * threadIdx.x;
        acc[0] = E2{acc5_reduce(w[0][0]), acc5_reduce(w[0][1])} * lo[0] + E2{acc5_reduce(w[1][0]), acc5_reduce(w[1][1])} * lo[1] +
                 E2{acc5_reduce(w[2][0]), acc5_reduce(w[2][1])} * lo[512] + E2{acc5_reduce(w[3][0]), acc5_reduce(w[3][1])} * lo[513];
    } else {
        // a table of fewer than 2^10 rows: one row of the split, entry by entry
        const size_t len = (size_t)1 << C.a;
        for (size_t e = threadIdx.x; e < len; e += NT) acc[0] = acc[0] + e2_mul_base(C.lo[e] * C.hi[0], C.col[e]);
    }
    red::block_sum<1, NT>(acc, smem);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc[0];
}
// one wave per table: the sum of its workgroups' partials
__global__ void __launch_bounds__(64) k_eval_cols_finish(const EvalCol* __restrict__ cols, const E2* __restrict__ partials, E2* __restrict__ out) {
    const EvalCol C = cols[blockIdx.x];
    E2 acc = e2_zero();
    for (unsigned k = threadIdx.x; k < C.n_chunks; k += 64) acc = acc + partials[C.wg_begin + k];
    acc = red::wave_sum(acc);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

extern "C" int ceno_hip_mle_evaluate_prefix_batch(ceno_hip_ctx* ctx, int n, ceno_hip_mle* const* cols, const uint64_t* point, int point_len, ceno_hip_stream s,
                                                  uint64_t* out) {
    CHECK_ARG(ctx, n >= 0 && (n == 0 || (cols && out && (point || point_len == 0))), "evaluate batch: NULL argument");
    CHECK_ARG(ctx, point_len >= 0 && point_len <= 40, "evaluate batch: point of %d elements", point_len);
    if (n == 0) return 0;
    hipStream_t st = ctx_stream(ctx, s);
    // half tables per distinct table size (every table is evaluated at the first num_vars elements of the point)
    size_t half_off[41];
    bool have[41] = {};
    size_t half_total = 0, wgs = 0;
    std::vector<EvalCol> desc((size_t)n);
    for (int j = 0; j < n; j++) {
        CHECK_ARG(ctx, cols[j] && !cols[j]->is_ext && cols[j]->num_vars <= point_len, "evaluate batch: table %d is not a base-field table of at most %d variables", j, point_len);
        const int nv = cols[j]->num_vars;
        const int a = std::min(nv, (int)EVAL_A);
        if (!have[nv]) {
            have[nv] = true;
            half_off[nv] = half_total;
            half_total += ((size_t)1 << a) + ((size_t)1 << (nv - a));
        }
        EvalCol& C = desc[(size_t)j];
        C.col = cols[j]->d;
        C.a = (uint32_t)a;
        C.n_hi = 1u << (nv - a);
        C.wg_begin = (uint32_t)wgs;
        C.n_chunks = a == (int)EVAL_A ? (C.n_hi + EVAL_HC - 1) / EVAL_HC : 1u;
        wgs += C.n_chunks;
    }
    CHECK_ARG(ctx, wgs < ((size_t)1 << 31), "evaluate batch too large");
    // one block: half tables | partials | results | descriptors | arrival counter of k_eq_halves
    const size_t bytes = (half_total + wgs + (size_t)n) * sizeof(E2) + (size_t)n * sizeof(EvalCol) + 64;
    void *scratch = nullptr, *hb = nullptr, *db = nullptr;
    TRY(ctx_alloc(ctx, bytes, &scratch));
    int rc = ctx_pinned_alloc(ctx, (size_t)n * (sizeof(EvalCol) + sizeof(E2)), &hb, &db);  // descriptors out, results back
    if (rc) {
        ctx_free_on(ctx, scratch, st);
        return rc;
    }
    E2* halves = (E2*)scratch;
    E2* partials = halves + half_total;
    E2* d_out = partials + wgs;
    EvalCol* d_desc = reinterpret_cast<EvalCol*>(d_out + n);
    unsigned* counter = reinterpret_cast<unsigned*>(d_desc + n);
    for (int j = 0; j < n; j++) {
        EvalCol& C = desc[(size_t)j];
        C.lo = halves + half_off[cols[j]->num_vars];
        C.hi = C.lo + ((size_t)1 << C.a);
    }
    memcpy(hb, desc.data(), (size_t)n * sizeof(EvalCol));
    hipError_t e = hipMemcpyAsync(d_desc, hb, (size_t)n * sizeof(EvalCol), hipMemcpyHostToDevice, st);
    PointArg pt;
    for (int k = 0; k < point_len; k++) pt.r[k] = E2{point[2 * k], point[2 * k + 1]};
    for (int nv = 0; nv <= point_len && e == hipSuccess; nv++) {
        if (!have[nv]) continue;
        const int a = std::min(nv, (int)EVAL_A), b = nv - a;
        E2* lo = halves + half_off[nv];
        hipLaunchKernelGGL(k_eq_halves, dim3(grid_for(((size_t)1 << a) + ((size_t)1 << b), NT, MAXB)), dim3(NT), 0, st, lo, a, lo + ((size_t)1 << a), b, pt, counter);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_eval_cols, dim3((unsigned)wgs), dim3(NT), 0, st, d_desc, n, partials);
        hipLaunchKernelGGL(k_eval_cols_finish, dim3((unsigned)n), dim3(64), 0, st, d_desc, partials, d_out);
        e = hipGetLastError();
    }
    char* h_res = (char*)hb + (size_t)n * sizeof(EvalCol);
    if (e == hipSuccess) e = hipMemcpyAsync(h_res, d_out, (size_t)n * sizeof(E2), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) memcpy(out, h_res, (size_t)n * sizeof(E2));
    ctx_pinned_free(ctx, hb);
    ctx_free_on(ctx, scratch, st);
    if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_HIP, "evaluate batch: %s", hipGetErrorString(e));
    return 0;
}

int ceno_hip_mle_fix_variables(ceno_hip_ctx* ctx, const ceno_hip_mle* m, const uint64_t* point, int n_fix, ceno_hip_stream s, ceno_hip_mle** out) {
    CHECK_ARG(ctx, m && out && (point || n_fix == 0), "NULL argument");
    CHECK_ARG(ctx, n_fix >= 0 && n_fix <= m->num_vars, "n_fix %d out of range", n_fix);
    hipStream_t st = ctx_stream(ctx, s);
    ceno_hip_mle* res = nullptr;
    TRY(ceno_hip_mle_alloc(ctx, m->num_vars - n_fix, 1, &res));
    if (n_fix == 0) {
        // ext copy (base inputs are widened)
        if (m->is_ext) {
            HIP_TRY(ctx, hipMemcpyAsync(res->d, m->d, m->bytes(), hipMemcpyDeviceToDevice, st));
        } else {
            ceno_hip_mle_free(ctx, res);
            return ctx_fail(ctx, CENO_HIP_ERR_UNSUPPORTED, "fix_variables with n_fix = 0 on a base table");
        }
        *out = res;
        return 0;
    }
    // ping-pong: scratch A holds 2^(nv-1) elements, scratch B 2^(nv-2); the last fold lands in res
    void* scratch = nullptr;
    size_t half = m->len() / 2;
    if (n_fix > 1) TRY(ctx_alloc(ctx, (half + half / 2 + 1) * sizeof(E2), &scratch));
    uint64_t* bufA = (uint64_t*)scratch;
    uint64_t* bufB = reinterpret_cast<uint64_t*>(reinterpret_cast<E2*>(scratch) + half);
    const uint64_t* cur = m->d;
    int cur_ext = m->is_ext;
    int rc = 0;
    for (int k = 0; k < n_fix && rc == 0; k++) {
        uint64_t* dst = (k == n_fix - 1) ? res->d : ((k & 1) ? bufB : bufA);
        rc = launch_fold(ctx, cur, cur_ext, dst, half, E2{point[2 * k], point[2 * k + 1]}, st);
        cur = dst;
        cur_ext = 1;
        half >>= 1;
    }
    // no wait: the result is ordered on `st` like every other table, and the scratch returns to the pool tagged with this stream
    // (another stream gets it only once this one has drained)
    hipError_t e = hipGetLastError();
    ctx_free(ctx, scratch);
    if (rc || e != hipSuccess) {
        ceno_hip_mle_free(ctx, res);
        return rc ? rc : ctx_fail(ctx, CENO_HIP_ERR_HIP, "fix_variables: %s", hipGetErrorString(e));
    }
    *out = res;
    return 0;
}


// out[i] = in[2 i + odd]  (filter_mle_even_odd_batch, ceno_zkvm/src/scheme/gpu/util.rs:186-266)
int ceno_hip_mle_filter_even_odd(ceno_hip_ctx* ctx, const ceno_hip_mle* m, int odd, ceno_hip_stream s, ceno_hip_mle** out) {
    CHECK_ARG(ctx, m && out, "NULL argument");
    CHECK_ARG(ctx, m->num_vars >= 1, "polynomial must have at least one variable");
    hipStream_t st = ctx_stream(ctx, s);
    ceno_hip_mle* o = nullptr;
    TRY(ceno_hip_mle_alloc(ctx, m->num_vars - 1, m->is_ext, &o));
    const size_t n = o->len();
    const int ew = m->is_ext ? 2 : 1;
    hipLaunchKernelGGL(k_take_stride2, dim3(grid_for(n * ew, NT, 2048)), dim3(NT), 0, st, m->d, o->d, n, ew, odd ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ceno_hip_mle_free(ctx, o);
        return ctx_fail(ctx, CENO_HIP_ERR_HIP, "filter_even_odd: %s", hipGetErrorString(e));
    }
    *out = o;
    return 0;
}

}  // extern "C"
