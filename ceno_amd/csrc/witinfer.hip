// Element-wise record inference and trace transpose.
//
// wit_infer: reference `wit_infer_by_monomial_expr` (EXT; CPU caller gkr_iop/src/cpu/mod.rs:119-176,
// GPU call site gkr_iop/src/gpu/mod.rs:599-609): out[o][x] = sum_t c_t prod_j f_j[x] with base-field
// witness columns in and extension-field record columns out (RLC with alpha, beta).
// Traffic: 8 B per input column + 16 B per output column per row -> HBM bound; column re-reads across
// outputs of the same row tile are served by L1/L2.
// transpose: reference `common::transpose::matrix_transpose` (ceno_zkvm/src/scheme/gpu/mod.rs:84,963-968):
// row-major RowMajorMatrix values[row*width+col] -> column-major device layout.
#include "common.hpp"
#include "witinfer_dev.hpp"

#include <algorithm>

using namespace gl;

static constexpr int NT = 256;
static constexpr unsigned MAXB = 2048;

// The plan (slots, coefficients, CSR offsets) is pulled into LDS once per workgroup: every lane walks the same records, and
// from global memory that walk is a chain of four dependent loads per factor.  Base-field factors of a term are multiplied
// together first (one 64-bit product each) and meet the extension-field coefficient once at the end.
__global__ void __launch_bounds__(NT) k_wit_infer(WiPlan pl, size_t len, int num_mles, int num_terms, int num_factors) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    const WiLds L = wi_stage<NT>(dyn, pl, num_mles, num_terms, num_factors, true);
    // one work item = one (record, row): a chip of 2^12 rows and 30 records is 120 k items, not 4 k lanes walking 30 records each — the
    // walk is a chain of dependent loads (plan word -> column pointer -> value), and what hides it is lanes, not loop iterations.  Items
    // of one record are consecutive rows (coalesced); the columns the records share come back from L2.  (len is a power of two.)
    const size_t stride = (size_t)gridDim.x * NT, items = (size_t)pl.num_outs * len, mask = len - 1;
    const int shift = __builtin_ctzll((unsigned long long)len);
    for (size_t it = (size_t)blockIdx.x * NT + threadIdx.x; it < items; it += stride) {
        const int o = (int)(it >> shift);
        const size_t x = it & mask;
        L.outs[o][x] = wi_eval(L, o, x);
    }
}
// MANY plans in one launch (ceno_hip_wit_infer_many): the records of all chips of a shard.  A workgroup belongs to one plan (BlkRef), stages
// that plan in LDS and walks its (record, row) items as k_wit_infer does.
struct WiJob {
    WiPlan pl;
    size_t len;
    int num_mles, num_terms, num_factors, pad_;
};
struct WiBlk {
    uint32_t job, blk, nblk;
};
__global__ void __launch_bounds__(NT) k_wit_infer_many(const WiJob* __restrict__ jobs, const WiBlk* __restrict__ blks) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    const WiBlk b = blks[blockIdx.x];
    const WiJob J = jobs[b.job];
    const WiLds L = wi_stage<NT>(dyn, J.pl, J.num_mles, J.num_terms, J.num_factors, true);
    const size_t len = J.len, stride = (size_t)b.nblk * NT, items = (size_t)J.pl.num_outs * len, mask = len - 1;
    const int shift = __builtin_ctzll((unsigned long long)len);
    for (size_t it = (size_t)b.blk * NT + threadIdx.x; it < items; it += stride) {
        const int o = (int)(it >> shift);
        const size_t x = it & mask;
        L.outs[o][x] = wi_eval(L, o, x);
    }
}
// plans too large for the LDS stage (thousands of terms) walk the records in global memory
__global__ void __launch_bounds__(NT) k_wit_infer_big(WiPlan pl, size_t len) {
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t x = (size_t)blockIdx.x * NT + threadIdx.x; x < len; x += stride) {
        for (int o = 0; o < pl.num_outs; o++) {
            E2 acc = e2_zero();
            for (uint32_t t = pl.out_term_off[o]; t < pl.out_term_off[o + 1]; t++) {
                E2 v = pl.coeffs[t];
                for (uint32_t k = pl.term_off[t]; k < pl.term_off[t + 1]; k++) {
                    const WiSlot sl = pl.mles[pl.term_idx[k]];
                    if (sl.is_ext) v = v * reinterpret_cast<const E2*>(sl.ptr)[x];
                    else v = e2_mul_base(v, sl.ptr[x]);
                }
                acc = acc + v;
            }
            pl.outs[o][x] = acc;
        }
    }
}

// 32x32 tile transpose of 64-bit words through LDS (+1 padding: conflict-free column reads)
// 1-D grid of tiles, column tiles fastest (consecutive workgroups read consecutive segments of the same rows): the grid's
// y dimension is limited to 65535, which 2^21 rows / 32 exceeds (the reference's add_op_21 bench, benches/riscv_add.rs:74-150)
__global__ void __launch_bounds__(256) k_transpose(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, size_t rows, size_t width,
                                                   unsigned col_tiles) {
    __shared__ uint64_t tile[32][33];
    const size_t col0 = (size_t)(blockIdx.x % col_tiles) * 32, row0 = (size_t)(blockIdx.x / col_tiles) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        size_t row = row0 + r, col = col0 + tx;
        if (row < rows && col < width) tile[r][tx] = in[row * width + col];
    }
    __syncthreads();
    for (int c = ty; c < 32; c += 8) {
        size_t col = col0 + c, row = row0 + tx;
        if (row < rows && col < width) out[col * rows + row] = tile[tx][c];
    }
}

template <typename T>
static int up(ceno_hip_ctx* ctx, const T* h, size_t n, hipStream_t st, std::vector<void*>& allocs, T** out) {
    void* p = nullptr;
    TRY(ctx_alloc(ctx, (n ? n : 1) * sizeof(T), &p));
    allocs.push_back(p);
    if (n) HIP_TRY(ctx, hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, st));
    *out = (T*)p;
    return 0;
}

extern "C" {

int ceno_hip_wit_infer(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, int num_mles, const uint64_t* term_coeffs,
                       const uint32_t* term_offsets, const uint32_t* term_mle_idx, int num_terms, const uint32_t* out_term_offsets,
                       int num_outs, int num_vars, ceno_hip_stream s, ceno_hip_mle** outs) {
    CENO_TIMED("wit_infer");
    CHECK_ARG(ctx, mles && term_coeffs && term_offsets && term_mle_idx && out_term_offsets && outs, "NULL argument");
    CHECK_ARG(ctx, num_mles >= 1 && num_terms >= 0 && num_outs >= 1, "empty wit_infer plan");
    CHECK_ARG(ctx, out_term_offsets[0] == 0 && (int)out_term_offsets[num_outs] == num_terms, "out_term_offsets must cover all terms");
    for (int j = 0; j < num_mles; j++) CHECK_ARG(ctx, mles[j] && mles[j]->num_vars == num_vars, "mle %d must have %d variables", j, num_vars);
    for (uint32_t k = 0; k < term_offsets[num_terms]; k++) CHECK_ARG(ctx, (int)term_mle_idx[k] < num_mles, "term factor %u out of range", term_mle_idx[k]);
    hipStream_t st = ctx_stream(ctx, s);
    std::vector<void*> allocs;
    std::vector<ceno_hip_mle*> res(num_outs, nullptr);
    int rc = 0;
    for (int o = 0; o < num_outs && !rc; o++) rc = ceno_hip_mle_alloc(ctx, num_vars, 1, &res[o]);
    // the whole plan travels as ONE blob in one copy (six small uploads cost ~10 us each in front of a ~100 us kernel);
    // the source is pageable, so the runtime has captured it when hipMemcpyAsync returns
    const size_t n_fac = term_offsets[num_terms];
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_slots = 0, o_coeffs = al(o_slots + (size_t)num_mles * sizeof(WiSlot)), o_outs = al(o_coeffs + (size_t)num_terms * sizeof(E2)),
                 o_toff = al(o_outs + (size_t)num_outs * sizeof(E2*)), o_tidx = al(o_toff + ((size_t)num_terms + 1) * 4),
                 o_ooff = al(o_tidx + n_fac * 4), total = al(o_ooff + ((size_t)num_outs + 1) * 4);
    std::vector<char> blob(total, 0);
    for (int j = 0; j < num_mles; j++) reinterpret_cast<WiSlot*>(blob.data() + o_slots)[j] = WiSlot{mles[j]->d, mles[j]->is_ext, 0};
    for (int t = 0; t < num_terms; t++) reinterpret_cast<E2*>(blob.data() + o_coeffs)[t] = E2{term_coeffs[2 * t], term_coeffs[2 * t + 1]};
    for (int o = 0; o < num_outs && !rc; o++) reinterpret_cast<E2**>(blob.data() + o_outs)[o] = reinterpret_cast<E2*>(res[o]->d);
    memcpy(blob.data() + o_toff, term_offsets, ((size_t)num_terms + 1) * 4);
    if (n_fac) memcpy(blob.data() + o_tidx, term_mle_idx, n_fac * 4);
    memcpy(blob.data() + o_ooff, out_term_offsets, ((size_t)num_outs + 1) * 4);
    char* d_blob = nullptr;
    rc = rc ? rc : up(ctx, blob.data(), blob.size(), st, allocs, &d_blob);
    const size_t lds = wit_infer_lds(num_mles, num_terms, (int)n_fac, num_outs);
    if (!rc) {
        WiPlan pl{};
        pl.mles = reinterpret_cast<const WiSlot*>(d_blob + o_slots);
        pl.coeffs = reinterpret_cast<const E2*>(d_blob + o_coeffs);
        pl.term_off = reinterpret_cast<const uint32_t*>(d_blob + o_toff);
        pl.term_idx = reinterpret_cast<const uint32_t*>(d_blob + o_tidx);
        pl.out_term_off = reinterpret_cast<const uint32_t*>(d_blob + o_ooff);
        pl.outs = reinterpret_cast<E2* const*>(d_blob + o_outs);
        pl.num_outs = num_outs;
        size_t len = (size_t)1 << num_vars;
        if (lds <= 60 * 1024) hipLaunchKernelGGL(k_wit_infer, dim3(grid_for(len * (size_t)num_outs, NT, 4 * MAXB)), dim3(NT), lds, st, pl, len, num_mles, num_terms, (int)n_fac);
        else hipLaunchKernelGGL(k_wit_infer_big, dim3(grid_for(len, NT, MAXB)), dim3(NT), 0, st, pl, len);
        // no wait: the outputs are ordered on `st` like every other result, and the plan blob returns to the pool tagged with
        // this stream (another stream gets it only after this one has drained)
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = ctx_fail(ctx, CENO_HIP_ERR_HIP, "wit_infer: %s", hipGetErrorString(e));
    }
    for (void* p : allocs) ctx_free(ctx, p);
    if (rc) {
        for (auto* m : res) ceno_hip_mle_free(ctx, m);
        return rc;
    }
    for (int o = 0; o < num_outs; o++) outs[o] = res[o];
    return 0;
}

int ceno_hip_wit_infer_many(ceno_hip_ctx* ctx, const ceno_hip_wit_plan* plans, int n, ceno_hip_stream s) {
    CENO_TIMED("wit_infer_many");
    CHECK_ARG(ctx, plans && n >= 1 && n <= 4096, "wit_infer_many: bad arguments");
    hipStream_t st = ctx_stream(ctx, s);
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    // pass 1: checks and the blob's layout; a plan too large for the LDS stage goes through the single-plan call
    struct Lay {
        size_t o_slots, o_coeffs, o_outs, o_toff, o_tidx, o_ooff;
        size_t n_fac, lds;
        bool many;
    };
    std::vector<Lay> lay((size_t)n);
    size_t total = al((size_t)n * sizeof(WiJob)), max_lds = 0;
    int n_many = 0;
    for (int i = 0; i < n; i++) {
        const ceno_hip_wit_plan& P = plans[i];
        CHECK_ARG(ctx, P.mles && P.term_coeffs && P.term_offsets && P.term_mle_idx && P.out_term_offsets && P.outs, "wit_infer_many: plan %d: NULL argument", i);
        CHECK_ARG(ctx, P.num_mles >= 1 && P.num_terms >= 0 && P.num_outs >= 1 && P.num_vars >= 0 && P.num_vars < 40, "wit_infer_many: plan %d is empty", i);
        CHECK_ARG(ctx, P.out_term_offsets[0] == 0 && (int)P.out_term_offsets[P.num_outs] == P.num_terms, "wit_infer_many: plan %d: out_term_offsets must cover all terms", i);
        for (int j = 0; j < P.num_mles; j++)
            CHECK_ARG(ctx, P.mles[j] && P.mles[j]->num_vars == P.num_vars, "wit_infer_many: plan %d: mle %d must have %d variables", i, j, P.num_vars);
        Lay& L = lay[(size_t)i];
        L.n_fac = P.term_offsets[P.num_terms];
        for (uint32_t k = 0; k < L.n_fac; k++) CHECK_ARG(ctx, (int)P.term_mle_idx[k] < P.num_mles, "wit_infer_many: plan %d: term factor %u out of range", i, P.term_mle_idx[k]);
        L.lds = wit_infer_lds(P.num_mles, P.num_terms, (int)L.n_fac, P.num_outs);
        L.many = L.lds <= 60 * 1024;
        if (!L.many) continue;
        n_many++;
        max_lds = std::max(max_lds, L.lds);
        L.o_slots = total;
        total = al(total + (size_t)P.num_mles * sizeof(WiSlot));
        L.o_coeffs = total;
        total = al(total + (size_t)P.num_terms * sizeof(E2));
        L.o_outs = total;
        total = al(total + (size_t)P.num_outs * sizeof(E2*));
        L.o_toff = total;
        total = al(total + ((size_t)P.num_terms + 1) * 4);
        L.o_tidx = total;
        total = al(total + L.n_fac * 4);
        L.o_ooff = total;
        total = al(total + ((size_t)P.num_outs + 1) * 4);
    }
    // outputs
    std::vector<ceno_hip_mle*> made;
    auto undo = [&]() {
        for (auto* m : made) ceno_hip_mle_free(ctx, m);
        for (int i = 0; i < n; i++)
            for (int o = 0; o < plans[i].num_outs; o++) plans[i].outs[o] = nullptr;
    };
    for (int i = 0; i < n; i++)
        for (int o = 0; o < plans[i].num_outs; o++) plans[i].outs[o] = nullptr;
    for (int i = 0; i < n; i++) {
        if (!lay[(size_t)i].many) continue;
        for (int o = 0; o < plans[i].num_outs; o++) {
            ceno_hip_mle* m = nullptr;
            const int rc = ceno_hip_mle_alloc(ctx, plans[i].num_vars, 1, &m);
            if (rc) {
                undo();
                return rc;
            }
            made.push_back(m);
            plans[i].outs[o] = m;
        }
    }
    if (n_many) {
        std::vector<WiBlk> blks;
        void* d_blob = nullptr;
        const size_t o_blks_guess = total;  // (block table follows the plans)
        // block table
        int job = 0;
        std::vector<int> job_of((size_t)n, -1);
        for (int i = 0; i < n; i++) {
            if (!lay[(size_t)i].many) continue;
            job_of[(size_t)i] = job;
            const size_t items = ((size_t)1 << plans[i].num_vars) * (size_t)plans[i].num_outs;
            const uint32_t nblk = (uint32_t)std::min<size_t>(std::max<size_t>((items + (size_t)NT * 2 - 1) / ((size_t)NT * 2), 1), 2048);
            for (uint32_t b = 0; b < nblk; b++) blks.push_back(WiBlk{(uint32_t)job, b, nblk});
            job++;
        }
        total = al(o_blks_guess + blks.size() * sizeof(WiBlk));
        int rc = ctx_alloc(ctx, total, &d_blob);
        if (rc) {
            undo();
            return rc;
        }
        std::vector<char> blob(total, 0);
        char* d = (char*)d_blob;
        for (int i = 0; i < n; i++) {
            const Lay& L = lay[(size_t)i];
            if (!L.many) continue;
            const ceno_hip_wit_plan& P = plans[i];
            for (int j = 0; j < P.num_mles; j++) reinterpret_cast<WiSlot*>(blob.data() + L.o_slots)[j] = WiSlot{P.mles[j]->d, P.mles[j]->is_ext, 0};
            for (int t = 0; t < P.num_terms; t++) reinterpret_cast<E2*>(blob.data() + L.o_coeffs)[t] = E2{P.term_coeffs[2 * t], P.term_coeffs[2 * t + 1]};
            for (int o = 0; o < P.num_outs; o++) reinterpret_cast<E2**>(blob.data() + L.o_outs)[o] = reinterpret_cast<E2*>(P.outs[o]->d);
            memcpy(blob.data() + L.o_toff, P.term_offsets, ((size_t)P.num_terms + 1) * 4);
            if (L.n_fac) memcpy(blob.data() + L.o_tidx, P.term_mle_idx, L.n_fac * 4);
            memcpy(blob.data() + L.o_ooff, P.out_term_offsets, ((size_t)P.num_outs + 1) * 4);
            WiJob& J = reinterpret_cast<WiJob*>(blob.data())[job_of[(size_t)i]];
            J.pl.mles = reinterpret_cast<const WiSlot*>(d + L.o_slots);
            J.pl.coeffs = reinterpret_cast<const E2*>(d + L.o_coeffs);
            J.pl.term_off = reinterpret_cast<const uint32_t*>(d + L.o_toff);
            J.pl.term_idx = reinterpret_cast<const uint32_t*>(d + L.o_tidx);
            J.pl.out_term_off = reinterpret_cast<const uint32_t*>(d + L.o_ooff);
            J.pl.outs = reinterpret_cast<E2* const*>(d + L.o_outs);
            J.pl.num_outs = P.num_outs;
            J.len = (size_t)1 << P.num_vars;
            J.num_mles = P.num_mles;
            J.num_terms = P.num_terms;
            J.num_factors = (int)L.n_fac;
        }
        memcpy(blob.data() + o_blks_guess, blks.data(), blks.size() * sizeof(WiBlk));
        // (pageable source: the runtime has captured it when hipMemcpyAsync returns)
        hipError_t e = hipMemcpyAsync(d_blob, blob.data(), total, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_wit_infer_many, dim3((unsigned)blks.size()), dim3(NT), max_lds, st, (const WiJob*)d, (const WiBlk*)(d + o_blks_guess));
            e = hipGetLastError();
        }
        ctx_free(ctx, d_blob);
        if (e != hipSuccess) {
            undo();
            return ctx_fail(ctx, CENO_HIP_ERR_HIP, "wit_infer_many: %s", hipGetErrorString(e));
        }
    }
    for (int i = 0; i < n; i++) {
        if (lay[(size_t)i].many) continue;
        const ceno_hip_wit_plan& P = plans[i];
        const int rc = ceno_hip_wit_infer(ctx, P.mles, P.num_mles, P.term_coeffs, P.term_offsets, P.term_mle_idx, P.num_terms, P.out_term_offsets, P.num_outs, P.num_vars,
                                          s, P.outs);
        if (rc) {
            for (int o = 0; o < P.num_outs; o++) P.outs[o] = nullptr;
            undo();
            return rc;
        }
        for (int o = 0; o < P.num_outs; o++) made.push_back(P.outs[o]);
    }
    return 0;
}

int ceno_hip_transpose(ceno_hip_ctx* ctx, const uint64_t* dev_row_major, size_t rows, size_t width, uint64_t* dev_col_major, ceno_hip_stream s) {
    CHECK_ARG(ctx, dev_row_major && dev_col_major && rows > 0 && width > 0, "bad transpose arguments");
    hipStream_t st = ctx_stream(ctx, s);
    const size_t col_tiles = (width + 31) / 32, row_tiles = (rows + 31) / 32;
    CHECK_ARG(ctx, col_tiles <= 0xFFFFFFFFull && col_tiles * row_tiles <= 0x7FFFFFFFull, "transpose: %zu x %zu is too large", rows, width);
    hipLaunchKernelGGL(k_transpose, dim3((unsigned)(col_tiles * row_tiles)), dim3(256), 0, st, dev_row_major, dev_col_major, rows, width, (unsigned)col_tiles);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
