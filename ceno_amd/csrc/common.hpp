// Internal definitions shared by the translation units of libceno_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <map>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <memory>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/ceno_hip.h"
#include "gl64.hpp"

using gl::E2;

struct PoseidonParams;  // poseidon2.hip
struct ceno_hip_ctx;
void merkle_drop_host_params(ceno_hip_ctx* ctx);  // poseidon2.hip: frees the host copy of the Poseidon2 table
// ctx.hip: counts a pipelined sumcheck in (waits for a trim in progress, and — on a thread that holds no other pipelined sumcheck — for a
// trimmer that is waiting for the gate).  The token is the BEGINNING thread's counter: _end gives it back, on whatever thread it runs
// (a handle released by a finaliser or handed to another thread must not touch the releaser's own count)
typedef std::shared_ptr<std::atomic<int>> PipelinedOwner;
PipelinedOwner ctx_pipelined_begin(ceno_hip_ctx* ctx);
void ctx_pipelined_end(ceno_hip_ctx* ctx, PipelinedOwner& owner);
bool ctx_trim_begin(ceno_hip_ctx* ctx);       // false: pipelined sumchecks are alive (or another trim runs) — nothing may be hipFree'd now
void ctx_trim_end(ceno_hip_ctx* ctx);

#include "pool_mutex.hpp"

struct ceno_hip_ctx {
    int device = 0;
    hipStream_t default_stream = nullptr;
    int num_cus = 256;
    // ---- pool (size-bucketed caching allocator over hipMalloc) ----
    PoolMutex mu;
    size_t pool_limit = 0;  // 0 = unlimited
    size_t pool_used = 0;   // bytes handed out
    size_t pool_peak = 0;   // high-water mark of pool_used since the last ceno_hip_mem_peak(reset)
    size_t pool_cached = 0; // bytes parked in free lists
    size_t pool_booked = 0; // bytes promised to scheduled-but-not-yet-running tasks (ceno_hip_mem_book)
    size_t pool_booked_peak = 0; // high-water mark of pool_booked (ceno_hip_mem_booked_peak)
    size_t pool_capacity = 0;  // booking capacity: pool_limit, or the device memory size when unlimited
    // a cached block remembers the stream its last user was working on (the freeing thread's current stream): it is handed to
    // a DIFFERENT stream only once that stream has drained (ctx_alloc), so a block freed with kernels still queued is never
    // overwritten early by another lane
    std::unordered_map<size_t, std::vector<std::pair<void*, hipStream_t>>> free_lists;
    std::unordered_map<void*, size_t> live;  // ptr -> bucket size
    std::vector<hipStream_t> streams;        // streams created through the C ABI that are still alive
    std::atomic<unsigned> stream_gen{0};     // bumped by ceno_hip_stream_destroy: a thread's "already adopted" shortcut is only valid within one generation
    bool xcd_private_l2 = true;              // gfx942 / gfx950: workgroup-scope atomics of one XCD meet in that XCD's L2 (witgen.hip lookup counters)
    std::vector<hipStream_t> lane_streams;   // the context's own lane streams (ceno_hip_lane_stream): created once, reused by every scheduler run
    // ---- pinned host memory cache (mailboxes of in-flight sumchecks; hipHostMalloc costs ~100 us) ----
    std::unordered_map<size_t, std::vector<void*>> pinned_free;
    std::unordered_map<void*, size_t> pinned_live;
    std::unordered_map<void*, void*> pinned_dev;  // device view of every pinned block the pool owns (asked of the runtime once per block)
    // ---- challenge mailboxes in host-writable device memory (large-BAR boxes): the device polls HBM, the host posts one
    // PCIe write per challenge; vram_state: 0 = not probed, 1 = available, -1 = unavailable (mailboxes stay in pinned memory)
    int vram_state = 0;
    char* vram_arena = nullptr;
    std::vector<int> vram_free_slots;
    // ---- persistent multi-workgroup kernels (k_mid) wait for each other and must ALL be resident: workgroups in flight ----
    // Booked in units of 1/64 of a compute unit: a launch of W workgroups at an occupancy of `nb` workgroups per CU (what
    // hipOccupancyMaxActiveBlocksPerMultiprocessor reports for its dynamic LDS) costs ceil(64 W / nb); the chip offers
    // 64 * num_cus minus headroom.  An atomic of its own: round enqueues never contend with the pool mutex.
    std::atomic<int> mid_wgs_in_flight{0};
    // live pipelined sumchecks (any lane): their queued round kernels wait for a HOST, and hipFree waits for every stream of the
    // device, so the pool's soft-cap trim (ctx_alloc) only runs while this is zero — two lanes each inside hipFree, each with
    // kernels that need the other's... host would otherwise wait for each other until the kernels' poll timeout
    std::atomic<int> pipelined_live{0};
    std::atomic<unsigned long long> eq_launches{0};  // eq-factored round launches so far (ceno_hip_stat_eq_launches: tests, A/B)
    // ... and the other half of that rule: a trim that has STARTED keeps new pipelined sumchecks from starting until it is done
    // (a lane that begins proving while another sits in hipFree has its round kernels waited for by that hipFree, and its own next
    // hipMalloc queues behind the same hipFree inside the runtime: host and kernel then wait for each other until the kernel's
    // poll timeout — seen at the start of a four-lane shard right after a 13 GB batch).  ctx_pipelined_begin / ctx_trim_begin.
    std::mutex gate_mu;
    std::condition_variable gate_cv;
    bool trimming = false;
    int trim_pending = 0;  // threads inside ctx_trim_begin_wait (under gate_mu): new pipelined sumchecks of threads that hold none wait for them
    // ---- errors ----
    std::string err;
    // CENO_HIP_PLAN_REPORT=1: which round kernels the size classes of the sumcheck built last on this context were given (one JSON object
    // per class; ceno_hip_plan_report — benches and tests)
    std::string plan_report;
    // ---- profiling of the dominant kernel (bench.py roofline) ----
    bool prof_on = false;
    bool prof_pipelined = false;  // ceno_hip_prof_enable(ctx, 2): pipelined sumchecks stay pipelined, events around their large dense rounds
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_event_pool;
    uint64_t prof_launches = 0;
    double prof_bytes = 0.0;
    // ---- NTT twiddle tables by (log_n, inverse): pool blocks (booked like every other allocation, freed with the context) ----
    std::mutex tw_mu;
    std::map<std::pair<int, int>, uint64_t*> twiddles;
    std::map<int, uint64_t*> fold_twiddles;  // Basefold fold coefficients by codeword height (a taller table serves as a prefix)
    // ---- poseidon2 parameters (device) ----
    PoseidonParams* poseidon_dev = nullptr;
    PoseidonParams* poseidon_host = nullptr;  // the same table for the host half of a Merkle tree top (merkle_finish_host)
    bool poseidon_pinned = false;  // true once ceno_hip_poseidon2_set_constants supplied a complete external table
};

struct ceno_hip_mle {
    uint64_t* d = nullptr;
    int num_vars = 0;
    int is_ext = 0;
    bool owned = false;  // backed by the ctx pool
    void* aux = nullptr; // pool block that must outlive the kernels that built this table (eq half tables); freed with the handle
    size_t len() const { return (size_t)1 << num_vars; }
    size_t bytes() const { return len() * (is_ext ? 16 : 8); }
};

int ctx_fail(ceno_hip_ctx* ctx, int code, const char* fmt, ...);

// CENO_HIP_HOST_TIMING=1: where the HOST threads of the library spend their time — calls and total microseconds per labelled scope,
// printed by ceno_hip_destroy (tools/dev/lanes_sc.cpp: what concurrent lanes contend for).  Off: one predictable branch per scope.
struct HostTimeSlot {
    const char* label = nullptr;
    std::atomic<unsigned long long> ns{0}, n{0}, seen{0}, max_ns{0};
    void add(unsigned long long d) {
        ns.fetch_add(d, std::memory_order_relaxed);
        n.fetch_add(1, std::memory_order_relaxed);
        unsigned long long m = max_ns.load(std::memory_order_relaxed);
        while (d > m && !max_ns.compare_exchange_weak(m, d, std::memory_order_relaxed)) {}
    }
};
unsigned long long host_timing_skip();  // CENO_HIP_HOST_TIMING_SKIP=k: the first k calls of every scope are warm-up and not counted
HostTimeSlot* host_time_slot(const char* label);
bool host_timing_on();
void host_timing_dump();
struct HostTimed {
    HostTimeSlot* s;
    timespec t0;
    explicit HostTimed(HostTimeSlot* slot) : s(host_timing_on() ? slot : nullptr) {
        if (s) clock_gettime(CLOCK_MONOTONIC, &t0);
    }
    ~HostTimed() {
        if (!s) return;
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (s->seen.fetch_add(1, std::memory_order_relaxed) < host_timing_skip()) return;
        s->add((unsigned long long)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec)));
    }
};
// a region that is not a scope: t0 = host_time_mark() in front of it, host_time_add(slot, t0) behind it (returns the new mark)
inline timespec host_time_mark() {
    timespec t{};
    if (host_timing_on()) clock_gettime(CLOCK_MONOTONIC, &t);
    return t;
}
inline timespec host_time_add(HostTimeSlot* s, const timespec& t0) {
    timespec t1{};
    if (!host_timing_on()) return t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (s->seen.fetch_add(1, std::memory_order_relaxed) >= host_timing_skip()) {
        s->add((unsigned long long)((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec)));
    }
    return t1;
}
#define CENO_TIMED_CAT2(a, b) a##b
#define CENO_TIMED_CAT(a, b) CENO_TIMED_CAT2(a, b)
#define CENO_TIMED(label)                                                                  \
    static HostTimeSlot* CENO_TIMED_CAT(_hts_, __LINE__) = host_time_slot(label);          \
    HostTimed CENO_TIMED_CAT(_ht_, __LINE__)(CENO_TIMED_CAT(_hts_, __LINE__))
int ctx_alloc(ceno_hip_ctx* ctx, size_t bytes, void** out);
void ctx_free(ceno_hip_ctx* ctx, void* p);                         // tag = the stream the calling thread resolved last
void ctx_free_on(ceno_hip_ctx* ctx, void* p, hipStream_t owner);   // tag = the stream that used the block
void ctx_free_many_on(ceno_hip_ctx* ctx, void* const* ptrs, size_t n, hipStream_t owner, bool ask_drained = true);  // the same for all blocks of one handle
// pinned, device-mapped host memory from a per-context cache; *dev_view is the device address of *host
// 64-byte slot of fine-grained device memory the host can write through the BAR (nullptr when unavailable)
void* ctx_vram_slot_alloc(ceno_hip_ctx* ctx);
void ctx_vram_slot_free(ceno_hip_ctx* ctx, void* slot);
int ctx_pinned_alloc(ceno_hip_ctx* ctx, size_t bytes, void** host, void** dev_view);
void ctx_pinned_free(ceno_hip_ctx* ctx, void* host);
// make ctx->device the calling thread's current device (reference: `ensure_context`, gkr_iop/src/gpu/mod.rs:91-92): a fresh
// thread starts on device 0, and allocations / stream creation / launches follow the CURRENT device, not the context's
void ctx_make_current(ceno_hip_ctx* ctx);
// every entry point that touches the device resolves its stream through here, which also makes the device current
extern thread_local hipStream_t ceno_tls_stream;   // the stream the calling thread resolved last (ctx.hip)
extern thread_local hipStream_t ceno_tls_adopted;  // the last caller-made stream this thread registered with the context
extern thread_local unsigned ceno_tls_adopted_gen; // ... and the context's stream generation at that moment
void ctx_adopt_stream(ceno_hip_ctx* ctx, hipStream_t s);
inline hipStream_t ctx_stream(ceno_hip_ctx* ctx, ceno_hip_stream s) {
    ctx_make_current(ctx);
    // streams the library did not create are adopted on first use.  The per-thread shortcut holds only while no stream has been destroyed
    // since: a handle destroyed on ANOTHER thread can come back at the same address, and a thread that still remembered it would skip the
    // adoption — the pool would then treat the stream as dead and hand its tagged blocks out while its work is still queued
    const unsigned gen = ctx->stream_gen.load(std::memory_order_acquire);
    if (s && ((hipStream_t)s != ceno_tls_adopted || gen != ceno_tls_adopted_gen)) {
        ctx_adopt_stream(ctx, (hipStream_t)s);
        ceno_tls_adopted = (hipStream_t)s;
        ceno_tls_adopted_gen = gen;
    }
    return ceno_tls_stream = (s ? (hipStream_t)s : ctx->default_stream);
}

// profiling hooks (ctx.hip)
void prof_begin(ceno_hip_ctx* ctx, hipStream_t st);
void prof_end(ceno_hip_ctx* ctx, hipStream_t st, double algorithmic_bytes, int launches = 1);
void prof_count(ceno_hip_ctx* ctx, double algorithmic_bytes);  // a launch inside an open event pair

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return ctx_fail((ctx), CENO_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                                        \
    } while (0)

#define CHECK_ARG(ctx, cond, ...)                                        \
    do {                                                                 \
        if (!(cond)) return ctx_fail((ctx), CENO_HIP_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define TRY(expr)                 \
    do {                          \
        int _rc = (expr);         \
        if (_rc != 0) return _rc; \
    } while (0)

// launch helper: grid for a grid-stride kernel over `work` items
inline unsigned grid_for(size_t work, unsigned block, unsigned max_blocks) {
    size_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (unsigned)g;
}

// Largest grid (<= max_blocks) of a grid-stride kernel at which EVERY workgroup is resident at once: occupancy (registers, LDS)
// x compute units.  A kernel bound by VALU issue whose launch exceeds it runs a second, partly filled dispatch wave: k_gen at
// 168 VGPRs holds 3 workgroups per CU = 768; launched as 1024 the last 256 run alone on their CUs at one wave per SIMD
// (batched main sumcheck 12.9 -> 11.8 ms, tools/dev/ab_gen_maxb.sh).  What is cached is the kernel's workgroups PER CU, keyed by
// everything the occupancy query depends on — (device, kernel, block size, exact dynamic LDS bytes) — and multiplied by the asking
// context's CU count at the call: two contexts on different devices, or two LDS sizes inside one KB, never share an entry.
template <typename F>
inline unsigned resident_grid(ceno_hip_ctx* ctx, F kernel, int block, size_t dyn_lds, unsigned max_blocks) {
    static PoolMutex mu;  // (a map lookup per launch, from every lane: a spin lock like the pool's)
    typedef std::tuple<int, const void*, int, size_t> Key;
    static std::map<Key, unsigned> cache;  // -> workgroups per CU (0: the query failed, no cap)
    const Key key{ctx->device, reinterpret_cast<const void*>(kernel), block, dyn_lds};
    unsigned per_cu = 0;
    bool hit = false;
    {
        std::lock_guard<PoolMutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) per_cu = it->second, hit = true;
    }
    if (!hit) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, dyn_lds) == hipSuccess && nb > 0) per_cu = (unsigned)nb;
        else (void)hipGetLastError();
        std::lock_guard<PoolMutex> lk(mu);
        cache[key] = per_cu;
    }
    return per_cu ? std::min(per_cu * (unsigned)ctx->num_cus, max_blocks) : max_blocks;
}

// ---- kernels exported across translation units ----
// fold: out[j] = in[2j] + r (in[2j+1] - in[2j]) for j < half ; `in` base or ext, `out` ext
int launch_fold(ceno_hip_ctx* ctx, const uint64_t* in, int in_is_ext, uint64_t* out, size_t half, E2 r, hipStream_t st);
int launch_eq_build(ceno_hip_ctx* ctx, const uint64_t* host_point, int n, E2 scalar, uint64_t* dev_out, hipStream_t st, void** keep_tmp);
// The set-up work of a sumcheck handle (zero its counter block, pull the plan blob out of the pinned block, arm the rows of the
// persistent mid-round kernel) as data: a tower layer hands it to the kernel that builds its eq table, which then does both — one
// dependent launch (~7 us of stream time) less in front of every layer.  dst == NULL: nothing deferred.
struct SetupJob {
    uint64_t* zero = nullptr;
    size_t zero_words = 0;
    uint64_t* dst = nullptr;
    const uint64_t* src_host_view = nullptr;
    size_t words = 0;
    uint64_t* ones = nullptr;
    size_t ones_words = 0;
};
// eq table of n <= the one-launch limit variables AND the set-up job in one launch; returns 1 (and launches nothing) when n is too large
int launch_eq_build_with_setup(ceno_hip_ctx* ctx, const uint64_t* host_point, int n, E2 scalar, uint64_t* dev_out, hipStream_t st, const SetupJob& job);
// ceno_hip_sumcheck_begin whose set-up launch is handed back in *job instead of being queued (sumcheck.hip)
struct ceno_hip_sumcheck;
struct ceno_hip_sumcheck_plan;
struct ceno_hip_mle;
int sumcheck_begin_deferred(ceno_hip_ctx* ctx, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan, hipStream_t st, ceno_hip_sumcheck** out,
                            SetupJob* job);
void launch_setup_job(const SetupJob& job, hipStream_t st);  // the stand-alone set-up kernel (sumcheck.hip)
