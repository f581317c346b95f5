// Context, memory pool, streams and MLE handles of libceno_hip.so.
// Reference counterpart: the process-global `CUDA_HAL` with its memory pool and stream binding
// (gkr_iop/src/gpu/mod.rs:53-154) and the alloc/copy entry points catalogued in SURVEY.md §2.2.
#include "common.hpp"
#include <chrono>

#include <algorithm>
#include <functional>

static thread_local std::string g_init_err;

int ctx_fail(ceno_hip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) {
        std::lock_guard<PoolMutex> g(ctx->mu);
        ctx->err = buf;
    } else {
        g_init_err = buf;
    }
    return code;
}

static HostTimeSlot g_host_time_slots[128];
static std::atomic<int> g_host_time_n{0};
bool host_timing_on() {
    static const bool on = getenv("CENO_HIP_HOST_TIMING") && atoi(getenv("CENO_HIP_HOST_TIMING")) != 0;
    return on;
}
unsigned long long host_timing_skip() {
    static const unsigned long long k = getenv("CENO_HIP_HOST_TIMING_SKIP") ? strtoull(getenv("CENO_HIP_HOST_TIMING_SKIP"), nullptr, 10) : 0;
    return k;
}
HostTimeSlot* host_time_slot(const char* label) {
    const int k = g_host_time_n.fetch_add(1);
    HostTimeSlot* s = &g_host_time_slots[k < 128 ? k : 127];
    s->label = label;
    return s;
}
extern "C" void ceno_hip_host_timing_dump(const char* what) {
    if (!host_timing_on()) return;
    if (what) fprintf(stderr, "[ceno_hip] host timing: ---- %s\n", what);
    host_timing_dump();
}
void host_timing_dump() {
    if (!host_timing_on()) return;
    const int n = std::min(g_host_time_n.load(), 128);
    for (int k = 0; k < n; k++) {
        HostTimeSlot& s = g_host_time_slots[k];
        const unsigned long long c = s.n.exchange(0), ns = s.ns.exchange(0), mx = s.max_ns.exchange(0);
        if (c) fprintf(stderr, "[ceno_hip] host timing: %-34s %8llu calls  %10.1f us total  %8.2f us each  %9.1f us the longest\n", s.label, c, ns / 1e3, ns / 1e3 / c, mx / 1e3);
    }
}

void ctx_make_current(ceno_hip_ctx* ctx) {
    // hipGetDevice is a thread-local read; other users of the runtime in this process (torch) may have switched the device
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) (void)hipSetDevice(ctx->device);
}

static size_t bucket_size(size_t bytes) {
    if (bytes < 256) return 256;
    if (bytes <= ((size_t)1 << 20)) {
        size_t p = 256;
        while (p < bytes) p <<= 1;
        return p;
    }
    const size_t MB = (size_t)1 << 20;
    return (bytes + MB - 1) / MB * MB;
}

thread_local hipStream_t ceno_tls_stream = nullptr;
thread_local hipStream_t ceno_tls_adopted = nullptr;
thread_local unsigned ceno_tls_adopted_gen = 0;

static bool stream_alive(ceno_hip_ctx* ctx, hipStream_t s) {
    if (s == ctx->default_stream) return true;
    for (hipStream_t t : ctx->streams)
        if (t == s) return true;
    return false;
}
// a stream the library did not create (a torch stream, a Rust-side stream) becomes known the first time a thread resolves it:
// blocks freed while it still has work queued then wait for it to drain like those of the library's own streams
void ctx_adopt_stream(ceno_hip_ctx* ctx, hipStream_t s) {
    std::lock_guard<PoolMutex> g(ctx->mu);
    if (!stream_alive(ctx, s)) ctx->streams.push_back(s);
}
// has everything queued on `s` finished?  A handle the runtime no longer knows (destroyed behind the library's back) has
// nothing queued either; the sticky error of that query is cleared.
static bool stream_drained(hipStream_t s) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return true;
    if (e == hipErrorNotReady) return false;
    (void)hipGetLastError();
    return true;
}

// pipelined sumchecks the CALLING thread has begun and not ended yet: a thread that holds none may wait for the trim gate (the lanes'
// sumchecks end within milliseconds); one that does must not — its own queued kernels are among those a trim would wait for.  The counter
// lives on the heap and every handle keeps a reference to the counter of the thread that BEGAN it, so that an end on another thread (a
// finaliser, a handle handed over) decrements the right one and never the releaser's.
static PipelinedOwner& tls_owner() {
    static thread_local PipelinedOwner c = std::make_shared<std::atomic<int>>(0);
    return c;
}
static int tls_pipelined() { return tls_owner()->load(std::memory_order_relaxed); }
PipelinedOwner ctx_pipelined_begin(ceno_hip_ctx* ctx) {
    CENO_TIMED("ctx_pipelined_begin");
    PipelinedOwner me = tls_owner();
    const bool holds = me->load(std::memory_order_relaxed) > 0;
    std::unique_lock<std::mutex> lk(ctx->gate_mu);
    // a trim in progress holds everybody back; a trimmer that is WAITING for the gate holds back the threads that have nothing in flight
    // (those that do must go on: the trimmer waits for exactly their sumchecks to end) — without this, four lanes proving tower layers back
    // to back never let pipelined_live reach 0 and the waiter starved until its timeout
    ctx->gate_cv.wait(lk, [&] { return !ctx->trimming && (holds || ctx->trim_pending == 0); });
    ctx->pipelined_live.fetch_add(1);
    me->fetch_add(1, std::memory_order_relaxed);
    return me;
}
void ctx_pipelined_end(ceno_hip_ctx* ctx, PipelinedOwner& owner) {
    CENO_TIMED("ctx_pipelined_end");
    {
        std::lock_guard<std::mutex> g(ctx->gate_mu);
        ctx->pipelined_live.fetch_sub(1);
    }
    if (owner) {
        owner->fetch_sub(1, std::memory_order_relaxed);
        owner.reset();
    }
    ctx->gate_cv.notify_all();  // a thread waiting to trim (ctx_trim_begin_wait)
}
// take the trim gate as soon as no pipelined sumcheck is alive anywhere; only for threads that hold none themselves.  While it waits,
// threads with nothing in flight do not begin new pipelined sumchecks (trim_pending), so the wait is bounded by the sumchecks already
// alive — milliseconds.  false: timed out
static bool ctx_trim_begin_wait(ceno_hip_ctx* ctx, int timeout_ms) {
    std::unique_lock<std::mutex> lk(ctx->gate_mu);
    ctx->trim_pending++;
    const bool got = ctx->gate_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return !ctx->trimming && ctx->pipelined_live.load() == 0; });
    ctx->trim_pending--;
    if (got) ctx->trimming = true;
    lk.unlock();
    if (!got) ctx->gate_cv.notify_all();  // the beginners this waiter held back
    return got;
}
// how long an over-the-limit request waits for the sumchecks in flight to end (they take milliseconds; CENO_HIP_TRIM_WAIT_MS)
static int trim_wait_ms() {
    static const int v = getenv("CENO_HIP_TRIM_WAIT_MS") ? std::max(1, atoi(getenv("CENO_HIP_TRIM_WAIT_MS"))) : 2000;
    return v;
}
bool ctx_trim_begin(ceno_hip_ctx* ctx) {
    std::lock_guard<std::mutex> g(ctx->gate_mu);
    if (ctx->trimming || ctx->pipelined_live.load() > 0) return false;
    ctx->trimming = true;
    return true;
}
void ctx_trim_end(ceno_hip_ctx* ctx) {
    {
        std::lock_guard<std::mutex> g(ctx->gate_mu);
        ctx->trimming = false;
    }
    ctx->gate_cv.notify_all();
}

static int ctx_alloc_impl(ceno_hip_ctx* ctx, size_t bytes, void** out) {
    CENO_TIMED("ctx_alloc");
    size_t b = bucket_size(bytes);
    // hipFree waits for every stream of the device, and a lane's queued round kernels wait for a host that may be waiting
    // for this mutex: blocks that go back to the driver are only COLLECTED under the mutex and released after it is dropped
    std::vector<void*> victims;
    struct Release {
        ceno_hip_ctx* ctx;
        std::vector<void*>& v;
        bool gate = false;  // this call holds the trim gate
        ~Release() {
            for (void* p : v) (void)hipFree(p);
            if (gate) ctx_trim_end(ctx);
        }
    } release{ctx, victims};
    int attempt = 0;
    bool need_gate;
again:
    need_gate = false;
    {
        std::lock_guard<PoolMutex> g(ctx->mu);
        auto it = ctx->free_lists.find(b);
        if (it != ctx->free_lists.end() && !it->second.empty()) {
            // A cached block was last used on the stream in its tag, which may still have that work queued.  Same stream:
            // stream order protects it.  Another stream: take the block only if the old stream has drained (hipStreamQuery);
            // NEVER wait for it — a pipelined sumcheck keeps kernels queued that wait for the host, and a stream ordered behind
            // them would stall until the host answers, which it may be unable to do while it waits for THIS stream.  A
            // destroyed stream was synchronised on the way out (ceno_hip_stream_destroy); streams the library does not know
            // are the caller's to synchronise before freeing (include/ceno_hip.h, "Memory").
            const hipStream_t cur = ceno_tls_stream ? ceno_tls_stream : ctx->default_stream;
            auto& fl = it->second;
            int pick = -1;
            // first choice: a block this stream (or nobody alive) used last — no runtime call; concurrent lanes mostly recycle
            // their own blocks, and a hipStreamQuery per candidate under the pool mutex would serialise them
            // (its OWN blocks before untagged ones: a stream that took the untagged blocks first — the most recently freed ones — left
            // its own tagged pile idle and grew it by what it absorbed, while the streams those untagged blocks had come from went
            // back to the driver for new ones: the chip proof on the flow's stream against the opening's two tree streams, +70 MB of
            // cache and 55 hipMalloc per chip flow, tools/dev/pool_growth_chip.py)
            for (int k = (int)fl.size() - 1; k >= 0 && pick < 0; k--)
                if (fl[k].second == cur) pick = k;
            for (int k = (int)fl.size() - 1; k >= 0 && pick < 0; k--) {
                const hipStream_t last = fl[k].second;
                if (!last || !stream_alive(ctx, last)) pick = k;
            }
            // second choice: any block whose stream has drained — one query per DISTINCT stream (a handful), not per block, so
            // that blocks tagged with a lane that no longer asks for this size do not pile up behind busy ones
            if (pick < 0) {
                CENO_TIMED("ctx_alloc: second choice (stream queries under the lock)");
                hipStream_t seen[16];
                bool idle[16];
                int n_seen = 0;
                for (int k = (int)fl.size() - 1; k >= 0 && pick < 0; k--) {
                    const hipStream_t last = fl[k].second;
                    int j = 0;
                    while (j < n_seen && seen[j] != last) j++;
                    if (j == n_seen) {
                        if (n_seen == 16) break;
                        seen[n_seen] = last;
                        idle[n_seen] = stream_drained(last);
                        n_seen++;
                    }
                    if (idle[j]) pick = k;
                }
            }
            if (pick >= 0) {
                void* p = fl[pick].first;
                fl.erase(fl.begin() + pick);
                ctx->pool_cached -= b;
                ctx->pool_used += b;
                    ctx->pool_peak = std::max(ctx->pool_peak, ctx->pool_used);
                ctx->live[p] = b;
                *out = p;
                return 0;
            }
        }
        if (ctx->pool_limit && ctx->pool_used + ctx->pool_cached + b > ctx->pool_limit) {
            // over the limit with the cache counted in: blocks of the cache have to go back to the driver — possible only while no
            // pipelined sumcheck is alive (the trim gate) — or a LARGER idle block of the cache serves the request as it is
            if (release.gate || (release.gate = ctx_trim_begin(ctx))) {
                for (auto& kv : ctx->free_lists) {
                    for (auto& p : kv.second) {
                        victims.push_back(p.first);
                        ctx->pool_cached -= kv.first;
                    }
                    kv.second.clear();
                }
            } else {
                const hipStream_t cur = ceno_tls_stream ? ceno_tls_stream : ctx->default_stream;
                size_t best = 0;
                int best_k = -1;
                hipStream_t seen[16];
                bool idle[16];
                int n_seen = 0;
                for (auto& kv : ctx->free_lists) {
                    if (kv.first < b || (best && kv.first >= best)) continue;
                    for (int k = (int)kv.second.size() - 1; k >= 0; k--) {
                        const hipStream_t last = kv.second[k].second;
                        bool ok = !last || last == cur || !stream_alive(ctx, last);
                        if (!ok) {
                            int j = 0;
                            while (j < n_seen && seen[j] != last) j++;
                            if (j == n_seen && n_seen < 16) {
                                seen[n_seen] = last;
                                idle[n_seen] = stream_drained(last);
                                n_seen++;
                            }
                            ok = j < n_seen && idle[j];
                        }
                        if (ok) {
                            best = kv.first;
                            best_k = k;
                            break;
                        }
                    }
                }
                if (best_k >= 0) {
                    auto& fl = ctx->free_lists[best];
                    void* p = fl[best_k].first;
                    fl.erase(fl.begin() + best_k);
                    ctx->pool_cached -= best;
                    ctx->pool_used += best;
                    ctx->pool_peak = std::max(ctx->pool_peak, ctx->pool_used);
                    ctx->live[p] = best;
                    *out = p;
                    return 0;
                }
                need_gate = true;
            }
        }
        if (!need_gate && ctx->pool_limit && ctx->pool_used + ctx->pool_cached + b > ctx->pool_limit) {
            ctx->err = "pool capacity exceeded";
            return CENO_HIP_ERR_OOM;
        }
    }
    if (need_gate) {
        // lanes are proving and no idle block is large enough.  A thread with no pipelined sumcheck of its own waits for them (their
        // rounds end within milliseconds) and trims then; one that has such a sumcheck alive cannot — a retryable failure
        if (attempt == 0 && tls_pipelined() == 0 && ctx_trim_begin_wait(ctx, trim_wait_ms())) {
            release.gate = true;
            attempt = 1;
            goto again;
        }
        bool hopeless;
        {
            std::lock_guard<PoolMutex> g(ctx->mu);
            hopeless = ctx->pool_used + b > ctx->pool_limit;
        }
        return ctx_fail(ctx, CENO_HIP_ERR_OOM, hopeless ? "pool capacity exceeded" : "pool capacity exceeded while other lanes are proving (cached blocks cannot be returned now): retry");
    }
    void* p = nullptr;
    CENO_TIMED("ctx_alloc: driver path (hipMalloc)");
    ctx_make_current(ctx);
    {   // soft cap on the cache: blocks parked without a tag (their stream had drained) go back to the driver once the cache is
        // several times what is in use — a backstop against slow growth under many lanes, far below any real footprint.
        // hipFree waits for EVERY stream of the device: it is only called while no pipelined sumcheck is alive anywhere (queued
        // round kernels wait for their host thread; a host thread inside hipFree with such kernels pending, and a second one
        // likewise, would wait for each other until the kernels give up — seen as "round finished without publishing its
        // message" after a 13 GB batch had left the cache over the cap in front of a four-lane shard flow)
        const size_t floor_ = (size_t)2 << 30;
        bool over = false;
        {
            std::lock_guard<PoolMutex> g(ctx->mu);
            over = ctx->pool_cached > 4 * std::max(ctx->pool_used + b, floor_);
        }
        if (over && !release.gate) release.gate = ctx_trim_begin(ctx);  // (gate before pool mutex, everywhere)
        std::lock_guard<PoolMutex> g(ctx->mu);
        if (over && release.gate) {
            // LARGEST blocks first, only blocks of >= 1 MB, only until the cache is back under HALF the cap: what puts a cache over
            // a cap of >= 8 GB are the tables of a big batch, not the working set of the flows that follow it — returning every
            // idle block made those flows re-allocate theirs from the driver on every use (measured after a 13.6 GB batch: 800-2000
            // hipMalloc and ~1000-2000 hipFree'd blocks per flow, chip proofs 2x slower, for as long as the big blocks — still tagged
            // with their stream — kept the cache over the cap).  A block counts as idle when it carries no tag or its stream has
            // drained or is gone (one query per distinct stream).
            const size_t target = 2 * std::max(ctx->pool_used + b, floor_);
            hipStream_t seen[16];
            bool idle[16];
            int n_seen = 0;
            std::vector<size_t> keys;
            for (auto& kv : ctx->free_lists)
                if (kv.first >= ((size_t)1 << 20) && !kv.second.empty()) keys.push_back(kv.first);
            std::sort(keys.begin(), keys.end(), std::greater<size_t>());
            for (size_t key : keys) {
                if (ctx->pool_cached <= target) break;
                auto it = ctx->free_lists.find(key);
                auto& fl = it->second;
                for (size_t k = 0; k < fl.size() && ctx->pool_cached > target;) {
                    const hipStream_t tag = fl[k].second;
                    bool free_now = tag == nullptr || !stream_alive(ctx, tag);
                    if (!free_now) {
                        int j = 0;
                        while (j < n_seen && seen[j] != tag) j++;
                        if (j == n_seen && n_seen < 16) {
                            seen[n_seen] = tag;
                            idle[n_seen] = stream_drained(tag);
                            n_seen++;
                        }
                        free_now = j < n_seen && idle[j];
                    }
                    if (free_now) {
                        victims.push_back(fl[k].first);
                        ctx->pool_cached -= it->first;
                        fl[k] = fl.back();
                        fl.pop_back();
                    } else {
                        k++;
                    }
                }
            }
        }
    }
    static const bool pool_trace = getenv("CENO_HIP_POOL_TRACE") != nullptr;  // one line per driver call (debugging cache behaviour)
    if (pool_trace && !victims.empty()) fprintf(stderr, "[ceno_hip] pool: hipFree of %zu cached blocks (cached %zu MB, used %zu MB)\n", victims.size(), ctx->pool_cached >> 20, ctx->pool_used >> 20);
    for (void* v : victims) (void)hipFree(v);  // outside the mutex (see above)
    victims.clear();
    if (pool_trace) {
        size_t n_same = 0, n_busy = 0;
        {
            std::lock_guard<PoolMutex> g(ctx->mu);
            auto it = ctx->free_lists.find(b);
            if (it != ctx->free_lists.end()) {
                n_same = it->second.size();
                std::map<hipStream_t, int> tags;
                for (auto& e : it->second) {
                    tags[e.second]++;
                    if (e.second && stream_alive(ctx, e.second) && !stream_drained(e.second)) n_busy++;
                }
                for (auto& kv : tags) fprintf(stderr, "[ceno_hip] pool:   tag %p x %d (alive %d)\n", (void*)kv.first, kv.second, kv.first ? (int)stream_alive(ctx, kv.first) : -1);
            }
        }
        fprintf(stderr, "[ceno_hip] pool: hipMalloc %zu KB (cached %zu MB, used %zu MB; %zu cached blocks of this size, %zu of them on a busy stream; stream %p)\n", b >> 10,
                ctx->pool_cached >> 20, ctx->pool_used >> 20, n_same, n_busy, (void*)(ceno_tls_stream ? ceno_tls_stream : ctx->default_stream));
    }
    if (release.gate) {
        ctx_trim_end(ctx);
        release.gate = false;
    }
    hipError_t e = hipMalloc(&p, b);
    if (e != hipSuccess) {
        // drop the cache and retry once (a thread without a pipelined sumcheck of its own waits for the lanes' to end first:
        // ceno_hip_mem_trim returns at once while any is alive)
        if (tls_pipelined() == 0 && ctx_trim_begin_wait(ctx, trim_wait_ms())) ctx_trim_end(ctx);
        ceno_hip_mem_trim(ctx);
        e = hipMalloc(&p, b);
        if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_OOM, "hipMalloc(%zu) failed: %s", b, hipGetErrorString(e));
    }
    std::lock_guard<PoolMutex> g(ctx->mu);
    ctx->pool_used += b;
                    ctx->pool_peak = std::max(ctx->pool_peak, ctx->pool_used);
    ctx->live[p] = b;
    *out = p;
    return 0;
}

// CENO_HIP_POOL_POISON=1 (debugging): every block the pool hands out is filled with a non-canonical pattern on the calling thread's stream — a
// kernel that reads memory nobody wrote then fails the same way in every run instead of depending on what the block held before
int ctx_alloc(ceno_hip_ctx* ctx, size_t bytes, void** out) {
    const int rc = ctx_alloc_impl(ctx, bytes, out);
    static const bool poison = getenv("CENO_HIP_POOL_POISON") != nullptr;
    if (rc == 0 && poison) {
        hipStream_t st = ceno_tls_stream ? ceno_tls_stream : ctx->default_stream;
        (void)hipMemsetAsync(*out, 0xA5, bucket_size(bytes), st);
        (void)hipStreamSynchronize(st);
    }
    return rc;
}

void ctx_free(ceno_hip_ctx* ctx, void* p) { ctx_free_on(ctx, p, ceno_tls_stream ? ceno_tls_stream : ctx->default_stream); }

// `owner` = the stream that used the block last (an object that carries its own stream frees with it, whatever stream the
// calling thread happened to resolve last)
void ctx_free_on(ceno_hip_ctx* ctx, void* p, hipStream_t owner) {
    if (!p) return;
    ctx_free_many_on(ctx, &p, 1, owner);
}

// Tag = the stream that used the block last.  A LARGE block is worth one runtime call: if that stream has already
// drained, the block is free for everybody (no tag) — otherwise blocks freed after a synchronisation by a thread that
// alternates between streams (commit_traces, the opening) could only ever go back to the stream of the tag, and every run
// would allocate the other stream's share afresh (measured: +230 MB of cache per shard flow).
// all blocks of one handle: one lock, one stream query (a sumcheck handle frees ~a dozen blocks, and the query alone is 1-2 us)
void ctx_free_many_on(ceno_hip_ctx* ctx, void* const* ptrs, size_t n, hipStream_t owner, bool ask_drained) {
    CENO_TIMED("ctx_free_many_on");
    if (!owner) owner = ctx->default_stream;
    // the stream query is a call into the runtime (1-6 us, longer when other threads are in there too): it is made with the pool's lock
    // RELEASED — lanes that free a handle each would otherwise queue up behind each other's queries, and so would every allocation
    // (ask_drained = false: a sumcheck handle at its release — its stream has just run its last kernel, the blocks go back to the same lane
    // for the next layer, and a stream that needs them later asks then: ctx_alloc's second choice)
    int drained = -1;  // not asked
    if (ask_drained) {
        bool ask = false;
        {
            std::lock_guard<PoolMutex> g(ctx->mu);
            for (size_t i = 0; i < n && !ask; i++) {
                auto it = ptrs[i] ? ctx->live.find(ptrs[i]) : ctx->live.end();
                ask = it != ctx->live.end() && it->second >= ((size_t)64 << 10);
            }
            ask = ask && stream_alive(ctx, owner);
        }
        if (ask) drained = stream_drained(owner) ? 1 : 0;
    }
    std::lock_guard<PoolMutex> g(ctx->mu);
    for (size_t i = 0; i < n; i++) {
        void* p = ptrs[i];
        if (!p) continue;
        auto it = ctx->live.find(p);
        if (it == ctx->live.end()) continue;
        const size_t b = it->second;
        ctx->live.erase(it);
        ctx->pool_used -= b;
        ctx->pool_cached += b;
        hipStream_t tag = owner;
        if (b >= ((size_t)64 << 10) && drained == 1) tag = nullptr;
        ctx->free_lists[b].push_back({p, tag});
    }
}

static constexpr int VRAM_SLOTS = 1024;
void* ctx_vram_slot_alloc(ceno_hip_ctx* ctx) {
    CENO_TIMED("ctx_vram_slot_alloc");
    ctx_make_current(ctx);
    std::lock_guard<PoolMutex> g(ctx->mu);
    if (ctx->vram_state == 0) {
        ctx->vram_state = -1;
        const char* env = getenv("CENO_HIP_VRAM_MAILBOX");  // 0 keeps the mailboxes in pinned host memory (A/B measurements)
        int large_bar = 0;
        if ((!env || atoi(env) != 0) && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, ctx->device) == hipSuccess && large_bar) {
            void* p = nullptr;
            if (hipExtMallocWithFlags(&p, (size_t)VRAM_SLOTS * 64, hipDeviceMallocFinegrained) == hipSuccess && p) {
                ctx->vram_arena = (char*)p;
                for (int i = VRAM_SLOTS - 1; i >= 0; i--) ctx->vram_free_slots.push_back(i);
                ctx->vram_state = 1;
            }
        }
    }
    if (ctx->vram_state != 1 || ctx->vram_free_slots.empty()) return nullptr;
    const int i = ctx->vram_free_slots.back();
    ctx->vram_free_slots.pop_back();
    return ctx->vram_arena + (size_t)i * 64;
}
void ctx_vram_slot_free(ceno_hip_ctx* ctx, void* slot) {
    if (!slot) return;
    std::lock_guard<PoolMutex> g(ctx->mu);
    ctx->vram_free_slots.push_back((int)(((char*)slot - ctx->vram_arena) / 64));
}

int ctx_pinned_alloc(ceno_hip_ctx* ctx, size_t bytes, void** host, void** dev_view) {
    CENO_TIMED("ctx_pinned_alloc");
    size_t b = 4096;
    while (b < bytes) b <<= 1;
    void *h = nullptr, *d = nullptr;
    {
        // a recycled block comes with the device view it was given when it was made: no runtime call (hipSetDevice and
        // hipHostGetDevicePointer take the runtime's own locks — ~14 us per sumcheck with four lanes asking at once)
        std::lock_guard<PoolMutex> g(ctx->mu);
        auto it = ctx->pinned_free.find(b);
        if (it != ctx->pinned_free.end() && !it->second.empty()) {
            h = it->second.back();
            it->second.pop_back();
            d = ctx->pinned_dev[h];
            ctx->pinned_live[h] = b;
        }
    }
    if (!h) {
        ctx_make_current(ctx);
        hipError_t e = hipHostMalloc(&h, b, hipHostMallocDefault);
        if (e != hipSuccess) return ctx_fail(ctx, CENO_HIP_ERR_OOM, "hipHostMalloc(%zu): %s", b, hipGetErrorString(e));
        e = hipHostGetDevicePointer(&d, h, 0);
        if (e != hipSuccess) {
            (void)hipHostFree(h);
            return ctx_fail(ctx, CENO_HIP_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e));
        }
        std::lock_guard<PoolMutex> g(ctx->mu);
        ctx->pinned_live[h] = b;
        ctx->pinned_dev[h] = d;
    }
    *host = h;
    *dev_view = d;
    return 0;
}

void ctx_pinned_free(ceno_hip_ctx* ctx, void* host) {
    CENO_TIMED("ctx_pinned_free");
    if (!host) return;
    std::lock_guard<PoolMutex> g(ctx->mu);
    auto it = ctx->pinned_live.find(host);
    if (it == ctx->pinned_live.end()) return;
    ctx->pinned_free[it->second].push_back(host);
    ctx->pinned_live.erase(it);
}

extern "C" {

const char* ceno_hip_version(void) { return "ceno_hip 0.1 (gfx950)"; }

int ceno_hip_init(int device, size_t pool_bytes, ceno_hip_ctx** out) {
    if (!out) return ctx_fail(nullptr, CENO_HIP_ERR_INVALID, "out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return ctx_fail(nullptr, CENO_HIP_ERR_HIP, "no HIP device available (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return ctx_fail(nullptr, CENO_HIP_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess) return ctx_fail(nullptr, CENO_HIP_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    auto* ctx = new ceno_hip_ctx();
    ctx->device = device;
    ctx->pool_limit = pool_bytes;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->num_cus = prop.multiProcessorCount;
        // the XCD-private lookup counters of the witness kernels rely on gfx942 / gfx950 behaviour (workgroup-scope atomics of the workgroups
        // of one XCD meet in that XCD's L2, HW_REG_XCC_ID names the XCD): any other target takes device-scope atomics
        ctx->xcd_private_l2 = strncmp(prop.gcnArchName, "gfx942", 6) == 0 || strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    }
    e = hipStreamCreateWithFlags(&ctx->default_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return ctx_fail(nullptr, CENO_HIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return 0;
}

void ceno_hip_destroy(ceno_hip_ctx* ctx) {
    if (!ctx) return;
    host_timing_dump();
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : ctx->free_lists)
        for (auto& p : kv.second) (void)hipFree(p.first);
    for (auto& kv : ctx->live) (void)hipFree(kv.first);
    for (auto& kv : ctx->pinned_free)
        for (void* p : kv.second) (void)hipHostFree(p);
    for (auto& kv : ctx->pinned_live) (void)hipHostFree(kv.first);
    for (auto& ev : ctx->prof_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : ctx->prof_event_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ctx->poseidon_dev) (void)hipFree(ctx->poseidon_dev);
    merkle_drop_host_params(ctx);
    if (ctx->vram_arena) (void)hipFree(ctx->vram_arena);
    for (hipStream_t ls : ctx->lane_streams)
        if (ls) (void)hipStreamDestroy(ls);
    if (ctx->default_stream) (void)hipStreamDestroy(ctx->default_stream);
    delete ctx;
}

const char* ceno_hip_last_error(ceno_hip_ctx* ctx) {
    if (!ctx) return g_init_err.c_str();
    // a copy per calling thread: another lane may be writing its own failure into ctx->err right now
    thread_local std::string mine;
    {
        std::lock_guard<PoolMutex> g(ctx->mu);
        mine = ctx->err;
    }
    return mine.c_str();
}

int ceno_hip_make_current(ceno_hip_ctx* ctx) {
    CHECK_ARG(ctx, ctx, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return 0;
}
int ceno_hip_device(const ceno_hip_ctx* ctx) { return ctx ? ctx->device : -1; }

int ceno_hip_stream_create(ceno_hip_ctx* ctx, ceno_hip_stream* out) {
    CHECK_ARG(ctx, out, "out is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    {
        std::lock_guard<PoolMutex> g(ctx->mu);
        ctx->streams.push_back(s);
    }
    *out = (ceno_hip_stream)s;
    return 0;
}
int ceno_hip_stream_create_lane(ceno_hip_ctx* ctx, int lane, ceno_hip_stream* out) {
    // Lanes (reference: thread-bound streams of the chip scheduler, gkr_iop/src/gpu/mod.rs:87-154,
    // ceno_zkvm/src/scheme/scheduler.rs:73-85) must land on DIFFERENT hardware queues to overlap: streams that share a
    // queue run back to back and pay a cross-stream barrier per kernel.  HIP gives each priority level queues of its
    // own, so consecutive lanes rotate through the levels (highest, normal, lowest, highest, ...).
    CHECK_ARG(ctx, out && lane >= 0, "bad lane");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int least = 0, greatest = 0;
    HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int levels[3] = {greatest, (least + greatest) / 2, least};
    hipStream_t s;
    HIP_TRY(ctx, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, levels[lane % 3]));
    {
        std::lock_guard<PoolMutex> g(ctx->mu);
        ctx->streams.push_back(s);
    }
    *out = (ceno_hip_stream)s;
    return 0;
}
int ceno_hip_lane_stream(ceno_hip_ctx* ctx, int lane, ceno_hip_stream* out) {
    // Streams are long-lived objects: hipStreamCreateWithPriority costs ~4 ms and hipStreamDestroy ~2 ms (rocprofv3 --hip-trace),
    // the size of a whole chip proof.  The lanes of the chip scheduler therefore belong to the CONTEXT: created on first use,
    // handed out again on every later run, destroyed with the context.
    CHECK_ARG(ctx, out && lane >= 0 && lane < 64, "bad lane");
    {
        std::lock_guard<PoolMutex> g(ctx->mu);
        if (lane < (int)ctx->lane_streams.size() && ctx->lane_streams[lane]) {
            *out = (ceno_hip_stream)ctx->lane_streams[lane];
            return 0;
        }
    }
    ceno_hip_stream s = nullptr;
    TRY(ceno_hip_stream_create_lane(ctx, lane, &s));
    std::lock_guard<PoolMutex> g(ctx->mu);
    if ((int)ctx->lane_streams.size() <= lane) ctx->lane_streams.resize(lane + 1, nullptr);
    if (ctx->lane_streams[lane]) {  // another thread was faster: keep its stream (ours is dropped from the bookkeeping below)
        hipStream_t mine = (hipStream_t)s;
        for (size_t i = 0; i < ctx->streams.size(); i++)
            if (ctx->streams[i] == mine) {
                ctx->streams.erase(ctx->streams.begin() + i);
                break;
            }
        (void)hipStreamDestroy(mine);
    } else {
        ctx->lane_streams[lane] = (hipStream_t)s;
    }
    *out = (ceno_hip_stream)ctx->lane_streams[lane];
    return 0;
}
int ceno_hip_stream_bind(ceno_hip_ctx* ctx, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "ctx is NULL");
    (void)ctx_stream(ctx, s);
    return 0;
}
int ceno_hip_stream_adopt(ceno_hip_ctx* ctx, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx && s, "bad stream");
    std::lock_guard<PoolMutex> g(ctx->mu);
    if (!stream_alive(ctx, (hipStream_t)s)) ctx->streams.push_back((hipStream_t)s);
    return 0;
}
int ceno_hip_stream_destroy(ceno_hip_ctx* ctx, ceno_hip_stream s) {
    if (!s) return 0;
    // blocks freed while this stream still had work queued carry its tag in the pool: drain it before the tag goes stale
    (void)hipStreamSynchronize((hipStream_t)s);
    {
        std::lock_guard<PoolMutex> g(ctx->mu);
        for (size_t i = 0; i < ctx->streams.size(); i++)
            if (ctx->streams[i] == (hipStream_t)s) {
                ctx->streams.erase(ctx->streams.begin() + i);
                break;
            }
    }
    ctx->stream_gen.fetch_add(1, std::memory_order_release);  // every thread re-adopts a caller-made stream at this address
    if (ceno_tls_stream == (hipStream_t)s) ceno_tls_stream = nullptr;
    if (ceno_tls_adopted == (hipStream_t)s) ceno_tls_adopted = nullptr;
    HIP_TRY(ctx, hipStreamDestroy((hipStream_t)s));
    return 0;
}
int ceno_hip_stream_sync(ceno_hip_ctx* ctx, ceno_hip_stream s) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx_stream(ctx, s)));
    return 0;
}

int ceno_hip_mem_info(ceno_hip_ctx* ctx, size_t* free_bytes, size_t* total_bytes, size_t* pool_used, size_t* pool_cached) {
    size_t f = 0, t = 0;
    ctx_make_current(ctx);
    HIP_TRY(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    std::lock_guard<PoolMutex> g(ctx->mu);
    if (pool_used) *pool_used = ctx->pool_used;
    if (pool_cached) *pool_cached = ctx->pool_cached;
    return 0;
}

// ---- VRAM booking for a chip scheduler (reference: mem_pool try_book_capacity / unbook_capacity / get_booked_total /
// reset_booking / init_booking_baseline, ceno_zkvm/src/scheme/scheduler.rs:342-347,390,438,622-652): a task books its
// ESTIMATED footprint before it is handed to a lane and unbooks it when done; booking fails — nothing is allocated —
// when live allocations + bookings would exceed the capacity, which is how the scheduler back-fills smaller tasks.
int ceno_hip_mem_book(ceno_hip_ctx* ctx, size_t bytes) {
    std::lock_guard<PoolMutex> g(ctx->mu);
    if (ctx->pool_capacity == 0) {
        size_t f = 0, t = 0;
        ctx->pool_capacity = ctx->pool_limit ? ctx->pool_limit : (hipMemGetInfo(&f, &t) == hipSuccess ? t : 0);
    }
    if (ctx->pool_capacity && ctx->pool_used + ctx->pool_booked + bytes > ctx->pool_capacity) {
        ctx->err = "booking refused: live allocations plus bookings would exceed the capacity";
        return CENO_HIP_ERR_OOM;
    }
    ctx->pool_booked += bytes;
    if (ctx->pool_booked > ctx->pool_booked_peak) ctx->pool_booked_peak = ctx->pool_booked;
    return 0;
}
int ceno_hip_mem_unbook(ceno_hip_ctx* ctx, size_t bytes) {
    std::lock_guard<PoolMutex> g(ctx->mu);
    ctx->pool_booked = bytes > ctx->pool_booked ? 0 : ctx->pool_booked - bytes;
    return 0;
}
size_t ceno_hip_mem_booked(ceno_hip_ctx* ctx) {
    std::lock_guard<PoolMutex> g(ctx->mu);
    return ctx->pool_booked;
}
size_t ceno_hip_mem_booked_peak(ceno_hip_ctx* ctx, int reset) {
    if (!ctx) return 0;
    std::lock_guard<PoolMutex> g(ctx->mu);
    const size_t v = ctx->pool_booked_peak;
    if (reset) ctx->pool_booked_peak = ctx->pool_booked;
    return v;
}

int ceno_hip_debug_state(ceno_hip_ctx* ctx, int* pipelined_live, int* mid_units_in_flight) {
    CHECK_ARG(ctx, ctx, "ctx is NULL");
    if (pipelined_live) *pipelined_live = ctx->pipelined_live.load();
    if (mid_units_in_flight) *mid_units_in_flight = ctx->mid_wgs_in_flight.load();
    return 0;
}

size_t ceno_hip_mem_peak(ceno_hip_ctx* ctx, int reset) {
    if (!ctx) return 0;
    std::lock_guard<PoolMutex> g(ctx->mu);
    const size_t v = ctx->pool_peak;
    if (reset) ctx->pool_peak = ctx->pool_used;
    return v;
}

int ceno_hip_mem_trim(ceno_hip_ctx* ctx) {
    std::vector<void*> victims;
    if (!ctx_trim_begin(ctx)) return 0;  // lanes are proving: nothing can go back to the driver now (hipFree would wait for their round kernels)
    struct End {
        ceno_hip_ctx* c;
        ~End() { ctx_trim_end(c); }
    } end{ctx};
    {
        std::lock_guard<PoolMutex> g(ctx->mu);
        for (auto& kv : ctx->free_lists) {
            for (auto& p : kv.second) {
                victims.push_back(p.first);
                ctx->pool_cached -= kv.first;
            }
            kv.second.clear();
        }
    }
    for (void* p : victims) (void)hipFree(p);  // hipFree waits for the device: never under the pool mutex
    return 0;
}

// ------------------------------------------------------------------------------------------------
// MLE handles
// ------------------------------------------------------------------------------------------------
int ceno_hip_mle_alloc(ceno_hip_ctx* ctx, int num_vars, int is_ext, ceno_hip_mle** out) {
    CHECK_ARG(ctx, out && num_vars >= 0 && num_vars < 40, "bad num_vars %d", num_vars);
    auto* m = new ceno_hip_mle();
    m->num_vars = num_vars;
    m->is_ext = is_ext ? 1 : 0;
    m->owned = true;
    void* p = nullptr;
    int rc = ctx_alloc(ctx, m->bytes(), &p);
    if (rc) {
        delete m;
        return rc;
    }
    m->d = (uint64_t*)p;
    *out = m;
    return 0;
}

int ceno_hip_mle_upload(ceno_hip_ctx* ctx, const uint64_t* host, int num_vars, int is_ext, ceno_hip_stream s, ceno_hip_mle** out) {
    CHECK_ARG(ctx, host, "host is NULL");
    (void)ctx_stream(ctx, s);  // allocations below belong to work on `s`: bind the thread first (pool tags, include/ceno_hip.h "Memory")
    ceno_hip_mle* m = nullptr;
    TRY(ceno_hip_mle_alloc(ctx, num_vars, is_ext, &m));
    hipStream_t st = ctx_stream(ctx, s);
    hipError_t e = hipMemcpyAsync(m->d, host, m->bytes(), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // host buffer is only borrowed for the call
    if (e != hipSuccess) {
        ceno_hip_mle_free(ctx, m);
        return ctx_fail(ctx, CENO_HIP_ERR_HIP, "upload: %s", hipGetErrorString(e));
    }
    *out = m;
    return 0;
}

int ceno_hip_mle_wrap(ceno_hip_ctx* ctx, uint64_t* device_ptr, int num_vars, int is_ext, ceno_hip_mle** out) {
    CHECK_ARG(ctx, out && device_ptr && num_vars >= 0 && num_vars < 40, "bad wrap arguments");
    CHECK_ARG(ctx, ((uintptr_t)device_ptr & 15) == 0, "device pointer must be 16-byte aligned");
    auto* m = new ceno_hip_mle();
    m->d = device_ptr;
    m->num_vars = num_vars;
    m->is_ext = is_ext ? 1 : 0;
    m->owned = false;
    *out = m;
    return 0;
}

int ceno_hip_mle_view_chunk(ceno_hip_ctx* ctx, ceno_hip_mle* parent, int sub_vars, size_t chunk, ceno_hip_mle** out) {
    CHECK_ARG(ctx, parent && out, "NULL argument");
    CHECK_ARG(ctx, sub_vars >= 0 && sub_vars <= parent->num_vars, "sub_vars %d out of range", sub_vars);
    CHECK_ARG(ctx, chunk < ((size_t)1 << (parent->num_vars - sub_vars)), "chunk out of range");
    auto* m = new ceno_hip_mle();
    m->num_vars = sub_vars;
    m->is_ext = parent->is_ext;
    m->owned = false;
    m->d = parent->d + chunk * ((size_t)1 << sub_vars) * (parent->is_ext ? 2 : 1);
    *out = m;
    return 0;
}

int ceno_hip_mle_download(ceno_hip_ctx* ctx, const ceno_hip_mle* m, uint64_t* host, ceno_hip_stream s) {
    CHECK_ARG(ctx, m && host, "NULL argument");
    hipStream_t st = ctx_stream(ctx, s);
    HIP_TRY(ctx, hipMemcpyAsync(host, m->d, m->bytes(), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return 0;
}

int ceno_hip_mle_free(ceno_hip_ctx* ctx, ceno_hip_mle* m) {
    CENO_TIMED("mle_free");
    if (!m) return 0;
    if (m->owned) ctx_free(ctx, m->d);
    if (m->aux) ctx_free(ctx, m->aux);
    delete m;
    return 0;
}
int ceno_hip_mle_num_vars(const ceno_hip_mle* m) { return m ? m->num_vars : -1; }
int ceno_hip_mle_is_ext(const ceno_hip_mle* m) { return m ? m->is_ext : -1; }
uint64_t* ceno_hip_mle_device_ptr(const ceno_hip_mle* m) { return m ? m->d : nullptr; }

// ------------------------------------------------------------------------------------------------
// profiling hooks: HIP events on the stream the kernel is launched on
// ------------------------------------------------------------------------------------------------
int ceno_hip_prof_enable(ceno_hip_ctx* ctx, int on) {
    ctx->prof_on = on != 0;
    ctx->prof_pipelined = on == 2;
    return 0;
}
int ceno_hip_prof_reset(ceno_hip_ctx* ctx) {
    for (auto& ev : ctx->prof_events) ctx->prof_event_pool.push_back(ev);
    ctx->prof_events.clear();
    ctx->prof_launches = 0;
    ctx->prof_bytes = 0.0;
    return 0;
}
int ceno_hip_prof_get(ceno_hip_ctx* ctx, double* kernel_ms, uint64_t* launches, double* algorithmic_bytes) {
    double tot = 0.0;
    for (auto& ev : ctx->prof_events) {
        HIP_TRY(ctx, hipEventSynchronize(ev.second));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
    }
    if (kernel_ms) *kernel_ms = tot;
    if (launches) *launches = ctx->prof_launches;
    if (algorithmic_bytes) *algorithmic_bytes = ctx->prof_bytes;
    return 0;
}

}  // extern "C"

void prof_begin(ceno_hip_ctx* ctx, hipStream_t st) {
    if (!ctx->prof_on) return;
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!ctx->prof_event_pool.empty()) {
        ev = ctx->prof_event_pool.back();
        ctx->prof_event_pool.pop_back();
    } else {
        (void)hipEventCreate(&ev.first);
        (void)hipEventCreate(&ev.second);
    }
    (void)hipEventRecord(ev.first, st);
    ctx->prof_events.push_back(ev);
}
void prof_end(ceno_hip_ctx* ctx, hipStream_t st, double algorithmic_bytes, int launches) {
    if (!ctx->prof_on || ctx->prof_events.empty()) return;
    (void)hipEventRecord(ctx->prof_events.back().second, st);
    ctx->prof_launches += launches;
    ctx->prof_bytes += algorithmic_bytes;
}
// a launch inside an open event pair (a span over several back-to-back launches)
void prof_count(ceno_hip_ctx* ctx, double algorithmic_bytes) {
    if (!ctx->prof_on) return;
    ctx->prof_launches += 1;
    ctx->prof_bytes += algorithmic_bytes;
}
