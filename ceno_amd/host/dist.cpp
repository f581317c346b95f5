// Hypercube-sharded sumcheck across the GPUs of one node, driven from C++.
//
// the reference has no intra-sumcheck distribution ("Distributed Sumcheck — TODO",
// docs/src/optimizations.md:3-5) — this is new design.  LSB-first binding pairs adjacent indices, so
// splitting every table by its TOP log2(world) index bits keeps every fold local for the first
// n_local = n - log2(world) rounds.  One process per GPU (launched by torch.distributed, which also carries
// the bootstrap of the communicator).  Per local round every rank produces d partial evaluations, the ranks
// exchange them, each host adds them mod p (no collective library has a mod-p reduction; ncclSum on uint64 would
// wrap mod 2^64) and the replicated transcript produces the challenge.  Two transports for that exchange:
//   * host shared memory (ceno_dist_comm_attach_shm, the default of bench.py): the partials are in host memory
//     anyway, a POSIX segment plus sequence words costs ~1-2 us; every rank runs its shard as an ordinary pipelined
//     sumcheck, the k final values are exchanged the same way and the last log2(world) rounds run on the host;
//   * RCCL (ceno_dist_comm_init): ONE ncclAllGather of world*d extension elements per round on the kernels' HIP
//     stream, and once a shard is down to 2^12 elements per table the shards themselves are all-gathered and every
//     rank finishes replicated.  RCCL is resolved with dlopen so that libceno_prover.so loads on machines without it.
// The mixed-size batched variant (ceno_dist_batched_sumcheck_prove) shards every size class along its own top bits.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ceno_prover.h"
#include "../csrc/gl64.hpp"

using gl::E2;

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // point-to-point (the row re-sharding of the commitment is an all-to-all of unequal blocks)
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};
Rccl g_rccl;
thread_local std::string g_dist_err;

int load_rccl() {
    if (g_rccl.h) return 0;
    // CENO_RCCL_PATH first: the launcher points it at the RCCL the process already uses (torch bundles its own copy;
    // two different RCCL builds in one process would each run their own bootstrap and transport setup)
    const char* names[] = {getenv("CENO_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) {
        g_dist_err = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?");
        return CENO_HIP_ERR_UNSUPPORTED;
    }
    *(void**)&g_rccl.GetUniqueId = dlsym(g_rccl.h, "ncclGetUniqueId");
    *(void**)&g_rccl.CommInitRank = dlsym(g_rccl.h, "ncclCommInitRank");
    *(void**)&g_rccl.AllGather = dlsym(g_rccl.h, "ncclAllGather");
    *(void**)&g_rccl.CommDestroy = dlsym(g_rccl.h, "ncclCommDestroy");
    *(void**)&g_rccl.GetErrorString = dlsym(g_rccl.h, "ncclGetErrorString");
    *(void**)&g_rccl.Send = dlsym(g_rccl.h, "ncclSend");
    *(void**)&g_rccl.Recv = dlsym(g_rccl.h, "ncclRecv");
    *(void**)&g_rccl.GroupStart = dlsym(g_rccl.h, "ncclGroupStart");
    *(void**)&g_rccl.GroupEnd = dlsym(g_rccl.h, "ncclGroupEnd");
    *(void**)&g_rccl.CommCount = dlsym(g_rccl.h, "ncclCommCount");
    *(void**)&g_rccl.CommUserRank = dlsym(g_rccl.h, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
        g_dist_err = "RCCL symbols missing";
        return CENO_HIP_ERR_UNSUPPORTED;
    }
    return 0;
}

int nccl_fail(ncclResult_t r, const char* what) {
    g_dist_err = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
    return CENO_HIP_ERR_HIP;
}

void tr_usize(ceno_transcript* t, uint64_t v) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
    t->append_label(t->self, b, 8);
}
E2 absorb_round(ceno_transcript* tr, const uint64_t* msg, int d) {
    for (int t = 0; t < d; t++) tr->append_ext(tr->self, msg + 2 * t);
    static const char lbl[] = "Internal round";
    tr->append_label(tr->self, (const uint8_t*)lbl, sizeof(lbl) - 1);
    uint64_t o[2];
    tr->sample_ext(tr->self, o);
    return E2{o[0], o[1]};
}

}  // namespace

extern "C" {

const char* ceno_dist_last_error(void) { return g_dist_err.c_str(); }

// Host shared-memory exchange between the ranks of one node.  The round messages are already in host memory (the
// transcript lives there), so the per-round "collective" of 16*d bytes per rank is cheapest as a store into a
// shared segment plus a spin on the peers' sequence words (~1-2 us) — an RCCL all-gather of the same 48 bytes costs
// a kernel launch on every rank plus a device-to-host copy (tens of us).  RCCL keeps the bulk transfers.
struct ShmRank {
    volatile uint64_t seq;             // last exchange this rank has published
    uint64_t pad[7];                   // one cache line per flag
    uint64_t slot[2][2 * 64];          // payload of exchanges seq (parity indexed), up to 64 ext
};
struct ShmSeg {
    uint64_t magic;
    uint64_t world;
    uint64_t bytes;                    // size of the whole segment as its creator laid it out (checked on attach, before anything past the header is touched)
    volatile uint64_t abort;           // != 0: a rank gave up (its own error, a time-out): every waiting peer returns at once
    uint64_t pad[4];
    ShmRank ranks[1];                  // `world` entries
};

// In-process group: `world` virtual ranks = threads of ONE process on one device, each with its own stream.  The bulk
// exchange is a device-to-device copy out of the peers' published send buffers between two barriers; small payloads go
// through host memory.  It exists so that the multi-rank commit path runs — bit for bit — on a single-GPU box (tests), and
// serves a single-process deployment; across processes the transports are RCCL (bulk) and the shared segment (messages).
struct ceno_dist_local_group {
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool broken = false;
    std::vector<const uint64_t*> send_base;       // per rank: packed send buffer (device)
    std::vector<std::vector<size_t>> send_off;    // per rank, per destination: offset in words
    std::vector<uint64_t> small;                  // per rank: up to 128 words
    bool barrier() {
        std::unique_lock<std::mutex> g(mu);
        if (broken) return false;
        const uint64_t my = gen;
        if (++arrived == world) {
            arrived = 0;
            gen++;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(g, std::chrono::seconds(120), [&] { return gen != my || broken; })) broken = true;  // a peer died
        if (broken) cv.notify_all();
        return !broken;
    }
};

struct ceno_dist_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    uint64_t* d_send = nullptr;  // device staging: up to 64 ext per rank
    uint64_t* d_recv = nullptr;
    uint64_t* h_recv = nullptr;  // pinned (RCCL path) or plain host memory (shm-only communicator)
    ShmSeg* shm = nullptr;
    size_t shm_bytes = 0;
    uint64_t shm_seq = 0;        // exchanges done so far (identical on every rank)
    bool h_recv_plain = false;
    struct ceno_dist_local_group* local = nullptr;  // in-process group of virtual ranks (threads sharing one device)
    int shares_device = -1;      // -1: not asked yet; 1: at least two ranks sit on one device (dist_comm_ranks_share_device)
    // what this rank put on the wire since the last reset (ceno_dist_comm_stats): [0] message exchanges (per-round partial sums, gathered
    // evaluations, digests: host words), [1] their bytes sent by this rank, [2] bulk exchanges (device buffers: codeword re-shard, gathered
    // tables), [3] their bytes sent by this rank
    uint64_t stats[4] = {0, 0, 0, 0};
};

int ceno_dist_unique_id(uint8_t* out128) {
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    memcpy(out128, &id, 128);
    return 0;
}

int ceno_dist_comm_init(int world, int rank, const uint8_t* id128, ceno_dist_comm** out) {
    int rc = load_rccl();
    if (rc) return rc;
    auto* c = new ceno_dist_comm();
    c->world = world;
    c->rank = rank;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return nccl_fail(r, "ncclCommInitRank");
    }
    const size_t per_rank = 64 * 2;  // words
    if (hipMalloc((void**)&c->d_send, per_rank * 8) != hipSuccess || hipMalloc((void**)&c->d_recv, per_rank * 8 * world) != hipSuccess ||
        hipHostMalloc((void**)&c->h_recv, per_rank * 8 * world, hipHostMallocDefault) != hipSuccess) {
        g_dist_err = "allocation of collective staging failed";
        return CENO_HIP_ERR_OOM;
    }
    *out = c;
    return 0;
}

// what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank): > 0 = number of ranks of the RCCL communicator,
// 0 = the communicator has no RCCL part (shared-memory or in-process exchange only), < 0 = error
int ceno_dist_comm_rccl_ranks(ceno_dist_comm* c, int* out_user_rank) {
    if (!c) return CENO_HIP_ERR_INVALID;
    if (!c->comm) return 0;
    if (!g_rccl.CommCount) {
        g_dist_err = "ncclCommCount is not available in the loaded RCCL";
        return CENO_HIP_ERR_UNSUPPORTED;
    }
    int n = 0, r = -1;
    ncclResult_t e = g_rccl.CommCount(c->comm, &n);
    if (e != ncclSuccess) return nccl_fail(e, "ncclCommCount");
    if (out_user_rank) {
        if (g_rccl.CommUserRank && g_rccl.CommUserRank(c->comm, &r) != ncclSuccess) r = -1;
        *out_user_rank = r;
    }
    return n;
}

void ceno_dist_comm_destroy(ceno_dist_comm* c) {
    if (!c) return;
    if (c->comm) g_rccl.CommDestroy(c->comm);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->h_recv_plain) free(c->h_recv);
    else if (c->h_recv) (void)hipHostFree(c->h_recv);
    if (c->shm) munmap((void*)c->shm, c->shm_bytes);
    delete c;
}

// ... followed by a BULK area: per rank two parity-indexed buffers of SHM_BULK_WORDS words, for the KB-sized gathers of the sharded GKR half and
// opening (tower tops: 64 KB per limb; folded tables; query answers) — the 128-word slots made such a gather hundreds of exchanges
static constexpr size_t SHM_BULK_WORDS = 16384;  // 128 KB per buffer
static size_t shm_head_size(int world) { return (sizeof(ShmSeg) + sizeof(ShmRank) * (size_t)(world > 0 ? world - 1 : 0) + 63) & ~(size_t)63; }
static size_t shm_size(int world) { return shm_head_size(world) + (size_t)world * 2 * SHM_BULK_WORDS * 8; }
static uint64_t* shm_bulk(ShmSeg* seg, int world, int rank, int parity) {
    return reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(seg) + shm_head_size(world)) + ((size_t)rank * 2 + (size_t)parity) * SHM_BULK_WORDS;
}
static const uint64_t SHM_MAGIC = 0x43454e4f53484d32ULL;  // "CENOSHM2" (2: size and abort word in the header, bulk area behind the ranks)

// Waiting for a peer: pause-spin, the clock every 1024 spins, the CPU given up every 4096 (more ranks than cores must not starve the rank that is
// waited for).  Bounded by WALL CLOCK (CENO_DIST_SHM_TIMEOUT_S, default 60 s), not by an iteration count; a rank that gives up — here or through
// dist_fail / ceno_dist_comm_abort — raises the segment's abort word, and every peer that waits returns at once instead of after its own time-out.
static double shm_timeout_s() {
    static const double v = [] {
        const char* e = getenv("CENO_DIST_SHM_TIMEOUT_S");
        const double x = e ? atof(e) : 60.0;
        return x > 0 ? x : 60.0;
    }();
    return v;
}
static thread_local ceno_dist_comm* tls_shm_comm = nullptr;  // the communicator this thread exchanged through last (dist_fail raises its abort word)
static int shm_wait(ceno_dist_comm* c, int g, uint64_t seq, const char* what) {
    volatile uint64_t* flag = &c->shm->ranks[g].seq;
    uint64_t spins = 0;
    timespec t0{0, 0};
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < seq) {
        if ((++spins & 1023) == 0) {
            if (c->shm->abort) {
                g_dist_err = std::string(what) + ": another rank gave up (abort word raised)";
                return CENO_HIP_ERR_STATE;
            }
            timespec now;
            clock_gettime(CLOCK_MONOTONIC, &now);
            if (t0.tv_sec == 0 && t0.tv_nsec == 0) t0 = now;
            if ((double)(now.tv_sec - t0.tv_sec) + 1e-9 * (double)(now.tv_nsec - t0.tv_nsec) > shm_timeout_s()) {
                c->shm->abort = 1;
                g_dist_err = std::string(what) + ": rank " + std::to_string(g) + " never published (timed out, CENO_DIST_SHM_TIMEOUT_S)";
                return CENO_HIP_ERR_STATE;
            }
            if ((spins & 4095) == 0) sched_yield();
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return 0;
}

/* Attach the host shared-memory exchange to a communicator (`c` may come from ceno_dist_comm_init, or be created
 * here without RCCL when *c is NULL).  Rank 0 passes create != 0 and must attach BEFORE the name is given to the
 * other ranks; it may shm_unlink the name once every rank has attached (ceno_dist_shm_unlink). */
int ceno_dist_comm_attach_shm(ceno_dist_comm** pc, int world, int rank, const char* name, int create) {
    if (!pc || !name || world < 1 || world > 64 || rank < 0 || rank >= world) {
        g_dist_err = "attach_shm: bad arguments";
        return CENO_HIP_ERR_INVALID;
    }
    ceno_dist_comm* c = *pc;
    if (c && (c->world != world || c->rank != rank)) {
        g_dist_err = "attach_shm: geometry differs from the communicator's";
        return CENO_HIP_ERR_INVALID;
    }
    const size_t bytes = shm_size(world);
    int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) {
        g_dist_err = std::string("shm_open(") + name + ") failed";
        return CENO_HIP_ERR_STATE;
    }
    if (create && ftruncate(fd, (off_t)bytes) != 0) {
        close(fd);
        g_dist_err = "ftruncate on the shared segment failed";
        return CENO_HIP_ERR_STATE;
    }
    if (!create) {  // a segment of another layout (an older library) is shorter: check the file's size before mapping `bytes` of it (SIGBUS otherwise)
        struct stat sb;
        if (fstat(fd, &sb) != 0 || (size_t)sb.st_size != bytes) {
            close(fd);
            g_dist_err = "shared segment has another size than this library lays out for this world size (mixed versions?)";
            return CENO_HIP_ERR_STATE;
        }
    }
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        g_dist_err = "mmap of the shared segment failed";
        return CENO_HIP_ERR_STATE;
    }
    ShmSeg* seg = (ShmSeg*)m;
    if (create) {
        memset(m, 0, bytes);
        seg->world = (uint64_t)world;
        seg->bytes = (uint64_t)bytes;
        __atomic_store_n(&seg->magic, SHM_MAGIC, __ATOMIC_RELEASE);
    } else if (__atomic_load_n(&seg->magic, __ATOMIC_ACQUIRE) != SHM_MAGIC || seg->world != (uint64_t)world || seg->bytes != (uint64_t)bytes) {
        munmap(m, bytes);
        g_dist_err = "shared segment is not initialised for this world size";
        return CENO_HIP_ERR_STATE;
    }
    if (!c) {
        c = new ceno_dist_comm();
        c->world = world;
        c->rank = rank;
        c->h_recv = (uint64_t*)calloc((size_t)64 * 2 * world, 8);
        c->h_recv_plain = true;
        *pc = c;
    }
    c->shm = seg;
    c->shm_bytes = bytes;
    c->shm_seq = 0;
    return 0;
}
int ceno_dist_comm_stats(ceno_dist_comm* c, uint64_t* out4, int reset) {
    if (!c || !out4) return CENO_HIP_ERR_INVALID;
    memcpy(out4, c->stats, sizeof c->stats);
    if (reset) memset(c->stats, 0, sizeof c->stats);
    return 0;
}
int ceno_dist_comm_abort(ceno_dist_comm* c) {
    if (!c) return CENO_HIP_ERR_INVALID;
    if (c->shm) c->shm->abort = 1;
    return 0;
}
int ceno_dist_shm_unlink(const char* name) { return name && shm_unlink(name) == 0 ? 0 : CENO_HIP_ERR_STATE; }

// all-gather `n_ext` extension elements per rank through the shared segment; result (world x n_ext) in c->h_recv
static int shm_gather_ext(ceno_dist_comm* c, const uint64_t* mine, int n_ext) {
    c->stats[0]++;
    c->stats[1] += (uint64_t)n_ext * 16;
    if (n_ext > 64) {
        g_dist_err = "shm_gather_ext: more than 64 elements per rank";
        return CENO_HIP_ERR_INVALID;
    }
    const uint64_t seq = ++c->shm_seq;
    ShmRank& me = c->shm->ranks[c->rank];
    memcpy((void*)me.slot[seq & 1], mine, (size_t)n_ext * 16);
    __atomic_store_n(&me.seq, seq, __ATOMIC_RELEASE);
    tls_shm_comm = c;
    for (int g = 0; g < c->world; g++) {
        ShmRank& r = c->shm->ranks[g];
        if (int rc = shm_wait(c, g, seq, "shm_gather_ext")) return rc;
        memcpy(c->h_recv + (size_t)g * n_ext * 2, (const void*)r.slot[seq & 1], (size_t)n_ext * 16);
    }
    return 0;
}

// all-gather of up to SHM_BULK_WORDS words per rank through the bulk area: out[g * n_words ..] = rank g's words
static int shm_gather_bulk(ceno_dist_comm* c, const uint64_t* mine, size_t n_words, uint64_t* out, size_t out_stride_words) {
    if (n_words > SHM_BULK_WORDS) {
        g_dist_err = "shm_gather_bulk: block too large";
        return CENO_HIP_ERR_INVALID;
    }
    const uint64_t seq = ++c->shm_seq;
    ShmRank& me = c->shm->ranks[c->rank];
    memcpy(shm_bulk(c->shm, c->world, c->rank, (int)(seq & 1)), mine, n_words * 8);
    __atomic_store_n(&me.seq, seq, __ATOMIC_RELEASE);
    tls_shm_comm = c;
    for (int g = 0; g < c->world; g++) {
        if (int rc = shm_wait(c, g, seq, "shm_gather_bulk")) return rc;
        memcpy(out + (size_t)g * out_stride_words, shm_bulk(c->shm, c->world, g, (int)(seq & 1)), n_words * 8);
    }
    return 0;
}

/* self-test of the exchange (CPU tests, no GPU involved): `iters` gathers of varying size; every rank checks every
 * payload word.  Returns 0 when all match. */
int ceno_dist_shm_selftest(ceno_dist_comm* c, int iters) {
    if (!c || !c->shm) return CENO_HIP_ERR_INVALID;
    uint64_t buf[128];
    for (int it = 0; it < iters; it++) {
        const int n = 1 + (it * 7) % 64;
        for (int k = 0; k < 2 * n; k++) buf[k] = gl::splitmix64_at(1000 + (uint64_t)c->rank, (uint64_t)it * 131 + k);
        int rc = shm_gather_ext(c, buf, n);
        if (rc) return rc;
        for (int g = 0; g < c->world; g++)
            for (int k = 0; k < 2 * n; k++)
                if (c->h_recv[(size_t)g * n * 2 + k] != gl::splitmix64_at(1000 + (uint64_t)g, (uint64_t)it * 131 + k)) {
                    g_dist_err = "shm selftest: payload mismatch";
                    return CENO_HIP_ERR_STATE;
                }
        if (it % 64 == 0) {  // ... and a bulk gather now and then (same sequence words, the big buffers)
            const size_t nb = 1 + (size_t)(it * 37) % SHM_BULK_WORDS;
            std::vector<uint64_t> mine(nb), all(nb * (size_t)c->world);
            for (size_t k = 0; k < nb; k++) mine[k] = gl::splitmix64_at(2000 + (uint64_t)c->rank, (uint64_t)it * 977 + k);
            rc = shm_gather_bulk(c, mine.data(), nb, all.data(), nb);
            if (rc) return rc;
            for (int g = 0; g < c->world; g++)
                for (size_t k = 0; k < nb; k += 97)
                    if (all[(size_t)g * nb + k] != gl::splitmix64_at(2000 + (uint64_t)g, (uint64_t)it * 977 + k)) {
                        g_dist_err = "shm selftest: bulk payload mismatch";
                        return CENO_HIP_ERR_STATE;
                    }
        }
    }
    return 0;
}

// all-gather `n_ext` extension elements per rank from device buffer c->d_send; result (world x n_ext) in c->h_recv
static int gather_ext(ceno_dist_comm* c, int n_ext, hipStream_t st) {
    c->stats[0]++;
    c->stats[1] += (uint64_t)n_ext * 16;
    if (n_ext > 64) {
        g_dist_err = "gather_ext: more than 64 elements per rank";
        return CENO_HIP_ERR_INVALID;
    }
    ncclResult_t r = g_rccl.AllGather(c->d_send, c->d_recv, (size_t)n_ext * 2, ncclUint64, c->comm, st);
    if (r != ncclSuccess) return nccl_fail(r, "ncclAllGather");
    if (hipMemcpyAsync(c->h_recv, c->d_recv, (size_t)n_ext * 16 * c->world, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        g_dist_err = "gather_ext: copy/sync failed";
        return CENO_HIP_ERR_HIP;
    }
    return 0;
}

// Replicated last log2(world) rounds on world-sized tables built from the per-rank final values (index = rank = the top
// bits).  The tables hold `world` <= 64 elements each: every rank evaluates these rounds directly on the host (a few
// hundred field multiplications) instead of paying a device sumcheck set-up for them; CENO_DIST_GPU_TAIL=1 keeps the
// device path for comparison.  Terms of a common-factor group list only their residual factors (include/ceno_hip.h),
// so the group's common factors are multiplied back in here.
static int dist_tail_host(ceno_dist_comm* c, const ceno_hip_sumcheck_plan* plan, int n_local, int log_w, int d, int k, ceno_transcript* tr,
                          uint64_t* ch, uint64_t* out_msgs, uint64_t* out_challenges, uint64_t* out_final_evals) {
    const int world = c->world;
    std::vector<std::vector<E2>> tab(k, std::vector<E2>(world));
    for (int j = 0; j < k; j++)
        for (int g = 0; g < world; g++) tab[j][g] = E2{c->h_recv[2 * ((size_t)g * k + j)], c->h_recv[2 * ((size_t)g * k + j) + 1]};
    // full factor list per term: residual factors + the common factors of its group
    std::vector<std::vector<uint32_t>> factors(plan->num_terms);
    for (int t = 0; t < plan->num_terms; t++)
        for (uint32_t q = plan->term_offsets[t]; q < plan->term_offsets[t + 1]; q++) factors[t].push_back(plan->term_mle_idx[q]);
    for (int g = 0; g < plan->num_groups; g++)
        for (uint32_t q = plan->group_term_offsets[g]; q < plan->group_term_offsets[g + 1]; q++)
            for (uint32_t cc = plan->common_offsets[g]; cc < plan->common_offsets[g + 1]; cc++)
                factors[plan->group_term_idx[q]].push_back(plan->common_mle_idx[cc]);
    std::vector<uint64_t> msg(2 * (size_t)d);
    size_t len = (size_t)world;
    for (int r = 0; r < log_w; r++) {
        const size_t pairs = len / 2;
        std::vector<E2> acc(d, gl::e2_zero());
        for (int t = 0; t < plan->num_terms; t++) {
            const E2 coeff{plan->term_coeffs[2 * t], plan->term_coeffs[2 * t + 1]};
            for (size_t p = 0; p < pairs; p++) {
                std::vector<E2> pr(d, coeff);
                for (uint32_t j : factors[t]) {
                    const E2 lo = tab[j][2 * p], hi = tab[j][2 * p + 1], delta = hi - lo;
                    E2 x = hi;
                    for (int e = 0; e < d; e++) {
                        pr[e] = pr[e] * x;
                        x = x + delta;
                    }
                }
                for (int e = 0; e < d; e++) acc[e] = acc[e] + pr[e];
            }
        }
        for (int e = 0; e < d; e++) {
            msg[2 * e] = acc[e].c0;
            msg[2 * e + 1] = acc[e].c1;
        }
        memcpy(out_msgs + (size_t)2 * d * (n_local + r), msg.data(), (size_t)16 * d);
        const E2 rr = absorb_round(tr, msg.data(), d);
        ch[0] = rr.c0;
        ch[1] = rr.c1;
        out_challenges[2 * (n_local + r)] = rr.c0;
        out_challenges[2 * (n_local + r) + 1] = rr.c1;
        for (int j = 0; j < k; j++)
            for (size_t p = 0; p < pairs; p++) tab[j][p] = tab[j][2 * p] + rr * (tab[j][2 * p + 1] - tab[j][2 * p]);
        len = pairs;
    }
    for (int j = 0; j < k; j++) {
        out_final_evals[2 * j] = tab[j][0].c0;
        out_final_evals[2 * j + 1] = tab[j][0].c1;
    }
    return 0;
}

static int dist_tail_device(ceno_hip_ctx* ctx, ceno_dist_comm* c, const ceno_hip_sumcheck_plan* plan_local, int n_local, int log_w, int d, int k,
                            ceno_transcript* tr, ceno_hip_stream s, uint64_t* ch, uint64_t* out_msgs, uint64_t* out_challenges,
                            uint64_t* out_final_evals) {
    const int world = c->world;
    std::vector<ceno_hip_mle*> tail(k, nullptr);
    std::vector<uint64_t> tab(2 * (size_t)world), msg(2 * (size_t)d);
    int rc = 0;
    for (int j = 0; j < k && !rc; j++) {
        for (int g = 0; g < world; g++) {
            tab[2 * g] = c->h_recv[2 * ((size_t)g * k + j)];
            tab[2 * g + 1] = c->h_recv[2 * ((size_t)g * k + j) + 1];
        }
        rc = ceno_hip_mle_upload(ctx, tab.data(), log_w, 1, s, &tail[j]);
    }
    if (!rc) {
        ceno_hip_sumcheck_plan tp = *plan_local;
        tp.max_num_vars = log_w;
        ceno_hip_sumcheck* ts = nullptr;
        rc = ceno_hip_sumcheck_begin(ctx, tail.data(), &tp, s, &ts);
        if (!rc) ceno_hip_sumcheck_set_pipelined(ctx, ts, 1);
        for (int r = 0; r < log_w && !rc; r++) {
            rc = ceno_hip_sumcheck_round(ctx, ts, r == 0 ? nullptr : ch, msg.data());
            if (rc) break;
            memcpy(out_msgs + (size_t)2 * d * (n_local + r), msg.data(), (size_t)16 * d);
            E2 rr = absorb_round(tr, msg.data(), d);
            ch[0] = rr.c0;
            ch[1] = rr.c1;
            out_challenges[2 * (n_local + r)] = rr.c0;
            out_challenges[2 * (n_local + r) + 1] = rr.c1;
        }
        if (!rc) rc = ceno_hip_sumcheck_finish(ctx, ts, ch, out_final_evals);
        if (ts) ceno_hip_sumcheck_free(ctx, ts);
    }
    for (auto* m : tail)
        if (m) ceno_hip_mle_free(ctx, m);
    if (rc) g_dist_err = ceno_hip_last_error(ctx);
    return rc;
}

static int dist_tail(ceno_hip_ctx* ctx, ceno_dist_comm* c, const ceno_hip_sumcheck_plan* plan_local, int n_local, int log_w, int d, int k,
                     ceno_transcript* tr, ceno_hip_stream s, uint64_t* ch, uint64_t* out_msgs, uint64_t* out_challenges,
                     uint64_t* out_final_evals) {
    if (getenv("CENO_DIST_GPU_TAIL"))
        return dist_tail_device(ctx, c, plan_local, n_local, log_w, d, k, tr, s, ch, out_msgs, out_challenges, out_final_evals);
    return dist_tail_host(c, plan_local, n_local, log_w, d, k, tr, ch, out_msgs, out_challenges, out_final_evals);
}

// Sharded rounds with the host shared-memory exchange: every rank runs the ordinary PIPELINED single-device round
// loop over its shard (message to pinned host memory, challenge through the mailbox) and the only addition per round
// is the ~1-2 us exchange of d ext partials between the host processes.  No device collective on the round path.
extern "C++" bool dist_comm_ranks_share_device(ceno_dist_comm* c, hipStream_t st);  // below: from the ranks' host names + PCI bus ids
// Rounds queued ahead of their challenges (pipelined) only where every rank has a device of its own.  Ranks that SHARE a device (several
// processes on one GPU: tests, a dry run of the multi-rank flow) launch every round when its challenge is known: a queued round kernel spins
// on the device until its challenge arrives, the challenge needs EVERY rank's message of the round before, and eight ranks' spinning kernels
// leave no room for the one kernel all of them wait for — found the first time the nv = 26 hypercube was split over 8 processes on one device
// (tests/test_gpu_dist_at_size.py): "sumcheck round finished without publishing its message (round 8 of 23)" after the 60 s pipe time-out;
// the toy sizes of the earlier tests ran those rounds on the host tail and never met it.  CENO_DIST_PIPELINE=1 / 0 overrides.
static bool dist_rounds_pipelined(ceno_dist_comm* c, hipStream_t st) {
    const bool shared = dist_comm_ranks_share_device(c, st);  // (collective: asked on every rank whatever the override says)
    const char* e = getenv("CENO_DIST_PIPELINE");
    return e ? atoi(e) != 0 : !shared;
}

static int dist_prove_shm(ceno_hip_ctx* ctx, ceno_dist_comm* c, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan_local,
                          int n_total, int n_local, int log_w, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_msgs,
                          uint64_t* out_challenges, uint64_t* out_final_evals) {
    const int world = c->world, d = plan_local->max_degree, k = plan_local->num_mles;
    if (n_local < 1) {
        g_dist_err = "shared-memory exchange needs at least one local variable per shard";
        return CENO_HIP_ERR_INVALID;
    }
    ceno_hip_sumcheck* sc = nullptr;
    int rc = ceno_hip_sumcheck_begin(ctx, mles, plan_local, s, &sc);
    if (rc) {
        g_dist_err = ceno_hip_last_error(ctx);
        return rc;
    }
    if (dist_rounds_pipelined(c, (hipStream_t)s)) ceno_hip_sumcheck_set_pipelined(ctx, sc, 1);
    uint64_t ch[2] = {0, 0};
    std::vector<uint64_t> part(2 * (size_t)d), msg(2 * (size_t)d);
    for (int r = 0; r < n_local && !rc; r++) {
        rc = ceno_hip_sumcheck_round(ctx, sc, r == 0 ? nullptr : ch, part.data());
        if (rc) {
            g_dist_err = ceno_hip_last_error(ctx);
            break;
        }
        rc = shm_gather_ext(c, part.data(), d);
        if (rc) break;
        for (int t = 0; t < d; t++) {
            E2 acc = gl::e2_zero();
            for (int g = 0; g < world; g++) acc = acc + E2{c->h_recv[2 * ((size_t)g * d + t)], c->h_recv[2 * ((size_t)g * d + t) + 1]};
            msg[2 * t] = acc.c0;
            msg[2 * t + 1] = acc.c1;
        }
        memcpy(out_msgs + (size_t)2 * d * r, msg.data(), (size_t)16 * d);
        E2 rr = absorb_round(tr, msg.data(), d);
        ch[0] = rr.c0;
        ch[1] = rr.c1;
        out_challenges[2 * r] = rr.c0;
        out_challenges[2 * r + 1] = rr.c1;
    }
    std::vector<uint64_t> fin_local(2 * (size_t)k);
    if (!rc) {
        rc = ceno_hip_sumcheck_finish(ctx, sc, ch, fin_local.data());
        if (rc) g_dist_err = ceno_hip_last_error(ctx);
    }
    ceno_hip_sumcheck_free(ctx, sc);
    if (rc) return rc;
    if (world == 1) {
        memcpy(out_final_evals, fin_local.data(), (size_t)16 * k);
        return 0;
    }
    if (k > 64) {
        g_dist_err = "more than 64 MLEs in a sharded sumcheck";
        return CENO_HIP_ERR_INVALID;
    }
    rc = shm_gather_ext(c, fin_local.data(), k);
    if (rc) return rc;
    return dist_tail(ctx, c, plan_local, n_local, log_w, d, k, tr, s, ch, out_msgs, out_challenges, out_final_evals);
}

}  // extern "C"

/* Mixed-size (front-loaded) batched sumcheck across ranks — SURVEY.md section 8(e), the C++ counterpart of
 * ceno_amd/dist.py::sharded_batched_sumcheck_prove (same protocol):
 *   sharded class    : every rank holds slice `rank` of each table (num_vars - log2(world) local variables); its local
 *                      rounds produce partial messages; when the local variables run out the per-rank values are
 *                      exchanged once and the class continues as world-sized replicated tables;
 *   replicated class : every rank holds the whole tables and runs the same rounds; counted once.
 * The engine of a replicated class is begun with max_num_vars = the rounds left, so it applies the front-load scalars
 * itself.  Exchanges go through the host shared-memory segment (ceno_dist_comm_attach_shm). */
namespace {
struct Seg {
    int cls = -1;
    ceno_hip_sumcheck* sc = nullptr;
    bool sharded = false;
    int start = 0, end = 0;                 // global rounds [start, end)
    std::vector<ceno_hip_mle*> owned;       // tail tables uploaded here
};
}  // namespace

extern "C" int ceno_dist_batched_sumcheck_prove(ceno_hip_ctx* ctx, ceno_dist_comm* c, const ceno_dist_class* classes, int n_classes, int n_total,
                                                int max_degree, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_msgs,
                                                uint64_t* out_challenges, uint64_t* out_final_evals) {
    if (!ctx || !c || !classes || !tr || !s || n_classes < 1 || n_total < 1 || max_degree < 1 || max_degree > 8) {
        g_dist_err = "batched: bad arguments (an explicit stream is required)";
        return CENO_HIP_ERR_INVALID;
    }
    if (!c->shm) {
        g_dist_err = "batched: the communicator needs the shared-memory exchange (ceno_dist_comm_attach_shm)";
        return CENO_HIP_ERR_STATE;
    }
    const int world = c->world, d = max_degree;
    int log_w = 0;
    while ((1 << log_w) < world) log_w++;
    if ((1 << log_w) != world) {
        g_dist_err = "batched: world size must be a power of two";
        return CENO_HIP_ERR_INVALID;
    }
    std::vector<size_t> fin_off(n_classes + 1, 0);
    for (int i = 0; i < n_classes; i++) fin_off[i + 1] = fin_off[i] + (size_t)classes[i].num_mles;
    std::vector<Seg> segs;
    int rc = 0;
    auto cleanup = [&]() {
        for (auto& g : segs) {
            if (g.sc) ceno_hip_sumcheck_free(ctx, g.sc);
            for (auto* m : g.owned) ceno_hip_mle_free(ctx, m);
        }
    };
    auto hip_fail = [&](int code) {
        g_dist_err = ceno_hip_last_error(ctx);
        cleanup();
        return code;
    };
    auto begin_seg = [&](int ci, ceno_hip_mle* const* mles, int max_nv, bool sharded, int start, int end, std::vector<ceno_hip_mle*> owned) {
        const ceno_dist_class& K = classes[ci];
        ceno_hip_sumcheck_plan plan;
        memset(&plan, 0, sizeof(plan));
        plan.num_mles = K.num_mles;
        plan.num_terms = K.num_terms;
        plan.term_coeffs = K.term_coeffs;
        plan.term_offsets = K.term_offsets;
        plan.term_mle_idx = K.term_mle_idx;
        plan.max_num_vars = max_nv;
        plan.max_degree = d;
        Seg g;
        g.cls = ci;
        g.sharded = sharded;
        g.start = start;
        g.end = end;
        g.owned = std::move(owned);
        int r = ceno_hip_sumcheck_begin(ctx, mles, &plan, s, &g.sc);
        segs.push_back(std::move(g));
        return r;
    };
    // per-rank values of a class (k ext) -> world-sized replicated tables that join at round `first`
    auto start_tail = [&](int ci, const uint64_t* local_vals, int first) -> int {
        const int k = classes[ci].num_mles;
        if (k > 64) {
            g_dist_err = "batched: more than 64 MLEs in a sharded class";
            return CENO_HIP_ERR_INVALID;
        }
        int r = shm_gather_ext(c, local_vals, k);
        if (r) return r;
        std::vector<ceno_hip_mle*> tabs(k, nullptr);
        std::vector<uint64_t> tab(2 * (size_t)world);
        for (int j = 0; j < k && !r; j++) {
            for (int g = 0; g < world; g++) {
                tab[2 * g] = c->h_recv[2 * ((size_t)g * k + j)];
                tab[2 * g + 1] = c->h_recv[2 * ((size_t)g * k + j) + 1];
            }
            r = ceno_hip_mle_upload(ctx, tab.data(), log_w, 1, s, &tabs[j]);
        }
        if (r) {
            for (auto* m : tabs)
                if (m) ceno_hip_mle_free(ctx, m);
            g_dist_err = ceno_hip_last_error(ctx);
            return r;
        }
        std::vector<ceno_hip_mle*> keep = tabs;
        r = begin_seg(ci, tabs.data(), n_total - first, false, first, n_total, std::move(keep));
        if (r) g_dist_err = ceno_hip_last_error(ctx);
        return r;
    };

    tr_usize(tr, (uint64_t)n_total);
    tr_usize(tr, (uint64_t)d);
    for (int ci = 0; ci < n_classes && !rc; ci++) {
        const ceno_dist_class& K = classes[ci];
        if (K.num_vars < 1 || K.num_vars > n_total || K.num_mles < 1 || K.num_terms < 1) {
            g_dist_err = "batched: bad class";
            cleanup();
            return CENO_HIP_ERR_INVALID;
        }
        if (K.sharded && world > 1) {
            const int nvl = K.num_vars - log_w;
            if (nvl < 1) {
                g_dist_err = "batched: a sharded class needs more than log2(world) variables (pass it replicated)";
                cleanup();
                return CENO_HIP_ERR_INVALID;
            }
            rc = begin_seg(ci, K.mles, nvl, true, 0, nvl, {});
        } else {
            rc = begin_seg(ci, K.mles, n_total, false, 0, n_total, {});
        }
    }
    if (rc) return hip_fail(rc);
    uint64_t ch[2] = {0, 0};
    std::vector<uint64_t> part, msg(2 * (size_t)d), tmp(2 * (size_t)d);
    for (int i = 0; i < n_total; i++) {
        std::vector<int> live_sh, live_rp;
        for (int g = 0; g < (int)segs.size(); g++)
            if (segs[g].start <= i && i < segs[g].end) (segs[g].sharded ? live_sh : live_rp).push_back(g);
        for (int t = 0; t < 2 * d; t++) msg[t] = 0;
        auto add_into_msg = [&](const uint64_t* m) {
            for (int t = 0; t < d; t++) {
                E2 a = E2{msg[2 * t], msg[2 * t + 1]} + E2{m[2 * t], m[2 * t + 1]};
                msg[2 * t] = a.c0;
                msg[2 * t + 1] = a.c1;
            }
        };
        if (!live_sh.empty()) {
            if ((int)live_sh.size() * d > 64) {
                g_dist_err = "batched: more than 64 partial evaluations per rank in one round";
                cleanup();
                return CENO_HIP_ERR_INVALID;
            }
            part.assign(2 * (size_t)d * live_sh.size(), 0);
            for (size_t q = 0; q < live_sh.size(); q++) {
                Seg& g = segs[live_sh[q]];
                rc = ceno_hip_sumcheck_round(ctx, g.sc, i == g.start ? nullptr : ch, part.data() + 2 * (size_t)d * q);
                if (rc) return hip_fail(rc);
            }
            rc = shm_gather_ext(c, part.data(), d * (int)live_sh.size());
            if (rc) {
                cleanup();
                return rc;
            }
            for (int g = 0; g < world; g++)
                for (size_t q = 0; q < live_sh.size(); q++) add_into_msg(c->h_recv + 2 * ((size_t)g * live_sh.size() + q) * d);
        }
        for (int gi : live_rp) {
            Seg& g = segs[gi];
            rc = ceno_hip_sumcheck_round(ctx, g.sc, i == g.start ? nullptr : ch, tmp.data());
            if (rc) return hip_fail(rc);
            add_into_msg(tmp.data());
        }
        memcpy(out_msgs + (size_t)2 * d * i, msg.data(), (size_t)16 * d);
        E2 rr = absorb_round(tr, msg.data(), d);
        ch[0] = rr.c0;
        ch[1] = rr.c1;
        out_challenges[2 * i] = rr.c0;
        out_challenges[2 * i + 1] = rr.c1;
        for (int gi : live_sh) {
            if (segs[gi].end != i + 1) continue;  // last local variable bound: one value per table per rank
            std::vector<uint64_t> fin(2 * (size_t)classes[segs[gi].cls].num_mles);
            rc = ceno_hip_sumcheck_finish(ctx, segs[gi].sc, ch, fin.data());
            if (rc) return hip_fail(rc);
            ceno_hip_sumcheck_free(ctx, segs[gi].sc);
            segs[gi].sc = nullptr;
            const int ci = segs[gi].cls;
            rc = start_tail(ci, fin.data(), i + 1);  // may reallocate `segs`
            if (rc) {
                cleanup();
                return rc;
            }
        }
    }
    for (auto& g : segs) {
        if (g.sharded || !g.sc) continue;
        rc = ceno_hip_sumcheck_finish(ctx, g.sc, ch, out_final_evals + 2 * fin_off[g.cls]);
        if (rc) return hip_fail(rc);
    }
    cleanup();
    return 0;
}

extern "C" {

/* Sumcheck of the plan's terms over a 2^n_total hypercube whose tables are sharded by their top
 * log2(world) index bits: `mles` are this rank's shards (n_total - log2(world) variables each).
 * Same transcript script as IOPProverState::prove, so the proof equals the single-device proof of the
 * unsharded tables.  Outputs (identical on every rank): msgs n_total*d ext, challenges n_total ext,
 * final_evals num_mles ext.  `s` must be an explicit stream created by ceno_hip_stream_create. */
int ceno_dist_sumcheck_prove(ceno_hip_ctx* ctx, ceno_dist_comm* c, ceno_hip_mle* const* mles, const ceno_hip_sumcheck_plan* plan_local,
                             int n_total, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_msgs, uint64_t* out_challenges,
                             uint64_t* out_final_evals) {
    if (!ctx || !c || !plan_local || !tr || !s) {
        g_dist_err = "NULL argument (an explicit stream is required)";
        return CENO_HIP_ERR_INVALID;
    }
    const int world = c->world;
    int log_w = 0;
    while ((1 << log_w) < world) log_w++;
    const int n_local = plan_local->max_num_vars, d = plan_local->max_degree, k = plan_local->num_mles;
    if ((1 << log_w) != world || n_local + log_w != n_total || k > 64 || d > 8) {
        g_dist_err = "bad sharding geometry";
        return CENO_HIP_ERR_INVALID;
    }
    hipStream_t st = (hipStream_t)s;
    tr_usize(tr, (uint64_t)n_total);
    tr_usize(tr, (uint64_t)d);
    if (c->shm && (!c->comm || !getenv("CENO_DIST_EXCHANGE_RCCL")))
        return dist_prove_shm(ctx, c, mles, plan_local, n_total, n_local, log_w, tr, s, out_msgs, out_challenges, out_final_evals);
    if (!c->comm) {
        g_dist_err = "communicator has neither RCCL nor the shared-memory exchange";
        return CENO_HIP_ERR_STATE;
    }
    ceno_hip_sumcheck* sc = nullptr;
    int rc = ceno_hip_sumcheck_begin(ctx, mles, plan_local, s, &sc);
    if (rc) {
        g_dist_err = ceno_hip_last_error(ctx);
        return rc;
    }
    uint64_t ch[2] = {0, 0};
    std::vector<uint64_t> msg(2 * (size_t)d);
    // Once a shard is down to 2^TAIL_VARS elements per table the per-round collective costs more than the
    // round itself: gather the small shards onto every rank ONCE and let each rank finish all remaining
    // rounds locally (replicated, pipelined, no further collectives).
    const int TAIL_VARS = 12;
    const bool force_gather = getenv("CENO_DIST_FORCE_GATHER") != nullptr;  // exercises the gather path at world size 1 (tests)
    int r = 0;
    for (; r < n_local; r++) {
        if ((world > 1 || force_gather) && r >= 1 && n_local - (r - 1) <= TAIL_VARS) break;  // live tables have n_local-(r-1) variables
        rc = ceno_hip_sumcheck_round_dev(ctx, sc, r == 0 ? nullptr : ch, c->d_send);
        if (!rc) rc = gather_ext(c, d, st);
        if (rc) {
            if (g_dist_err.empty()) g_dist_err = ceno_hip_last_error(ctx);
            ceno_hip_sumcheck_free(ctx, sc);
            return rc;
        }
        for (int t = 0; t < d; t++) {
            E2 acc = gl::e2_zero();
            for (int g = 0; g < world; g++) acc = acc + E2{c->h_recv[2 * ((size_t)g * d + t)], c->h_recv[2 * ((size_t)g * d + t) + 1]};
            msg[2 * t] = acc.c0;
            msg[2 * t + 1] = acc.c1;
        }
        memcpy(out_msgs + (size_t)2 * d * r, msg.data(), (size_t)16 * d);
        E2 rr = absorb_round(tr, msg.data(), d);
        ch[0] = rr.c0;
        ch[1] = rr.c1;
        out_challenges[2 * r] = rr.c0;
        out_challenges[2 * r + 1] = rr.c1;
    }
    if (r < n_local) {
        // ---- early gather: tables T_{r-1} (m variables per shard), messages 0..r-1 sent, challenge r-1 in `ch` ----
        const int m = n_local - (r - 1), nv2 = m + log_w;
        std::vector<ceno_hip_mle*> full(k, nullptr);
        auto drop = [&]() {
            for (auto* x : full) if (x) ceno_hip_mle_free(ctx, x);
        };
        for (int j = 0; j < k && !rc; j++) {
            uint64_t* src = nullptr;
            int is_ext = 0, nvj = 0;
            rc = ceno_hip_sumcheck_table(ctx, sc, j, &src, &is_ext, &nvj);
            if (!rc && nvj != m) { g_dist_err = "sharded tables of unequal size"; rc = CENO_HIP_ERR_INVALID; }
            if (!rc) rc = ceno_hip_stream_bind(ctx, s);
            if (!rc) rc = ceno_hip_mle_alloc(ctx, nv2, is_ext, &full[j]);
            if (!rc) {
                ncclResult_t nr = g_rccl.AllGather(src, ceno_hip_mle_device_ptr(full[j]), ((size_t)1 << m) * (is_ext ? 2 : 1), ncclUint64, c->comm, st);
                if (nr != ncclSuccess) rc = nccl_fail(nr, "ncclAllGather(tables)");
            }
        }
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = CENO_HIP_ERR_HIP;
        ceno_hip_sumcheck_free(ctx, sc);
        if (rc) {
            if (g_dist_err.empty()) g_dist_err = ceno_hip_last_error(ctx);
            drop();
            return rc;
        }
        ceno_hip_sumcheck_plan p2 = *plan_local;
        p2.max_num_vars = nv2;
        ceno_hip_sumcheck* s2 = nullptr;
        rc = ceno_hip_sumcheck_begin(ctx, full.data(), &p2, s, &s2);
        if (!rc) ceno_hip_sumcheck_set_pipelined(ctx, s2, 1);
        // its round 0 recomputes message r-1 from the gathered tables: a free consistency check of the gather
        if (!rc) rc = ceno_hip_sumcheck_round(ctx, s2, nullptr, msg.data());
        if (!rc && memcmp(msg.data(), out_msgs + (size_t)2 * d * (r - 1), (size_t)16 * d) != 0) {
            g_dist_err = "gathered tables do not reproduce the last sharded message";
            rc = CENO_HIP_ERR_STATE;
        }
        for (int g = r; g < n_total && !rc; g++) {
            rc = ceno_hip_sumcheck_round(ctx, s2, ch, msg.data());
            if (rc) break;
            memcpy(out_msgs + (size_t)2 * d * g, msg.data(), (size_t)16 * d);
            E2 rr = absorb_round(tr, msg.data(), d);
            ch[0] = rr.c0;
            ch[1] = rr.c1;
            out_challenges[2 * g] = rr.c0;
            out_challenges[2 * g + 1] = rr.c1;
        }
        if (!rc) rc = ceno_hip_sumcheck_finish(ctx, s2, ch, out_final_evals);
        if (s2) ceno_hip_sumcheck_free(ctx, s2);
        drop();
        if (rc && g_dist_err.empty()) g_dist_err = ceno_hip_last_error(ctx);
        return rc;
    }
    std::vector<uint64_t> fin_local(2 * (size_t)k);
    rc = ceno_hip_sumcheck_finish(ctx, sc, n_local > 0 ? ch : nullptr, fin_local.data());
    ceno_hip_sumcheck_free(ctx, sc);
    if (rc) {
        g_dist_err = ceno_hip_last_error(ctx);
        return rc;
    }
    if (world == 1) {
        memcpy(out_final_evals, fin_local.data(), (size_t)16 * k);
        return 0;
    }
    // one value per table per rank -> world-sized tables (index = rank = the top bits), replicated tail
    if (hipMemcpyAsync(c->d_send, fin_local.data(), (size_t)16 * k, hipMemcpyHostToDevice, st) != hipSuccess) return CENO_HIP_ERR_HIP;
    rc = gather_ext(c, k, st);
    if (rc) return rc;
    std::vector<ceno_hip_mle*> tail(k, nullptr);
    std::vector<uint64_t> tab(2 * (size_t)world);
    for (int j = 0; j < k && !rc; j++) {
        for (int g = 0; g < world; g++) {
            tab[2 * g] = c->h_recv[2 * ((size_t)g * k + j)];
            tab[2 * g + 1] = c->h_recv[2 * ((size_t)g * k + j) + 1];
        }
        rc = ceno_hip_mle_upload(ctx, tab.data(), log_w, 1, s, &tail[j]);
    }
    if (!rc) {
        ceno_hip_sumcheck_plan tp = *plan_local;
        tp.max_num_vars = log_w;
        ceno_hip_sumcheck* ts = nullptr;
        rc = ceno_hip_sumcheck_begin(ctx, tail.data(), &tp, s, &ts);
        for (int r = 0; r < log_w && !rc; r++) {
            rc = ceno_hip_sumcheck_round(ctx, ts, (n_local + r) == 0 ? nullptr : (r == 0 ? nullptr : ch), msg.data());
            if (rc) break;
            memcpy(out_msgs + (size_t)2 * d * (n_local + r), msg.data(), (size_t)16 * d);
            E2 rr = absorb_round(tr, msg.data(), d);
            ch[0] = rr.c0;
            ch[1] = rr.c1;
            out_challenges[2 * (n_local + r)] = rr.c0;
            out_challenges[2 * (n_local + r) + 1] = rr.c1;
        }
        if (!rc) rc = ceno_hip_sumcheck_finish(ctx, ts, ch, out_final_evals);
        if (ts) ceno_hip_sumcheck_free(ctx, ts);
    }
    for (auto* m : tail)
        if (m) ceno_hip_mle_free(ctx, m);
    if (rc) g_dist_err = ceno_hip_last_error(ctx);
    return rc;
}

}  // extern "C"

// ==================================================================================================================
// Trace commitment across ranks (SURVEY.md §8(e) "Commit path"; reference single-device flow: commit_traces,
// ceno_zkvm/src/scheme/gpu/mod.rs:927-1020).  Columns are independent, so every rank RS-encodes ITS columns (no exchange);
// a Merkle leaf hashes one codeword row across ALL columns, so the codeword is re-sharded by rows with ONE all-to-all — the
// only step of the whole proving path whose cost is xGMI bandwidth ((world-1)/world of the codeword leaves each rank; with
// point-to-point links every pair of ranks exchanges its block directly, RCCL grouped send/recv) — after which rank g owns
// rows [g R/world, (g+1) R/world) = the sub-tree under node g of level log2(world).  The sub-tree roots are gathered and
// the top log2(world) levels are hashed on every rank.  Equal to the single-device commitment bit for bit.
// ==================================================================================================================
namespace {

int dist_fail(int code, const std::string& msg) {
    g_dist_err = msg;
    // a rank that fails on its own (a copy, a validation) will not publish what its peers wait for: let them return now
    if (tls_shm_comm && tls_shm_comm->shm) tls_shm_comm->shm->abort = 1;
    return code;
}

// all-to-all of unequal blocks of 64-bit words.  send block for destination g: send + soff[g], scnt[g] words;
// block from source g lands at recv + roff[g], rcnt[g] words.  The own block is copied on the device.
// The same exchange between PROCESSES without RCCL (several ranks on one GPU, or a node whose RCCL is unusable): through the shared segment's bulk
// area, staged on the host.  Step k = 1 .. W - 1 moves the block for rank me + k and takes the one from rank me - k, 128 KB per publication; every
// rank makes the same number of publications per step (the largest block of anyone sets it), so the sequence words stay in lock step and a buffer is
// rewritten only after every rank has passed the publication that read it.  A correctness path (~10 GB/s of host copies), not a fast one.
static int shm_exchange_blocks(ceno_dist_comm* c, const uint64_t* send, const size_t* soff, const size_t* scnt, uint64_t* recv, const size_t* roff,
                               const size_t* rcnt, hipStream_t st) {
    const int W = c->world, me = c->rank;
    if (scnt[me] && hipMemcpyAsync(recv + roff[me], send + soff[me], scnt[me] * 8, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return dist_fail(CENO_HIP_ERR_HIP, "exchange_blocks: device copy failed");
    uint64_t mx = 0;
    for (int g = 0; g < W; g++) mx = std::max<uint64_t>(mx, g == me ? 0 : scnt[g]);
    std::vector<uint64_t> all_mx((size_t)W);
    if (int rc = shm_gather_bulk(c, &mx, 1, all_mx.data(), 1)) return dist_fail(rc, g_dist_err.c_str());
    for (uint64_t v : all_mx) mx = std::max(mx, v);
    const size_t n_chunks = (size_t)((mx + SHM_BULK_WORDS - 1) / SHM_BULK_WORDS);
    std::vector<std::vector<uint64_t>> hs((size_t)W), hr((size_t)W);
    for (int g = 0; g < W; g++) {
        if (g == me) continue;
        hs[(size_t)g].resize(scnt[g]);
        hr[(size_t)g].resize(rcnt[g]);
        if (scnt[g] && hipMemcpyAsync(hs[(size_t)g].data(), send + soff[g], scnt[g] * 8, hipMemcpyDeviceToHost, st) != hipSuccess)
            return dist_fail(CENO_HIP_ERR_HIP, "exchange_blocks: download failed");
    }
    if (hipStreamSynchronize(st) != hipSuccess) return dist_fail(CENO_HIP_ERR_HIP, "exchange_blocks: sync failed");
    for (int k = 1; k < W; k++) {
        const int to = (me + k) % W, from = (me - k + W) % W;
        for (size_t ch = 0; ch < n_chunks; ch++) {
            const size_t o = ch * SHM_BULK_WORDS;
            const size_t ns = scnt[to] > o ? std::min<size_t>(SHM_BULK_WORDS, scnt[to] - o) : 0;
            const size_t nr = rcnt[from] > o ? std::min<size_t>(SHM_BULK_WORDS, rcnt[from] - o) : 0;
            const uint64_t seq = ++c->shm_seq;
            if (ns) memcpy(shm_bulk(c->shm, W, me, (int)(seq & 1)), hs[(size_t)to].data() + o, ns * 8);
            __atomic_store_n(&c->shm->ranks[me].seq, seq, __ATOMIC_RELEASE);
            tls_shm_comm = c;
            for (int g = 0; g < W; g++)
                if (int rc = shm_wait(c, g, seq, "exchange_blocks")) return rc;
            if (nr) memcpy(hr[(size_t)from].data() + o, shm_bulk(c->shm, W, from, (int)(seq & 1)), nr * 8);
        }
    }
    for (int g = 0; g < W; g++)
        if (g != me && rcnt[g] && hipMemcpyAsync(recv + roff[g], hr[(size_t)g].data(), rcnt[g] * 8, hipMemcpyHostToDevice, st) != hipSuccess)
            return dist_fail(CENO_HIP_ERR_HIP, "exchange_blocks: upload failed");
    if (hipStreamSynchronize(st) != hipSuccess) return dist_fail(CENO_HIP_ERR_HIP, "exchange_blocks: sync failed");  // (the staging vectors go out of scope)
    return 0;
}

int exchange_blocks(ceno_dist_comm* c, const uint64_t* send, const size_t* soff, const size_t* scnt, uint64_t* recv, const size_t* roff,
                    const size_t* rcnt, hipStream_t st) {
    const int W = c->world, me = c->rank;
    c->stats[2]++;
    for (int g = 0; g < W; g++)
        if (g != me) c->stats[3] += (uint64_t)scnt[g] * 8;
    if (scnt[me] != rcnt[me]) return dist_fail(CENO_HIP_ERR_INVALID, "exchange_blocks: own block size mismatch");
    if (c->local) {
        ceno_dist_local_group* G = c->local;
        if (hipStreamSynchronize(st) != hipSuccess) return dist_fail(CENO_HIP_ERR_HIP, "exchange_blocks: sync before publish failed");
        {
            std::lock_guard<std::mutex> g(G->mu);
            G->send_base[me] = send;
            G->send_off[me].assign(soff, soff + W);
        }
        if (!G->barrier()) return dist_fail(CENO_HIP_ERR_STATE, "exchange_blocks: a peer of the local group is gone");
        for (int g = 0; g < W; g++) {
            const uint64_t* src;
            {
                std::lock_guard<std::mutex> lk(G->mu);
                src = G->send_base[g] + G->send_off[g][me];
            }
            if (rcnt[g] && hipMemcpyAsync(recv + roff[g], src, rcnt[g] * 8, hipMemcpyDeviceToDevice, st) != hipSuccess)
                return dist_fail(CENO_HIP_ERR_HIP, "exchange_blocks: device copy failed");
        }
        if (hipStreamSynchronize(st) != hipSuccess) return dist_fail(CENO_HIP_ERR_HIP, "exchange_blocks: sync failed");
        if (!G->barrier()) return dist_fail(CENO_HIP_ERR_STATE, "exchange_blocks: a peer of the local group is gone");  // send buffers may be reused now
        return 0;
    }
    if (!c->comm && c->shm) return shm_exchange_blocks(c, send, soff, scnt, recv, roff, rcnt, st);
    if (!c->comm) return dist_fail(CENO_HIP_ERR_STATE, "exchange_blocks: the communicator has no transport");
    if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd)
        return dist_fail(CENO_HIP_ERR_UNSUPPORTED, "this RCCL has no ncclSend / ncclRecv");
    const bool self_p2p = getenv("CENO_DIST_SELF_P2P") && atoi(getenv("CENO_DIST_SELF_P2P")) != 0;  // tests: own block through RCCL too
    if (!self_p2p && scnt[me] &&
        hipMemcpyAsync(recv + roff[me], send + soff[me], scnt[me] * 8, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return dist_fail(CENO_HIP_ERR_HIP, "exchange_blocks: device copy failed");
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) return nccl_fail(r, "ncclGroupStart");
    for (int k = 0; k < W && r == ncclSuccess; k++) {
        // pairwise order (me + k, me - k): every link of the fully connected xGMI mesh carries one block per direction
        const int to = (me + k) % W, from = (me - k + W) % W;
        if (k == 0 && !self_p2p) continue;
        if (scnt[to]) r = g_rccl.Send(send + soff[to], scnt[to], ncclUint64, to, c->comm, st);
        if (r == ncclSuccess && rcnt[from]) r = g_rccl.Recv(recv + roff[from], rcnt[from], ncclUint64, from, c->comm, st);
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r != ncclSuccess) return nccl_fail(r, "ncclSend/ncclRecv");
    if (r2 != ncclSuccess) return nccl_fail(r2, "ncclGroupEnd");
    return 0;
}

// every rank contributes 4 words, everyone gets world x 4 (host)
int gather_digests(ceno_dist_comm* c, const uint64_t* mine4, uint64_t* out, hipStream_t st) {
    const int W = c->world;
    if (c->local) {
        ceno_dist_local_group* G = c->local;
        {
            std::lock_guard<std::mutex> g(G->mu);
            memcpy(&G->small[(size_t)c->rank * 128], mine4, 32);
        }
        if (!G->barrier()) return dist_fail(CENO_HIP_ERR_STATE, "gather_digests: a peer of the local group is gone");
        {
            std::lock_guard<std::mutex> g(G->mu);
            for (int r = 0; r < W; r++) memcpy(out + 4 * r, &G->small[(size_t)r * 128], 32);
        }
        if (!G->barrier()) return dist_fail(CENO_HIP_ERR_STATE, "gather_digests: a peer of the local group is gone");
        return 0;
    }
    if (c->shm) {
        if (int rc = shm_gather_ext(c, mine4, 2)) return rc;
        memcpy(out, c->h_recv, (size_t)W * 32);
        return 0;
    }
    if (!c->comm) return dist_fail(CENO_HIP_ERR_STATE, "gather_digests: communicator without a transport");
    if (hipMemcpyAsync(c->d_send, mine4, 32, hipMemcpyHostToDevice, st) != hipSuccess) return dist_fail(CENO_HIP_ERR_HIP, "gather_digests: upload failed");
    if (int rc = gather_ext(c, 2, st)) return rc;
    memcpy(out, c->h_recv, (size_t)W * 32);
    return 0;
}

}  // namespace

// ---- helpers for the row-sharded GKR half (dist_gkr.cpp) ----
int dist_comm_world(const ceno_dist_comm* c) { return c ? c->world : 1; }
int dist_comm_rank(const ceno_dist_comm* c) { return c ? c->rank : 0; }
bool dist_comm_has_rccl(const ceno_dist_comm* c) { return c && c->comm != nullptr; }
int dist_allgather_words(ceno_dist_comm* c, const uint64_t* mine, size_t n_words, uint64_t* out, hipStream_t st);
// Do two ranks of the communicator sit on ONE device?  Decided from the placement itself — every rank's host name and the PCI bus id of its
// current device, gathered once per communicator — not from which transports happen to be attached: ranks that share a device share its
// hardware queues (a queued round kernel of one can hold up the tree kernels another waits for: docs/rounds/r05.md "A hazard found late"), whatever
// carries their messages.  COLLECTIVE on first use (every rank must ask).  An in-process group is one device by construction.
bool dist_comm_ranks_share_device(ceno_dist_comm* c, hipStream_t st) {
    if (!c || c->world == 1) return false;
    if (c->local) return true;
    if (c->shares_device >= 0) return c->shares_device == 1;
    uint64_t mine[2] = {0, 0};  // FNV-1a of "<host name>/<pci bus id>", twice with different seeds
    {
        char host[256] = {0}, bus[64] = {0};
        (void)gethostname(host, sizeof host - 1);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) snprintf(bus, sizeof bus, "dev%d", dev);
        const std::string key = std::string(host) + "/" + bus;
        uint64_t h0 = 0xcbf29ce484222325ULL, h1 = 0x84222325cbf29ce4ULL;
        for (unsigned char ch : key) {
            h0 = (h0 ^ ch) * 0x100000001b3ULL;
            h1 = (h1 ^ (ch + 0x9e)) * 0x100000001b3ULL;
        }
        mine[0] = h0 >> 1;  // (the small-message transports carry field words: keep them below 2^63)
        mine[1] = h1 >> 1;
    }
    std::vector<uint64_t> all((size_t)2 * c->world);
    if (dist_allgather_words(c, mine, 2, all.data(), st) != 0) return true;  // cannot tell: take the safe (serial) side
    int shared = 0;
    for (int a = 0; a < c->world && !shared; a++)
        for (int b = a + 1; b < c->world; b++)
            if (all[2 * a] == all[2 * b] && all[2 * a + 1] == all[2 * b + 1]) shared = 1;
    c->shares_device = shared;
    return shared == 1;
}
// all-gather `n_words` 64-bit words per rank (host memory in, host memory out: out[g * n_words + k]) over the communicator's small-message
// transport — in-process group, shared segment or RCCL — in chunks of 128 words.  Bulk data (MBs) belongs on exchange_blocks; this carries the
// per-round partial sums and the folded tables (KBs) of the sharded tower prover.
int dist_allgather_words(ceno_dist_comm* c, const uint64_t* mine, size_t n_words, uint64_t* out, hipStream_t st) {
    if (!c || c->world == 1) {
        if (n_words) memcpy(out, mine, n_words * 8);
        return 0;
    }
    const int W = c->world;
    c->stats[0]++;   // one message exchange, whatever it is chunked into underneath
    c->stats[1] += (uint64_t)n_words * 8;
    if (c->shm && !c->local) {  // the shared segment's bulk area: up to SHM_BULK_WORDS words per exchange
        for (size_t off = 0; off < n_words; off += SHM_BULK_WORDS) {
            const size_t nb = std::min<size_t>(SHM_BULK_WORDS, n_words - off);
            if (int rc = shm_gather_bulk(c, mine + off, nb, out + off, n_words)) return rc;
        }
        return 0;
    }
    uint64_t buf[128];
    for (size_t off = 0; off < n_words; off += 128) {
        const size_t n = std::min<size_t>(128, n_words - off);
        memset(buf, 0, sizeof buf);
        memcpy(buf, mine + off, n * 8);
        if (c->local) {
            ceno_dist_local_group* G = c->local;
            {
                std::lock_guard<std::mutex> g(G->mu);
                memcpy(&G->small[(size_t)c->rank * 128], buf, sizeof buf);
            }
            if (!G->barrier()) return dist_fail(CENO_HIP_ERR_STATE, "allgather: a peer of the local group is gone");
            {
                std::lock_guard<std::mutex> g(G->mu);
                for (int r = 0; r < W; r++) memcpy(out + (size_t)r * n_words + off, &G->small[(size_t)r * 128], n * 8);
            }
            if (!G->barrier()) return dist_fail(CENO_HIP_ERR_STATE, "allgather: a peer of the local group is gone");
        } else if (c->comm) {
            if (hipMemcpyAsync(c->d_send, buf, sizeof buf, hipMemcpyHostToDevice, st) != hipSuccess) return dist_fail(CENO_HIP_ERR_HIP, "allgather: upload failed");
            if (int rc = gather_ext(c, 64, st)) return rc;
            c->stats[0]--;   // (gather_ext counted the chunk: the exchange was counted once above)
            c->stats[1] -= 64 * 16;
            for (int r = 0; r < W; r++) memcpy(out + (size_t)r * n_words + off, c->h_recv + (size_t)r * 128, n * 8);
        } else {
            return dist_fail(CENO_HIP_ERR_STATE, "allgather: communicator without a transport");
        }
    }
    return 0;
}

// all-gather of `n_words` words per rank between DEVICE buffers over the communicator's bulk transport (in-process group: device copies; RCCL:
// grouped send / recv): recv[g * n_words ..] = rank g's block
int dist_allgather_device(ceno_dist_comm* c, const uint64_t* send_dev, size_t n_words, uint64_t* recv_dev, hipStream_t st) {
    if (c) {
        c->stats[2]++;
        c->stats[3] += (uint64_t)n_words * 8;
    }
    if (!c || c->world == 1) {
        if (n_words && hipMemcpyAsync(recv_dev, send_dev, n_words * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) return dist_fail(CENO_HIP_ERR_HIP, "allgather: device copy failed");
        return 0;
    }
    const int W = c->world;
    std::vector<size_t> soff((size_t)W, 0), scnt((size_t)W, n_words), roff((size_t)W), rcnt((size_t)W, n_words);
    for (int g = 0; g < W; g++) roff[(size_t)g] = (size_t)g * n_words;
    return exchange_blocks(c, send_dev, soff.data(), scnt.data(), recv_dev, roff.data(), rcnt.data(), st);
}

extern "C" {

ceno_dist_local_group* ceno_dist_local_group_create(int world) {
    if (world < 1 || (world & (world - 1)) != 0) return nullptr;
    auto* G = new ceno_dist_local_group();
    G->world = world;
    G->send_base.assign((size_t)world, nullptr);
    G->send_off.assign((size_t)world, std::vector<size_t>((size_t)world, 0));
    G->small.assign((size_t)world * 128, 0);
    return G;
}
void ceno_dist_local_group_destroy(ceno_dist_local_group* g) { delete g; }
int ceno_dist_comm_init_local(ceno_dist_local_group* group, int rank, ceno_dist_comm** out) {
    if (!group || !out || rank < 0 || rank >= group->world) return dist_fail(CENO_HIP_ERR_INVALID, "comm_init_local: bad arguments");
    auto* c = new ceno_dist_comm();
    c->world = group->world;
    c->rank = rank;
    c->local = group;
    *out = c;
    return 0;
}

// The multi-rank form of commit_traces: SEVERAL matrices of several heights under ONE root that equals the single-device
// mixed-height commitment (ceno_hip_mmcs_commit) bit for bit.  Rank g holds widths[m * world + g] columns of every matrix m.
//   1. every rank RS-encodes its columns of every matrix (no exchange);
//   2. per matrix one all-to-all of unequal blocks re-shards the codeword by rows (rank g gets rows [g R_m / W, (g+1) R_m / W) of
//      ALL columns); a matrix with fewer codeword rows than ranks is gathered whole on every rank instead;
//   3. rank g's sub-tree = the mixed-height tree over its row shards: it is the sub-tree under node g of the level with W nodes
//      of the global tree, injected classes included (a class with exactly W rows joins at the sub-tree's root);
//   4. the W sub-tree roots are gathered and the top log2 W levels are built on every rank over that digest layer, with the
//      classes shorter than W joining there (ceno_hip_mmcs_commit_over).
int ceno_dist_commit_traces_mmcs(ceno_hip_ctx* ctx, ceno_dist_comm* c, int n_mats, const int* log_rows, const int* widths,
                                 const uint64_t* const* local_cols_dev, int log_blowup, ceno_hip_stream s, uint64_t* const* out_rows_dev,
                                 ceno_hip_merkle** out_subtree, ceno_hip_merkle** out_top, uint64_t* out_subtree_roots, uint64_t* out_root) {
    if (!ctx || !c || !log_rows || !widths || !local_cols_dev || !s || !out_rows_dev || !out_subtree || !out_root || n_mats < 1)
        return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: NULL argument (an explicit stream is required)");
    const int W = c->world, me = c->rank;
    int log_w = 0;
    while ((1 << log_w) < W) log_w++;
    if ((1 << log_w) != W) return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: world must be a power of two");
    if (log_blowup < 0) return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: bad log_blowup");
    int log_max = 0;
    std::vector<size_t> w_total((size_t)n_mats, 0);
    for (int m = 0; m < n_mats; m++) {
        if (log_rows[m] < 0 || log_rows[m] + log_blowup > 40) return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: bad log_rows");
        log_max = std::max(log_max, log_rows[m] + log_blowup);
        for (int g = 0; g < W; g++) {
            if (widths[(size_t)m * W + g] < 0) return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: negative width");
            w_total[m] += (size_t)widths[(size_t)m * W + g];
        }
        if (w_total[m] == 0 || w_total[m] > (1u << 20)) return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: bad total width");
        if (widths[(size_t)m * W + me] && !local_cols_dev[m]) return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: local_cols_dev is NULL");
        if (!out_rows_dev[m]) return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: out_rows_dev is NULL");
    }
    if (log_max < log_w) return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: the tallest codeword has fewer rows than there are ranks");
    hipStream_t st = (hipStream_t)s;
    (void)ceno_hip_stream_bind(ctx, s);  // every allocation below is for work on `s`
    auto fail_hip = [&](int rc) {
        g_dist_err = ceno_hip_last_error(ctx);
        return rc;
    };
    std::vector<ceno_hip_mle*> scratch;  // pool blocks through the C ABI (the host layer has no other access to the pool)
    auto pool_words = [&](size_t words, uint64_t** out) -> int {
        int nv = 0;
        while (((size_t)1 << nv) < words) nv++;
        ceno_hip_mle* mm = nullptr;
        int rc = ceno_hip_mle_alloc(ctx, nv, 0, &mm);
        if (rc) return rc;
        scratch.push_back(mm);
        *out = ceno_hip_mle_device_ptr(mm);
        return 0;
    };
    auto release = [&]() {
        for (auto* mm : scratch) ceno_hip_mle_free(ctx, mm);
        scratch.clear();
    };
    int rc = 0;
    // a one-rank communicator normally skips the exchange; CENO_DIST_SELF_P2P=1 (tests) sends the own block through RCCL
    const bool exchange = W > 1 || (c->comm && getenv("CENO_DIST_SELF_P2P") && atoi(getenv("CENO_DIST_SELF_P2P")) != 0);
    for (int m = 0; m < n_mats; m++) {
        const size_t w_local = (size_t)widths[(size_t)m * W + me];
        const int lr = log_rows[m] + log_blowup;
        const size_t R = (size_t)1 << lr;
        const bool tiny = lr < log_w;                 // fewer rows than ranks: every rank gets the whole codeword
        const size_t rl = tiny ? R : (R >> log_w);    // rows this rank ends up with
        uint64_t *d_cw = nullptr, *d_pack = nullptr;
        if (w_local) {
            if ((rc = pool_words(w_local * R, &d_cw)) || (exchange && !tiny && (rc = pool_words(w_local * R, &d_pack)))) {
                release();
                return fail_hip(rc);
            }
            if ((rc = ceno_hip_rs_encode(ctx, local_cols_dev[m], log_rows[m], (int)w_local, log_blowup, d_cw, s))) {
                release();
                return fail_hip(rc);
            }
        }
        std::vector<size_t> soff((size_t)W), scnt((size_t)W), roff((size_t)W), rcnt((size_t)W);
        size_t col0 = 0;
        for (int g = 0; g < W; g++) {
            soff[g] = tiny ? 0 : (size_t)g * w_local * rl;  // tiny: the same block (all my columns, all rows) goes to everybody
            scnt[g] = w_local * rl;
            roff[g] = col0 * rl;  // source ranks in order = global column order
            rcnt[g] = (size_t)widths[(size_t)m * W + g] * rl;
            col0 += (size_t)widths[(size_t)m * W + g];
        }
        if (!exchange) {
            if (w_local && hipMemcpyAsync(out_rows_dev[m], d_cw, w_local * R * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                release();
                return dist_fail(CENO_HIP_ERR_HIP, "dist_commit_traces: copy failed");
            }
        } else {
            // pack: block for destination h = rows [h rl, (h+1) rl) of each of my columns (one strided copy per destination, the
            // rows of one column are contiguous); a tiny matrix needs no packing
            if (w_local && !tiny)
                for (int h = 0; h < W; h++)
                    if (hipMemcpy2DAsync(d_pack + soff[h], rl * 8, d_cw + (size_t)h * rl, R * 8, rl * 8, w_local, hipMemcpyDeviceToDevice, st) != hipSuccess) {
                        release();
                        return dist_fail(CENO_HIP_ERR_HIP, "dist_commit_traces: pack failed");
                    }
            if ((rc = exchange_blocks(c, tiny ? d_cw : d_pack, soff.data(), scnt.data(), out_rows_dev[m], roff.data(), rcnt.data(), st))) {
                (void)hipStreamSynchronize(st);
                release();
                return rc;
            }
        }
    }
    // 3. sub-tree over my row shards of every matrix with at least one row per rank
    std::vector<const uint64_t*> sp, tp;
    std::vector<int> slr, sw, tlr, tw;
    for (int m = 0; m < n_mats; m++) {
        const int lr = log_rows[m] + log_blowup;
        if (lr >= log_w) {
            sp.push_back(out_rows_dev[m]);
            slr.push_back(lr - log_w);
            sw.push_back((int)w_total[m]);
        } else {
            tp.push_back(out_rows_dev[m]);
            tlr.push_back(lr);
            tw.push_back((int)w_total[m]);
        }
    }
    ceno_hip_merkle* sub = nullptr;
    if ((rc = ceno_hip_mmcs_commit(ctx, sp.data(), slr.data(), sw.data(), (int)sp.size(), s, &sub))) {
        (void)hipStreamSynchronize(st);
        release();
        return fail_hip(rc);
    }
    uint64_t mine[4];
    if ((rc = ceno_hip_merkle_root(ctx, sub, mine, s))) {  // synchronises: encode, pack and exchange are complete behind it
        ceno_hip_merkle_free(ctx, sub);
        release();
        return fail_hip(rc);
    }
    release();
    // 4. gather the sub-tree roots; the top log2 W levels (with the classes shorter than W) on every rank
    std::vector<uint64_t> level((size_t)W * 4);
    if ((rc = gather_digests(c, mine, level.data(), st))) {
        ceno_hip_merkle_free(ctx, sub);
        return rc;
    }
    if (out_subtree_roots) memcpy(out_subtree_roots, level.data(), (size_t)W * 32);
    ceno_hip_merkle* top = nullptr;
    if (W > 1 || !tp.empty()) {
        uint64_t* d_level = nullptr;
        if ((rc = pool_words((size_t)W * 4, &d_level))) {
            ceno_hip_merkle_free(ctx, sub);
            return fail_hip(rc);
        }
        if (hipMemcpyAsync(d_level, level.data(), (size_t)W * 32, hipMemcpyHostToDevice, st) != hipSuccess) rc = dist_fail(CENO_HIP_ERR_HIP, "dist_commit_traces: upload of the sub-tree roots failed");
        if (!rc && (rc = ceno_hip_mmcs_commit_over(ctx, d_level, log_w, tp.data(), tlr.data(), tw.data(), (int)tp.size(), s, &top))) rc = fail_hip(rc);
        if (!rc && (rc = ceno_hip_merkle_root(ctx, top, out_root, s))) rc = fail_hip(rc);
        release();
        if (rc) {
            if (top) ceno_hip_merkle_free(ctx, top);
            ceno_hip_merkle_free(ctx, sub);
            return rc;
        }
    } else {
        memcpy(out_root, mine, 32);
    }
    if (out_top) *out_top = top;
    else if (top) ceno_hip_merkle_free(ctx, top);
    *out_subtree = sub;
    return 0;
}

// one matrix: the round-1/2 entry point
int ceno_dist_commit_traces(ceno_hip_ctx* ctx, ceno_dist_comm* c, const uint64_t* local_cols_dev, const int* widths, int log_rows, int log_blowup,
                            ceno_hip_stream s, uint64_t* out_rows_dev, ceno_hip_merkle** out_subtree, uint64_t* out_subtree_roots,
                            uint64_t* out_root) {
    if (!c || !widths) return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: NULL argument (an explicit stream is required)");
    if (log_rows + log_blowup < 0 || (1 << std::min(log_rows + log_blowup, 30)) < c->world)
        return dist_fail(CENO_HIP_ERR_INVALID, "dist_commit_traces: bad log_rows / log_blowup");
    const uint64_t* cols[1] = {local_cols_dev};
    uint64_t* outs[1] = {out_rows_dev};
    return ceno_dist_commit_traces_mmcs(ctx, c, 1, &log_rows, widths, cols, log_blowup, s, outs, out_subtree, nullptr, out_subtree_roots, out_root);
}

}  // extern "C"
