// The GKR half of a chip proof across the GPUs of one node: record inference, tower witness and tower proof over ROW-SHARDED witness
// columns, bit for bit the proof ceno_prover_create_chip_proof produces from the whole columns.
//
// Reference flow (single device): ZKVMProver::create_chip_proof ceno_zkvm/src/scheme/prover.rs:717-833 -> build_tower_witness
// scheme/cpu/mod.rs:608-757 (GPU arm scheme/gpu/mod.rs:2136-2407) -> CpuTowerProver::create_proof scheme/cpu/mod.rs:346-554.  The reference
// has no distribution (docs/src/optimizations.md:3-5); SURVEY.md section 8(e) lists a3 / a5-a10 as shardable: this is that design.
//
// WHICH rows a rank holds.  A tower layer is indexed by x; the layer above pairs x with x + half (the TOP bit), the layer sumcheck binds
// x LSB first.  A split by the top bits of x would make every product layer a cross-rank exchange; a split by bits in the MIDDLE keeps
// both directions local: rank g holds the rows whose bits [q, q + k) equal g (k = log2 world; block-cyclic, blocks of 2^q rows), its local
// table is those rows in order.  Then
//   * record inference is elementwise: local;
//   * the interleaved leaf limbs and every product / LogUp layer built from the local records ARE the shards of the global layers as long as
//     the global layer still contains the rank bits: the local tower of a rank is the shard of the global tower for the layers
//     l >= s_t + k, where s_t = q + ceil_log2(records of tower t) is the position of the rank bits in that tower's index;
//   * at global layer s_t + k the rank bits are the TOP bits: an all-gather of the local layer s_t is that global layer (general layer: an
//     interleave), from which every rank builds the small top of the tower itself (replicated; 2^(s_t + k) entries per limb);
//   * a layer sumcheck of r > r_rep variables runs its first s_min = min_t s_t rounds on the local shards — one engine per group of towers
//     with the same s_t, because a group's eq table is eq over the point with ITS rank coordinates taken out, times the scalar
//     eq(g, rt[s_t .. s_t + k)) (folded into the alpha powers) — the 3 partial evaluations per round are summed across ranks (one small
//     all-gather per round, as ceno_dist_sumcheck_prove); then the folded tables (2^(r - k - s_min) entries per rank) are gathered,
//     interleaved into the global tables of 2^(r - s_min) entries and the remaining rounds run replicated on the host;
//   * the rotation argument of a keccak-style chip (prover_prove_rotation_sharded, prover.cpp): the rotation pairs rows inside blocks of 32 / 64,
//     which a row block (2^q >= 64 rows) keeps on one rank; its sumcheck runs q local rounds, then the gathered tail replicated;
//   * the transcript is replicated: every rank appends the same words and draws the same challenges.
// Per-rank work is 1 / world of the large layers; what is replicated is the tower top (<= 2^(s_max + k) entries) and the tails.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/ceno_prover.h"
#include "../csrc/gl64.hpp"
#include "tower_hook.hpp"

using gl::E2;

int prover_set_error(int code, const char* msg);  // prover.cpp
int prover_tower_create_proof_hooked(ceno_hip_ctx* ctx, ceno_hip_tower* const* prod, int n_prod, ceno_hip_tower* const* logup, int n_logup,
                                     ceno_transcript* tr, ceno_hip_stream s, ceno_tower_proof* out, const TowerDistHook* hook);
void prover_host_tower_rounds(int n, std::vector<std::vector<E2>>& tabs, int n_prod_active, int n_logup_active, const std::vector<E2>& alpha_prod,
                              const std::vector<E2>& alpha_num, const std::vector<E2>& alpha_den, ceno_transcript* tr, uint64_t* msgs, uint64_t* chal,
                              uint64_t* fin);
int dist_comm_world(const ceno_dist_comm* c);  // dist.cpp
int dist_comm_rank(const ceno_dist_comm* c);
int dist_allgather_words(ceno_dist_comm* c, const uint64_t* mine, size_t n_words, uint64_t* out, hipStream_t st);

namespace {

int ceil_log2(size_t x) {
    int l = 0;
    while (((size_t)1 << l) < x) l++;
    return l;
}
int fail_ctx(ceno_hip_ctx* ctx, int rc) { return prover_set_error(rc, ceno_hip_last_error(ctx)); }
void tr_usize(ceno_transcript* t, uint64_t v) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
    t->append_label(t->self, b, 8);
}

// global table from per-rank tables: entry (hi, g, lo) with `lo_bits` low bits kept local, then the rank, then the local high bits
void interleave(const std::vector<uint64_t>& gathered /* world x len_loc ext */, size_t len_loc, int world, int lo_bits, std::vector<E2>& out) {
    const int k = ceil_log2((size_t)world);
    out.resize(len_loc * (size_t)world);
    const size_t lo_mask = ((size_t)1 << lo_bits) - 1;
    for (int g = 0; g < world; g++) {
        const E2* src = reinterpret_cast<const E2*>(gathered.data()) + (size_t)g * len_loc;
        for (size_t j = 0; j < len_loc; j++) {
            const size_t lo = j & lo_mask, hi = j >> lo_bits;
            out[(hi << (k + lo_bits)) | ((size_t)g << lo_bits) | lo] = src[j];
        }
    }
}

struct DistTowers {
    ceno_hip_ctx* ctx;
    ceno_dist_comm* comm;
    hipStream_t st;
    int world, rank, k;
    // local towers (shards) in the global order: product towers, then LogUp towers
    std::vector<ceno_hip_tower*> loc_prod, loc_logup;
    std::vector<int> s_of;       // per tower (prod then logup): position of the rank bits in the tower's index
    std::vector<int> nv_global;  // per tower
    int s_min = 0;
};

// hook: the layer sumcheck of `round` variables over the sharded layers
int dist_layer(void* self, int round, const uint64_t* out_rt, const uint64_t* alpha, ceno_transcript* tr, uint64_t* msgs, uint64_t* chal, uint64_t* fin) {
    DistTowers& D = *static_cast<DistTowers*>(self);
    ceno_hip_ctx* ctx = D.ctx;
    const int n_prod = (int)D.loc_prod.size(), n_logup = (int)D.loc_logup.size(), k = D.k, W = D.world, s = D.s_min;
    auto rt = [&](int j) { return E2{out_rt[2 * j], out_rt[2 * j + 1]}; };
    // groups of active towers by the position of their rank bits
    struct Group {
        int s_t;
        std::vector<int> prod, logup;  // tower indices
        ceno_hip_sumcheck* sc = nullptr;
        int n_mles = 1;
    };
    std::map<int, Group> groups;
    for (int i = 0; i < n_prod; i++)
        if (D.nv_global[i] > round) groups[D.s_of[i]].prod.push_back(i);
    for (int i = 0; i < n_logup; i++)
        if (D.nv_global[n_prod + i] > round) groups[D.s_of[n_prod + i]].logup.push_back(i);
    if (groups.empty()) return prover_set_error(CENO_HIP_ERR_INVALID, "sharded tower: no tower has this layer");
    auto cleanup = [&]() {
        for (auto& kv : groups)
            if (kv.second.sc) ceno_hip_sumcheck_free(ctx, kv.second.sc);
    };
    const int layer_loc = round - k;
    for (auto& kv : groups) {
        Group& G = kv.second;
        G.s_t = kv.first;
        // the group's local point: rt without the coordinates [s_t, s_t + k); its scalar eq(g, rt[s_t ..]) rides on the alpha powers
        std::vector<uint64_t> rt_loc;
        for (int j = 0; j < round; j++)
            if (j < G.s_t || j >= G.s_t + k) {
                rt_loc.push_back(out_rt[2 * j]);
                rt_loc.push_back(out_rt[2 * j + 1]);
            }
        E2 eq_g = gl::e2_one();
        for (int j = 0; j < k; j++) {
            const E2 c = rt(G.s_t + j);
            eq_g = eq_g * (((D.rank >> j) & 1) ? c : gl::e2_one() - c);
        }
        std::vector<ceno_hip_tower*> tp, tl;
        std::vector<uint64_t> al;
        for (int i : G.prod) {
            tp.push_back(D.loc_prod[(size_t)i]);
            const E2 a = E2{alpha[2 * i], alpha[2 * i + 1]} * eq_g;
            al.push_back(a.c0);
            al.push_back(a.c1);
        }
        for (int i : G.logup) {
            tl.push_back(D.loc_logup[(size_t)i]);
            for (int h = 0; h < 2; h++) {
                const E2 a = E2{alpha[2 * (n_prod + 2 * i + h)], alpha[2 * (n_prod + 2 * i + h) + 1]} * eq_g;
                al.push_back(a.c0);
                al.push_back(a.c1);
            }
        }
        G.n_mles = 1 + 2 * (int)tp.size() + 4 * (int)tl.size();
        int rc = ceno_hip_tower_layer_sumcheck_begin(ctx, tp.data(), (int)tp.size(), tl.data(), (int)tl.size(), layer_loc, rt_loc.data(), al.data(), D.st, &G.sc);
        if (rc) {
            cleanup();
            return fail_ctx(ctx, rc);
        }
    }
    // ---- the local rounds: every group's partial message, summed over the groups, then over the ranks ----
    tr_usize(tr, (uint64_t)round);
    tr_usize(tr, 3);
    uint64_t ch[2] = {0, 0};
    std::vector<uint64_t> all((size_t)W * 6);
    for (int i = 0; i < s; i++) {
        E2 part[3] = {gl::e2_zero(), gl::e2_zero(), gl::e2_zero()};
        for (auto& kv : groups) {
            uint64_t m[6];
            int rc = ceno_hip_sumcheck_round(ctx, kv.second.sc, i == 0 ? nullptr : ch, m);
            if (rc) {
                cleanup();
                return fail_ctx(ctx, rc);
            }
            for (int e = 0; e < 3; e++) part[e] = part[e] + E2{m[2 * e], m[2 * e + 1]};
        }
        uint64_t mine[6];
        for (int e = 0; e < 3; e++) {
            mine[2 * e] = part[e].c0;
            mine[2 * e + 1] = part[e].c1;
        }
        if (int rc = dist_allgather_words(D.comm, mine, 6, all.data(), D.st)) {
            cleanup();
            return prover_set_error(rc, ceno_dist_last_error());
        }
        uint64_t* msg = msgs + (size_t)6 * i;
        for (int e = 0; e < 3; e++) {
            E2 v = gl::e2_zero();
            for (int g = 0; g < W; g++) v = v + E2{all[(size_t)g * 6 + 2 * e], all[(size_t)g * 6 + 2 * e + 1]};
            msg[2 * e] = v.c0;
            msg[2 * e + 1] = v.c1;
            tr->append_ext(tr->self, msg + 2 * e);
        }
        static const char lbl[] = "Internal round";
        tr->append_label(tr->self, (const uint8_t*)lbl, sizeof(lbl) - 1);
        tr->sample_ext(tr->self, ch);
        chal[2 * i] = ch[0];
        chal[2 * i + 1] = ch[1];
    }
    // ---- the folded tables of every group: as the next round would read them (folded s - 1 times), folded once more here with the last
    // challenge, gathered and interleaved into the global tables of 2^(round - s) entries ----
    const int nv_rem = round - s;
    const size_t len_loc = (size_t)1 << (layer_loc - s);
    const E2 r_last{ch[0], ch[1]};
    // global MLE order: [eq, (a, b) per active product tower, (p1, p2, q1, q2) per active LogUp tower]
    std::vector<std::vector<E2>> tabs;
    std::vector<E2> a_prod, a_num, a_den;
    std::vector<std::pair<int, int>> where_prod((size_t)n_prod, {-1, -1}), where_logup((size_t)n_logup, {-1, -1});  // (group key, first MLE in the group's handle)
    for (auto& kv : groups) {
        int cur = 1;
        for (int i : kv.second.prod) {
            where_prod[(size_t)i] = {kv.first, cur};
            cur += 2;
        }
        for (int i : kv.second.logup) {
            where_logup[(size_t)i] = {kv.first, cur};
            cur += 4;
        }
    }
    std::vector<uint64_t> mine, gathered;
    auto fetch = [&](Group& G, int mle, bool is_eq, std::vector<E2>& out) -> int {
        int nv = 0;
        std::vector<E2> t((size_t)2 * len_loc);
        int rc = ceno_hip_sumcheck_table_host(ctx, G.sc, mle, reinterpret_cast<uint64_t*>(t.data()), t.size(), &nv);
        if (rc) return fail_ctx(ctx, rc);
        if (nv != layer_loc - s + 1) return prover_set_error(CENO_HIP_ERR_STATE, "sharded tower: unexpected table shape after the local rounds");
        mine.resize(2 * len_loc);
        E2 scale = gl::e2_one();
        if (is_eq)  // the handle's eq table carries no rank factor (it went into the alpha powers): the global table does
            for (int j = 0; j < k; j++) {
                const E2 c = rt(G.s_t + j);
                scale = scale * (((D.rank >> j) & 1) ? c : gl::e2_one() - c);
            }
        for (size_t j = 0; j < len_loc; j++) {
            E2 v = t[2 * j] + r_last * (t[2 * j + 1] - t[2 * j]);
            if (is_eq) v = v * scale;
            mine[2 * j] = v.c0;
            mine[2 * j + 1] = v.c1;
        }
        gathered.resize((size_t)W * 2 * len_loc);
        if (int rc2 = dist_allgather_words(D.comm, mine.data(), 2 * len_loc, gathered.data(), D.st)) return prover_set_error(rc2, ceno_dist_last_error());
        interleave(gathered, len_loc, W, G.s_t - s, out);
        return 0;
    };
    int rc = 0;
    {
        std::vector<E2> eq;
        rc = fetch(groups.begin()->second, 0, true, eq);
        tabs.push_back(std::move(eq));
    }
    for (int i = 0; i < n_prod && !rc; i++) {
        if (where_prod[(size_t)i].first < 0) continue;
        Group& G = groups[where_prod[(size_t)i].first];
        for (int b = 0; b < 2 && !rc; b++) {
            std::vector<E2> t;
            rc = fetch(G, where_prod[(size_t)i].second + b, false, t);
            tabs.push_back(std::move(t));
        }
        a_prod.push_back(E2{alpha[2 * i], alpha[2 * i + 1]});
    }
    for (int i = 0; i < n_logup && !rc; i++) {
        if (where_logup[(size_t)i].first < 0) continue;
        Group& G = groups[where_logup[(size_t)i].first];
        for (int b = 0; b < 4 && !rc; b++) {
            std::vector<E2> t;
            rc = fetch(G, where_logup[(size_t)i].second + b, false, t);
            tabs.push_back(std::move(t));
        }
        a_num.push_back(E2{alpha[2 * (n_prod + 2 * i)], alpha[2 * (n_prod + 2 * i) + 1]});
        a_den.push_back(E2{alpha[2 * (n_prod + 2 * i + 1)], alpha[2 * (n_prod + 2 * i + 1) + 1]});
    }
    cleanup();
    if (rc) return rc;
    // ---- the remaining rounds, replicated, on the host ----
    prover_host_tower_rounds(nv_rem, tabs, (int)a_prod.size(), (int)a_num.size(), a_prod, a_num, a_den, tr, msgs + (size_t)6 * s, chal + (size_t)2 * s, fin);
    return 0;
}

}  // namespace

extern "C" {

int ceno_dist_chip_block_log(void) {
    const char* e = getenv("CENO_DIST_ROW_BLOCK_LOG");  // q: rank g holds the rows whose bits [q, q + log2 world) equal g
    return e ? std::max(1, atoi(e)) : 10;
}

int ceno_dist_create_chip_proof(ceno_hip_ctx* ctx, ceno_dist_comm* comm, const ceno_chip_task* task, int log2_num_instances_global,
                                int row_block_log, const uint64_t* challenges4, ceno_transcript* tr, ceno_hip_stream s, ceno_chip_proof* out) {
    if (!ctx || !task || !challenges4 || !tr || !out) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_create_chip_proof: NULL argument");
    const int W = dist_comm_world(comm), rank = dist_comm_rank(comm), k = ceil_log2((size_t)W);
    if (W == 1) return ceno_prover_create_chip_proof(ctx, task, challenges4, tr, s, out);
    if (((size_t)1 << k) != (size_t)W) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_create_chip_proof: the number of ranks must be a power of two");
    const int q = row_block_log > 0 ? row_block_log : ceno_dist_chip_block_log();
    // a keccak-style chip has 2^rotation_vars rows per instance (prover.rs:728-729): the ROWS are what is sharded
    const int rot_vars = std::max(task->rotation_vars, 0);
    const int n = log2_num_instances_global + rot_vars, n_loc = n - k;
    if (n_loc < q + 1) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_create_chip_proof: the chip is too small for this block size (needs log2 rows >= q + log2 world + 1)");
    if (task->log2_num_instances + rot_vars != n_loc)
        return prover_set_error(CENO_HIP_ERR_INVALID, "dist_create_chip_proof: task->log2_num_instances (+ rotation_vars) must be the LOCAL height");
    if (task->n_rotation_pairs > 0 && (q < task->cyclic_group_log2 || !task->rotation_source_idx || !task->rotation_target_idx))
        return prover_set_error(CENO_HIP_ERR_INVALID, "dist_create_chip_proof: the row blocks must hold whole cyclic groups of the rotation (q >= cyclic_group_log2)");
    memset(out, 0, sizeof(*out));
    hipStream_t st = (hipStream_t)s;
    const int n_mles = task->n_witin + task->n_fixed + task->n_structural;
    const int n_lk_num = task->num_lk_tables, n_lk_den = task->num_lk_tables > 0 ? task->num_lk_tables : task->num_lk;
    const int n_records = task->num_reads + task->num_writes + n_lk_num + n_lk_den;
    if (n_records < 1 || n_mles < 1 || !task->mles) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_create_chip_proof: bad task");
    out->num_instances = task->num_instances;
    // ---- record inference on the local rows (elementwise) ----
    std::vector<ceno_hip_mle*> present;
    std::vector<uint32_t> remap((size_t)n_mles, UINT32_MAX), ridx;
    for (int j = 0; j < n_mles; j++)
        if (task->mles[j]) {
            if (ceno_hip_mle_num_vars(task->mles[j]) != n_loc) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_create_chip_proof: a local table has the wrong height");
            remap[(size_t)j] = (uint32_t)present.size();
            present.push_back(task->mles[j]);
        }
    const uint32_t n_factors = task->record_term_offsets[task->n_record_terms];
    for (uint32_t x = 0; x < n_factors; x++) {
        const uint32_t j = task->record_term_mle_idx[x];
        if ((int)j >= n_mles || remap[j] == UINT32_MAX) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_create_chip_proof: a record expression reads an absent table");
        ridx.push_back(remap[j]);
    }
    std::vector<ceno_hip_mle*> records((size_t)n_records, nullptr);
    int rc = ceno_hip_wit_infer(ctx, present.data(), (int)present.size(), task->record_coeffs, task->record_term_offsets, ridx.data(), task->n_record_terms,
                                task->record_out_term_offsets, n_records, n_loc, s, records.data());
    if (rc) return fail_ctx(ctx, rc);
    // ---- local towers = the shards of the global towers' large layers ----
    ceno_tower_witness tw_loc;
    rc = ceno_prover_build_tower_witness(ctx, records.data(), task->num_reads, task->num_writes, task->num_lk_tables, task->num_lk, n_loc - rot_vars, rot_vars,
                                         challenges4, s, &tw_loc);
    for (auto* m : records)
        if (m) ceno_hip_mle_free(ctx, m);
    if (rc) return rc;
    DistTowers D;
    D.ctx = ctx;
    D.comm = comm;
    D.st = st;
    D.world = W;
    D.rank = rank;
    D.k = k;
    const int c_r = ceil_log2((size_t)std::max(task->num_reads, 1)), c_w = ceil_log2((size_t)std::max(task->num_writes, 1)), c_lk = ceil_log2((size_t)std::max(n_lk_den, 1));
    {
        int ip = 0;
        if (tw_loc.has_r) {
            D.loc_prod.push_back(tw_loc.prod[ip++]);
            D.s_of.push_back(q + c_r);
        }
        if (tw_loc.has_w) {
            D.loc_prod.push_back(tw_loc.prod[ip++]);
            D.s_of.push_back(q + c_w);
        }
        if (tw_loc.has_lk) {
            D.loc_logup.push_back(tw_loc.logup[0]);
            D.s_of.push_back(q + c_lk);
        }
    }
    const int n_prod = (int)D.loc_prod.size(), n_logup = (int)D.loc_logup.size(), n_t = n_prod + n_logup;
    int s_max = 0;
    D.s_min = 1 << 30;
    for (int t = 0; t < n_t; t++) {
        ceno_hip_tower* T = t < n_prod ? D.loc_prod[(size_t)t] : D.loc_logup[(size_t)(t - n_prod)];
        D.nv_global.push_back(ceno_hip_tower_num_vars(T) + k);
        s_max = std::max(s_max, D.s_of[(size_t)t]);
        D.s_min = std::min(D.s_min, D.s_of[(size_t)t]);
    }
    // ---- the replicated tops: global layer G_t = min(r_rep, nv_t - 1) of every tower, gathered from the local layer G_t - k ----
    const int r_rep = s_max + k;
    std::vector<ceno_hip_tower*> top((size_t)n_t, nullptr);
    std::vector<ceno_hip_mle*> keep;  // limbs uploaded for from_last_layer (the towers copy them)
    auto free_all = [&]() {
        for (auto* m : keep) ceno_hip_mle_free(ctx, m);
        keep.clear();
        for (auto*& T : top) {
            if (T) ceno_hip_tower_free(ctx, T);
            T = nullptr;
        }
        ceno_tower_witness_free(ctx, &tw_loc);
    };
    std::vector<uint64_t> mine, gathered;
    std::vector<E2> glob;
    for (int t = 0; t < n_t && !rc; t++) {
        ceno_hip_tower* T = t < n_prod ? D.loc_prod[(size_t)t] : D.loc_logup[(size_t)(t - n_prod)];
        const int n_limbs = ceno_hip_tower_num_limbs(T);
        const int G_t = std::min(r_rep, D.nv_global[(size_t)t] - 1), l_loc = G_t - k;
        if (l_loc < D.s_of[(size_t)t]) {  // cannot happen with n_loc >= q + 1 (the leaf layer has at least s_t variables)
            rc = prover_set_error(CENO_HIP_ERR_STATE, "dist_create_chip_proof: a tower is shorter than its shard position");
            break;
        }
        const size_t len_loc = (size_t)1 << l_loc;
        std::vector<ceno_hip_mle*> limbs((size_t)n_limbs, nullptr);
        for (int b = 0; b < n_limbs && !rc; b++) {
            ceno_hip_mle* L = nullptr;
            rc = ceno_hip_tower_layer(ctx, T, l_loc, b, &L);
            if (rc) { rc = fail_ctx(ctx, rc); break; }
            mine.resize(2 * len_loc);
            if (hipMemcpyAsync(mine.data(), ceno_hip_mle_device_ptr(L), len_loc * 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                rc = prover_set_error(CENO_HIP_ERR_HIP, "dist_create_chip_proof: layer download failed");
                break;
            }
            gathered.resize((size_t)W * 2 * len_loc);
            if (int rc2 = dist_allgather_words(comm, mine.data(), 2 * len_loc, gathered.data(), st)) { rc = prover_set_error(rc2, ceno_dist_last_error()); break; }
            interleave(gathered, len_loc, W, D.s_of[(size_t)t], glob);
            rc = ceno_hip_mle_upload(ctx, reinterpret_cast<const uint64_t*>(glob.data()), G_t, 1, s, &limbs[(size_t)b]);
            if (rc) { rc = fail_ctx(ctx, rc); break; }
            keep.push_back(limbs[(size_t)b]);
        }
        if (!rc) {
            rc = ceno_hip_tower_from_last_layer(ctx, limbs.data(), n_limbs, s, &top[(size_t)t]);
            if (rc) rc = fail_ctx(ctx, rc);
        }
    }
    if (rc) {
        free_all();
        return rc;
    }
    // ---- the proof: out-evaluations from the replicated tops, layers up to r_rep on them, the large layers through the hook ----
    int max_nv = 0;
    for (int v : D.nv_global) max_nv = std::max(max_nv, v);
    const int R = max_nv - 1;
    out->tower_num_vars = max_nv;
    out->n_prod = n_prod;
    out->n_logup = n_logup;
    out->tower.msgs = (uint64_t*)calloc(std::max<size_t>(1, ceno_tower_msgs_words(max_nv)), 8);
    out->tower.prod_evals = (uint64_t*)calloc((size_t)std::max(1, n_prod) * std::max(1, R) * 4, 8);
    out->tower.logup_evals = (uint64_t*)calloc((size_t)std::max(1, n_logup) * std::max(1, R) * 8, 8);
    out->tower.point = (uint64_t*)calloc((size_t)2 * (max_nv + 1), 8);
    out->rt_main = (uint64_t*)calloc((size_t)2 * std::max(1, n), 8);
    if (!out->tower.msgs || !out->tower.prod_evals || !out->tower.logup_evals || !out->tower.point || !out->rt_main) {
        free_all();
        ceno_chip_proof_free(out);
        return prover_set_error(CENO_HIP_ERR_OOM, "dist_create_chip_proof: out of host memory");
    }
    {
        rc = ceno_hip_tower_prefetch_tops(ctx, top.data(), n_t, 12, s);
        // prove_tower_relation: the out-evaluations first (cpu/mod.rs:783-786), in the order r, w, lk
        int ip = 0;
        if (!rc && tw_loc.has_r) rc = ceno_hip_tower_out_evals(ctx, top[(size_t)ip++], out->r_out_evals, s);
        if (!rc && tw_loc.has_w) rc = ceno_hip_tower_out_evals(ctx, top[(size_t)ip++], out->w_out_evals, s);
        if (!rc && tw_loc.has_lk) rc = ceno_hip_tower_out_evals(ctx, top[(size_t)ip++], out->lk_out_evals, s);
        if (rc) {
            rc = fail_ctx(ctx, rc);
            free_all();
            ceno_chip_proof_free(out);
            return rc;
        }
        out->n_r_out = tw_loc.has_r ? 2 : 0;
        out->n_w_out = tw_loc.has_w ? 2 : 0;
        out->n_lk_out = tw_loc.has_lk ? 4 : 0;
        if (tw_loc.has_r) for (int e = 0; e < 2; e++) tr->append_ext(tr->self, out->r_out_evals + 2 * e);
        if (tw_loc.has_w) for (int e = 0; e < 2; e++) tr->append_ext(tr->self, out->w_out_evals + 2 * e);
        if (tw_loc.has_lk) for (int e = 0; e < 4; e++) tr->append_ext(tr->self, out->lk_out_evals + 2 * e);
    }
    TowerDistHook hook;
    hook.r_rep = r_rep;
    hook.nv_global = D.nv_global.data();
    hook.layer = dist_layer;
    hook.self = &D;
    rc = prover_tower_create_proof_hooked(ctx, top.data(), n_prod, top.data() + n_prod, n_logup, tr, s, &out->tower, &hook);
    free_all();
    if (rc) {
        ceno_chip_proof_free(out);
        return rc;
    }
    out->num_var_with_rotation = n;
    if (max_nv < n) {
        ceno_chip_proof_free(out);
        return prover_set_error(CENO_HIP_ERR_STATE, "tower challenge point is shorter than the main point");
    }
    memcpy(out->rt_main, out->tower.point + (size_t)2 * (max_nv - n), (size_t)16 * n);
    // ---- prove_rotation (prover.rs:771-776; keccak-style chips): local rotations and local rounds, the tail replicated ----
    if (task->n_rotation_pairs > 0) {
        const int np = task->n_rotation_pairs;
        out->n_rotation_pairs = np;
        out->rotation_msgs = (uint64_t*)calloc((size_t)n * 2 * 2, 8);
        out->rotation_evals = (uint64_t*)calloc((size_t)3 * np * 2, 8);
        out->rotation_points = (uint64_t*)calloc((size_t)3 * n * 2, 8);  // origin | left | right
        if (!out->rotation_msgs || !out->rotation_evals || !out->rotation_points) {
            ceno_chip_proof_free(out);
            return prover_set_error(CENO_HIP_ERR_OOM, "dist_create_chip_proof: out of host memory");
        }
        struct Gather {
            ceno_dist_comm* comm;
            hipStream_t st;
        } gth{comm, st};
        RotationShard sh{W, rank, k, q,
                         [](void* self, const uint64_t* mine, size_t n_words, uint64_t* all) -> int {
                             auto* G = static_cast<Gather*>(self);
                             if (int rc2 = dist_allgather_words(G->comm, mine, n_words, all, G->st)) return prover_set_error(rc2, ceno_dist_last_error());
                             return 0;
                         },
                         &gth};
        rc = prover_prove_rotation_sharded(ctx, task->mles, task->rotation_source_idx, task->rotation_target_idx, np, task->cyclic_subgroup_size,
                                           task->cyclic_group_log2, out->rt_main, n, tr, s, out->rotation_msgs, out->rotation_evals, out->rotation_points,
                                           out->rotation_points + (size_t)2 * n, out->rotation_points + (size_t)4 * n, &sh);
        if (rc) {
            ceno_chip_proof_free(out);
            return rc;
        }
    }
    return 0;
}

// prove_batched_main_constraints over tables in the block layout of ceno_dist_create_chip_proof (main_constraints.cpp
// prover_main_constraints_sharded): every job's tables hold this rank's rows, J.num_vars stays the global number of variables
int ceno_dist_prove_batched_main_constraints(ceno_hip_ctx* ctx, ceno_dist_comm* comm, const ceno_main_job* jobs_local, int n_jobs, int row_block_log,
                                             const uint64_t* global_challenges4, ceno_transcript* tr, ceno_hip_stream s, uint64_t* out_claimed_sum,
                                             uint64_t* out_msgs, uint64_t* out_global_rt, uint64_t* out_evals, int* out_num_vars, int* out_degree) {
    if (!ctx || !jobs_local || n_jobs < 1 || !tr) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_prove_batched_main_constraints: NULL argument");
    const int W = dist_comm_world(comm), rank = dist_comm_rank(comm), k = ceil_log2((size_t)W);
    if (W == 1)
        return ceno_prover_prove_batched_main_constraints(ctx, jobs_local, n_jobs, global_challenges4, tr, s, out_claimed_sum, out_msgs, out_global_rt, out_evals,
                                                          out_num_vars, out_degree);
    if (((size_t)1 << k) != (size_t)W) return prover_set_error(CENO_HIP_ERR_INVALID, "dist_prove_batched_main_constraints: the number of ranks must be a power of two");
    const int q = row_block_log > 0 ? row_block_log : ceno_dist_chip_block_log();
    struct Gather {
        ceno_dist_comm* comm;
        hipStream_t st;
    } gth{comm, (hipStream_t)s};
    RotationShard sh{W, rank, k, q,
                     [](void* self, const uint64_t* mine, size_t n_words, uint64_t* all) -> int {
                         auto* G = static_cast<Gather*>(self);
                         if (int rc2 = dist_allgather_words(G->comm, mine, n_words, all, G->st)) return prover_set_error(rc2, ceno_dist_last_error());
                         return 0;
                     },
                     &gth};
    return prover_main_constraints_sharded(ctx, jobs_local, n_jobs, global_challenges4, tr, s, out_claimed_sum, out_msgs, out_global_rt, out_evals, out_num_vars,
                                           out_degree, &sh);
}

}  // extern "C"
