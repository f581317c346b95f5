// On-device witness generation for the chips ADD / SUB, AND / OR / XOR (R-type) and ADDI, ANDI / ORI / XORI, LUI, AUIPC (I-type base), JAL, SLT / SLTU, SLTI / SLTIU, the six branches, JALR, the six shifts, the five loads, the three stores, the four multiplications and the four divisions.
//
// One lane per instance: read the step record, compute the 22 witness words of the row exactly as the reference's
// CPU assignment does (ceno_zkvm/src/instructions/riscv/arith.rs:101-142, r_insn.rs:67-86, insn_base.rs:61-77,
// 112-145, 223-257, 337-400; AssertLt diff limbs gkr_iop/src/gadgets/is_lt.rs:243-287), scatter them into the
// column-major matrix through the caller's column map, and count the lookups of the row.
// HBM-bound integer work: 136 B read + 8 * num_cols B written per row; consecutive lanes write consecutive rows of a
// column (coalesced), the 136-byte records are fetched with 8-byte loads that the L2 merges.
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"

#include <algorithm>

namespace {

constexpr int NT = 256;
constexpr unsigned MAXB = 4096;
constexpr uint64_t SUBCYCLES_PER_INSN = 4;  // ceno_emul/src/tracer.rs:574-578
constexpr uint64_t SUBCYCLE_RS1 = 0, SUBCYCLE_RS2 = 1, SUBCYCLE_RD = 2;
constexpr int MAX_TS_BITS = 29;  // gkr_iop/src/circuit_builder/ram.rs:13

// byte offsets inside ceno_emul::StepRecord (#[repr(C)], tracer.rs:33-60; Instruction rv32im.rs:115-128; MemOp tracer.rs:634-644)
constexpr int OFF_CYCLE = 0, OFF_PC_BEFORE = 8;
constexpr int OFF_RS1 = 48, OFF_RS2 = 64, OFF_RD = 80;  // ReadOp {addr u32, value u32, previous_cycle u64}; WriteOp {addr, before, after, pad, previous_cycle}

struct Map {  // common layout of ceno_hip_add_column_map / ceno_hip_sub_column_map
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t a[2], b[2], carries[2];
    uint32_t num_cols;
};
static_assert(sizeof(Map) == sizeof(ceno_hip_add_column_map) && sizeof(Map) == sizeof(ceno_hip_sub_column_map), "column map layout");

// ShardContext::aligned_prev_ts (ceno_zkvm/src/e2e.rs:435-441)
__device__ __forceinline__ uint64_t aligned_prev_ts(uint64_t prev_cycle, uint64_t offset) {
    uint64_t ts = prev_cycle > offset ? prev_cycle - offset : 0;
    return ts < SUBCYCLES_PER_INSN ? 0 : ts;
}
// cal_lt_diff (gkr_iop/src/gadgets/is_lt.rs:277-287): lhs - rhs, plus 2^max_bits when lhs < rhs
__device__ __forceinline__ uint64_t lt_diff(uint64_t lhs, uint64_t rhs) { return (lhs < rhs ? (1ull << MAX_TS_BITS) : 0ull) + lhs - rhs; }

// counter += 1 per lane, merged per wave (lk_count below).
// XCD_LOCAL: `table` is this XCD's private copy, so the add only has to be atomic inside the XCD's L2 (workgroup-scope
// read-modify-write, no sc1: it never leaves the L2); a device-scope atomic is executed on the fabric side of the L2 at
// ~15 G/s chip-wide, which is what bounds this kernel (5 distinct-slot counts per instance).
template <bool XCD_LOCAL>
__device__ __forceinline__ void lk_add(uint32_t* p, uint32_t v) {
    if (XCD_LOCAL) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAdd(p, v);
}
// One atomic per DISTINCT slot of the wave.  First the common case: every active lane on the slot of the first one (timestamp-difference high
// limbs are constant across a chip) — one ballot, one atomic.  Otherwise each lane finds the lanes that hold its own slot, bit by bit (a ballot per
// bit in which the wave's slots differ at all: __match_any over wave64), and the lowest lane of every group adds the group's size.  Few-valued keys
// (a 3-bit range check, the carries of a shift by few bits, the zero of a special case) put a whole chip on a handful of L2 lines: with only the
// first lane's group merged, SRA's counts cost 1.74 ms per 2^20 instances (0.11 ms for its witness columns), DIV's 1.9 ms; now 0.80 and 0.58 ms,
// ADD's spread keys unchanged at 0.20 ms.  (A workgroup-level LDS cache of hot slots in front of the table was measured too: the extra LDS atomics
// of the spread keys cost more than the hot lines save — ADD 0.29 ms, SRA 0.90 ms — and was dropped.)
template <bool XCD_LOCAL>
__device__ __forceinline__ void lk_count(uint32_t* table, uint32_t slot) {
    if (!table) return;
    const uint64_t active = __ballot(1);
    const uint32_t first = __builtin_amdgcn_readfirstlane(slot);
    const uint64_t same = __ballot(slot == first);
    if (same == active) {  // wave-uniform
        if ((int)__lane_id() == __ffsll((long long)active) - 1) lk_add<XCD_LOCAL>(table + slot, (uint32_t)__popcll(active));
        return;
    }
    uint64_t peers = active;
#pragma unroll
    for (int b = 0; b < 32; b++) {
        const bool bit = (slot >> b) & 1u;
        const uint64_t m = __ballot(bit);
        if (m == 0 || m == active) continue;  // wave-uniform bit: nothing to separate
        peers &= bit ? m : ~m;
    }
    if ((int)__lane_id() == __ffsll((long long)peers) - 1) lk_add<XCD_LOCAL>(table + slot, (uint32_t)__popcll(peers));
}
// dst[i] += sum over the 8 per-XCD copies (device-scope atomics: other chips of the shard may be adding to dst concurrently)
// (`used` <= `slots`: the chip only counts into the first `used` entries of each copy)
__global__ void __launch_bounds__(NT) k_lk_merge(const uint32_t* __restrict__ copies, size_t slots, size_t used, uint32_t* dst) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= used) return;
    uint32_t s = 0;
#pragma unroll
    for (int x = 0; x < 8; x++) s += copies[(size_t)x * slots + i];
    if (s) atomicAdd(dst + i, s);
}

// the `mlt` witness column of a table circuit from its counters: column[i] = counters[i] as a field element (a count is < 2^32 < p), zero
// beyond the table (InstancePaddingStrategy::Default); ceno_zkvm/src/tables/ops/ops_impl.rs:85-100, tables/range/range_impl.rs:60-96
__global__ void __launch_bounds__(NT) k_lk_to_mlt(const uint32_t* __restrict__ counters, size_t n, uint64_t* __restrict__ column, size_t rows) {
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < rows; i += stride) column[i] = i < n ? (uint64_t)counters[i] : 0ull;
}

// XCD_LOCAL: lk_dyn / lk_fetch point at 8 consecutive copies of the tables, one per XCD (HW_REG_XCC_ID picks this wave's)
template <bool SUB, bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_arith(Map m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                     uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w,
                                                     size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch) {
    if (XCD_LOCAL) {
        // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, 4 bits): simm16 = id | offset << 6 | (size - 1) << 11
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
        if (lk_dyn) lk_dyn += (size_t)xcc * CENO_HIP_LK_DYNAMIC_SLOTS;
        if (lk_fetch) lk_fetch += (size_t)xcc * fetch_slots;
    }
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        if (r >= n) {  // padding rows: every mapped column is zero
            const uint32_t* cols = &m.pc;
#pragma unroll
            for (int c = 0; c < 22; c++) w[(size_t)cols[c] * rows + r] = 0;
            continue;
        }
        const uint64_t* q = reinterpret_cast<const uint64_t*>(recs + (size_t)idx[r] * CENO_HIP_STEP_RECORD_BYTES);
        const uint64_t cycle = q[OFF_CYCLE / 8];
        const uint32_t pc = (uint32_t)q[OFF_PC_BEFORE / 8];
        const uint64_t rs1_av = q[OFF_RS1 / 8], rs1_prev = q[OFF_RS1 / 8 + 1];
        const uint64_t rs2_av = q[OFF_RS2 / 8], rs2_prev = q[OFF_RS2 / 8 + 1];
        const uint64_t rd_ab = q[OFF_RD / 8], rd_after_w = q[OFF_RD / 8 + 1], rd_prev = q[OFF_RD / 8 + 2];
        const uint32_t rs1_addr = (uint32_t)rs1_av, rs1_val = (uint32_t)(rs1_av >> 32);  // rs1_val: ADD operand (unused by SUB)
        const uint32_t rs2_addr = (uint32_t)rs2_av, rs2_val = (uint32_t)(rs2_av >> 32);
        const uint32_t rd_addr = (uint32_t)rd_ab, rd_before = (uint32_t)(rd_ab >> 32), rd_after = (uint32_t)rd_after_w;
        const uint64_t ts = cycle - offset;
        auto put = [&](uint32_t col, uint64_t v) { w[(size_t)col * rows + r] = v; };
        put(m.pc, pc);
        put(m.ts, ts);
        // register index = (word address * 4) >> 8 as u8 (ceno_emul/src/platform.rs:120-128, tracer.rs:656-658)
        const uint64_t p1 = aligned_prev_ts(rs1_prev, offset), p2 = aligned_prev_ts(rs2_prev, offset), pd = aligned_prev_ts(rd_prev, offset);
        const uint64_t d1 = lt_diff(p1, ts + SUBCYCLE_RS1), d2 = lt_diff(p2, ts + SUBCYCLE_RS2), dd = lt_diff(pd, ts + SUBCYCLE_RD);
        put(m.rs1_id, ((rs1_addr << 2) >> 8) & 0xff);
        put(m.rs1_prev_ts, p1);
        put(m.rs1_lt_diff[0], d1 & 0xffff);
        put(m.rs1_lt_diff[1], (d1 >> 16) & 0xffff);
        put(m.rs2_id, ((rs2_addr << 2) >> 8) & 0xff);
        put(m.rs2_prev_ts, p2);
        put(m.rs2_lt_diff[0], d2 & 0xffff);
        put(m.rs2_lt_diff[1], (d2 >> 16) & 0xffff);
        put(m.rd_id, ((rd_addr << 2) >> 8) & 0xff);
        put(m.rd_prev_ts, pd);
        put(m.rd_prev_val[0], rd_before & 0xffff);
        put(m.rd_prev_val[1], rd_before >> 16);
        put(m.rd_lt_diff[0], dd & 0xffff);
        put(m.rd_lt_diff[1], (dd >> 16) & 0xffff);
        // ADD: rs1 + rs2 = rd (limbs of rs1, rs2; carries of the sum).  SUB: rs2 + rd = rs1 (limbs of rs2, rd; carries of that sum)
        const uint32_t x = SUB ? rs2_val : rs1_val, y = SUB ? rd_after : rs2_val;
        put(m.a[0], x & 0xffff);
        put(m.a[1], x >> 16);
        put(m.b[0], y & 0xffff);
        put(m.b[1], y >> 16);
        const uint32_t s0 = (x & 0xffff) + (y & 0xffff);
        const uint32_t s1 = (x >> 16) + (y >> 16) + (s0 >> 16);
        put(m.carries[0], s0 >> 16);  // Value::add with_overflow (ceno_zkvm/src/uint.rs:762-785)
        put(m.carries[1], s1 >> 16);
        // lookups: fetch, per register access u16(diff limb 0) + 13-bit range(diff limb 1), u16 limbs of the sum
        // (SUB: also of rd, Value::new uint.rs:684-688)
        if (lk_fetch) {
            const uint32_t slot = (pc - fetch_base) >> 2;
            if (slot < fetch_slots) lk_count<XCD_LOCAL>(lk_fetch, slot);
        }
        constexpr uint32_t U16 = 1u << 16, R13 = 1u << (MAX_TS_BITS - 16);
        lk_count<XCD_LOCAL>(lk_dyn, U16 + (uint32_t)(d1 & 0xffff));
        lk_count<XCD_LOCAL>(lk_dyn, R13 + (uint32_t)((d1 >> 16) & 0xffff));
        lk_count<XCD_LOCAL>(lk_dyn, U16 + (uint32_t)(d2 & 0xffff));
        lk_count<XCD_LOCAL>(lk_dyn, R13 + (uint32_t)((d2 >> 16) & 0xffff));
        lk_count<XCD_LOCAL>(lk_dyn, U16 + (uint32_t)(dd & 0xffff));
        lk_count<XCD_LOCAL>(lk_dyn, R13 + (uint32_t)((dd >> 16) & 0xffff));
        lk_count<XCD_LOCAL>(lk_dyn, U16 + (s0 & 0xffff));  // limbs of the addition's result, range-checked inside Value::add
        lk_count<XCD_LOCAL>(lk_dyn, U16 + (s1 & 0xffff));
        if (SUB) {
            lk_count<XCD_LOCAL>(lk_dyn, U16 + (rd_after & 0xffff));
            lk_count<XCD_LOCAL>(lk_dyn, U16 + (rd_after >> 16));
        }
    }
}

// ---- pieces every chip shares (the new chips below are written with them; k_witgen_arith above keeps its hand-scheduled body) ----
struct Row {  // one instance's row of the column-major witness
    uint64_t* w;
    size_t rows, r;
    __device__ __forceinline__ void put(uint32_t col, uint64_t v) const { w[(size_t)col * rows + r] = v; }
};
struct Step {  // the fields of a StepRecord the chips read
    uint64_t cycle, rs1_prev, rs2_prev, rd_prev;
    uint32_t pc, imm, rs1_addr, rs1_val, rs2_addr, rs2_val, rd_addr, rd_before, rd_after;
};
__device__ __forceinline__ Step load_step(const unsigned char* recs, uint32_t index) {
    const uint64_t* q = reinterpret_cast<const uint64_t*>(recs + (size_t)index * CENO_HIP_STEP_RECORD_BYTES);
    Step s;
    s.cycle = q[OFF_CYCLE / 8];
    s.pc = (uint32_t)q[OFF_PC_BEFORE / 8];
    s.imm = (uint32_t)(q[4] >> 32);  // Instruction { kind u8, rs1 u8, rs2 u8, rd u8, imm i32, raw u32 } at byte 32: imm = bytes 36..39 (rv32im.rs:115-128)
    const uint64_t rs1_av = q[OFF_RS1 / 8], rs2_av = q[OFF_RS2 / 8], rd_ab = q[OFF_RD / 8];
    s.rs1_prev = q[OFF_RS1 / 8 + 1];
    s.rs2_prev = q[OFF_RS2 / 8 + 1];
    s.rd_prev = q[OFF_RD / 8 + 2];
    s.rs1_addr = (uint32_t)rs1_av;
    s.rs1_val = (uint32_t)(rs1_av >> 32);
    s.rs2_addr = (uint32_t)rs2_av;
    s.rs2_val = (uint32_t)(rs2_av >> 32);
    s.rd_addr = (uint32_t)rd_ab;
    s.rd_before = (uint32_t)(rd_ab >> 32);
    s.rd_after = (uint32_t)q[OFF_RD / 8 + 1];
    return s;
}
// this wave's copy of a per-XCD table (s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, 4 bits))
template <bool XCD_LOCAL>
__device__ __forceinline__ uint32_t* xcd_copy(uint32_t* table, size_t slots) {
    if (!XCD_LOCAL || !table) return table;
    return table + (size_t)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u) * slots;
}
// ReadRS1 / ReadRS2 (insn_base.rs:112-145,223-257): register id, aligned previous timestamp, AssertLt(prev_ts < ts + sub_cycle) limbs
template <bool XCD_LOCAL>
__device__ __forceinline__ void emit_read(const Row& o, uint32_t id_col, uint32_t prev_col, const uint32_t (&diff_cols)[2], uint32_t addr,
                                          uint64_t prev_cycle, uint64_t offset, uint64_t ts_sub, uint32_t* lk_dyn) {
    const uint64_t p = aligned_prev_ts(prev_cycle, offset), d = lt_diff(p, ts_sub);
    o.put(id_col, ((addr << 2) >> 8) & 0xff);  // register index = (word address * 4) >> 8 as u8 (platform.rs:120-128, tracer.rs:656-658)
    o.put(prev_col, p);
    o.put(diff_cols[0], d & 0xffff);
    o.put(diff_cols[1], (d >> 16) & 0xffff);
    lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)(d & 0xffff));
    lk_count<XCD_LOCAL>(lk_dyn, (1u << (MAX_TS_BITS - 16)) + (uint32_t)((d >> 16) & 0xffff));
}
// WriteRD (insn_base.rs:337-400): as a read, plus the previous value as two u16 limbs
template <bool XCD_LOCAL>
__device__ __forceinline__ void emit_write(const Row& o, uint32_t id_col, uint32_t prev_col, const uint32_t (&prev_val_cols)[2],
                                           const uint32_t (&diff_cols)[2], uint32_t addr, uint32_t before, uint64_t prev_cycle, uint64_t offset,
                                           uint64_t ts_sub, uint32_t* lk_dyn) {
    o.put(prev_val_cols[0], before & 0xffff);
    o.put(prev_val_cols[1], before >> 16);
    emit_read<XCD_LOCAL>(o, id_col, prev_col, diff_cols, addr, prev_cycle, offset, ts_sub, lk_dyn);
}
template <bool XCD_LOCAL>
__device__ __forceinline__ void emit_fetch(uint32_t* lk_fetch, uint32_t pc, uint32_t fetch_base, uint32_t fetch_slots) {
    if (!lk_fetch) return;
    const uint32_t slot = (pc - fetch_base) >> 2;
    if (slot < fetch_slots) lk_count<XCD_LOCAL>(lk_fetch, slot);
}
template <int N_COLS>
__device__ __forceinline__ void zero_row(const Row& o, const uint32_t* cols) {  // padding rows: every mapped column is zero
#pragma unroll
    for (int c = 0; c < N_COLS; c++) o.put(cols[c], 0);
}

// ---- host side shared by every chip --------------------------------------------------------------------------------------------
int witgen_check(ceno_hip_ctx* ctx, const uint32_t* cols, int n_cols, uint32_t num_cols, const void* recs, size_t num_records, const uint32_t* idx,
                 size_t n, const uint64_t* w, size_t rows, const uint32_t* lk_fetch, uint32_t fetch_slots) {
    CHECK_ARG(ctx, cols && w && rows > 0 && n <= rows, "bad witgen arguments");
    CHECK_ARG(ctx, n == 0 || (recs && idx && num_records > 0), "witgen: records / indices missing");
    CHECK_ARG(ctx, num_cols >= (uint32_t)n_cols, "witgen: this chip has %d mapped columns", n_cols);
    uint64_t seen[4] = {0, 0, 0, 0};
    for (int c = 0; c < n_cols; c++) {
        CHECK_ARG(ctx, cols[c] < num_cols, "witgen: column id out of range");
        if (cols[c] < 256) {
            CHECK_ARG(ctx, !(seen[cols[c] >> 6] >> (cols[c] & 63) & 1), "witgen: duplicate column id");
            seen[cols[c] >> 6] |= 1ull << (cols[c] & 63);
        }
    }
    CHECK_ARG(ctx, lk_fetch == nullptr || fetch_slots > 0, "witgen: fetch table without slots");
    return 0;
}
// The lookup counts go through per-XCD copies of the tables (zeroed scratch, L2-local atomics, one merge per table) unless
// CENO_HIP_WITGEN_XCD=0; `launch(xcd_local, t0, t1, t2, t3)` starts the chip's kernel on the tables it is given.  With tables the call
// synchronises the stream (the scratch goes back to the pool only after the stream has consumed it).
struct LkTab {
    uint32_t* user;
    size_t slots;
    size_t used = 0;  // entries of each copy that are cleared and merged (0 = all): the dynamic table has 2^19 entries (range checks of up to
                      // DYNAMIC_RANGE_MAX_BITS = 18 bits), every chip but the multiplications stays below 2^17
};
constexpr size_t DYN_USED_16 = (size_t)1 << 17;
// ---- a SHARD's witness generation as one session (ceno_hip_witgen_session_begin / _end, include/ceno_hip.h) ----
// A shard runs ~45 chips against the same handful of lookup tables.  Chip by chip, every call clears, merges and waits for its own eight
// per-XCD copies (16 MB of memset, four merge launches and a stream synchronisation per chip).  Inside a session the copies of the
// registered tables live as long as the session: cleared once, counted into by every chip's kernel back to back on the stream (no wait,
// no merge), merged once at the end.  The per-chip entry points are unchanged: witgen_run recognises a registered table by its pointer.
struct WitgenSession {
    struct Tab {
        uint32_t* user;
        size_t slots;
        uint32_t* copies;  // 8 x slots
    };
    std::vector<Tab> tabs;
    void* scratch = nullptr;
    hipStream_t stream = nullptr;
    // chips may run on OTHER streams too (a shard's ~45 witness kernels are small: one after the other they are ~1.3 ms of device time, over four
    // streams a third of that): a stream seen for the first time waits for the session's set-up (`begun`), the session's end waits for all of them
    hipEvent_t begun = nullptr;
    std::vector<hipStream_t> others;
};
std::mutex g_sessions_mu;
std::map<ceno_hip_ctx*, WitgenSession> g_sessions;

template <class Launch>
int witgen_run(ceno_hip_ctx* ctx, hipStream_t st, size_t n, const LkTab (&tabs)[4], Launch&& launch) {
    static const bool xcd_wanted = [] { const char* e = getenv("CENO_HIP_WITGEN_XCD"); return !(e && atoi(e) == 0); }();
    const bool xcd_local = xcd_wanted && ctx->xcd_private_l2;  // (gfx942 / gfx950 only: ctx.hip)
    const bool any = tabs[0].user || tabs[1].user || tabs[2].user || tabs[3].user;
    if (xcd_local && any && n > 0) {
        std::lock_guard<std::mutex> g(g_sessions_mu);
        auto it = g_sessions.find(ctx);
        if (it != g_sessions.end()) {
            WitgenSession& S = it->second;
            uint32_t* copy[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int t = 0; t < 4; t++) {
                if (!tabs[t].user) continue;
                for (const auto& T : S.tabs)
                    if (T.user == tabs[t].user) copy[t] = T.copies;
                CHECK_ARG(ctx, copy[t] != nullptr, "witgen: a lookup table that is not registered with the open witgen session");
                for (const auto& T : S.tabs)
                    if (T.user == tabs[t].user) CHECK_ARG(ctx, T.slots == tabs[t].slots, "witgen: a session table registered with another slot count");
            }
            if (st != S.stream && std::find(S.others.begin(), S.others.end(), st) == S.others.end()) {
                HIP_TRY(ctx, hipStreamWaitEvent(st, S.begun, 0));  // (the copies' zero fill, and whatever the caller queued on the session's stream before)
                S.others.push_back(st);
            }
            launch(true, copy[0], copy[1], copy[2], copy[3]);
            HIP_TRY(ctx, hipGetLastError());
            return 0;
        }
    }
    if (!(xcd_local && any && n > 0)) {
        launch(false, tabs[0].user, tabs[1].user, tabs[2].user, tabs[3].user);
        HIP_TRY(ctx, hipGetLastError());
        return 0;
    }
    size_t slots[4], total = 0;
    for (int t = 0; t < 4; t++) total += slots[t] = tabs[t].user ? tabs[t].slots : 0;
    void* scratch = nullptr;
    TRY(ctx_alloc(ctx, 8 * total * sizeof(uint32_t), &scratch));
    uint32_t* copy[4];
    copy[0] = (uint32_t*)scratch;
    for (int t = 1; t < 4; t++) copy[t] = copy[t - 1] + 8 * slots[t - 1];
    hipError_t e = hipSuccess;
    size_t used[4];
    for (int t = 0; t < 4 && e == hipSuccess; t++) {
        used[t] = tabs[t].used && tabs[t].used < slots[t] ? tabs[t].used : slots[t];
        if (!tabs[t].user) continue;
        if (used[t] == slots[t]) e = hipMemsetAsync(copy[t], 0, 8 * slots[t] * sizeof(uint32_t), st);
        else e = hipMemset2DAsync(copy[t], slots[t] * sizeof(uint32_t), 0, used[t] * sizeof(uint32_t), 8, st);
    }
    if (e == hipSuccess) {
        launch(true, tabs[0].user ? copy[0] : nullptr, tabs[1].user ? copy[1] : nullptr, tabs[2].user ? copy[2] : nullptr, tabs[3].user ? copy[3] : nullptr);
        for (int t = 0; t < 4; t++)
            if (tabs[t].user)
                hipLaunchKernelGGL(k_lk_merge, dim3((unsigned)((used[t] + NT - 1) / NT)), dim3(NT), 0, st, copy[t], slots[t], used[t], tabs[t].user);
        e = hipGetLastError();
    }
    const hipError_t e2 = hipStreamSynchronize(st);
    ctx_free(ctx, scratch);
    HIP_TRY(ctx, e);
    HIP_TRY(ctx, e2);
    return 0;
}
constexpr size_t LOGIC_SLOTS = (size_t)1 << 16;

// ---- R-type logic chips AND / OR / XOR (LogicInstruction, ceno_zkvm/src/instructions/riscv/logic/logic_circuit.rs:30-160): the same
// R-instruction base (state, rs1, rs2, rd: r_insn.rs:67-86) and then the three registers as 4 BYTES each (UInt8, split_to_u8);
// lookups: fetch, the six timestamp-difference limbs, and one entry of the op's 2^16-entry table per byte pair, key a | b << 8
// (UInt8::logic_assign uint/logic.rs:26-32, OpsTable::pack gkr_iop/src/tables/mod.rs:29-31).  28 mapped columns.
struct LogicMap {  // ceno_hip_logic_r_column_map = ceno_gpu's LogicRColumnMap (chips/logic_r.rs:25-42)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_bytes[4], rs2_bytes[4], rd_bytes[4];
    uint32_t num_cols;
};
static_assert(sizeof(LogicMap) == sizeof(ceno_hip_logic_r_column_map), "column map layout");
constexpr int LOGIC_COLS = 28;

template <bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_logic(LogicMap m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                     uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w,
                                                     size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_logic) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    lk_logic = xcd_copy<XCD_LOCAL>(lk_logic, LOGIC_SLOTS);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<LOGIC_COLS>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_read<XCD_LOCAL>(o, m.rs2_id, m.rs2_prev_ts, m.rs2_lt_diff, st.rs2_addr, st.rs2_prev, offset, ts + SUBCYCLE_RS2, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t x = (st.rs1_val >> (8 * b)) & 0xff, y = (st.rs2_val >> (8 * b)) & 0xff;
            o.put(m.rs1_bytes[b], x);
            o.put(m.rs2_bytes[b], y);
            o.put(m.rd_bytes[b], (st.rd_after >> (8 * b)) & 0xff);
            lk_count<XCD_LOCAL>(lk_logic, x | (y << 8));
        }
    }
}

// ---- I-type ADDI (AddiInstruction, ceno_zkvm/src/instructions/riscv/arith_imm/arith_imm_circuit_v2.rs:85-117): the I-instruction base
// (state, rs1, rd: i_insn.rs:66-82; no rs2), rs1 as two u16 limbs, the low 16 bits of the immediate, its sign, and the carries of
// rs1 + sign_extend(imm) (imm_sign_extend utils.rs:139-148; Value::add with overflow uint.rs:762-785: both result limbs are
// range-checked).  18 mapped columns; lookups: fetch, four timestamp-difference limbs, two u16 result limbs.
struct AddiMap {  // ceno_hip_addi_column_map = ceno_gpu's AddiColumnMap (chips/addi.rs:27-42)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_limbs[2], imm, imm_sign, rd_carries[2];
    uint32_t num_cols;
};
static_assert(sizeof(AddiMap) == sizeof(ceno_hip_addi_column_map), "column map layout");
constexpr int ADDI_COLS = 18;

template <bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_addi(AddiMap m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                    uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                    uint32_t* lk_dyn, uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<ADDI_COLS>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        // imm as i16 as u16; sign extension into the second limb
        const uint32_t imm16 = st.imm & 0xffff, neg = (imm16 >> 15) & 1u, ext = neg ? 0xffffu : 0u;
        o.put(m.rs1_limbs[0], st.rs1_val & 0xffff);
        o.put(m.rs1_limbs[1], st.rs1_val >> 16);
        o.put(m.imm, imm16);
        o.put(m.imm_sign, neg);
        const uint32_t s0 = (st.rs1_val & 0xffff) + imm16;
        const uint32_t s1 = (st.rs1_val >> 16) + ext + (s0 >> 16);
        o.put(m.rd_carries[0], s0 >> 16);
        o.put(m.rd_carries[1], s1 >> 16);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (s0 & 0xffff));
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (s1 & 0xffff));
    }
}

// ---- I-type logic chips ANDI / ORI / XORI (logic_imm/logic_imm_circuit_v2.rs:105-130,195-224): the I-instruction base, rs1 and rd as four
// bytes each, the immediate as two low bytes (imm & 0xffff) and two high bytes (0xff,0xff when bit 31 of the sign-extended
// immediate is set, else 0: InsnRecord::imm_internal / imm_signed_internal, tables/program.rs:111-152); four entries of the
// operation's byte table per instance, keys rs1_byte | imm_byte << 8.  24 mapped columns.
struct LogicIMap {  // ceno_hip_logic_i_column_map = ceno_gpu's LogicIColumnMap (chips/logic_i.rs:26-41)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_bytes[4], rd_bytes[4], imm_lo_bytes[2], imm_hi_bytes[2];
    uint32_t num_cols;
};
static_assert(sizeof(LogicIMap) == sizeof(ceno_hip_logic_i_column_map), "column map layout");
constexpr int LOGIC_I_COLS = 24;

template <bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_logic_i(LogicIMap m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                       uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w,
                                                       size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_logic) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    lk_logic = xcd_copy<XCD_LOCAL>(lk_logic, LOGIC_SLOTS);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<LOGIC_I_COLS>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        // the immediate as the circuit sees it: low half as is, high half = the sign (bit 31) spread over 16 bits
        const uint32_t imm_eff = (st.imm & 0xffffu) | ((st.imm >> 31) ? 0xffff0000u : 0u);
        o.put(m.imm_lo_bytes[0], imm_eff & 0xff);
        o.put(m.imm_lo_bytes[1], (imm_eff >> 8) & 0xff);
        o.put(m.imm_hi_bytes[0], (imm_eff >> 16) & 0xff);
        o.put(m.imm_hi_bytes[1], imm_eff >> 24);
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t x = (st.rs1_val >> (8 * b)) & 0xff;
            o.put(m.rs1_bytes[b], x);
            o.put(m.rd_bytes[b], (st.rd_after >> (8 * b)) & 0xff);
            lk_count<XCD_LOCAL>(lk_logic, x | (((imm_eff >> (8 * b)) & 0xff) << 8));
        }
    }
}

// ---- LUI (LuiInstruction, ceno_zkvm/src/instructions/riscv/lui.rs:100-120): the I-instruction base (rs1 is read as decoded: x0), bytes 1..3 of
// rd (byte 0 is zero by construction), each range-checked as a byte of the dynamic table (assert_ux::<8>), and imm = insn.imm as u32 >> 12
// (InsnRecord::imm_internal, U type: tables/program.rs:126-129).  16 mapped columns.
struct LuiMap {  // ceno_hip_lui_column_map = ceno_gpu's LuiColumnMap (chips/lui.rs:29-41)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rd_bytes[3], imm;
    uint32_t num_cols;
};
static_assert(sizeof(LuiMap) == sizeof(ceno_hip_lui_column_map), "column map layout");
constexpr int LUI_COLS = 16;

template <bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_lui(LuiMap m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                   uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                   uint32_t* lk_dyn, uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<LUI_COLS>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        o.put(m.imm, st.imm >> 12);
#pragma unroll
        for (int b = 1; b < 4; b++) {
            const uint32_t v = (st.rd_after >> (8 * b)) & 0xff;
            o.put(m.rd_bytes[b - 1], v);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 8) + v);
        }
    }
}

// ---- JAL (JalInstruction, ceno_zkvm/src/instructions/riscv/jump/jal_v2.rs:99-127; J-instruction base j_insn.rs:58-73: state with next_pc,
// rd write, fetch): rd = pc + 4 as four bytes, range-checked pairwise in the double-byte table (key a << 8 | b), the top byte also XORed
// with 0xC0 in the XOR table (PC_BITS = 30: the two bits above the program counter's range must be clear).  13 mapped columns.
constexpr uint32_t PC_BITS = 30;  // riscv/constants.rs
constexpr uint32_t PC_MSB_MASK = 0xC0;  // sum of 2^x for x in PC_BITS - 24 .. 8 (riscv/constants.rs:29, jal_v2.rs:120-124)
struct JalMap {  // ceno_hip_jal_column_map = ceno_gpu's JalColumnMap (chips/jal.rs:21-31)
    uint32_t pc, next_pc, ts;
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rd_bytes[4];
    uint32_t num_cols;
};
static_assert(sizeof(JalMap) == sizeof(ceno_hip_jal_column_map), "column map layout");
constexpr int JAL_COLS = 13;
constexpr int OFF_PC_AFTER = 12;

template <bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_jal(JalMap m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                   uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                   uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_du8, uint32_t* lk_xor) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    lk_du8 = xcd_copy<XCD_LOCAL>(lk_du8, LOGIC_SLOTS);
    lk_xor = xcd_copy<XCD_LOCAL>(lk_xor, LOGIC_SLOTS);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<JAL_COLS>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint32_t pc_after = *reinterpret_cast<const uint32_t*>(recs + (size_t)idx[r] * CENO_HIP_STEP_RECORD_BYTES + OFF_PC_AFTER);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.next_pc, pc_after);
        o.put(m.ts, ts);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint32_t b0 = st.rd_after & 0xff, b1 = (st.rd_after >> 8) & 0xff, b2 = (st.rd_after >> 16) & 0xff, b3 = st.rd_after >> 24;
        o.put(m.rd_bytes[0], b0);
        o.put(m.rd_bytes[1], b1);
        o.put(m.rd_bytes[2], b2);
        o.put(m.rd_bytes[3], b3);
        lk_count<XCD_LOCAL>(lk_du8, (b0 << 8) + b1);  // assert_double_u8 (lk_multiplicity.rs:200-203)
        lk_count<XCD_LOCAL>(lk_du8, (b2 << 8) + b3);
        lk_count<XCD_LOCAL>(lk_xor, b3 | (PC_MSB_MASK << 8));
    }
}

// ---- AUIPC (AuipcInstruction, ceno_zkvm/src/instructions/riscv/auipc.rs:149-187): the I-instruction base (rs1 = x0 as decoded), rd as four
// bytes (double-byte table), the middle bytes 1, 2 of pc and the three bytes of imm = insn.imm as u32 >> 8 (imm_internal, AUIPC:
// tables/program.rs:119-124) each as a byte of the dynamic table, and pc's top byte XORed with 0xC0.  21 mapped columns.
struct AuipcMap {  // ceno_hip_auipc_column_map = ceno_gpu's AuipcColumnMap (chips/auipc.rs:28-44)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rd_bytes[4], pc_limbs[2], imm_limbs[3];
    uint32_t num_cols;
};
static_assert(sizeof(AuipcMap) == sizeof(ceno_hip_auipc_column_map), "column map layout");
constexpr int AUIPC_COLS = 21;

template <bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_auipc(AuipcMap m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                     uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                     uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_du8, uint32_t* lk_xor) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    lk_du8 = xcd_copy<XCD_LOCAL>(lk_du8, LOGIC_SLOTS);
    lk_xor = xcd_copy<XCD_LOCAL>(lk_xor, LOGIC_SLOTS);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<AUIPC_COLS>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint32_t b0 = st.rd_after & 0xff, b1 = (st.rd_after >> 8) & 0xff, b2 = (st.rd_after >> 16) & 0xff, b3 = st.rd_after >> 24;
        o.put(m.rd_bytes[0], b0);
        o.put(m.rd_bytes[1], b1);
        o.put(m.rd_bytes[2], b2);
        o.put(m.rd_bytes[3], b3);
        lk_count<XCD_LOCAL>(lk_du8, (b0 << 8) + b1);
        lk_count<XCD_LOCAL>(lk_du8, (b2 << 8) + b3);
#pragma unroll
        for (int k = 0; k < 2; k++) {  // pc bytes 1, 2
            const uint32_t v = (st.pc >> (8 * (k + 1))) & 0xff;
            o.put(m.pc_limbs[k], v);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 8) + v);
        }
        const uint32_t imm = st.imm >> 8;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t v = (imm >> (8 * k)) & 0xff;
            o.put(m.imm_limbs[k], v);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 8) + v);
        }
        lk_count<XCD_LOCAL>(lk_xor, (st.pc >> 24) | (PC_MSB_MASK << 8));
    }
}

// ---- SLT / SLTU (SetLessThanInstruction, ceno_zkvm/src/instructions/riscv/slt/slt_circuit_v2.rs:86-119): the R-instruction base, rs1 and rs2 as
// u16 limbs, and the UIntLimbsLT comparison (gadgets/signed_limbs.rs:150-236): cmp_lt, the most significant limbs as FIELD elements of
// their signed value (limb - 2^16 when negative in a signed comparison), a one-hot marker of the most significant differing limb and
// the (positive) difference there; u16 range lookups of diff - 1 (or 0 when equal) and of the two shifted top limbs.  26 mapped columns.
constexpr uint64_t GOLDILOCKS_P = 0xFFFFFFFF00000001ULL;
// UIntLimbsLT::assign (gadgets/signed_limbs.rs:150-222) over run_cmp (:226-236) for two 32-bit values as u16 limbs (a0, a1), (b0, b1)
template <bool XCD_LOCAL>
__device__ __forceinline__ void emit_uint_lt(const Row& o, uint32_t cmp_lt_col, uint32_t a_msb_col, uint32_t b_msb_col, const uint32_t (&marker_cols)[2],
                                             uint32_t diff_val_col, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, bool is_signed, uint32_t* lk_dyn) {
    // most significant differing limb; the sign bits flip the outcome
    const bool a_neg = is_signed && (a1 >> 15), b_neg = is_signed && (b1 >> 15);
    const int diff_idx = a1 != b1 ? 1 : (a0 != b0 ? 0 : 2);
    const bool lt = diff_idx == 2 ? false : (((diff_idx == 1 ? a1 < b1 : a0 < b0) ? 1 : 0) ^ (a_neg ? 1 : 0) ^ (b_neg ? 1 : 0)) != 0;
    o.put(cmp_lt_col, lt ? 1 : 0);
    o.put(marker_cols[0], diff_idx == 0);
    o.put(marker_cols[1], diff_idx == 1);
    // top limbs as signed values; in the field a negative one is p - (2^16 - limb)
    const int64_t sa = a_neg ? (int64_t)a1 - 65536 : (int64_t)a1, sb = b_neg ? (int64_t)b1 - 65536 : (int64_t)b1;
    o.put(a_msb_col, sa < 0 ? GOLDILOCKS_P - (uint64_t)(-sa) : (uint64_t)sa);
    o.put(b_msb_col, sb < 0 ? GOLDILOCKS_P - (uint64_t)(-sb) : (uint64_t)sb);
    uint32_t diff_val = 0;
    if (diff_idx == 1) diff_val = (uint32_t)(lt ? sb - sa : sa - sb) & 0xffff;
    else if (diff_idx == 0) diff_val = (lt ? b0 - a0 : a0 - b0) & 0xffff;
    o.put(diff_val_col, diff_val);
    constexpr uint32_t U16 = 1u << 16;
    lk_count<XCD_LOCAL>(lk_dyn, U16 + (diff_idx == 2 ? 0u : ((diff_val - 1) & 0xffff)));
    lk_count<XCD_LOCAL>(lk_dyn, U16 + (a_neg ? a1 - 0x8000u : a1 + ((uint32_t)is_signed << 15)));
    lk_count<XCD_LOCAL>(lk_dyn, U16 + (b_neg ? b1 - 0x8000u : b1 + ((uint32_t)is_signed << 15)));
}

struct SltMap {  // ceno_hip_slt_column_map = ceno_gpu's SltColumnMap (chips/slt.rs:33-55)
    uint32_t rs1_limbs[2], rs2_limbs[2], cmp_lt, a_msb_f, b_msb_f, diff_marker[2], diff_val;
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t num_cols;
};
static_assert(sizeof(SltMap) == sizeof(ceno_hip_slt_column_map), "column map layout");
constexpr int SLT_COLS = 26;

template <bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_slt(SltMap m, int is_signed, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                   uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                   uint32_t* lk_dyn, uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<SLT_COLS>(o, &m.rs1_limbs[0]);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_read<XCD_LOCAL>(o, m.rs2_id, m.rs2_prev_ts, m.rs2_lt_diff, st.rs2_addr, st.rs2_prev, offset, ts + SUBCYCLE_RS2, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint32_t a0 = st.rs1_val & 0xffff, a1 = st.rs1_val >> 16, b0 = st.rs2_val & 0xffff, b1 = st.rs2_val >> 16;
        o.put(m.rs1_limbs[0], a0);
        o.put(m.rs1_limbs[1], a1);
        o.put(m.rs2_limbs[0], b0);
        o.put(m.rs2_limbs[1], b1);
        emit_uint_lt<XCD_LOCAL>(o, m.cmp_lt, m.a_msb_f, m.b_msb_f, m.diff_marker, m.diff_val, a0, a1, b0, b1, is_signed != 0, lk_dyn);
    }
}

// ---- SLTI / SLTIU (riscv/slti/slti_circuit_v2.rs:104-140): the I-instruction base, rs1 as u16 limbs, imm (low 16 bits) and its sign, and the
// same comparison of rs1 with the sign-extended immediate [imm, sign ? 0xffff : 0].  22 mapped columns.
struct SltiMap {  // ceno_hip_slti_column_map = ceno_gpu's SltiColumnMap (chips/slti.rs:32-51)
    uint32_t rs1_limbs[2], imm, imm_sign, cmp_lt, a_msb_f, b_msb_f, diff_marker[2], diff_val;
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t num_cols;
};
static_assert(sizeof(SltiMap) == sizeof(ceno_hip_slti_column_map), "column map layout");
constexpr int SLTI_COLS = 22;

template <bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_slti(SltiMap m, int is_signed, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                    uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                    uint32_t* lk_dyn, uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<SLTI_COLS>(o, &m.rs1_limbs[0]);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint32_t imm16 = st.imm & 0xffff, neg = (imm16 >> 15) & 1u;
        o.put(m.rs1_limbs[0], st.rs1_val & 0xffff);
        o.put(m.rs1_limbs[1], st.rs1_val >> 16);
        o.put(m.imm, imm16);
        o.put(m.imm_sign, neg);
        emit_uint_lt<XCD_LOCAL>(o, m.cmp_lt, m.a_msb_f, m.b_msb_f, m.diff_marker, m.diff_val, st.rs1_val & 0xffff, st.rs1_val >> 16, imm16, neg ? 0xffffu : 0u,
                                is_signed != 0, lk_dyn);
    }
}

// ---- branches (BranchCircuit, ceno_zkvm/src/instructions/riscv/branch/branch_circuit_v2.rs:143-209 over BInstructionConfig b_insn.rs:92-116: state
// with next_pc, rs1, rs2, the immediate as a FIELD element of its signed value, fetch): BLT / BGE / BLTU / BGEU carry the UIntLimbsLT
// comparison of rs1 and rs2 (22 mapped columns); BEQ / BNE carry the taken bit and, at the first differing limb, the field inverse of
// the limb difference (19 mapped columns; gl::inv — one exponentiation per instance).
__device__ __forceinline__ uint64_t signed_to_field(uint32_t imm32) {  // i64_to_base(insn.imm as i64)
    const int32_t v = (int32_t)imm32;
    return v < 0 ? GOLDILOCKS_P - (uint64_t)(-(int64_t)v) : (uint64_t)v;
}
struct BranchCmpMap {  // ceno_hip_branch_cmp_column_map = ceno_gpu's BranchCmpColumnMap (chips/branch_cmp.rs:35-54)
    uint32_t rs1_limbs[2], rs2_limbs[2], cmp_lt, a_msb_f, b_msb_f, diff_marker[2], diff_val;
    uint32_t pc, next_pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t imm;
    uint32_t num_cols;
};
static_assert(sizeof(BranchCmpMap) == sizeof(ceno_hip_branch_cmp_column_map), "column map layout");
constexpr int BRANCH_CMP_COLS = 22;
struct BranchEqMap {  // ceno_hip_branch_eq_column_map = ceno_gpu's BranchEqColumnMap (chips/branch_eq.rs:27-43)
    uint32_t rs1_limbs[2], rs2_limbs[2], branch_taken, diff_inv_marker[2];
    uint32_t pc, next_pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t imm;
    uint32_t num_cols;
};
static_assert(sizeof(BranchEqMap) == sizeof(ceno_hip_branch_eq_column_map), "column map layout");
constexpr int BRANCH_EQ_COLS = 19;

// MODE 0: comparison branches (flag = is_signed), MODE 1: equality branches (flag = is_beq)
template <bool XCD_LOCAL, int MODE, class MapT>
__global__ void __launch_bounds__(NT) k_witgen_branch(MapT m, int flag, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                      uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                      uint32_t* lk_dyn, uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<(MODE == 0 ? BRANCH_CMP_COLS : BRANCH_EQ_COLS)>(o, &m.rs1_limbs[0]);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint32_t pc_after = *reinterpret_cast<const uint32_t*>(recs + (size_t)idx[r] * CENO_HIP_STEP_RECORD_BYTES + OFF_PC_AFTER);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.next_pc, pc_after);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_read<XCD_LOCAL>(o, m.rs2_id, m.rs2_prev_ts, m.rs2_lt_diff, st.rs2_addr, st.rs2_prev, offset, ts + SUBCYCLE_RS2, lk_dyn);
        o.put(m.imm, signed_to_field(st.imm));
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint32_t a0 = st.rs1_val & 0xffff, a1 = st.rs1_val >> 16, b0 = st.rs2_val & 0xffff, b1 = st.rs2_val >> 16;
        o.put(m.rs1_limbs[0], a0);
        o.put(m.rs1_limbs[1], a1);
        o.put(m.rs2_limbs[0], b0);
        o.put(m.rs2_limbs[1], b1);
        if constexpr (MODE == 0) {
            emit_uint_lt<XCD_LOCAL>(o, m.cmp_lt, m.a_msb_f, m.b_msb_f, m.diff_marker, m.diff_val, a0, a1, b0, b1, flag != 0, lk_dyn);
        } else {
            // run_eq (branch_circuit_v2.rs:164-182): the FIRST differing limb from the least significant one; inverse of the field difference
            const int diff_idx = a0 != b0 ? 0 : (a1 != b1 ? 1 : -1);
            const bool taken = diff_idx < 0 ? flag != 0 : flag == 0;
            uint64_t inv = 0;
            if (diff_idx >= 0) {
                const uint32_t x = diff_idx == 0 ? a0 : a1, y = diff_idx == 0 ? b0 : b1;
                inv = gl::inv(x > y ? (uint64_t)(x - y) : GOLDILOCKS_P - (uint64_t)(y - x));
            }
            o.put(m.branch_taken, taken ? 1 : 0);
            o.put(m.diff_inv_marker[0], diff_idx == 0 ? inv : 0);
            o.put(m.diff_inv_marker[1], diff_idx == 1 ? inv : 0);
        }
    }
}

// ---- word memory access: LW (riscv/memory/load_v2.rs:197-255 over IMInstructionConfig im_insn.rs:71-90: state, rs1, rd, memory read, fetch) and SW
// (riscv/memory/store_v2.rs:138-177 over SInstructionConfig s_insn.rs:77-96: state, rs1, rs2, memory write, fetch).  The address rs1 +
// sign_extend(imm) is witnessed as two u16 limbs and range-checked by MemAddr::assign_instance (insn_base.rs:880-905, MEM_BITS = 30, word-aligned:
// no low-bit columns): the low limb without its two low bits as a 14-bit value, the high limb as a 14-bit value.  23 mapped columns each.
constexpr int OFF_MEM = 104;  // memory_op: WriteOp {addr u32 (word address), before u32, after u32, pad, previous_cycle u64}
constexpr uint64_t SUBCYCLE_MEM = 3;
struct LwMap {  // ceno_hip_lw_column_map = ceno_gpu's LwColumnMap (chips/lw.rs:35-53)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], imm, imm_sign, mem_addr_limbs[2], mem_read_limbs[2];
    uint32_t num_cols;
};
static_assert(sizeof(LwMap) == sizeof(ceno_hip_lw_column_map), "column map layout");
struct SwMap {  // ceno_hip_sw_column_map = ceno_gpu's SwColumnMap (chips/sw.rs:31-49)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], rs2_limbs[2], imm, imm_sign, prev_mem_val[2], mem_addr[2];
    uint32_t num_cols;
};
static_assert(sizeof(SwMap) == sizeof(ceno_hip_sw_column_map), "column map layout");
// SH / SB (StoreConfig<E, 1> / <E, 0>, store_v2.rs:100-177): the SW columns, the address bits the width leaves free (bit 1; bits 0 and 1) and, for SB,
// MemWordUtil's byte columns (riscv/memory/gadget.rs:134-185): the two bytes of the previous word's addressed limb, the stored byte, the limb after it
struct ShMap {  // ceno_hip_sh_column_map = ceno_gpu's ShColumnMap (chips/sh.rs:39-58)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], rs2_limbs[2], imm, imm_sign, prev_mem_val[2], mem_addr[2];
    uint32_t mem_addr_bit_1;
    uint32_t num_cols;
};
static_assert(sizeof(ShMap) == sizeof(ceno_hip_sh_column_map), "column map layout");
struct SbMap {  // ceno_hip_sb_column_map = ceno_gpu's SbColumnMap (chips/sb.rs:57-81)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], rs2_limbs[2], imm, imm_sign, prev_mem_val[2], mem_addr[2];
    uint32_t mem_addr_bit_0, mem_addr_bit_1, prev_limb_bytes[2], rs2_limb_byte, expected_limb;
    uint32_t num_cols;
};
static_assert(sizeof(SbMap) == sizeof(ceno_hip_sb_column_map), "column map layout");
constexpr int MEM_COLS = 23, SH_COLS = 24, SB_COLS = 29;
constexpr int mem_cols(int kind) { return kind == 2 ? SH_COLS : kind == 3 ? SB_COLS : MEM_COLS; }

// the memory access itself (ReadMEM / WriteMEM::assign_op, insn_base.rs:517-545,650-680) and the address range checks
template <bool XCD_LOCAL>
__device__ __forceinline__ void emit_mem(const Row& o, uint32_t prev_col, const uint32_t (&diff_cols)[2], const uint32_t (&addr_cols)[2], uint32_t addr,
                                         uint64_t prev_cycle, uint64_t offset, uint64_t ts, uint32_t* lk_dyn) {
    const uint64_t p = aligned_prev_ts(prev_cycle, offset), d = lt_diff(p, ts + SUBCYCLE_MEM);
    o.put(prev_col, p);
    o.put(diff_cols[0], d & 0xffff);
    o.put(diff_cols[1], (d >> 16) & 0xffff);
    lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)(d & 0xffff));
    lk_count<XCD_LOCAL>(lk_dyn, (1u << (MAX_TS_BITS - 16)) + (uint32_t)((d >> 16) & 0xffff));
    o.put(addr_cols[0], addr & 0xffff);
    o.put(addr_cols[1], addr >> 16);
    lk_count<XCD_LOCAL>(lk_dyn, (1u << 14) + ((addr & 0xffff) >> 2));  // assert_ux::<14>(mid_u14)
    lk_count<XCD_LOCAL>(lk_dyn, (1u << 14) + (addr >> 16));            // assert_const_range(high_u16, MEM_BITS - 16)
}

// KIND 0: LW, 1: SW, 2: SH, 3: SB
template <bool XCD_LOCAL, int KIND, class MapT>
__global__ void __launch_bounds__(NT) k_witgen_mem(MapT m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n, uint64_t offset,
                                                   uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows, uint32_t* lk_dyn,
                                                   uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        constexpr bool STORE = KIND != 0;
        if (r >= n) {
            zero_row<mem_cols(KIND)>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t* q = reinterpret_cast<const uint64_t*>(recs + (size_t)idx[r] * CENO_HIP_STEP_RECORD_BYTES);
        const uint32_t mem_before = (uint32_t)(q[OFF_MEM / 8] >> 32);
        const uint64_t mem_prev = q[OFF_MEM / 8 + 2];
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        const uint32_t imm16 = st.imm & 0xffff, neg = (imm16 >> 15) & 1u;
        const uint32_t addr = st.rs1_val + (imm16 | (neg ? 0xffff0000u : 0u));  // rs1.wrapping_add_signed(imm as i16)
        o.put(m.rs1_limbs[0], st.rs1_val & 0xffff);
        o.put(m.rs1_limbs[1], st.rs1_val >> 16);
        o.put(m.imm, imm16);
        o.put(m.imm_sign, neg);
        if constexpr (STORE) {
            emit_read<XCD_LOCAL>(o, m.rs2_id, m.rs2_prev_ts, m.rs2_lt_diff, st.rs2_addr, st.rs2_prev, offset, ts + SUBCYCLE_RS2, lk_dyn);
            o.put(m.rs2_limbs[0], st.rs2_val & 0xffff);
            o.put(m.rs2_limbs[1], st.rs2_val >> 16);
            o.put(m.prev_mem_val[0], mem_before & 0xffff);
            o.put(m.prev_mem_val[1], mem_before >> 16);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (mem_before & 0xffff));  // Value::new(memory_op.value.before, lkm)
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (mem_before >> 16));
            emit_mem<XCD_LOCAL>(o, m.mem_prev_ts, m.mem_lt_diff, m.mem_addr, addr, mem_prev, offset, ts, lk_dyn);
            if constexpr (KIND == 2) o.put(m.mem_addr_bit_1, (addr >> 1) & 1u);
            if constexpr (KIND == 3) {
                const uint32_t bit0 = addr & 1u, bit1 = (addr >> 1) & 1u;
                const uint32_t prev_limb = (mem_before >> (16 * bit1)) & 0xffff, rs2_limb = st.rs2_val & 0xffff;
                const uint32_t p_lo = prev_limb & 0xff, p_hi = prev_limb >> 8, s_lo = rs2_limb & 0xff, s_hi = rs2_limb >> 8;
                o.put(m.mem_addr_bit_0, bit0);
                o.put(m.mem_addr_bit_1, bit1);
                o.put(m.prev_limb_bytes[0], p_lo);
                o.put(m.prev_limb_bytes[1], p_hi);
                o.put(m.rs2_limb_byte, s_lo);
                o.put(m.expected_limb, bit0 ? (s_lo << 8) + p_lo : (p_hi << 8) + s_lo);
                lk_count<XCD_LOCAL>(lk_dyn, (1u << 8) + p_lo);  // assert_ux::<8> of both bytes of the previous limb and of rs2's low limb
                lk_count<XCD_LOCAL>(lk_dyn, (1u << 8) + p_hi);
                lk_count<XCD_LOCAL>(lk_dyn, (1u << 8) + s_lo);
                lk_count<XCD_LOCAL>(lk_dyn, (1u << 8) + s_hi);
            }
        } else {
            emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
            o.put(m.mem_read_limbs[0], mem_before & 0xffff);
            o.put(m.mem_read_limbs[1], mem_before >> 16);
            emit_mem<XCD_LOCAL>(o, m.mem_prev_ts, m.mem_lt_diff, m.mem_addr_limbs, addr, mem_prev, offset, ts, lk_dyn);
        }
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
    }
}

// ---- JALR (JalrInstruction, riscv/jump/jalr_v2.rs:146-190 over IInstructionConfig with a branching state: pc, next_pc, ts): rs1 as limbs, the
// offset's low 16 bits and its sign, the jump target rs1 + sign_extend(imm) as a MemAddr with max_bits = PC_BITS = 30 and BOTH low bits
// witnessed (construct_with_max_bits(cb, 0, PC_BITS); next_pc is the target rounded down to an even address), rd = pc + 4 with its high limb
// witnessed and both limbs range-checked (16 bits, PC_BITS - 16 bits).  22 mapped columns.
struct JalrMap {  // ceno_hip_jalr_column_map = ceno_gpu's JalrColumnMap (chips/jalr.rs:31-49)
    uint32_t pc, next_pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_limbs[2], imm, imm_sign, jump_pc_addr[2], jump_pc_addr_bit[2], rd_high;
    uint32_t num_cols;
};
static_assert(sizeof(JalrMap) == sizeof(ceno_hip_jalr_column_map), "column map layout");
constexpr int JALR_COLS = 22;

template <bool XCD_LOCAL>
__global__ void __launch_bounds__(NT) k_witgen_jalr(JalrMap m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                    uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                    uint32_t* lk_dyn, uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<JALR_COLS>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint32_t pc_after = *reinterpret_cast<const uint32_t*>(recs + (size_t)idx[r] * CENO_HIP_STEP_RECORD_BYTES + OFF_PC_AFTER);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.next_pc, pc_after);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint32_t imm16 = st.imm & 0xffff, neg = (imm16 >> 15) & 1u;
        const uint32_t target = st.rs1_val + (imm16 | (neg ? 0xffff0000u : 0u));  // rs1.overflowing_add_signed(sign-extended imm)
        o.put(m.rs1_limbs[0], st.rs1_val & 0xffff);
        o.put(m.rs1_limbs[1], st.rs1_val >> 16);
        o.put(m.imm, imm16);
        o.put(m.imm_sign, neg);
        o.put(m.jump_pc_addr[0], target & 0xffff);
        o.put(m.jump_pc_addr[1], target >> 16);
        o.put(m.jump_pc_addr_bit[0], target & 1u);
        o.put(m.jump_pc_addr_bit[1], (target >> 1) & 1u);
        o.put(m.rd_high, st.rd_after >> 16);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (st.rd_after & 0xffff));             // assert_const_range(rd_limb[0], 16)
        lk_count<XCD_LOCAL>(lk_dyn, (1u << (PC_BITS - 16)) + (st.rd_after >> 16));    // assert_const_range(rd_limb[1], PC_BITS - 16)
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 14) + ((target & 0xffff) >> 2));           // MemAddr: assert_ux::<14>(mid_u14)
        lk_count<XCD_LOCAL>(lk_dyn, (1u << (PC_BITS - 16)) + (target >> 16));         // MemAddr: assert_const_range(high limb, PC_BITS - 16)
    }
}

// ---- shifts: SLL / SRL / SRA (ShiftLogicalInstruction, riscv/shift/shift_circuit_v2.rs:359-396 over RInstructionConfig) and SLLI / SRLI / SRAI
// (ShiftImmInstruction, :485-521 over IInstructionConfig).  Operands and result as BYTES (the result's pairs in the double-byte table), and the
// ShiftBase gadget (:242-293): the shift amount c[0] mod 32 split into limb_shift (c / 8) and bit_shift (c mod 8) as one-hot markers, the power
// 2^bit_shift in the left (SLL) or right (SRL / SRA) multiplier column, per operand byte the bits that cross into the neighbour byte (range
// lookups of bit_shift bits), (c[0] - shift) >> 5 as a 3-bit range lookup, and for SRA the operand's sign with its top byte XORed with 128.
// 47 (R) / 40 (I) mapped columns.
struct ShiftRMap {  // ceno_hip_shift_r_column_map = ceno_gpu's ShiftRColumnMap (chips/shift_r.rs:36-59)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_bytes[4], rs2_bytes[4], rd_bytes[4];
    uint32_t bit_shift_marker[8], limb_shift_marker[4], bit_multiplier_left, bit_multiplier_right, b_sign, bit_shift_carry[4];
    uint32_t num_cols;
};
static_assert(sizeof(ShiftRMap) == sizeof(ceno_hip_shift_r_column_map), "column map layout");
struct ShiftIMap {  // ceno_hip_shift_i_column_map = ceno_gpu's ShiftIColumnMap (chips/shift_i.rs:33-53)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_bytes[4], rd_bytes[4], imm;
    uint32_t bit_shift_marker[8], limb_shift_marker[4], bit_multiplier_left, bit_multiplier_right, b_sign, bit_shift_carry[4];
    uint32_t num_cols;
};
static_assert(sizeof(ShiftIMap) == sizeof(ceno_hip_shift_i_column_map), "column map layout");
constexpr int SHIFT_R_COLS = 47, SHIFT_I_COLS = 40;

// KIND 0: shift left, 1: logical right, 2: arithmetic right (GpuWitgenKind::ShiftR / ShiftI's argument)
template <bool XCD_LOCAL, bool IMM, class MapT>
__global__ void __launch_bounds__(NT) k_witgen_shift(MapT m, int kind, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                     uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                     uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_du8, uint32_t* lk_xor) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    lk_du8 = xcd_copy<XCD_LOCAL>(lk_du8, LOGIC_SLOTS);
    lk_xor = xcd_copy<XCD_LOCAL>(lk_xor, LOGIC_SLOTS);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<IMM ? SHIFT_I_COLS : SHIFT_R_COLS>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        uint32_t c;
        if constexpr (IMM) {
            c = st.imm & 0xffff;  // insn.imm as i16 as u16
            o.put(m.imm, c);
        } else {
            c = st.rs2_val;
            emit_read<XCD_LOCAL>(o, m.rs2_id, m.rs2_prev_ts, m.rs2_lt_diff, st.rs2_addr, st.rs2_prev, offset, ts + SUBCYCLE_RS2, lk_dyn);
#pragma unroll
            for (int k = 0; k < 4; k++) o.put(m.rs2_bytes[k], (c >> (8 * k)) & 0xff);
        }
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint32_t b = st.rs1_val, d = st.rd_after;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            o.put(m.rs1_bytes[k], (b >> (8 * k)) & 0xff);
            o.put(m.rd_bytes[k], (d >> (8 * k)) & 0xff);
        }
        lk_count<XCD_LOCAL>(lk_du8, ((d & 0xff) << 8) + ((d >> 8) & 0xff));  // assert_double_u8 per byte pair of the result
        lk_count<XCD_LOCAL>(lk_du8, (((d >> 16) & 0xff) << 8) + (d >> 24));
        const uint32_t c0 = c & 0xff, shift = c0 & 31, limb_shift = shift >> 3, bit_shift = shift & 7;
        o.put(m.bit_multiplier_left, kind == 0 ? (1u << bit_shift) : 0u);
        o.put(m.bit_multiplier_right, kind == 0 ? 0u : (1u << bit_shift));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t byte = (b >> (8 * k)) & 0xff;
            const uint32_t carry = kind == 0 ? (byte >> (8 - bit_shift)) : (byte & ((1u << bit_shift) - 1));
            o.put(m.bit_shift_carry[k], carry);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << bit_shift) + carry);  // assert_dynamic_range(carry, bit_shift)
        }
#pragma unroll
        for (int k = 0; k < 8; k++) o.put(m.bit_shift_marker[k], k == (int)bit_shift ? 1u : 0u);
#pragma unroll
        for (int k = 0; k < 4; k++) o.put(m.limb_shift_marker[k], k == (int)limb_shift ? 1u : 0u);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 3) + ((c0 - shift) >> 5));  // assert_const_range((c[0] - bit_shift - 8 limb_shift) >> 5, 3)
        uint32_t sign = 0;
        if (kind == 2) {
            sign = b >> 31;
            lk_count<XCD_LOCAL>(lk_xor, (b >> 24) | (128u << 8));  // lookup_xor_byte(top byte, 1 << 7)
        }
        o.put(m.b_sign, sign);
    }
}

// ---- sub-word loads LH / LHU / LB / LBU (LoadInstruction, riscv/memory/load_v2.rs:197-255): LW's columns, then the address bit that selects
// the 16-bit limb of the memory word and that limb; for byte loads the bit that selects the byte, the addressed byte and the other one (both
// byte-range lookups); for signed loads the sign bit of the loaded value with SignedExtendConfig's range lookup of 2 val - (msb << n_bits)
// (gadgets/signed_ext.rs:92-103).  Columns a variant does not have carry CENO_HIP_NO_COLUMN in the map.
struct LoadSubMap {  // ceno_hip_load_sub_column_map = ceno_gpu's LoadSubColumnMap (chips/load_sub.rs:67-91)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t mem_prev_ts, mem_lt_diff[2];
    uint32_t rs1_limbs[2], imm, imm_sign, mem_addr[2], mem_read[2];
    uint32_t addr_bit_1, target_limb, addr_bit_0, target_byte, dummy_byte, msb;
    uint32_t num_cols;
};
static_assert(sizeof(LoadSubMap) == sizeof(ceno_hip_load_sub_column_map), "column map layout");
constexpr int LOAD_SUB_COMMON = 25;  // the fields every variant has (up to target_limb)

template <bool XCD_LOCAL, bool BYTE, bool SIGNED>
__global__ void __launch_bounds__(NT) k_witgen_load_sub(LoadSubMap m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                        uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                        uint32_t* lk_dyn, uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<LOAD_SUB_COMMON>(o, &m.pc);
            if (BYTE) {
                o.put(m.addr_bit_0, 0);
                o.put(m.target_byte, 0);
                o.put(m.dummy_byte, 0);
            }
            if (SIGNED) o.put(m.msb, 0);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t* q = reinterpret_cast<const uint64_t*>(recs + (size_t)idx[r] * CENO_HIP_STEP_RECORD_BYTES);
        const uint32_t word = (uint32_t)(q[OFF_MEM / 8] >> 32);
        const uint64_t mem_prev = q[OFF_MEM / 8 + 2];
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        const uint32_t imm16 = st.imm & 0xffff, neg = (imm16 >> 15) & 1u;
        const uint32_t addr = st.rs1_val + (imm16 | (neg ? 0xffff0000u : 0u));
        o.put(m.rs1_limbs[0], st.rs1_val & 0xffff);
        o.put(m.rs1_limbs[1], st.rs1_val >> 16);
        o.put(m.imm, imm16);
        o.put(m.imm_sign, neg);
        o.put(m.mem_read[0], word & 0xffff);
        o.put(m.mem_read[1], word >> 16);
        emit_mem<XCD_LOCAL>(o, m.mem_prev_ts, m.mem_lt_diff, m.mem_addr, addr, mem_prev, offset, ts, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint32_t bit0 = addr & 1u, bit1 = (addr >> 1) & 1u;
        const uint32_t limb = (word >> (16 * bit1)) & 0xffff;
        o.put(m.addr_bit_1, bit1);
        o.put(m.target_limb, limb);
        uint32_t val = limb;
        if (BYTE) {
            const uint32_t target = (limb >> (8 * bit0)) & 0xff, other = (limb >> (8 * (1 - bit0))) & 0xff;
            o.put(m.addr_bit_0, bit0);
            o.put(m.target_byte, target);
            o.put(m.dummy_byte, other);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 8) + target);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 8) + other);
            val = target;
        }
        if (SIGNED) {
            constexpr uint32_t BITS = BYTE ? 8 : 16;
            const uint32_t msb = val >> (BITS - 1);
            o.put(m.msb, msb);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << BITS) + (2 * val - (msb << BITS)));  // assert_const_range(2 val - (msb << n_bits), n_bits)
        }
    }
}

// ---- MUL / MULH / MULHU / MULHSU (MulhInstructionBase, riscv/mulh/mulh_circuit_v2.rs:234-333 over RInstructionConfig): operands as u16 limbs, the
// low product limbs, and for the high forms the high product limbs with the operands' sign extensions; run_mulh (:427-487) is a schoolbook product
// over 16-bit limbs whose carries (< 2^18) and result limbs are range-checked, plus the operands' sign tests.  The register write is the circuit's
// own expression of rd_low / rd_high: StepRecord.rd.value.after is not read.  22 (MUL) / 26 mapped columns.
struct MulMap {  // ceno_hip_mul_column_map = ceno_gpu's MulColumnMap (chips/mul.rs:36-56); rd_high, rs1_ext, rs2_ext = CENO_HIP_NO_COLUMN for MUL
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t rs1_limbs[2], rs2_limbs[2], rd_low[2], rd_high[2], rs1_ext, rs2_ext;
    uint32_t num_cols;
};
static_assert(sizeof(MulMap) == sizeof(ceno_hip_mul_column_map), "column map layout");
constexpr int MUL_COMMON = 22;

// kind 0: MUL, 1: MULH, 2: MULHU, 3: MULHSU (GpuWitgenKind::Mul's argument)
template <bool XCD_LOCAL, bool HIGH>
__global__ void __launch_bounds__(NT) k_witgen_mul(MulMap m, int kind, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n,
                                                   uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows,
                                                   uint32_t* lk_dyn, uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<MUL_COMMON>(o, &m.pc);
            if (HIGH) {
                o.put(m.rd_high[0], 0);
                o.put(m.rd_high[1], 0);
                o.put(m.rs1_ext, 0);
                o.put(m.rs2_ext, 0);
            }
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_read<XCD_LOCAL>(o, m.rs2_id, m.rs2_prev_ts, m.rs2_lt_diff, st.rs2_addr, st.rs2_prev, offset, ts + SUBCYCLE_RS2, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint64_t x0 = st.rs1_val & 0xffff, x1 = st.rs1_val >> 16, y0 = st.rs2_val & 0xffff, y1 = st.rs2_val >> 16;
        o.put(m.rs1_limbs[0], x0);
        o.put(m.rs1_limbs[1], x1);
        o.put(m.rs2_limbs[0], y0);
        o.put(m.rs2_limbs[1], y1);
        // run_mulh: low half
        const uint64_t m0 = x0 * y0, c0 = m0 >> 16;
        const uint64_t m1 = c0 + x0 * y1 + x1 * y0, c1 = m1 >> 16;
        o.put(m.rd_low[0], m0 & 0xffff);
        o.put(m.rd_low[1], m1 & 0xffff);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)(m0 & 0xffff));
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 18) + (uint32_t)c0);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)(m1 & 0xffff));
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 18) + (uint32_t)c1);
        if (HIGH) {
            const uint64_t x_ext = (x1 >> 15) * (kind == 2 ? 0u : 0xffffu), y_ext = (y1 >> 15) * (kind == 1 ? 0xffffu : 0u);
            const uint64_t h0 = c1 + x0 * y_ext + y0 * x_ext + x1 * y1, c2 = h0 >> 16;
            const uint64_t h1 = c2 + (x0 + x1) * y_ext + (y0 + y1) * x_ext, c3 = h1 >> 16;
            o.put(m.rd_high[0], h0 & 0xffff);
            o.put(m.rd_high[1], h1 & 0xffff);
            o.put(m.rs1_ext, x_ext);
            o.put(m.rs2_ext, y_ext);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)(h0 & 0xffff));
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 18) + (uint32_t)c2);
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)(h1 & 0xffff));
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 18) + (uint32_t)c3);
            const uint32_t s1 = (uint32_t)(x_ext / 0xffff), s2 = (uint32_t)(y_ext / 0xffff);
            if (kind == 1) {  // MULH: both operands' top limbs without their sign bit, doubled
                lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + 2 * ((uint32_t)x1 - s1 * 0x8000u));
                lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + 2 * ((uint32_t)y1 - s2 * 0x8000u));
            } else if (kind == 3) {  // MULHSU: rs1 signed, rs2 unsigned (its extension is zero)
                lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + 2 * ((uint32_t)x1 - s1 * 0x8000u));
                lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + ((uint32_t)y1 - s2 * 0x8000u));
            }
        }
    }
}

// ---- DIV / DIVU / REM / REMU (DivRemInstruction, riscv/div/div_circuit_v2.rs:391-536 over RInstructionConfig): dividend = quotient * divisor + remainder
// witnessed over u16 limbs (run_divrem :628-697: RISC-V's results for a zero divisor and for the signed overflow), the operands' and the
// quotient's signs, zero flags with the field inverses that prove them, the carries of divisor * quotient + remainder as 18-bit range lookups
// (run_mul_carries :711-752), and |remainder| < |divisor| through remainder' (the remainder with the divisor's sign) and an unsigned comparison's
// marker / difference (run_sltu_diff_idx :699-709).  Field-element columns (the inverses) are canonical Goldilocks values.  39 mapped columns.
struct DivMap {  // ceno_hip_div_column_map = ceno_gpu's DivColumnMap (chips/div.rs:54-84)
    uint32_t pc, ts;
    uint32_t rs1_id, rs1_prev_ts, rs1_lt_diff[2];
    uint32_t rs2_id, rs2_prev_ts, rs2_lt_diff[2];
    uint32_t rd_id, rd_prev_ts, rd_prev_val[2], rd_lt_diff[2];
    uint32_t dividend[2], divisor[2], quotient[2], remainder[2];
    uint32_t dividend_sign, divisor_sign, quotient_sign, remainder_zero, divisor_zero;
    uint32_t divisor_sum_inv, remainder_sum_inv, remainder_inv[2], sign_xor, remainder_prime[2], lt_marker[2], lt_diff;
    uint32_t num_cols;
};
static_assert(sizeof(DivMap) == sizeof(ceno_hip_div_column_map), "column map layout");
constexpr int DIV_COLS = 39;

// SIGNED: DIV / REM (div_kind 0 / 2), otherwise DIVU / REMU (1 / 3): the quotient and the remainder are both witnessed, the kinds of one signedness
// write the same row
template <bool XCD_LOCAL, bool SIGNED>
__global__ void __launch_bounds__(NT) k_witgen_div(DivMap m, const unsigned char* __restrict__ recs, const uint32_t* __restrict__ idx, size_t n, uint64_t offset,
                                                   uint32_t fetch_base, uint32_t fetch_slots, uint64_t* __restrict__ w, size_t rows, uint32_t* lk_dyn,
                                                   uint32_t* lk_fetch) {
    lk_dyn = xcd_copy<XCD_LOCAL>(lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS);
    lk_fetch = xcd_copy<XCD_LOCAL>(lk_fetch, fetch_slots);
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t r = (size_t)blockIdx.x * NT + threadIdx.x; r < rows; r += stride) {
        const Row o{w, rows, r};
        if (r >= n) {
            zero_row<DIV_COLS>(o, &m.pc);
            continue;
        }
        const Step st = load_step(recs, idx[r]);
        const uint64_t ts = st.cycle - offset;
        o.put(m.pc, st.pc);
        o.put(m.ts, ts);
        emit_read<XCD_LOCAL>(o, m.rs1_id, m.rs1_prev_ts, m.rs1_lt_diff, st.rs1_addr, st.rs1_prev, offset, ts + SUBCYCLE_RS1, lk_dyn);
        emit_read<XCD_LOCAL>(o, m.rs2_id, m.rs2_prev_ts, m.rs2_lt_diff, st.rs2_addr, st.rs2_prev, offset, ts + SUBCYCLE_RS2, lk_dyn);
        emit_write<XCD_LOCAL>(o, m.rd_id, m.rd_prev_ts, m.rd_prev_val, m.rd_lt_diff, st.rd_addr, st.rd_before, st.rd_prev, offset, ts + SUBCYCLE_RD, lk_dyn);
        emit_fetch<XCD_LOCAL>(lk_fetch, st.pc, fetch_base, fetch_slots);
        const uint32_t x = st.rs1_val, y = st.rs2_val;
        // run_divrem
        const bool x_sign = SIGNED && (x >> 31), y_sign = SIGNED && (y >> 31);
        const bool zero_divisor = y == 0, overflow = SIGNED && x == 0x80000000u && y == 0xffffffffu;
        uint32_t q, rem;
        bool q_sign;
        if (zero_divisor) {
            q = 0xffffffffu;
            rem = x;
            q_sign = SIGNED;
        } else if (overflow) {
            q = x;
            rem = 0;
            q_sign = false;
        } else {
            const uint32_t xa = x_sign ? 0u - x : x, ya = y_sign ? 0u - y : y;
            const uint32_t qb = xa / ya, rb = xa % ya;
            q = (x_sign != y_sign) ? 0u - qb : qb;
            q_sign = SIGNED && (q >> 31);
            rem = x_sign ? 0u - rb : rb;
        }
        const uint64_t d0 = y & 0xffff, d1 = y >> 16, q0 = q & 0xffff, q1 = q >> 16, r0 = rem & 0xffff, r1 = rem >> 16;
        o.put(m.dividend[0], x & 0xffff);
        o.put(m.dividend[1], x >> 16);
        o.put(m.divisor[0], d0);
        o.put(m.divisor[1], d1);
        o.put(m.quotient[0], q0);
        o.put(m.quotient[1], q1);
        o.put(m.remainder[0], r0);
        o.put(m.remainder[1], r1);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)q0);  // Value::new(quotient), Value::new(remainder): every limb a u16
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)q1);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)r0);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (uint32_t)r1);
        o.put(m.dividend_sign, x_sign);
        o.put(m.divisor_sign, y_sign);
        o.put(m.quotient_sign, q_sign);
        o.put(m.divisor_zero, zero_divisor);
        // run_mul_carries: d * q + r over 16-bit limbs, sign-extended to four limbs
        const uint64_t c0 = (r0 + d0 * q0) >> 16;
        const uint64_t c1 = (r1 + c0 + d0 * q1 + d1 * q0) >> 16;
        const uint64_t q_ext = (q_sign && SIGNED) ? 0xffffu : 0u, d_ext = (d1 >> 15) * (SIGNED ? 0xffffu : 0u), r_ext = (r1 >> 15) * (SIGNED ? 0xffffu : 0u);
        const uint64_t c2 = (c1 + d0 * q_ext + q0 * d_ext + r_ext + d1 * q1) >> 16;
        const uint64_t c3 = (c2 + (d0 + d1) * q_ext + (q0 + q1) * d_ext + r_ext) >> 16;
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 18) + (uint32_t)c0);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 18) + (uint32_t)c2);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 18) + (uint32_t)c1);
        lk_count<XCD_LOCAL>(lk_dyn, (1u << 18) + (uint32_t)c3);
        const bool sign_xor = x_sign != y_sign;
        const uint32_t rp = sign_xor ? 0u - rem : rem;  // remainder_prime
        const uint32_t rp0 = rp & 0xffff, rp1 = rp >> 16;
        const bool remainder_zero = rem == 0 && !zero_divisor;
        o.put(m.remainder_zero, remainder_zero);
        if (SIGNED) {
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (((x >> 16) - (x_sign ? 0x8000u : 0u)) << 1));
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (((uint32_t)d1 - (y_sign ? 0x8000u : 0u)) << 1));
        }
        o.put(m.divisor_sum_inv, gl::inv(d0 + d1));
        o.put(m.remainder_sum_inv, gl::inv(r0 + r1));
        o.put(m.remainder_inv[0], gl::inv(GOLDILOCKS_P - (0x10000u - rp0)));
        o.put(m.remainder_inv[1], gl::inv(GOLDILOCKS_P - (0x10000u - rp1)));
        int lt_idx = 2;
        uint32_t lt_val = 0;
        if (!zero_divisor && !overflow && !remainder_zero) {
            // run_sltu_diff_idx(divisor, remainder', divisor_sign): the most significant limb in which they differ
            lt_idx = (uint32_t)d1 != rp1 ? 1 : 0;
            const uint32_t dl = lt_idx ? (uint32_t)d1 : (uint32_t)d0, rl = lt_idx ? rp1 : rp0;
            lt_val = y_sign ? rl - dl : dl - rl;
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + (lt_val - 1));
        } else {
            lk_count<XCD_LOCAL>(lk_dyn, (1u << 16) + 0u);
        }
        o.put(m.lt_marker[0], lt_idx == 0);
        o.put(m.lt_marker[1], lt_idx == 1);
        o.put(m.sign_xor, sign_xor);
        o.put(m.remainder_prime[0], rp0);
        o.put(m.remainder_prime[1], rp1);
        o.put(m.lt_diff, lt_val);
    }
}

// one launcher for every chip: K<true> counts into per-XCD table copies, K<false> into the caller's tables
#define WITGEN_LAUNCH(KERNEL, ...)                                                                                        \
    [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t* t2, uint32_t* t3) {                                               \
        (void)t2;                                                                                                         \
        (void)t3;                                                                                                         \
        if (xcd) hipLaunchKernelGGL((KERNEL<true>), dim3(grid), dim3(NT), 0, st, __VA_ARGS__);                            \
        else hipLaunchKernelGGL((KERNEL<false>), dim3(grid), dim3(NT), 0, st, __VA_ARGS__);                               \
    }

int witgen_arith(ceno_hip_ctx* ctx, const Map* map, bool sub, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
                 uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->pc, 22, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);  // unmapped columns (num_cols > 22) are left to the caller; mapped ones are fully written
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t*, uint32_t*) {
        if (sub && xcd) hipLaunchKernelGGL((k_witgen_arith<true, true>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
        else if (sub) hipLaunchKernelGGL((k_witgen_arith<true, false>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
        else if (xcd) hipLaunchKernelGGL((k_witgen_arith<false, true>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
        else hipLaunchKernelGGL((k_witgen_arith<false, false>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
    });
}

int witgen_logic(ceno_hip_ctx* ctx, const LogicMap* map, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
                 uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_logic,
                 ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->pc, LOGIC_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {lk_logic, LOGIC_SLOTS}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, WITGEN_LAUNCH(k_witgen_logic, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1, t2));
}

int witgen_logic_i(ceno_hip_ctx* ctx, const LogicIMap* map, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
                   uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_logic,
                   ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->pc, LOGIC_I_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {lk_logic, LOGIC_SLOTS}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, WITGEN_LAUNCH(k_witgen_logic_i, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1, t2));
}

int witgen_addi(ceno_hip_ctx* ctx, const AddiMap* map, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
                uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->pc, ADDI_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, WITGEN_LAUNCH(k_witgen_addi, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1));
}

int witgen_lui(ceno_hip_ctx* ctx, const LuiMap* map, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
               uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->pc, LUI_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, WITGEN_LAUNCH(k_witgen_lui, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1));
}
int witgen_jal(ceno_hip_ctx* ctx, const JalMap* map, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
               uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_du8, uint32_t* lk_xor,
               ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->pc, JAL_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {lk_du8, LOGIC_SLOTS}, {lk_xor, LOGIC_SLOTS}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t* t2, uint32_t* t3) {
        if (xcd) hipLaunchKernelGGL((k_witgen_jal<true>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1, t2, t3);
        else hipLaunchKernelGGL((k_witgen_jal<false>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1, t2, t3);
    });
}

int witgen_auipc(ceno_hip_ctx* ctx, const AuipcMap* map, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
                 uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_du8,
                 uint32_t* lk_xor, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->pc, AUIPC_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {lk_du8, LOGIC_SLOTS}, {lk_xor, LOGIC_SLOTS}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t* t2, uint32_t* t3) {
        if (xcd) hipLaunchKernelGGL((k_witgen_auipc<true>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1, t2, t3);
        else hipLaunchKernelGGL((k_witgen_auipc<false>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1, t2, t3);
    });
}
int witgen_slt(ceno_hip_ctx* ctx, const SltMap* map, int is_signed, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
               uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->rs1_limbs[0], SLT_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, WITGEN_LAUNCH(k_witgen_slt, *map, is_signed, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1));
}
int witgen_slti(ceno_hip_ctx* ctx, const SltiMap* map, int is_signed, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
                uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->rs1_limbs[0], SLTI_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, WITGEN_LAUNCH(k_witgen_slti, *map, is_signed, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1));
}
template <int MODE, class MapT>
int witgen_branch(ceno_hip_ctx* ctx, const MapT* map, int n_cols, int flag, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
                  uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->rs1_limbs[0], n_cols, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t*, uint32_t*) {
        if (xcd) hipLaunchKernelGGL((k_witgen_branch<true, MODE, MapT>), dim3(grid), dim3(NT), 0, st, *map, flag, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
        else hipLaunchKernelGGL((k_witgen_branch<false, MODE, MapT>), dim3(grid), dim3(NT), 0, st, *map, flag, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
    });
}
int witgen_div(ceno_hip_ctx* ctx, const DivMap* map, int kind, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
               uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    CHECK_ARG(ctx, kind >= 0 && kind <= 3, "witgen_div: kind is 0 (DIV), 1 (DIVU), 2 (REM) or 3 (REMU)");
    TRY(witgen_check(ctx, &map->pc, DIV_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const bool is_signed = kind == 0 || kind == 2;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t*, uint32_t*) {
#define DV(X, S) hipLaunchKernelGGL((k_witgen_div<X, S>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1)
        if (xcd) { if (is_signed) DV(true, true); else DV(true, false); }
        else { if (is_signed) DV(false, true); else DV(false, false); }
#undef DV
    });
}

int witgen_mul(ceno_hip_ctx* ctx, const MulMap* map, int kind, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
               uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    CHECK_ARG(ctx, kind >= 0 && kind <= 3, "witgen_mul: kind is 0 (MUL), 1 (MULH), 2 (MULHU) or 3 (MULHSU)");
    const bool high = kind != 0;
    uint32_t cols[MUL_COMMON + 4];
    int nc = MUL_COMMON;
    memcpy(cols, &map->pc, sizeof(uint32_t) * MUL_COMMON);
    if (high) {
        cols[nc++] = map->rd_high[0];
        cols[nc++] = map->rd_high[1];
        cols[nc++] = map->rs1_ext;
        cols[nc++] = map->rs2_ext;
    } else {
        CHECK_ARG(ctx, map->rd_high[0] == CENO_HIP_NO_COLUMN && map->rd_high[1] == CENO_HIP_NO_COLUMN && map->rs1_ext == CENO_HIP_NO_COLUMN &&
                           map->rs2_ext == CENO_HIP_NO_COLUMN,
                  "witgen_mul: MUL has no rd_high / rs1_ext / rs2_ext columns (CENO_HIP_NO_COLUMN)");
    }
    TRY(witgen_check(ctx, cols, nc, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t*, uint32_t*) {
#define ML(X, H) hipLaunchKernelGGL((k_witgen_mul<X, H>), dim3(grid), dim3(NT), 0, st, *map, kind, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1)
        if (xcd) { if (high) ML(true, true); else ML(true, false); }
        else { if (high) ML(false, true); else ML(false, false); }
#undef ML
    });
}

int witgen_load_sub(ceno_hip_ctx* ctx, const LoadSubMap* map, int load_width, int is_signed, const void* recs, size_t num_records, const uint32_t* idx,
                    size_t n, uint64_t offset, uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch,
                    ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    CHECK_ARG(ctx, (load_width == 8 || load_width == 16) && (is_signed == 0 || is_signed == 1), "witgen_load_sub: load_width is 8 or 16, is_signed 0 or 1");
    const bool byte = load_width == 8;
    // the columns this variant has, in map order; the others must be marked absent
    uint32_t cols[LOAD_SUB_COMMON + 4];
    int nc = LOAD_SUB_COMMON;
    memcpy(cols, &map->pc, sizeof(uint32_t) * LOAD_SUB_COMMON);
    if (byte) {
        cols[nc++] = map->addr_bit_0;
        cols[nc++] = map->target_byte;
        cols[nc++] = map->dummy_byte;
    } else {
        CHECK_ARG(ctx, map->addr_bit_0 == CENO_HIP_NO_COLUMN && map->target_byte == CENO_HIP_NO_COLUMN && map->dummy_byte == CENO_HIP_NO_COLUMN,
                  "witgen_load_sub: a halfword load has no byte columns (CENO_HIP_NO_COLUMN)");
    }
    if (is_signed) cols[nc++] = map->msb;
    else CHECK_ARG(ctx, map->msb == CENO_HIP_NO_COLUMN, "witgen_load_sub: an unsigned load has no msb column (CENO_HIP_NO_COLUMN)");
    TRY(witgen_check(ctx, cols, nc, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t*, uint32_t*) {
#define LS(X, B, S) hipLaunchKernelGGL((k_witgen_load_sub<X, B, S>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1)
        if (xcd) { if (byte) { if (is_signed) LS(true, true, true); else LS(true, true, false); } else { if (is_signed) LS(true, false, true); else LS(true, false, false); } }
        else { if (byte) { if (is_signed) LS(false, true, true); else LS(false, true, false); } else { if (is_signed) LS(false, false, true); else LS(false, false, false); } }
#undef LS
    });
}

template <bool IMM, class MapT>
int witgen_shift(ceno_hip_ctx* ctx, const MapT* map, int kind, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset,
                 uint32_t fetch_base, uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, uint32_t* lk_du8, uint32_t* lk_xor,
                 ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    CHECK_ARG(ctx, kind >= 0 && kind <= 2, "witgen_shift: kind is 0 (left), 1 (logical right) or 2 (arithmetic right)");
    TRY(witgen_check(ctx, &map->pc, IMM ? SHIFT_I_COLS : SHIFT_R_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {lk_du8, LOGIC_SLOTS}, {lk_xor, LOGIC_SLOTS}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t* t2, uint32_t* t3) {
        if (xcd) hipLaunchKernelGGL((k_witgen_shift<true, IMM, MapT>), dim3(grid), dim3(NT), 0, st, *map, kind, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1, t2, t3);
        else hipLaunchKernelGGL((k_witgen_shift<false, IMM, MapT>), dim3(grid), dim3(NT), 0, st, *map, kind, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1, t2, t3);
    });
}

int witgen_jalr(ceno_hip_ctx* ctx, const JalrMap* map, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset, uint32_t fetch_base,
                uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->pc, JALR_COLS, map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t*, uint32_t*) {
        if (xcd) hipLaunchKernelGGL(k_witgen_jalr<true>, dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
        else hipLaunchKernelGGL(k_witgen_jalr<false>, dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
    });
}
template <int KIND, class MapT>
int witgen_mem(ceno_hip_ctx* ctx, const MapT* map, const void* recs, size_t num_records, const uint32_t* idx, size_t n, uint64_t offset, uint32_t fetch_base,
               uint32_t fetch_slots, uint64_t* w, size_t rows, uint32_t* lk_dyn, uint32_t* lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, map, "NULL column map");
    TRY(witgen_check(ctx, &map->pc, mem_cols(KIND), map->num_cols, recs, num_records, idx, n, w, rows, lk_fetch, fetch_slots));
    hipStream_t st = ctx_stream(ctx, s);
    const unsigned grid = grid_for(rows, NT, MAXB);
    const unsigned char* rp = (const unsigned char*)recs;
    const LkTab tabs[4] = {{lk_dyn, CENO_HIP_LK_DYNAMIC_SLOTS, DYN_USED_16}, {lk_fetch, fetch_slots}, {nullptr, 0}, {nullptr, 0}};
    return witgen_run(ctx, st, n, tabs, [&](bool xcd, uint32_t* t0, uint32_t* t1, uint32_t*, uint32_t*) {
        if (xcd) hipLaunchKernelGGL((k_witgen_mem<true, KIND, MapT>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
        else hipLaunchKernelGGL((k_witgen_mem<false, KIND, MapT>), dim3(grid), dim3(NT), 0, st, *map, rp, idx, n, offset, fetch_base, fetch_slots, w, rows, t0, t1);
    });
}
#undef WITGEN_LAUNCH

}  // namespace

extern "C" {

int ceno_hip_witgen_add(ceno_hip_ctx* ctx, const ceno_hip_add_column_map* map, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                        uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                        uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_arith(ctx, reinterpret_cast<const Map*>(map), false, dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                        fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_sub(ceno_hip_ctx* ctx, const ceno_hip_sub_column_map* map, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                        uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                        uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_arith(ctx, reinterpret_cast<const Map*>(map), true, dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                        fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_addi(ceno_hip_ctx* ctx, const ceno_hip_addi_column_map* map, const void* dev_step_records, size_t num_records,
                         const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                         uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                         uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_addi(ctx, reinterpret_cast<const AddiMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                       fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_jal(ceno_hip_ctx* ctx, const ceno_hip_jal_column_map* map, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, uint32_t* dev_lk_double_u8,
                        uint32_t* dev_lk_xor, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_jal(ctx, reinterpret_cast<const JalMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle, fetch_base_pc,
                      fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, dev_lk_double_u8, dev_lk_xor, s);
}

int ceno_hip_witgen_auipc(ceno_hip_ctx* ctx, const ceno_hip_auipc_column_map* map, const void* dev_step_records, size_t num_records,
                          const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                          uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch,
                          uint32_t* dev_lk_double_u8, uint32_t* dev_lk_xor, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_auipc(ctx, reinterpret_cast<const AuipcMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle, fetch_base_pc,
                        fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, dev_lk_double_u8, dev_lk_xor, s);
}

int ceno_hip_witgen_slt(ceno_hip_ctx* ctx, const ceno_hip_slt_column_map* map, int is_signed, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    CHECK_ARG(ctx, is_signed == 0 || is_signed == 1, "witgen_slt: is_signed is 1 (SLT) or 0 (SLTU)");
    return witgen_slt(ctx, reinterpret_cast<const SltMap*>(map), is_signed, dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                      fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_slti(ceno_hip_ctx* ctx, const ceno_hip_slti_column_map* map, int is_signed, const void* dev_step_records, size_t num_records,
                         const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                         uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    CHECK_ARG(ctx, is_signed == 0 || is_signed == 1, "witgen_slti: is_signed is 1 (SLTI) or 0 (SLTIU)");
    return witgen_slti(ctx, reinterpret_cast<const SltiMap*>(map), is_signed, dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                       fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_branch_cmp(ceno_hip_ctx* ctx, const ceno_hip_branch_cmp_column_map* map, int is_signed, const void* dev_step_records, size_t num_records,
                               const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                               uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    CHECK_ARG(ctx, is_signed == 0 || is_signed == 1, "witgen_branch_cmp: is_signed is 1 (BLT / BGE) or 0 (BLTU / BGEU)");
    return witgen_branch<0>(ctx, reinterpret_cast<const BranchCmpMap*>(map), BRANCH_CMP_COLS, is_signed, dev_step_records, num_records, dev_step_indices, n,
                            shard_offset_cycle, fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_branch_eq(ceno_hip_ctx* ctx, const ceno_hip_branch_eq_column_map* map, int is_beq, const void* dev_step_records, size_t num_records,
                              const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                              uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    CHECK_ARG(ctx, is_beq == 0 || is_beq == 1, "witgen_branch_eq: is_beq is 1 (BEQ) or 0 (BNE)");
    return witgen_branch<1>(ctx, reinterpret_cast<const BranchEqMap*>(map), BRANCH_EQ_COLS, is_beq, dev_step_records, num_records, dev_step_indices, n,
                            shard_offset_cycle, fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_div(ceno_hip_ctx* ctx, const ceno_hip_div_column_map* map, int div_kind, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_div(ctx, reinterpret_cast<const DivMap*>(map), div_kind, dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                      fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_mul(ceno_hip_ctx* ctx, const ceno_hip_mul_column_map* map, int mul_kind, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_mul(ctx, reinterpret_cast<const MulMap*>(map), mul_kind, dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                      fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_load_sub(ceno_hip_ctx* ctx, const ceno_hip_load_sub_column_map* map, int load_width, int is_signed, const void* dev_step_records,
                             size_t num_records, const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                             uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch,
                             ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_load_sub(ctx, reinterpret_cast<const LoadSubMap*>(map), load_width, is_signed, dev_step_records, num_records, dev_step_indices, n,
                           shard_offset_cycle, fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_sh(ceno_hip_ctx* ctx, const ceno_hip_sh_column_map* map, const void* dev_step_records, size_t num_records,
                       const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                       uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_mem<2>(ctx, reinterpret_cast<const ShMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle, fetch_base_pc,
                         fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_sb(ceno_hip_ctx* ctx, const ceno_hip_sb_column_map* map, const void* dev_step_records, size_t num_records,
                       const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                       uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_mem<3>(ctx, reinterpret_cast<const SbMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle, fetch_base_pc,
                         fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_shift_r(ceno_hip_ctx* ctx, const ceno_hip_shift_r_column_map* map, int kind, const void* dev_step_records, size_t num_records,
                            const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                            uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch,
                            uint32_t* dev_lk_double_u8, uint32_t* dev_lk_xor, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_shift<false>(ctx, reinterpret_cast<const ShiftRMap*>(map), kind, dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                               fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, dev_lk_double_u8, dev_lk_xor, s);
}

int ceno_hip_witgen_shift_i(ceno_hip_ctx* ctx, const ceno_hip_shift_i_column_map* map, int kind, const void* dev_step_records, size_t num_records,
                            const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                            uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch,
                            uint32_t* dev_lk_double_u8, uint32_t* dev_lk_xor, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_shift<true>(ctx, reinterpret_cast<const ShiftIMap*>(map), kind, dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                              fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, dev_lk_double_u8, dev_lk_xor, s);
}

int ceno_hip_witgen_jalr(ceno_hip_ctx* ctx, const ceno_hip_jalr_column_map* map, const void* dev_step_records, size_t num_records,
                         const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                         uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_jalr(ctx, reinterpret_cast<const JalrMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle, fetch_base_pc,
                       fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_lw(ceno_hip_ctx* ctx, const ceno_hip_lw_column_map* map, const void* dev_step_records, size_t num_records,
                       const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                       uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_mem<0>(ctx, reinterpret_cast<const LwMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle, fetch_base_pc,
                             fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_sw(ceno_hip_ctx* ctx, const ceno_hip_sw_column_map* map, const void* dev_step_records, size_t num_records,
                       const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                       uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_mem<1>(ctx, reinterpret_cast<const SwMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle, fetch_base_pc,
                            fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_lui(ceno_hip_ctx* ctx, const ceno_hip_lui_column_map* map, const void* dev_step_records, size_t num_records,
                        const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc, uint32_t fetch_num_slots,
                        uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic, uint32_t* dev_lk_fetch, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    return witgen_lui(ctx, reinterpret_cast<const LuiMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle, fetch_base_pc,
                      fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, s);
}

int ceno_hip_witgen_logic_i(ceno_hip_ctx* ctx, const ceno_hip_logic_i_column_map* map, int logic_kind, const void* dev_step_records,
                            size_t num_records, const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                            uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                            uint32_t* dev_lk_fetch, uint32_t* dev_lk_logic, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    CHECK_ARG(ctx, logic_kind >= 0 && logic_kind <= 2, "witgen_logic_i: kind %d is not ANDI (0) / ORI (1) / XORI (2)", logic_kind);
    return witgen_logic_i(ctx, reinterpret_cast<const LogicIMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                          fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, dev_lk_logic, s);
}

int ceno_hip_witgen_logic_r(ceno_hip_ctx* ctx, const ceno_hip_logic_r_column_map* map, int logic_kind, const void* dev_step_records,
                            size_t num_records, const uint32_t* dev_step_indices, size_t n, uint64_t shard_offset_cycle, uint32_t fetch_base_pc,
                            uint32_t fetch_num_slots, uint64_t* dev_witness_col_major, size_t rows_padded, uint32_t* dev_lk_dynamic,
                            uint32_t* dev_lk_fetch, uint32_t* dev_lk_logic, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    // the witness does not depend on the operation (rd comes from the step record); the kind only names the table dev_lk_logic counts
    CHECK_ARG(ctx, logic_kind >= 0 && logic_kind <= 2, "witgen_logic_r: kind %d is not AND (0) / OR (1) / XOR (2)", logic_kind);
    return witgen_logic(ctx, reinterpret_cast<const LogicMap*>(map), dev_step_records, num_records, dev_step_indices, n, shard_offset_cycle,
                        fetch_base_pc, fetch_num_slots, dev_witness_col_major, rows_padded, dev_lk_dynamic, dev_lk_fetch, dev_lk_logic, s);
}


int ceno_hip_witgen_session_begin(ceno_hip_ctx* ctx, uint32_t* const* dev_tables, const size_t* slots, int n_tables, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    CHECK_ARG(ctx, dev_tables && slots && n_tables >= 1 && n_tables <= 16, "witgen_session_begin: 1 .. 16 tables");
    size_t total = 0;
    for (int t = 0; t < n_tables; t++) {
        CHECK_ARG(ctx, dev_tables[t] && slots[t] > 0, "witgen_session_begin: NULL table / no slots");
        for (int u = 0; u < t; u++) CHECK_ARG(ctx, dev_tables[u] != dev_tables[t], "witgen_session_begin: a table registered twice");
        total += slots[t];
    }
    static const bool xcd_wanted = [] { const char* e = getenv("CENO_HIP_WITGEN_XCD"); return !(e && atoi(e) == 0); }();
    if (!(xcd_wanted && ctx->xcd_private_l2)) return 0;  // the chips count into the caller's tables directly: nothing to hold
    std::lock_guard<std::mutex> g(g_sessions_mu);
    CHECK_ARG(ctx, g_sessions.find(ctx) == g_sessions.end(), "witgen_session_begin: this context has an open session");
    hipStream_t st = ctx_stream(ctx, s);
    WitgenSession S;
    S.stream = st;
    TRY(ctx_alloc(ctx, 8 * total * sizeof(uint32_t), &S.scratch));
    const hipError_t e = hipMemsetAsync(S.scratch, 0, 8 * total * sizeof(uint32_t), st);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(st);
        ctx_free(ctx, S.scratch);
        HIP_TRY(ctx, e);
    }
    hipError_t e_ev = hipEventCreateWithFlags(&S.begun, hipEventDisableTiming);
    if (e_ev == hipSuccess) e_ev = hipEventRecord(S.begun, st);
    if (e_ev != hipSuccess) {
        (void)hipStreamSynchronize(st);
        if (S.begun) (void)hipEventDestroy(S.begun);
        ctx_free(ctx, S.scratch);
        HIP_TRY(ctx, e_ev);
    }
    uint32_t* p = (uint32_t*)S.scratch;
    for (int t = 0; t < n_tables; t++) {
        S.tabs.push_back({dev_tables[t], slots[t], p});
        p += 8 * slots[t];
    }
    g_sessions[ctx] = std::move(S);
    return 0;
}

int ceno_hip_witgen_session_end(ceno_hip_ctx* ctx, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    WitgenSession S;
    {
        std::lock_guard<std::mutex> g(g_sessions_mu);
        auto it = g_sessions.find(ctx);
        if (it == g_sessions.end()) return 0;  // (no per-XCD copies on this device: begin held nothing)
        S = std::move(it->second);
        g_sessions.erase(it);
    }
    hipStream_t st = ctx_stream(ctx, s);
    hipError_t e = st == S.stream ? hipSuccess : hipErrorInvalidValue;
    // the merge follows every chip: the session's stream waits for the other streams the chips ran on (the event is reused: a wait captures the
    // record before it)
    for (hipStream_t o : S.others) {
        if (e != hipSuccess) break;
        e = hipEventRecord(S.begun, o);
        if (e == hipSuccess) e = hipStreamWaitEvent(S.stream, S.begun, 0);
    }
    if (e == hipSuccess) {
        for (const auto& T : S.tabs)
            hipLaunchKernelGGL(k_lk_merge, dim3((unsigned)((T.slots + NT - 1) / NT)), dim3(NT), 0, st, T.copies, T.slots, T.slots, T.user);
        e = hipGetLastError();
    }
    hipError_t e2 = hipStreamSynchronize(S.stream);  // the copies go back to the pool only after the stream has consumed them
    for (hipStream_t o : S.others) {  // (after an error above the other streams may still be writing the copies)
        const hipError_t eo = e == hipSuccess ? hipSuccess : hipStreamSynchronize(o);
        if (e2 == hipSuccess) e2 = eo;
    }
    if (S.begun) (void)hipEventDestroy(S.begun);
    ctx_free(ctx, S.scratch);
    HIP_TRY(ctx, e);
    HIP_TRY(ctx, e2);
    return 0;
}

int ceno_hip_lk_to_mlt_column(ceno_hip_ctx* ctx, const uint32_t* dev_counters, size_t n, uint64_t* dev_column, size_t rows_padded, ceno_hip_stream s) {
    CHECK_ARG(ctx, ctx, "NULL context");
    CHECK_ARG(ctx, dev_counters && dev_column && n <= rows_padded && rows_padded > 0, "lk_to_mlt_column: bad arguments");
    hipStream_t st = ctx_stream(ctx, s);
    hipLaunchKernelGGL(k_lk_to_mlt, dim3(grid_for(rows_padded, NT, MAXB)), dim3(NT), 0, st, dev_counters, n, dev_column, rows_padded);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
