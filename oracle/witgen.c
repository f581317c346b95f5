/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's witness assignment for the R-type
 * chips ADD / SUB (and, at the end of the file, AND / OR / XOR), one instance at a time, in the order the reference's CPU path does it:
 *   ArithInstruction::assign_instance        ceno_zkvm/src/instructions/riscv/arith.rs:101-142
 *   RInstructionConfig::assign_instance      ceno_zkvm/src/instructions/riscv/r_insn.rs:67-86
 *   StateInOut / ReadRS1 / ReadRS2 / WriteRD ceno_zkvm/src/instructions/riscv/insn_base.rs:61-77,112-145,223-257,337-400
 *   InnerLtConfig::assign_instance_field     gkr_iop/src/gadgets/is_lt.rs:243-274 (cal_lt_diff :277-287)
 *   Value::add / Value::new                  ceno_zkvm/src/uint.rs:684-688,762-785
 *   LkMultiplicity keys                      gkr_iop/src/utils/lk_multiplicity.rs:181-198,264-266
 *   ShardContext::aligned_prev_ts            ceno_zkvm/src/e2e.rs:435-451
 * The StepRecord layout is the emulator's #[repr(C)] struct (ceno_emul/src/tracer.rs:33-60, 136 bytes).
 * PARITY: the reference's own test of this path (chips/add.rs:119-188) compares its GPU kernel with this CPU
 * assignment at run time and holds no literal vectors, so this restatement is pinned only by construction
 * ("parity unpinned" beyond the cited code); the step generator of that test is reproduced in tests/.
 */
#include <string.h>
#include "oracle.h"

typedef struct {
    uint32_t addr, value;
    uint64_t previous_cycle;
} orc_read_op;
typedef struct {
    uint32_t addr, before, after, pad_;
    uint64_t previous_cycle;
} orc_write_op;
typedef struct {
    uint64_t cycle;
    uint32_t pc_before, pc_after;
    uint32_t heap_before, heap_after, hint_before, hint_after;
    uint8_t kind, rs1_idx, rs2_idx, rd_idx;
    int32_t imm;
    uint32_t raw;
    uint8_t has_rs1, has_rs2, has_rd, has_memory_op;
    orc_read_op rs1, rs2;
    orc_write_op rd, memory_op;
    uint32_t syscall_index;
    uint8_t future_access_mask, padding_[3];
} orc_step_record;
_Static_assert(sizeof(orc_step_record) == 136, "StepRecord is 136 bytes");

size_t orc_step_record_bytes(void) { return sizeof(orc_step_record); }

/* StepRecord::new_r_instruction (ceno_emul/src/tracer.rs:1164-1186,1355-1407): the shape the reference's tests build */
void orc_step_record_r(void* out, uint64_t cycle, uint32_t pc, uint8_t kind, uint8_t rs1, uint8_t rs2, uint8_t rd, uint32_t rs1_val,
                       uint32_t rs2_val, uint32_t rd_before, uint32_t rd_after, uint64_t prev_cycle) {
    orc_step_record r;
    memset(&r, 0, sizeof(r));
    r.cycle = cycle;
    r.pc_before = pc;
    r.pc_after = pc + 4;
    r.kind = kind; r.rs1_idx = rs1; r.rs2_idx = rs2; r.rd_idx = rd;
    r.has_rs1 = r.has_rs2 = r.has_rd = 1;
    /* Platform::register_vma(idx) = idx << 8 as a byte address; WordAddr = byte address / 4 (platform.rs:120-123, addr.rs:61-65) */
    r.rs1.addr = ((uint32_t)rs1 << 8) / 4; r.rs1.value = rs1_val; r.rs1.previous_cycle = prev_cycle;
    r.rs2.addr = ((uint32_t)rs2 << 8) / 4; r.rs2.value = rs2_val; r.rs2.previous_cycle = prev_cycle;
    r.rd.addr = ((uint32_t)rd << 8) / 4; r.rd.before = rd_before; r.rd.after = rd_after; r.rd.previous_cycle = prev_cycle;
    r.syscall_index = 0xFFFFFFFFu;
    memcpy(out, &r, sizeof(r));
}

static uint64_t aligned_prev_ts(uint64_t prev_cycle, uint64_t offset) {
    uint64_t ts = prev_cycle > offset ? prev_cycle - offset : 0; /* saturating_sub */
    if (ts < 4) ts = 0;                                          /* FullTracer::SUBCYCLES_PER_INSN */
    return ts;
}
static void lk_dyn(uint32_t* t, uint64_t v, unsigned bits) {
    if (t) t[((uint64_t)1 << bits) + v] += 1;
}
/* AssertLtConfig::assign_instance with max_bits = 29: one u16 limb, one 13-bit limb */
static void assign_lt(uint64_t* row, const uint32_t diff_cols[2], uint32_t* lkd, uint64_t lhs, uint64_t rhs) {
    const uint64_t diff = (lhs < rhs ? ((uint64_t)1 << 29) : 0) + lhs - rhs;
    row[diff_cols[0]] = diff & 0xffff;
    lk_dyn(lkd, diff & 0xffff, 16);
    row[diff_cols[1]] = (diff >> 16) & 0xffff;
    lk_dyn(lkd, (diff >> 16) & 0xffff, 13);
}
static uint8_t register_index(uint32_t waddr) { return (uint8_t)((waddr * 4u) >> 8); }

/* cols[23]: the column map in AddColumnMap / SubColumnMap field order (num_cols last).  out: ROW-major n x num_cols
 * (as cpu_assign_instances produces it); lk_dynamic: 2^19 counters or NULL; lk_fetch: slots or NULL. */
int orc_witgen_arith(const uint32_t* cols, int is_sub, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                     uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint32_t num_cols = cols[22];
    for (int c = 0; c < 22; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rs2 || !st->has_rd) return -2; /* step.rs1().expect(..) */
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        /* rs1 */
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        /* rs2 */
        p = aligned_prev_ts(st->rs2.previous_cycle, shard_offset);
        row[cols[6]] = register_index(st->rs2.addr);
        row[cols[7]] = p;
        assign_lt(row, cols + 8, lk_dynamic, p, ts + 1);
        /* rd */
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[10]] = register_index(st->rd.addr);
        row[cols[11]] = p;
        row[cols[12]] = st->rd.before & 0xffff;
        row[cols[13]] = st->rd.before >> 16;
        assign_lt(row, cols + 14, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        /* the addition: ADD rs1 + rs2, SUB rs2 + rd_written; limb-wise with carries, each result limb range-checked */
        uint32_t x, y;
        if (!is_sub) {
            x = st->rs1.value; y = st->rs2.value;
        } else {
            x = st->rs2.value; y = st->rd.after;
            lk_dyn(lk_dynamic, y & 0xffff, 16); /* Value::new(rd.after, lkm) */
            lk_dyn(lk_dynamic, y >> 16, 16);
        }
        row[cols[16]] = x & 0xffff; row[cols[17]] = x >> 16;
        row[cols[18]] = y & 0xffff; row[cols[19]] = y >> 16;
        uint32_t carry = 0;
        for (int l = 0; l < 2; l++) {
            const uint32_t a = (x >> (16 * l)) & 0xffff, b = (y >> (16 * l)) & 0xffff;
            const uint32_t s = a + b + carry;
            carry = s >> 16;
            lk_dyn(lk_dynamic, s & 0xffff, 16);
            row[cols[20 + l]] = carry;
        }
    }
    return 0;
}

/* LogicInstruction::assign_instance (ceno_zkvm/src/instructions/riscv/logic/logic_circuit.rs:66-79,137-159): UInt8::logic_assign
 * counts one entry of the operation's table per byte pair (uint/logic.rs:26-32, key a | b << 8: gkr_iop/src/tables/mod.rs:29-31),
 * then the R-instruction base (r_insn.rs:67-86) and the bytes of rs1, rs2 and rd.value.after (split_to_u8).
 * cols[29]: LogicRColumnMap field order (chips/logic_r.rs:25-42), num_cols last.  lk_logic: 2^16 counters of the op's table or NULL. */
int orc_witgen_logic_r(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                       uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_logic) {
    const uint32_t num_cols = cols[28];
    for (int c = 0; c < 28; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rs2 || !st->has_rd) return -2;
        for (int b = 0; b < 4; b++) {
            const uint32_t x = (st->rs1.value >> (8 * b)) & 0xff, y = (st->rs2.value >> (8 * b)) & 0xff;
            if (lk_logic) lk_logic[x | (y << 8)] += 1;
        }
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rs2.previous_cycle, shard_offset);
        row[cols[6]] = register_index(st->rs2.addr);
        row[cols[7]] = p;
        assign_lt(row, cols + 8, lk_dynamic, p, ts + 1);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[10]] = register_index(st->rd.addr);
        row[cols[11]] = p;
        row[cols[12]] = st->rd.before & 0xffff;
        row[cols[13]] = st->rd.before >> 16;
        assign_lt(row, cols + 14, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        for (int b = 0; b < 4; b++) {
            row[cols[16 + b]] = (st->rs1.value >> (8 * b)) & 0xff;
            row[cols[20 + b]] = (st->rs2.value >> (8 * b)) & 0xff;
            row[cols[24 + b]] = (st->rd.after >> (8 * b)) & 0xff;
        }
    }
    return 0;
}

/* StepRecord::new_i_instruction (ceno_emul/src/tracer.rs:1210-1230): rs1 read, rd written, no rs2; the immediate travels in insn.imm */
void orc_step_record_i(void* out, uint64_t cycle, uint32_t pc, uint8_t kind, uint8_t rs1, uint8_t rd, int32_t imm, uint32_t rs1_val,
                       uint32_t rd_before, uint32_t rd_after, uint64_t prev_cycle) {
    orc_step_record r;
    memset(&r, 0, sizeof(r));
    r.cycle = cycle;
    r.pc_before = pc;
    r.pc_after = pc + 4;
    r.kind = kind; r.rs1_idx = rs1; r.rs2_idx = 0; r.rd_idx = rd;
    r.imm = imm;
    r.has_rs1 = r.has_rd = 1;
    r.rs1.addr = ((uint32_t)rs1 << 8) / 4; r.rs1.value = rs1_val; r.rs1.previous_cycle = prev_cycle;
    r.rd.addr = ((uint32_t)rd << 8) / 4; r.rd.before = rd_before; r.rd.after = rd_after; r.rd.previous_cycle = prev_cycle;
    r.syscall_index = 0xFFFFFFFFu;
    memcpy(out, &r, sizeof(r));
}

/* AddiInstruction::assign_instance (arith_imm/arith_imm_circuit_v2.rs:85-117): imm = insn.imm as i16 as u16, imm_sign = its sign,
 * rs1 + [imm, sign ? 0xffff : 0] with overflow (every result limb range-checked: uint.rs:762-785), then the I-instruction base
 * (i_insn.rs:66-82: state, rs1, rd, fetch).  cols[19]: AddiColumnMap field order (chips/addi.rs:27-42), num_cols last. */
int orc_witgen_addi(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                    uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint32_t num_cols = cols[18];
    for (int c = 0; c < 18; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rd) return -2;
        const uint32_t x = st->rs1.value;
        const uint16_t imm = (uint16_t)(int16_t)st->imm;
        const int negative = (int16_t)st->imm < 0;
        row[cols[14]] = imm;
        row[cols[15]] = negative ? 1 : 0;
        const uint32_t ext[2] = {imm, negative ? 0xffffu : 0u};
        row[cols[12]] = x & 0xffff;
        row[cols[13]] = x >> 16;
        uint32_t carry = 0;
        for (int l = 0; l < 2; l++) {
            const uint32_t s = ((x >> (16 * l)) & 0xffff) + ext[l] + carry;
            carry = s >> 16;
            lk_dyn(lk_dynamic, s & 0xffff, 16);
            row[cols[16 + l]] = carry;
        }
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[6]] = register_index(st->rd.addr);
        row[cols[7]] = p;
        row[cols[8]] = st->rd.before & 0xffff;
        row[cols[9]] = st->rd.before >> 16;
        assign_lt(row, cols + 10, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
    }
    return 0;
}

/* LogicInstruction (imm) ::assign_instance (logic_imm/logic_imm_circuit_v2.rs:105-130) and LogicConfig::assign_instance (:195-224):
 * imm_lo = imm_internal(insn).0 as u32 & 0xffff, imm_hi = (imm_signed_internal(insn).0 as u32 >> 16) & 0xffff with
 * imm_internal = imm as i16 as i64 and imm_signed_internal = (imm >> 16) as i16 as i64 for ANDI / ORI / XORI (tables/program.rs:115,140-143);
 * logic_assign over the two bytes of (rs1_lo, imm_lo) and of (rs1_hi, imm_hi); then the I-instruction base and the byte columns.
 * cols[25]: LogicIColumnMap field order (chips/logic_i.rs:26-41), num_cols last. */
int orc_witgen_logic_i(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                       uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_logic) {
    const uint32_t num_cols = cols[24];
    for (int c = 0; c < 24; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rd) return -2;
        const uint32_t rs1_lo = st->rs1.value & 0xffff, rs1_hi = (st->rs1.value >> 16) & 0xffff;
        const uint32_t imm_internal = (uint32_t)(int64_t)(int16_t)st->imm;          /* imm as i16 as i64, then as u32 */
        const uint32_t imm_signed = (uint32_t)(int64_t)(int16_t)(st->imm >> 16);    /* (imm >> LIMB_BITS) as i16 as i64, then as u32 */
        const uint32_t imm_lo = imm_internal & 0xffff, imm_hi = (imm_signed >> 16) & 0xffff;
        for (int b = 0; b < 2; b++)
            if (lk_logic) lk_logic[((rs1_lo >> (8 * b)) & 0xff) | (((imm_lo >> (8 * b)) & 0xff) << 8)] += 1;
        for (int b = 0; b < 2; b++)
            if (lk_logic) lk_logic[((rs1_hi >> (8 * b)) & 0xff) | (((imm_hi >> (8 * b)) & 0xff) << 8)] += 1;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[6]] = register_index(st->rd.addr);
        row[cols[7]] = p;
        row[cols[8]] = st->rd.before & 0xffff;
        row[cols[9]] = st->rd.before >> 16;
        assign_lt(row, cols + 10, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        for (int b = 0; b < 4; b++) {
            row[cols[12 + b]] = (st->rs1.value >> (8 * b)) & 0xff;
            row[cols[16 + b]] = (st->rd.after >> (8 * b)) & 0xff;
        }
        row[cols[20]] = imm_internal & 0xff;          /* split_to_u8(imm_internal as u32)[..2] */
        row[cols[21]] = (imm_internal >> 8) & 0xff;
        row[cols[22]] = (imm_signed >> 16) & 0xff;    /* split_to_u8(imm_signed_internal as u32)[2..] */
        row[cols[23]] = imm_signed >> 24;
    }
    return 0;
}

/* LuiInstruction::assign_instance (riscv/lui.rs:100-120): the I-instruction base, then bytes 1..3 of rd.value.after — each
 * assert_ux::<8> (LookupTable::Dynamic key 2^8 + v) — and imm = imm_internal(insn).0 = insn.imm as u32 >> 12 (U type).
 * cols[17]: LuiColumnMap field order (chips/lui.rs:29-41), num_cols last. */
int orc_witgen_lui(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint32_t num_cols = cols[16];
    for (int c = 0; c < 16; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rd) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[6]] = register_index(st->rd.addr);
        row[cols[7]] = p;
        row[cols[8]] = st->rd.before & 0xffff;
        row[cols[9]] = st->rd.before >> 16;
        assign_lt(row, cols + 10, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        for (int b = 1; b < 4; b++) {
            const uint32_t v = (st->rd.after >> (8 * b)) & 0xff;
            lk_dyn(lk_dynamic, v, 8);
            row[cols[12 + (b - 1)]] = v;
        }
        row[cols[15]] = (uint32_t)st->imm >> 12;
    }
    return 0;
}

/* StepRecord::new_j_instruction (ceno_emul/src/tracer.rs:1285-1304): only rd is written; pc changes to the jump target */
void orc_step_record_j(void* out, uint64_t cycle, uint32_t pc, uint32_t pc_after, uint8_t kind, uint8_t rd, int32_t imm, uint32_t rd_before,
                       uint32_t rd_after, uint64_t prev_cycle) {
    orc_step_record r;
    memset(&r, 0, sizeof(r));
    r.cycle = cycle;
    r.pc_before = pc;
    r.pc_after = pc_after;
    r.kind = kind; r.rd_idx = rd;
    r.imm = imm;
    r.has_rd = 1;
    r.rd.addr = ((uint32_t)rd << 8) / 4; r.rd.before = rd_before; r.rd.after = rd_after; r.rd.previous_cycle = prev_cycle;
    r.syscall_index = 0xFFFFFFFFu;
    memcpy(out, &r, sizeof(r));
}

#define ORC_PC_MSB_MASK 0xC0u /* sum of 2^x, x = PC_BITS - 24 .. 7 with PC_BITS = 30 (riscv/constants.rs:29; jal_v2.rs:120-124, auipc.rs:180-184) */

/* JalInstruction::assign_instance (riscv/jump/jal_v2.rs:99-127) over JInstructionConfig (j_insn.rs:58-73: pc, next_pc, ts, rd, fetch):
 * rd.value.after as four bytes, assert_double_u8 per pair (key a << 8 | b, lk_multiplicity.rs:200-203), logic_u8::<XorTable>(byte 3, 0xC0).
 * cols[14]: JalColumnMap field order (chips/jal.rs:21-31), num_cols last. */
int orc_witgen_jal(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_double_u8, uint32_t* lk_xor) {
    const uint32_t num_cols = cols[13];
    for (int c = 0; c < 13; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rd) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = st->pc_after;
        row[cols[2]] = ts;
        const uint64_t p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[3]] = register_index(st->rd.addr);
        row[cols[4]] = p;
        row[cols[5]] = st->rd.before & 0xffff;
        row[cols[6]] = st->rd.before >> 16;
        assign_lt(row, cols + 7, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        uint32_t b[4];
        for (int k = 0; k < 4; k++) row[cols[9 + k]] = b[k] = (st->rd.after >> (8 * k)) & 0xff;
        if (lk_double_u8) { lk_double_u8[(b[0] << 8) + b[1]] += 1; lk_double_u8[(b[2] << 8) + b[3]] += 1; }
        if (lk_xor) lk_xor[b[3] | (ORC_PC_MSB_MASK << 8)] += 1;
    }
    return 0;
}

/* AuipcInstruction::assign_instance (riscv/auipc.rs:149-187): the I-instruction base, rd bytes (double_u8 pairs), pc bytes 1 and 2 and the
 * three bytes of imm_internal = insn.imm as u32 >> 8 (tables/program.rs:119-124), each assert_ux::<8>, and logic_u8::<XorTable>(pc byte 3, 0xC0).
 * cols[22]: AuipcColumnMap field order (chips/auipc.rs:28-44), num_cols last. */
int orc_witgen_auipc(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                     uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_double_u8, uint32_t* lk_xor) {
    const uint32_t num_cols = cols[21];
    for (int c = 0; c < 21; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rd) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[6]] = register_index(st->rd.addr);
        row[cols[7]] = p;
        row[cols[8]] = st->rd.before & 0xffff;
        row[cols[9]] = st->rd.before >> 16;
        assign_lt(row, cols + 10, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        uint32_t b[4];
        for (int k = 0; k < 4; k++) row[cols[12 + k]] = b[k] = (st->rd.after >> (8 * k)) & 0xff;
        if (lk_double_u8) { lk_double_u8[(b[0] << 8) + b[1]] += 1; lk_double_u8[(b[2] << 8) + b[3]] += 1; }
        for (int k = 0; k < 2; k++) {
            const uint32_t v = (st->pc_before >> (8 * (k + 1))) & 0xff;
            lk_dyn(lk_dynamic, v, 8);
            row[cols[16 + k]] = v;
        }
        const uint32_t imm = (uint32_t)st->imm >> 8;
        for (int k = 0; k < 3; k++) {
            const uint32_t v = (imm >> (8 * k)) & 0xff;
            lk_dyn(lk_dynamic, v, 8);
            row[cols[18 + k]] = v;
        }
        if (lk_xor) lk_xor[(st->pc_before >> 24) | (ORC_PC_MSB_MASK << 8)] += 1;
    }
    return 0;
}

/* UIntLimbsLT::assign (gadgets/signed_limbs.rs:150-222) over run_cmp (:226-236).  lt_cols: cmp_lt, a_msb_f, b_msb_f, diff_marker[2], diff_val */
static void assign_uint_lt(uint64_t* row, const uint32_t* lt_cols, uint32_t* lk_dynamic, const uint16_t a[2], const uint16_t b[2], int is_signed) {
    const uint64_t P = 0xFFFFFFFF00000001ULL;
    const int is_a_neg = (a[1] >> 15) == 1 && is_signed, is_b_neg = (b[1] >> 15) == 1 && is_signed;
    int cmp_lt = 0, diff_idx = 2;
    for (int k = 1; k >= 0; k--)
        if (a[k] != b[k]) { cmp_lt = (a[k] < b[k]) ^ is_a_neg ^ is_b_neg; diff_idx = k; break; }
    row[lt_cols[3]] = diff_idx == 0;
    row[lt_cols[4]] = diff_idx == 1;
    row[lt_cols[0]] = (uint64_t)cmp_lt;
    uint64_t a_msb_f, b_msb_f;
    uint16_t a_range, b_range;
    if (is_a_neg) { a_msb_f = P - (uint64_t)((1u << 16) - a[1]); a_range = (uint16_t)(a[1] - (1u << 15)); }
    else { a_msb_f = a[1]; a_range = (uint16_t)(a[1] + ((uint16_t)(is_signed != 0) << 15)); }
    if (is_b_neg) { b_msb_f = P - (uint64_t)((1u << 16) - b[1]); b_range = (uint16_t)(b[1] - (1u << 15)); }
    else { b_msb_f = b[1]; b_range = (uint16_t)(b[1] + ((uint16_t)(is_signed != 0) << 15)); }
    row[lt_cols[1]] = a_msb_f;
    row[lt_cols[2]] = b_msb_f;
    uint16_t diff_val;
    if (diff_idx == 2) diff_val = 0;
    else if (diff_idx == 1) { /* field subtraction, canonical, then `as u16` */
        const uint64_t x = cmp_lt ? b_msb_f : a_msb_f, y = cmp_lt ? a_msb_f : b_msb_f;
        const uint64_t d = x >= y ? x - y : x + (P - y);
        diff_val = (uint16_t)d;
    } else diff_val = (uint16_t)(cmp_lt ? b[0] - a[0] : a[0] - b[0]);
    row[lt_cols[5]] = diff_val;
    lk_dyn(lk_dynamic, diff_idx != 2 ? (uint16_t)(diff_val - 1) : 0, 16);
    lk_dyn(lk_dynamic, a_range, 16);
    lk_dyn(lk_dynamic, b_range, 16);
}

/* SetLessThanInstruction::assign_instance (riscv/slt/slt_circuit_v2.rs:86-119): the R-instruction base, the u16 limbs of rs1 and rs2, then
 * UIntLimbsLT::assign (gadgets/signed_limbs.rs:150-222) over run_cmp (:226-236).  Field elements are Goldilocks-canonical words.
 * cols[27]: SltColumnMap field order (chips/slt.rs:33-55), num_cols last. */
int orc_witgen_slt(const uint32_t* cols, int is_signed, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                   uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint32_t num_cols = cols[26];
    for (int c = 0; c < 26; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rs2 || !st->has_rd) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[10]] = st->pc_before;
        row[cols[11]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[12]] = register_index(st->rs1.addr);
        row[cols[13]] = p;
        assign_lt(row, cols + 14, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rs2.previous_cycle, shard_offset);
        row[cols[16]] = register_index(st->rs2.addr);
        row[cols[17]] = p;
        assign_lt(row, cols + 18, lk_dynamic, p, ts + 1);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[20]] = register_index(st->rd.addr);
        row[cols[21]] = p;
        row[cols[22]] = st->rd.before & 0xffff;
        row[cols[23]] = st->rd.before >> 16;
        assign_lt(row, cols + 24, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        const uint16_t a[2] = {(uint16_t)st->rs1.value, (uint16_t)(st->rs1.value >> 16)}, b[2] = {(uint16_t)st->rs2.value, (uint16_t)(st->rs2.value >> 16)};
        row[cols[0]] = a[0]; row[cols[1]] = a[1]; row[cols[2]] = b[0]; row[cols[3]] = b[1];
        assign_uint_lt(row, cols + 4, lk_dynamic, a, b, is_signed);
    }
    return 0;
}

/* SetLessThanImmInstruction::assign_instance (riscv/slti/slti_circuit_v2.rs:104-140): the I-instruction base, rs1 limbs, imm = insn.imm as i16
 * as u16 and its sign, then UIntLimbsLT::assign(rs1, imm_sign_extend(true, imm), is SLTI).  cols[23]: SltiColumnMap order (chips/slti.rs:32-51). */
int orc_witgen_slti(const uint32_t* cols, int is_signed, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                    uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint32_t num_cols = cols[22];
    for (int c = 0; c < 22; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rd) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[10]] = st->pc_before;
        row[cols[11]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[12]] = register_index(st->rs1.addr);
        row[cols[13]] = p;
        assign_lt(row, cols + 14, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[16]] = register_index(st->rd.addr);
        row[cols[17]] = p;
        row[cols[18]] = st->rd.before & 0xffff;
        row[cols[19]] = st->rd.before >> 16;
        assign_lt(row, cols + 20, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        const uint16_t a[2] = {(uint16_t)st->rs1.value, (uint16_t)(st->rs1.value >> 16)};
        const uint16_t imm = (uint16_t)(int16_t)st->imm;
        const uint16_t b[2] = {imm, (int16_t)st->imm < 0 ? 0xffff : 0};
        row[cols[0]] = a[0]; row[cols[1]] = a[1];
        row[cols[2]] = imm;
        row[cols[3]] = b[1] > 0;
        assign_uint_lt(row, cols + 4, lk_dynamic, a, b, is_signed);
    }
    return 0;
}

/* StepRecord::new_b_instruction (ceno_emul/src/tracer.rs): rs1 and rs2 read, no rd; pc moves to pc + imm when the branch is taken */
void orc_step_record_b(void* out, uint64_t cycle, uint32_t pc, uint32_t pc_after, uint8_t kind, uint8_t rs1, uint8_t rs2, int32_t imm, uint32_t rs1_val,
                       uint32_t rs2_val, uint64_t prev_cycle) {
    orc_step_record r;
    memset(&r, 0, sizeof(r));
    r.cycle = cycle;
    r.pc_before = pc;
    r.pc_after = pc_after;
    r.kind = kind; r.rs1_idx = rs1; r.rs2_idx = rs2;
    r.imm = imm;
    r.has_rs1 = r.has_rs2 = 1;
    r.rs1.addr = ((uint32_t)rs1 << 8) / 4; r.rs1.value = rs1_val; r.rs1.previous_cycle = prev_cycle;
    r.rs2.addr = ((uint32_t)rs2 << 8) / 4; r.rs2.value = rs2_val; r.rs2.previous_cycle = prev_cycle;
    r.syscall_index = 0xFFFFFFFFu;
    memcpy(out, &r, sizeof(r));
}

static uint64_t gl_mul_(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % 0xFFFFFFFF00000001ULL); }
static uint64_t gl_inv_(uint64_t a) { /* a^(p-2) */
    uint64_t e = 0xFFFFFFFF00000001ULL - 2, r = 1;
    while (e) { if (e & 1) r = gl_mul_(r, a); a = gl_mul_(a, a); e >>= 1; }
    return r;
}

/* BranchCircuit::assign_instance (riscv/branch/branch_circuit_v2.rs:143-209) over BInstructionConfig::assign_instance (b_insn.rs:92-116: pc, next_pc,
 * ts, rs1, rs2, imm = imm_internal(insn).1 = i64_to_base(insn.imm), fetch).  is_eq = 0: BLT / BGE / BLTU / BGEU, cols[23] in BranchCmpColumnMap order
 * (chips/branch_cmp.rs:35-54), flag = is_signed.  is_eq = 1: BEQ / BNE, cols[20] in BranchEqColumnMap order (chips/branch_eq.rs:27-43), flag = is_beq. */
int orc_witgen_branch(const uint32_t* cols, int is_eq, int flag, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                      uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint64_t P = 0xFFFFFFFF00000001ULL;
    const int nc = is_eq ? 19 : 22, base = is_eq ? 7 : 10; /* first column of the b_insn block: pc */
    const uint32_t num_cols = cols[nc];
    for (int c = 0; c < nc; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rs2) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[base]] = st->pc_before;
        row[cols[base + 1]] = st->pc_after;
        row[cols[base + 2]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[base + 3]] = register_index(st->rs1.addr);
        row[cols[base + 4]] = p;
        assign_lt(row, cols + base + 5, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rs2.previous_cycle, shard_offset);
        row[cols[base + 7]] = register_index(st->rs2.addr);
        row[cols[base + 8]] = p;
        assign_lt(row, cols + base + 9, lk_dynamic, p, ts + 1);
        row[cols[base + 11]] = st->imm < 0 ? P - (uint64_t)(-(int64_t)st->imm) : (uint64_t)st->imm;
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        const uint16_t a[2] = {(uint16_t)st->rs1.value, (uint16_t)(st->rs1.value >> 16)}, b[2] = {(uint16_t)st->rs2.value, (uint16_t)(st->rs2.value >> 16)};
        row[cols[0]] = a[0]; row[cols[1]] = a[1]; row[cols[2]] = b[0]; row[cols[3]] = b[1];
        if (!is_eq) {
            assign_uint_lt(row, cols + 4, lk_dynamic, a, b, flag);
        } else { /* run_eq: the first differing limb from the least significant one */
            int taken = flag, diff_idx = 0;
            uint64_t inv = 0;
            for (int k = 0; k < 2; k++)
                if (a[k] != b[k]) {
                    taken = !flag;
                    diff_idx = k;
                    inv = gl_inv_(a[k] > b[k] ? (uint64_t)(a[k] - b[k]) : P - (uint64_t)(b[k] - a[k]));
                    break;
                }
            row[cols[4]] = (uint64_t)taken;
            row[cols[5 + diff_idx]] = inv; /* the other marker stays zero */
        }
    }
    return 0;
}

/* StepRecord::new_im_instruction / new_s_instruction (ceno_emul/src/tracer.rs:1232-1260,1306-1330): a load reads rs1, writes rd and reads memory; a
 * store reads rs1 and rs2 and writes memory.  memory_op.addr is a WORD address. */
void orc_step_record_mem(void* out, int is_store, uint64_t cycle, uint32_t pc, uint8_t kind, uint8_t rs1, uint8_t rs2_or_rd, int32_t imm, uint32_t rs1_val,
                         uint32_t rs2_val, uint32_t rd_before, uint32_t rd_after, uint32_t mem_byte_addr, uint32_t mem_before, uint32_t mem_after,
                         uint64_t prev_cycle, uint64_t mem_prev_cycle) {
    orc_step_record r;
    memset(&r, 0, sizeof(r));
    r.cycle = cycle;
    r.pc_before = pc;
    r.pc_after = pc + 4;
    r.kind = kind; r.rs1_idx = rs1;
    r.imm = imm;
    r.has_rs1 = 1;
    r.rs1.addr = ((uint32_t)rs1 << 8) / 4; r.rs1.value = rs1_val; r.rs1.previous_cycle = prev_cycle;
    if (is_store) {
        r.rs2_idx = rs2_or_rd; r.has_rs2 = 1;
        r.rs2.addr = ((uint32_t)rs2_or_rd << 8) / 4; r.rs2.value = rs2_val; r.rs2.previous_cycle = prev_cycle;
    } else {
        r.rd_idx = rs2_or_rd; r.has_rd = 1;
        r.rd.addr = ((uint32_t)rs2_or_rd << 8) / 4; r.rd.before = rd_before; r.rd.after = rd_after; r.rd.previous_cycle = prev_cycle;
    }
    r.has_memory_op = 1;
    r.memory_op.addr = mem_byte_addr / 4; r.memory_op.before = mem_before; r.memory_op.after = mem_after; r.memory_op.previous_cycle = mem_prev_cycle;
    r.syscall_index = 0xFFFFFFFFu;
    memcpy(out, &r, sizeof(r));
}

/* MemAddr::assign_instance for a word-aligned address with max_bits = MEM_BITS = 30 (insn_base.rs:880-905) + ReadMEM / WriteMEM::assign_op (:517-545) */
static void assign_mem(uint64_t* row, uint32_t prev_col, const uint32_t diff_cols[2], const uint32_t addr_cols[2], uint32_t* lkd, uint32_t addr,
                       uint64_t prev_cycle, uint64_t shard_offset, uint64_t ts) {
    const uint64_t p = aligned_prev_ts(prev_cycle, shard_offset);
    row[prev_col] = p;
    assign_lt(row, diff_cols, lkd, p, ts + 3); /* Tracer::SUBCYCLE_MEM */
    row[addr_cols[0]] = addr & 0xffff;
    row[addr_cols[1]] = addr >> 16;
    lk_dyn(lkd, (addr & 0xffff) >> 2, 14);
    lk_dyn(lkd, addr >> 16, 14); /* min(30 - 16, 16) */
}

/* LW: LoadInstruction::assign_instance (riscv/memory/load_v2.rs:197-255) + IMInstructionConfig (im_insn.rs:71-90); cols[24] in LwColumnMap order.
 * SW: StoreInstruction::assign_instance (riscv/memory/store_v2.rs:138-177) + SInstructionConfig (s_insn.rs:77-96); cols[24] in SwColumnMap order. */
/* is_store 2 = SH (cols[25]: + mem_addr_bit_1), 3 = SB (cols[30]: + bit_0, bit_1, prev_limb_bytes[2], rs2_limb_byte, expected_limb): StoreConfig<E, 1> / <E, 0>
 * with MemWordUtil::assign_instance (riscv/memory/gadget.rs:134-185) */
int orc_witgen_mem(const uint32_t* cols, int is_store, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const int nc = is_store == 2 ? 24 : is_store == 3 ? 29 : 23;
    const uint32_t num_cols = cols[nc];
    for (int c = 0; c < nc; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_memory_op || (is_store ? !st->has_rs2 : !st->has_rd)) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        const uint16_t imm = (uint16_t)(int16_t)st->imm;
        const int negative = (int16_t)st->imm < 0;
        const uint32_t addr = st->rs1.value + (uint32_t)(int32_t)(int16_t)st->imm; /* wrapping_add_signed(imm_internal.0 as i32) */
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        if (!is_store) {
            p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
            row[cols[6]] = register_index(st->rd.addr);
            row[cols[7]] = p;
            row[cols[8]] = st->rd.before & 0xffff;
            row[cols[9]] = st->rd.before >> 16;
            assign_lt(row, cols + 10, lk_dynamic, p, ts + 2);
            row[cols[15]] = st->rs1.value & 0xffff; row[cols[16]] = st->rs1.value >> 16;
            row[cols[17]] = imm;
            row[cols[18]] = negative ? 1 : 0;
            assign_mem(row, cols[12], cols + 13, cols + 19, lk_dynamic, addr, st->memory_op.previous_cycle, shard_offset, ts);
            row[cols[21]] = st->memory_op.before & 0xffff; row[cols[22]] = st->memory_op.before >> 16;
        } else {
            p = aligned_prev_ts(st->rs2.previous_cycle, shard_offset);
            row[cols[6]] = register_index(st->rs2.addr);
            row[cols[7]] = p;
            assign_lt(row, cols + 8, lk_dynamic, p, ts + 1);
            row[cols[13]] = st->rs1.value & 0xffff; row[cols[14]] = st->rs1.value >> 16;
            row[cols[15]] = st->rs2.value & 0xffff; row[cols[16]] = st->rs2.value >> 16;
            row[cols[17]] = imm;
            row[cols[18]] = negative ? 1 : 0;
            row[cols[19]] = st->memory_op.before & 0xffff; row[cols[20]] = st->memory_op.before >> 16;
            lk_dyn(lk_dynamic, st->memory_op.before & 0xffff, 16); /* Value::new(memory_op.value.before, lkm) */
            lk_dyn(lk_dynamic, st->memory_op.before >> 16, 16);
            assign_mem(row, cols[10], cols + 11, cols + 21, lk_dynamic, addr, st->memory_op.previous_cycle, shard_offset, ts);
            if (is_store == 2) row[cols[23]] = (addr >> 1) & 1;
            if (is_store == 3) {
                const uint32_t bit0 = addr & 1, bit1 = (addr >> 1) & 1;
                const uint32_t prev_limb = (st->memory_op.before >> (16 * bit1)) & 0xffff, rs2_limb = st->rs2.value & 0xffff;
                row[cols[23]] = bit0;
                row[cols[24]] = bit1;
                row[cols[25]] = prev_limb & 0xff;
                row[cols[26]] = prev_limb >> 8;
                row[cols[27]] = rs2_limb & 0xff;
                row[cols[28]] = bit0 ? ((rs2_limb & 0xff) << 8) + (prev_limb & 0xff) : ((prev_limb >> 8) << 8) + (rs2_limb & 0xff);
                lk_dyn(lk_dynamic, prev_limb & 0xff, 8);
                lk_dyn(lk_dynamic, prev_limb >> 8, 8);
                lk_dyn(lk_dynamic, rs2_limb & 0xff, 8);
                lk_dyn(lk_dynamic, rs2_limb >> 8, 8);
            }
        }
    }
    return 0;
}


/* JalrInstruction::assign_instance (riscv/jump/jalr_v2.rs:146-190): imm = insn.imm as i16 as u16 with its sign, rd.after as limbs (low: 16-bit range,
 * high: PC_BITS - 16 = 14-bit range, witnessed as rd_high), rs1 limbs, the target rs1 + sign-extended imm as a MemAddr with both low bits witnessed and
 * max_bits = PC_BITS (insn_base.rs:880-905), then the I-instruction base with a branching state (pc, next_pc, ts).  cols[23]: JalrColumnMap order. */
int orc_witgen_jalr(const uint32_t* cols, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                    uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint32_t num_cols = cols[22];
    for (int c = 0; c < 22; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rd) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = st->pc_after;
        row[cols[2]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[3]] = register_index(st->rs1.addr);
        row[cols[4]] = p;
        assign_lt(row, cols + 5, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[7]] = register_index(st->rd.addr);
        row[cols[8]] = p;
        row[cols[9]] = st->rd.before & 0xffff;
        row[cols[10]] = st->rd.before >> 16;
        assign_lt(row, cols + 11, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        const uint32_t target = st->rs1.value + (uint32_t)(int32_t)(int16_t)st->imm;
        row[cols[13]] = st->rs1.value & 0xffff;
        row[cols[14]] = st->rs1.value >> 16;
        row[cols[15]] = (uint16_t)(int16_t)st->imm;
        row[cols[16]] = (int16_t)st->imm < 0 ? 1 : 0;
        row[cols[17]] = target & 0xffff;
        row[cols[18]] = target >> 16;
        row[cols[19]] = target & 1;
        row[cols[20]] = (target >> 1) & 1;
        row[cols[21]] = st->rd.after >> 16;
        lk_dyn(lk_dynamic, st->rd.after & 0xffff, 16);
        lk_dyn(lk_dynamic, st->rd.after >> 16, 14); /* PC_BITS - 16 */
        lk_dyn(lk_dynamic, (target & 0xffff) >> 2, 14);
        lk_dyn(lk_dynamic, target >> 16, 14);
    }
    return 0;
}


/* ShiftLogicalInstruction / ShiftImmInstruction::assign_instance (riscv/shift/shift_circuit_v2.rs:359-396,485-521) with ShiftBaseConfig::assign_instances
 * (:242-293) over byte limbs: kind 0 = SLL / SLLI, 1 = SRL / SRLI, 2 = SRA / SRAI.  cols[48] in ShiftRColumnMap order or cols[41] in ShiftIColumnMap order. */
int orc_witgen_shift(const uint32_t* cols, int is_imm, int kind, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                     uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch, uint32_t* lk_double_u8,
                     uint32_t* lk_xor) {
    const int nc = is_imm ? 40 : 47;
    const uint32_t num_cols = cols[nc];
    if (kind < 0 || kind > 2) return -3;
    for (int c = 0; c < nc; c++)
        if (cols[c] >= num_cols) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rd || (!is_imm && !st->has_rs2)) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        const uint32_t* k = cols;
        row[k[0]] = st->pc_before;
        row[k[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[k[2]] = register_index(st->rs1.addr);
        row[k[3]] = p;
        assign_lt(row, k + 4, lk_dynamic, p, ts + 0);
        k += 6;
        uint32_t c;
        if (!is_imm) {
            p = aligned_prev_ts(st->rs2.previous_cycle, shard_offset);
            row[k[0]] = register_index(st->rs2.addr);
            row[k[1]] = p;
            assign_lt(row, k + 2, lk_dynamic, p, ts + 1);
            k += 4;
            c = st->rs2.value;
        } else {
            c = (uint16_t)(int16_t)st->imm;
        }
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[k[0]] = register_index(st->rd.addr);
        row[k[1]] = p;
        row[k[2]] = st->rd.before & 0xffff;
        row[k[3]] = st->rd.before >> 16;
        assign_lt(row, k + 4, lk_dynamic, p, ts + 2);
        k += 6;
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        const uint32_t b = st->rs1.value, d = st->rd.after;
        for (int j = 0; j < 4; j++) row[k[j]] = (b >> (8 * j)) & 0xff;
        k += 4;
        if (!is_imm) {
            for (int j = 0; j < 4; j++) row[k[j]] = (c >> (8 * j)) & 0xff;
            k += 4;
        }
        for (int j = 0; j < 4; j++) row[k[j]] = (d >> (8 * j)) & 0xff;
        k += 4;
        if (lk_double_u8) {
            lk_double_u8[((d & 0xff) << 8) + ((d >> 8) & 0xff)] += 1;
            lk_double_u8[(((d >> 16) & 0xff) << 8) + (d >> 24)] += 1;
        }
        if (is_imm) { row[k[0]] = c; k += 1; }
        const uint32_t c0 = c & 0xff, shift = c0 % 32, limb_shift = shift / 8, bit_shift = shift % 8;
        for (uint32_t j = 0; j < 8; j++) row[k[j]] = j == bit_shift;
        k += 8;
        for (uint32_t j = 0; j < 4; j++) row[k[j]] = j == limb_shift;
        k += 4;
        row[k[0]] = kind == 0 ? (1u << bit_shift) : 0;  /* bit_multiplier_left: only the shift's own column is set (:254-265) */
        row[k[1]] = kind == 0 ? 0 : (1u << bit_shift);
        uint32_t sign = 0;
        if (kind == 2) {
            sign = b >> 31;
            if (lk_xor) lk_xor[(b >> 24) | (128u << 8)] += 1;
        }
        row[k[2]] = sign;
        k += 3;
        for (int j = 0; j < 4; j++) {
            const uint32_t byte = (b >> (8 * j)) & 0xff;
            const uint32_t carry = kind == 0 ? (byte >> (8 - bit_shift)) : (byte % (1u << bit_shift));
            row[k[j]] = carry;
            if (lk_dynamic) lk_dynamic[(1u << bit_shift) + carry] += 1; /* assert_dynamic_range(carry, bit_shift): no skip at 0 or 1 bits */
        }
        lk_dyn(lk_dynamic, (c0 - bit_shift - limb_shift * 8) >> 5, 3);
    }
    return 0;
}


/* LoadInstruction::assign_instance for LH / LHU / LB / LBU (riscv/memory/load_v2.rs:197-255): LW's assignment, the selected limb, (bytes: the addressed and
 * the other byte of it, byte-range lookups), (signed: SignedExtendConfig::assign_instance, gadgets/signed_ext.rs:92-103).  cols[30]: LoadSubColumnMap field
 * order, 0xFFFFFFFF in the Option fields the variant lacks, num_cols last. */
int orc_witgen_load_sub(const uint32_t* cols, int load_width, int is_signed, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset,
                        uint32_t fetch_base_pc, uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint32_t num_cols = cols[29];
    const int byte = load_width == 8;
    if ((load_width != 8 && load_width != 16) || (is_signed != 0 && is_signed != 1)) return -3;
    for (int c = 0; c < 29; c++) {
        const int present = c < 25 || (c < 28 ? byte : is_signed);
        if (present ? cols[c] >= num_cols : cols[c] != 0xFFFFFFFFu) return -1;
    }
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rd || !st->has_memory_op) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        const uint32_t addr = st->rs1.value + (uint32_t)(int32_t)(int16_t)st->imm, word = st->memory_op.before;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[6]] = register_index(st->rd.addr);
        row[cols[7]] = p;
        row[cols[8]] = st->rd.before & 0xffff;
        row[cols[9]] = st->rd.before >> 16;
        assign_lt(row, cols + 10, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        row[cols[15]] = st->rs1.value & 0xffff; row[cols[16]] = st->rs1.value >> 16;
        row[cols[17]] = (uint16_t)(int16_t)st->imm;
        row[cols[18]] = (int16_t)st->imm < 0 ? 1 : 0;
        assign_mem(row, cols[12], cols + 13, cols + 19, lk_dynamic, addr, st->memory_op.previous_cycle, shard_offset, ts);
        row[cols[21]] = word & 0xffff; row[cols[22]] = word >> 16;
        const uint32_t bit0 = addr & 1, bit1 = (addr >> 1) & 1, limb = (word >> (16 * bit1)) & 0xffff;
        row[cols[23]] = bit1;
        row[cols[24]] = limb;
        uint32_t val = limb;
        if (byte) {
            const uint32_t target = (limb >> (8 * bit0)) & 0xff, other = (limb >> (8 * (1 - bit0))) & 0xff;
            row[cols[25]] = bit0;
            row[cols[26]] = target;
            row[cols[27]] = other;
            lk_dyn(lk_dynamic, target, 8);
            lk_dyn(lk_dynamic, other, 8);
            val = target;
        }
        if (is_signed) {
            const int bits = byte ? 8 : 16;
            const uint32_t msb = val >> (bits - 1);
            row[cols[28]] = msb;
            lk_dyn(lk_dynamic, 2 * val - (msb << bits), bits);
        }
    }
    return 0;
}


/* MulhInstructionBase::assign_instance (riscv/mulh/mulh_circuit_v2.rs:234-333) with run_mulh (:427-487) for two 16-bit limbs: kind 0 = MUL, 1 = MULH,
 * 2 = MULHU, 3 = MULHSU.  cols[27]: MulColumnMap field order (0xFFFFFFFF in rd_high[2], rs1_ext, rs2_ext for MUL), num_cols last. */
int orc_witgen_mul(const uint32_t* cols, int kind, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint32_t num_cols = cols[26];
    if (kind < 0 || kind > 3) return -3;
    for (int c = 0; c < 26; c++)
        if ((c < 22 || kind != 0) ? cols[c] >= num_cols : cols[c] != 0xFFFFFFFFu) return -1;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rs2 || !st->has_rd) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rs2.previous_cycle, shard_offset);
        row[cols[6]] = register_index(st->rs2.addr);
        row[cols[7]] = p;
        assign_lt(row, cols + 8, lk_dynamic, p, ts + 1);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[10]] = register_index(st->rd.addr);
        row[cols[11]] = p;
        row[cols[12]] = st->rd.before & 0xffff;
        row[cols[13]] = st->rd.before >> 16;
        assign_lt(row, cols + 14, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        const uint32_t x[2] = {st->rs1.value & 0xffff, st->rs1.value >> 16}, y[2] = {st->rs2.value & 0xffff, st->rs2.value >> 16};
        row[cols[16]] = x[0]; row[cols[17]] = x[1];
        row[cols[18]] = y[0]; row[cols[19]] = y[1];
        uint64_t mul[2] = {0, 0}, carry[4] = {0, 0, 0, 0};
        for (int a = 0; a < 2; a++) {
            if (a > 0) mul[a] = carry[a - 1];
            for (int j = 0; j <= a; j++) mul[a] += (uint64_t)(x[j] * y[a - j]);
            carry[a] = mul[a] >> 16;
            mul[a] %= 1 << 16;
        }
        const uint32_t x_ext = (x[1] >> 15) * (kind == 2 ? 0u : 0xffffu), y_ext = (y[1] >> 15) * (kind == 1 ? 0xffffu : 0u);
        uint64_t mulh[2] = {0, 0};
        uint32_t xp = 0, yp = 0;
        for (int a = 0; a < 2; a++) {
            xp += x[a];
            yp += y[a];
            mulh[a] = carry[2 + a - 1] + (uint64_t)xp * y_ext + (uint64_t)yp * x_ext;
            for (int j = a + 1; j < 2; j++) mulh[a] += (uint64_t)(x[j] * y[2 + a - j]);
            carry[2 + a] = mulh[a] >> 16;
            mulh[a] %= 1 << 16;
        }
        for (int a = 0; a < 2; a++) {
            row[cols[20 + a]] = mul[a];
            lk_dyn(lk_dynamic, mul[a], 16);
            lk_dyn(lk_dynamic, carry[a], 18);
        }
        if (kind != 0) {
            for (int a = 0; a < 2; a++) {
                row[cols[22 + a]] = mulh[a];
                lk_dyn(lk_dynamic, mulh[a], 16);
                lk_dyn(lk_dynamic, carry[2 + a], 18);
            }
            row[cols[24]] = x_ext;
            row[cols[25]] = y_ext;
            const uint32_t s1 = x_ext / 0xffff, s2 = y_ext / 0xffff;
            if (kind == 1) {
                lk_dyn(lk_dynamic, 2 * (x[1] - s1 * 0x8000), 16);
                lk_dyn(lk_dynamic, 2 * (y[1] - s2 * 0x8000), 16);
            } else if (kind == 3) {
                lk_dyn(lk_dynamic, 2 * (x[1] - s1 * 0x8000), 16);
                lk_dyn(lk_dynamic, y[1] - s2 * 0x8000, 16);
            }
        }
    }
    return 0;
}


/* DivRemInstruction::assign_instance (riscv/div/div_circuit_v2.rs:391-536) with run_divrem (:628-697), run_mul_carries (:711-752) and run_sltu_diff_idx
 * (:699-709) over two 16-bit limbs: kind 0 = DIV, 1 = DIVU, 2 = REM, 3 = REMU.  cols[40]: DivColumnMap field order, num_cols last. */
static void div_negate(const uint32_t x[2], uint32_t out[2]) { /* negate (:776-783) */
    uint32_t carry = 1;
    for (int i = 0; i < 2; i++) {
        const uint32_t val = (1u << 16) + carry - 1 - x[i];
        carry = val >> 16;
        out[i] = val % (1u << 16);
    }
}
int orc_witgen_div(const uint32_t* cols, int kind, const void* records, const uint32_t* indices, size_t n, uint64_t shard_offset, uint32_t fetch_base_pc,
                   uint32_t fetch_num_slots, uint64_t* out_row_major, uint32_t* lk_dynamic, uint32_t* lk_fetch) {
    const uint64_t P = 0xFFFFFFFF00000001ULL;
    const uint32_t num_cols = cols[39];
    if (kind < 0 || kind > 3) return -3;
    for (int c = 0; c < 39; c++)
        if (cols[c] >= num_cols) return -1;
    const int is_signed = kind == 0 || kind == 2;
    const orc_step_record* recs = (const orc_step_record*)records;
    for (size_t i = 0; i < n; i++) {
        const orc_step_record* st = &recs[indices[i]];
        uint64_t* row = out_row_major + i * num_cols;
        if (!st->has_rs1 || !st->has_rs2 || !st->has_rd) return -2;
        const uint64_t ts = st->cycle - shard_offset;
        row[cols[0]] = st->pc_before;
        row[cols[1]] = ts;
        uint64_t p = aligned_prev_ts(st->rs1.previous_cycle, shard_offset);
        row[cols[2]] = register_index(st->rs1.addr);
        row[cols[3]] = p;
        assign_lt(row, cols + 4, lk_dynamic, p, ts + 0);
        p = aligned_prev_ts(st->rs2.previous_cycle, shard_offset);
        row[cols[6]] = register_index(st->rs2.addr);
        row[cols[7]] = p;
        assign_lt(row, cols + 8, lk_dynamic, p, ts + 1);
        p = aligned_prev_ts(st->rd.previous_cycle, shard_offset);
        row[cols[10]] = register_index(st->rd.addr);
        row[cols[11]] = p;
        row[cols[12]] = st->rd.before & 0xffff;
        row[cols[13]] = st->rd.before >> 16;
        assign_lt(row, cols + 14, lk_dynamic, p, ts + 2);
        if (lk_fetch) {
            const uint32_t slot = (st->pc_before - fetch_base_pc) / 4;
            if (slot < fetch_num_slots) lk_fetch[slot] += 1;
        }
        const uint32_t x[2] = {st->rs1.value & 0xffff, st->rs1.value >> 16}, y[2] = {st->rs2.value & 0xffff, st->rs2.value >> 16};
        /* run_divrem */
        const int x_sign = is_signed && (x[1] >> 15) == 1, y_sign = is_signed && (y[1] >> 15) == 1;
        const int zero_divisor = y[0] == 0 && y[1] == 0;
        const int overflow = x[1] == 0x8000 && x[0] == 0 && y[0] == 0xffff && y[1] == 0xffff && x_sign && y_sign;
        uint32_t q[2], r[2];
        int q_sign, special = zero_divisor ? 1 : overflow ? 2 : 0;
        if (zero_divisor) {
            q[0] = q[1] = 0xffff; r[0] = x[0]; r[1] = x[1]; q_sign = is_signed;
        } else if (overflow) {
            q[0] = x[0]; q[1] = x[1]; r[0] = r[1] = 0; q_sign = 0;
        } else {
            uint32_t xa[2] = {x[0], x[1]}, ya[2] = {y[0], y[1]};
            if (x_sign) div_negate(x, xa);
            if (y_sign) div_negate(y, ya);
            const uint32_t xb = xa[0] + (xa[1] << 16), yb = ya[0] + (ya[1] << 16), qb = xb / yb, rb = xb % yb;
            const uint32_t ql[2] = {qb & 0xffff, qb >> 16}, rl[2] = {rb & 0xffff, rb >> 16};
            if (x_sign ^ y_sign) div_negate(ql, q); else { q[0] = ql[0]; q[1] = ql[1]; }
            q_sign = is_signed && (q[1] >> 15) == 1;
            if (x_sign) div_negate(rl, r); else { r[0] = rl[0]; r[1] = rl[1]; }
        }
        row[cols[16]] = x[0]; row[cols[17]] = x[1];
        row[cols[18]] = y[0]; row[cols[19]] = y[1];
        for (int a = 0; a < 2; a++) {
            row[cols[20 + a]] = q[a];
            row[cols[22 + a]] = r[a];
        }
        lk_dyn(lk_dynamic, q[0], 16); lk_dyn(lk_dynamic, q[1], 16);   /* Value::new(quotient, lkm) */
        lk_dyn(lk_dynamic, r[0], 16); lk_dyn(lk_dynamic, r[1], 16);   /* Value::new(remainder, lkm) */
        row[cols[24]] = x_sign;
        row[cols[25]] = y_sign;
        row[cols[26]] = q_sign;
        row[cols[28]] = special == 1;
        /* run_mul_carries(signed, d = divisor, q, r, q_sign) */
        uint32_t carry[4] = {0, 0, 0, 0};
        for (int a = 0; a < 2; a++) {
            uint64_t val = (uint64_t)r[a] + (a > 0 ? carry[a - 1] : 0);
            for (int j = 0; j <= a; j++) val += (uint64_t)y[j] * q[a - j];
            carry[a] = (uint32_t)(val >> 16);
        }
        const uint32_t q_ext = (q_sign && is_signed) ? 0xffff : 0, d_ext = (y[1] >> 15) * (is_signed ? 0xffffu : 0u), r_ext = (r[1] >> 15) * (is_signed ? 0xffffu : 0u);
        uint32_t d_prefix = 0, q_prefix = 0;
        for (int a = 0; a < 2; a++) {
            d_prefix += y[a];
            q_prefix += q[a];
            uint64_t val = (uint64_t)carry[2 + a - 1] + (uint64_t)d_prefix * q_ext + (uint64_t)q_prefix * d_ext + r_ext;
            for (int j = a + 1; j < 2; j++) val += (uint64_t)y[j] * q[2 + a - j];
            carry[2 + a] = (uint32_t)(val >> 16);
        }
        for (int a = 0; a < 2; a++) {
            lk_dyn(lk_dynamic, carry[a], 18);
            lk_dyn(lk_dynamic, carry[a + 2], 18);
        }
        const int sign_xor = x_sign ^ y_sign;
        uint32_t rp[2] = {r[0], r[1]};
        if (sign_xor) div_negate(r, rp);
        const int remainder_zero = r[0] == 0 && r[1] == 0 && special != 1;
        row[cols[27]] = remainder_zero;
        if (is_signed) {
            lk_dyn(lk_dynamic, ((uint64_t)x[1] - (x_sign ? 0x8000 : 0)) << 1, 16);
            lk_dyn(lk_dynamic, ((uint64_t)y[1] - (y_sign ? 0x8000 : 0)) << 1, 16);
        }
        row[cols[29]] = gl_inv_((uint64_t)y[0] + y[1]);
        row[cols[30]] = gl_inv_((uint64_t)r[0] + r[1]);
        row[cols[31]] = gl_inv_(P - (0x10000 - (uint64_t)rp[0]));
        row[cols[32]] = gl_inv_(P - (0x10000 - (uint64_t)rp[1]));
        int lt_idx = 2;
        uint32_t lt_val = 0;
        if (special == 0 && !remainder_zero) {
            for (int a = 1; a >= 0; a--)
                if (y[a] != rp[a]) { lt_idx = a; break; }
            if (lt_idx == 2) return -4; /* |remainder| = |divisor|: the reference panics here */
            lt_val = y_sign ? rp[lt_idx] - y[lt_idx] : y[lt_idx] - rp[lt_idx];
            lk_dyn(lk_dynamic, (uint64_t)lt_val - 1, 16);
        } else {
            lk_dyn(lk_dynamic, 0, 16);
        }
        row[cols[33]] = sign_xor;
        row[cols[34]] = rp[0];
        row[cols[35]] = rp[1];
        row[cols[36]] = lt_idx == 0;
        row[cols[37]] = lt_idx == 1;
        row[cols[38]] = lt_val;
    }
    return 0;
}
