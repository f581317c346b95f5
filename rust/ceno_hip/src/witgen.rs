//! On-device witness generation (`hal.witgen.*` of the reference's CUDA HAL as called from
//! `ceno_zkvm/src/instructions/gpu/dispatch.rs:509-650`): one call per chip kind turns the shard's `StepRecord`s — already on the
//! device, in the emulator's `#[repr(C)]` layout (`ceno_emul/src/tracer.rs:33-60`, 136 bytes) — into the chip's COLUMN-major
//! witness matrix and adds the chip's lookup multiplicities to the shard's tables.
//!
//! The column maps are the `#[repr(C)]` structs of `ceno_hip_sys`, field for field those of `ceno_gpu::common::witgen::types`
//! (`AddColumnMap` `chips/add.rs:29-46`, `SubColumnMap` `chips/sub.rs:28-45`, `LogicRColumnMap` `chips/logic_r.rs:25-42`,
//! `AddiColumnMap` `chips/addi.rs:27-42`, `LogicIColumnMap` `chips/logic_i.rs:26-41`, `LuiColumnMap` `chips/lui.rs:29-41`,
//! `AuipcColumnMap` `chips/auipc.rs:28-44`, `JalColumnMap` `chips/jal.rs:21-31`): the
//! dispatch arm builds them with the reference's own `extract_*_column_map` helpers and passes them through.
use std::sync::Arc;

use ceno_hip_sys as sys;

use crate::{
    error::Result,
    hal::{raw_stream, HipHal, HipStream},
};

/// Device pointers of the lookup-multiplicity tables one shard accumulates into (`LkMultiplicity`, one counter per key); a null
/// pointer skips that table.  `logic` is the 2^16-entry table of the operation being generated (`LookupTable::And` / `Or` / `Xor`).
#[derive(Clone, Copy)]
pub struct LkTables {
    /// `LookupTable::Dynamic`: 2^19 counters (DYNAMIC_RANGE_MAX_BITS = 18), key `(1 << bits) + value`
    pub dynamic: *mut u32,
    /// `LookupTable::Instruction`: one counter per program slot, key `(pc - fetch_base_pc) / 4`
    pub fetch: *mut u32,
    pub fetch_base_pc: u32,
    pub fetch_num_slots: u32,
    pub logic: *mut u32,
    /// `LookupTable::DoubleU8`: 2^16 counters, key `a << 8 | b` (AUIPC, JAL)
    pub double_u8: *mut u32,
    /// `LookupTable::Xor`: 2^16 counters, key `a | b << 8` (the program-counter range check of AUIPC and JAL)
    pub xor: *mut u32,
}

/// The step records of a shard on the device and the steps that belong to one chip.
pub struct ChipSteps<'a> {
    /// `num_records` x 136 bytes
    pub dev_records: *const u8,
    pub num_records: usize,
    /// device array of `n` indices into the records
    pub dev_indices: *const u32,
    pub n: usize,
    /// `ShardContext::current_shard_offset_cycle()`
    pub shard_offset_cycle: u64,
    pub stream: Option<&'a HipStream>,
}

/// `AND` / `OR` / `XOR` (and `ANDI` / `ORI` / `XORI`): the payload of `GpuWitgenKind::LogicR` / `LogicI`
#[derive(Clone, Copy, PartialEq, Eq)]
#[repr(i32)]
pub enum LogicKind {
    And = 0,
    Or = 1,
    Xor = 2,
}

pub struct Witgen {
    hal: Arc<HipHal>,
}

impl Witgen {
    pub fn new(hal: &Arc<HipHal>) -> Self {
        Self { hal: hal.clone() }
    }

    /// `witgen_add`: `dev_witness` = `map.num_cols` columns of `rows_padded` words; rows `>= steps.n` are zeroed
    pub fn add(&self, map: &sys::ceno_hip_add_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_add(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                     lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
    /// `witgen_sub`
    pub fn sub(&self, map: &sys::ceno_hip_sub_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_sub(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                     lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
    /// `witgen_logic_r`
    pub fn logic_r(&self, map: &sys::ceno_hip_logic_r_column_map, kind: LogicKind, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
                   lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_logic_r(self.hal.ctx, map, kind as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                         steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                         lk.logic, raw_stream(steps.stream))
        })
    }
    /// `witgen_addi`
    pub fn addi(&self, map: &sys::ceno_hip_addi_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_addi(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                      lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
    /// `witgen_logic_i`
    pub fn logic_i(&self, map: &sys::ceno_hip_logic_i_column_map, kind: LogicKind, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
                   lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_logic_i(self.hal.ctx, map, kind as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                         steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                         lk.logic, raw_stream(steps.stream))
        })
    }
    /// `witgen_auipc`
    pub fn auipc(&self, map: &sys::ceno_hip_auipc_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_auipc(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                       lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, lk.double_u8, lk.xor,
                                       raw_stream(steps.stream))
        })
    }
    /// `witgen_jal`
    pub fn jal(&self, map: &sys::ceno_hip_jal_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_jal(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                     lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, lk.double_u8, lk.xor,
                                     raw_stream(steps.stream))
        })
    }
    /// `witgen_slt`: `is_signed` = the payload of `GpuWitgenKind::Slt` (true: SLT, false: SLTU)
    pub fn slt(&self, map: &sys::ceno_hip_slt_column_map, is_signed: bool, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
               lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_slt(self.hal.ctx, map, is_signed as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                     steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                     raw_stream(steps.stream))
        })
    }
    /// `witgen_slti`: `is_signed` = the payload of `GpuWitgenKind::Slti` (true: SLTI, false: SLTIU)
    pub fn slti(&self, map: &sys::ceno_hip_slti_column_map, is_signed: bool, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
                lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_slti(self.hal.ctx, map, is_signed as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                      steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                      raw_stream(steps.stream))
        })
    }
    /// `witgen_branch_cmp`: BLT / BGE (`is_signed`) and BLTU / BGEU
    pub fn branch_cmp(&self, map: &sys::ceno_hip_branch_cmp_column_map, is_signed: bool, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
                      lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_branch_cmp(self.hal.ctx, map, is_signed as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                            steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                            raw_stream(steps.stream))
        })
    }
    /// `witgen_branch_eq`: BEQ (`is_beq`) and BNE
    pub fn branch_eq(&self, map: &sys::ceno_hip_branch_eq_column_map, is_beq: bool, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
                     lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_branch_eq(self.hal.ctx, map, is_beq as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                           steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                           raw_stream(steps.stream))
        })
    }
    /// `witgen_div`: DIV (`div_kind` 0), DIVU (1), REM (2), REMU (3)
    pub fn div(&self, map: &sys::ceno_hip_div_column_map, div_kind: u32, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
               lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_div(self.hal.ctx, map, div_kind as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                     steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                     raw_stream(steps.stream))
        })
    }
    /// `witgen_mul`: MUL (`mul_kind` 0), MULH (1), MULHU (2), MULHSU (3); `rd_high`, `rs1_ext`, `rs2_ext` hold `sys::CENO_HIP_NO_COLUMN` for MUL
    pub fn mul(&self, map: &sys::ceno_hip_mul_column_map, mul_kind: u32, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
               lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_mul(self.hal.ctx, map, mul_kind as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                     steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                     raw_stream(steps.stream))
        })
    }
    /// `witgen_load_sub`: LH / LHU (`load_width` 16) and LB / LBU (8); Option columns the variant lacks hold `sys::CENO_HIP_NO_COLUMN`
    pub fn load_sub(&self, map: &sys::ceno_hip_load_sub_column_map, load_width: u32, is_signed: bool, steps: &ChipSteps, dev_witness: *mut u64,
                    rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_load_sub(self.hal.ctx, map, load_width as i32, is_signed as i32, steps.dev_records.cast(), steps.num_records,
                                          steps.dev_indices, steps.n, steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness,
                                          rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
    /// `witgen_sh`: the halfword store
    pub fn sh(&self, map: &sys::ceno_hip_sh_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_sh(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                    lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
    /// `witgen_sb`: the byte store
    pub fn sb(&self, map: &sys::ceno_hip_sb_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_sb(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                    lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
    /// `witgen_jalr`: the indirect jump (target as a MemAddr with both low bits witnessed, rd = pc + 4)
    pub fn jalr(&self, map: &sys::ceno_hip_jalr_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_jalr(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                      lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
    /// `witgen_shift_r`: SLL (`kind` 0), SRL (1), SRA (2)
    pub fn shift_r(&self, map: &sys::ceno_hip_shift_r_column_map, kind: u32, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
                   lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_shift_r(self.hal.ctx, map, kind as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                         steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                         lk.double_u8, lk.xor, raw_stream(steps.stream))
        })
    }
    /// `witgen_shift_i`: SLLI (`kind` 0), SRLI (1), SRAI (2)
    pub fn shift_i(&self, map: &sys::ceno_hip_shift_i_column_map, kind: u32, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize,
                   lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_shift_i(self.hal.ctx, map, kind as i32, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n,
                                         steps.shard_offset_cycle, lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch,
                                         lk.double_u8, lk.xor, raw_stream(steps.stream))
        })
    }
    /// `witgen_lw`: the word load (register read, register write, memory read, address range checks)
    pub fn lw(&self, map: &sys::ceno_hip_lw_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_lw(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                    lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
    /// `witgen_sw`: the word store (two register reads, memory write, address range checks)
    pub fn sw(&self, map: &sys::ceno_hip_sw_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_sw(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                    lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
    /// `witgen_lui`
    pub fn lui(&self, map: &sys::ceno_hip_lui_column_map, steps: &ChipSteps, dev_witness: *mut u64, rows_padded: usize, lk: &LkTables) -> Result<()> {
        self.hal.check(unsafe {
            sys::ceno_hip_witgen_lui(self.hal.ctx, map, steps.dev_records.cast(), steps.num_records, steps.dev_indices, steps.n, steps.shard_offset_cycle,
                                     lk.fetch_base_pc, lk.fetch_num_slots, dev_witness, rows_padded, lk.dynamic, lk.fetch, raw_stream(steps.stream))
        })
    }
}
