//! Host-layer entry points (`libceno_prover.so`, `include/ceno_prover.h`) driven with a RUST transcript: the C side sees a
//! `ceno_transcript` function table whose callbacks land in an `impl FsTranscript`.
//!   `create_chip_proof`                 <-> `ZKVMProver::create_chip_proof`        `ceno_zkvm/src/scheme/prover.rs:717-833`
//!   `prove_batched_main_constraints`    <-> `BatchedMainConstraintProver`           `ceno_zkvm/src/scheme/cpu/mod.rs:1052-1390`
//!   `lanes_run`                         <-> `ChipScheduler::execute`                `ceno_zkvm/src/scheme/scheduler.rs:231-336`
use std::{ffi::c_void, marker::PhantomData, os::raw::c_int, ptr, slice, sync::Arc};

use ceno_hip_sys as sys;

use crate::{
    error::Result,
    hal::{raw_stream, HipHal, HipStream},
    mle::HipMle,
    tower::TowerProofWords,
    ExtWords, FsTranscript,
};

/// `ceno_transcript` view of a Rust transcript, valid while the borrow lives
pub struct CTranscript<'a, T: FsTranscript> {
    table: sys::ceno_transcript,
    _borrow: PhantomData<&'a mut T>,
}
impl<'a, T: FsTranscript> CTranscript<'a, T> {
    pub fn new(t: &'a mut T) -> Self {
        unsafe extern "C" fn label<T: FsTranscript>(s: *mut c_void, b: *const u8, n: usize) {
            (*(s as *mut T)).append_bytes(slice::from_raw_parts(b, n));
        }
        unsafe extern "C" fn ext<T: FsTranscript>(s: *mut c_void, e: *const u64) {
            (*(s as *mut T)).append_ext([*e, *e.add(1)]);
        }
        // the C loops call append_label(label) and then sample_ext(), i.e. `sample_and_append_challenge` in two steps
        unsafe extern "C" fn sample<T: FsTranscript>(s: *mut c_void, o: *mut u64) {
            let r = (*(s as *mut T)).sample();
            *o = r[0];
            *o.add(1) = r[1];
        }
        unsafe extern "C" fn base<T: FsTranscript>(s: *mut c_void, v: u64) {
            (*(s as *mut T)).append_base(v);
        }
        unsafe extern "C" fn bits<T: FsTranscript>(s: *mut c_void, bits: c_int) -> u64 {
            (*(s as *mut T)).sample_bits(bits as usize) as u64
        }
        unsafe extern "C" fn grind<T: FsTranscript>(s: *mut c_void, bits: c_int) -> u64 {
            (*(s as *mut T)).grind(bits as usize)
        }
        unsafe extern "C" fn export<T: FsTranscript>(s: *mut c_void, out16: *mut u64) -> c_int {
            match (*(s as *mut T)).export_duplex_state() {
                Some(w) => {
                    ptr::copy_nonoverlapping(w.as_ptr(), out16, 16);
                    sys::CENO_TRANSCRIPT_DUPLEX8 as c_int
                }
                None => 0,
            }
        }
        Self {
            // no fork / import: a borrowed Rust transcript is neither cloned nor overwritten from the C side
            table: sys::ceno_transcript { append_label: Some(label::<T>), append_ext: Some(ext::<T>), sample_ext: Some(sample::<T>), self_: t as *mut T as *mut c_void,
                                          destroy: None, append_base: Some(base::<T>), sample_bits: Some(bits::<T>), fork: None, fork_free: None,
                                          export_state: Some(export::<T>), import_state: None, grind: Some(grind::<T>) },
            _borrow: PhantomData,
        }
    }
    pub fn raw(&mut self) -> *mut sys::ceno_transcript {
        &mut self.table
    }
}

/// one chip of `create_chip_proof`: tables and the record expressions in monomial form (coefficients already evaluated at
/// the two global challenges); outputs in the order reads, writes, lookup numerators, lookup denominators
pub struct ChipTask<'a> {
    pub circuit_idx: usize,
    pub num_instances: usize,
    pub log2_num_instances: usize,
    pub rotation_vars: usize,
    pub n_witin: usize,
    pub n_fixed: usize,
    pub n_structural: usize,
    pub mles: Vec<Option<&'a HipMle>>, // witness ++ fixed ++ structural (structural may be absent at this stage)
    pub num_reads: usize,
    pub num_writes: usize,
    pub num_lk_tables: usize,
    pub num_lk: usize,
    pub record_coeffs: Vec<ExtWords>,
    pub record_terms: Vec<Vec<usize>>,
    pub record_out_term_offsets: Vec<u32>,
    pub rotation: Option<(Vec<(usize, usize)>, usize, usize)>, // (source, target) pairs, cyclic_subgroup_size, cyclic_group_log2
}

/// `ZKVMChipProof` (`ceno_zkvm/src/scheme.rs:59-76`) in boundary words + the main point for the deferred `MainConstraintJob`
pub struct ChipProofWords {
    pub r_out_evals: Vec<ExtWords>,
    pub w_out_evals: Vec<ExtWords>,
    pub lk_out_evals: Vec<ExtWords>,
    pub tower: TowerProofWords,
    pub rt_tower: Vec<ExtWords>,
    pub rt_main: Vec<ExtWords>,
    pub rotation: Option<(Vec<Vec<ExtWords>>, Vec<ExtWords>, [Vec<ExtWords>; 3])>, // msgs, [left, right, target] evals, origin / left / right points
}

unsafe fn ext_slice(p: *const u64, n: usize) -> Vec<ExtWords> {
    if p.is_null() || n == 0 {
        return vec![];
    }
    slice::from_raw_parts(p as *const ExtWords, n).to_vec()
}

pub fn create_chip_proof(hal: &Arc<HipHal>, task: &ChipTask, challenges: &[ExtWords; 2], transcript: &mut impl FsTranscript,
                         stream: Option<&HipStream>) -> Result<ChipProofWords> {
    let handles: Vec<*mut sys::ceno_hip_mle> = task.mles.iter().map(|m| m.map_or(ptr::null_mut(), |m| m.raw())).collect();
    let (toff, tidx) = crate::sumcheck::csr(&task.record_terms);
    let (src, tgt): (Vec<i32>, Vec<i32>) = task.rotation.as_ref().map_or((vec![], vec![]), |r| r.0.iter().map(|&(s, t)| (s as i32, t as i32)).unzip());
    let c_task = sys::ceno_chip_task {
        circuit_idx: task.circuit_idx as i32,
        num_instances: task.num_instances,
        log2_num_instances: task.log2_num_instances as i32,
        rotation_vars: task.rotation_vars as i32,
        n_witin: task.n_witin as i32,
        n_fixed: task.n_fixed as i32,
        n_structural: task.n_structural as i32,
        mles: handles.as_ptr(),
        num_reads: task.num_reads as i32,
        num_writes: task.num_writes as i32,
        num_lk_tables: task.num_lk_tables as i32,
        num_lk: task.num_lk as i32,
        n_record_terms: task.record_terms.len() as i32,
        record_coeffs: task.record_coeffs.as_ptr() as *const u64,
        record_term_offsets: toff.as_ptr(),
        record_term_mle_idx: tidx.as_ptr(),
        record_out_term_offsets: task.record_out_term_offsets.as_ptr(),
        n_rotation_pairs: src.len() as i32,
        rotation_source_idx: src.as_ptr(),
        rotation_target_idx: tgt.as_ptr(),
        cyclic_subgroup_size: task.rotation.as_ref().map_or(0, |r| r.1 as i32),
        cyclic_group_log2: task.rotation.as_ref().map_or(0, |r| r.2 as i32),
    };
    let mut tr = CTranscript::new(transcript);
    let mut out: sys::ceno_chip_proof = unsafe { std::mem::zeroed() };
    hal.check_prover(unsafe { sys::ceno_prover_create_chip_proof(hal.raw(), &c_task, challenges.as_ptr() as *const u64, tr.raw(), raw_stream(stream), &mut out) })?;
    // copy out of the C structure (freed below)
    let nv = out.tower_num_vars as usize;
    let rounds = nv.saturating_sub(1);
    let proof = unsafe {
        let mut tower = TowerProofWords::default();
        let mut off = 0usize;
        for r in 1..=rounds {
            let m = ext_slice(out.tower.msgs.add(2 * off), r * 3);
            tower.proofs.push(m.chunks(3).map(|c| c.to_vec()).collect());
            off += r * 3;
        }
        // zero rows mark rounds in which a shorter spec is no longer active: the reference omits them
        for i in 0..out.n_prod as usize {
            let all = ext_slice(out.tower.prod_evals.add(2 * i * rounds * 2), rounds * 2);
            tower.prod_specs_eval.push(all.chunks(2).filter(|c| c.iter().any(|e| e != &[0, 0])).map(|c| c.to_vec()).collect());
        }
        for i in 0..out.n_logup as usize {
            let all = ext_slice(out.tower.logup_evals.add(2 * i * rounds * 4), rounds * 4);
            tower.logup_specs_eval.push(all.chunks(4).filter(|c| c.iter().any(|e| e != &[0, 0])).map(|c| c.to_vec()).collect());
        }
        let n = out.num_var_with_rotation as usize;
        let rotation = (out.n_rotation_pairs > 0).then(|| {
            let msgs = ext_slice(out.rotation_msgs, n * 2).chunks(2).map(|c| c.to_vec()).collect();
            let evals = ext_slice(out.rotation_evals, 3 * out.n_rotation_pairs as usize);
            let pts = ext_slice(out.rotation_points, 3 * n);
            (msgs, evals, [pts[..n].to_vec(), pts[n..2 * n].to_vec(), pts[2 * n..].to_vec()])
        });
        ChipProofWords {
            r_out_evals: ext_slice(out.r_out_evals.as_ptr(), out.n_r_out as usize),
            w_out_evals: ext_slice(out.w_out_evals.as_ptr(), out.n_w_out as usize),
            lk_out_evals: ext_slice(out.lk_out_evals.as_ptr(), out.n_lk_out as usize),
            tower,
            rt_tower: ext_slice(out.tower.point, nv),
            rt_main: ext_slice(out.rt_main, n),
            rotation,
        }
    };
    unsafe { sys::ceno_chip_proof_free(&mut out) };
    Ok(proof)
}

/// scheduler back end: run chip-proof closures on `n_lanes` streams, largest estimate first, booked against the pool
/// (`ceno_prover_lanes_run`; reference `ChipScheduler::execute`).  Returns the per-task status.
pub fn lanes_run<F>(hal: &Arc<HipHal>, n_lanes: usize, tasks: Vec<(usize, F)>) -> Result<Vec<i32>>
where
    F: FnMut(i32, sys::ceno_hip_stream) -> i32 + Send,
{
    unsafe extern "C" fn tramp<F: FnMut(i32, sys::ceno_hip_stream) -> i32>(arg: *mut c_void, lane: i32, s: sys::ceno_hip_stream) -> i32 {
        (*(arg as *mut F))(lane, s)
    }
    let mut closures: Vec<F> = vec![];
    let mut est = vec![];
    for (e, f) in tasks {
        est.push(e);
        closures.push(f);
    }
    let c_tasks: Vec<sys::ceno_lane_task> = closures
        .iter_mut()
        .zip(&est)
        .map(|(f, &e)| sys::ceno_lane_task { fn_: Some(tramp::<F>), arg: f as *mut F as *mut c_void, estimated_bytes: e })
        .collect();
    let mut status = vec![0i32; c_tasks.len()];
    hal.check_prover(unsafe { sys::ceno_prover_lanes_run(hal.raw(), n_lanes as i32, c_tasks.as_ptr(), c_tasks.len() as i32, status.as_mut_ptr(), ptr::null_mut()) })?;
    Ok(status)
}
