//! Tower witness and tower proof.
//! Reference: `build_prod_tower_from_virtual_ext_batch` / `build_logup_tower_from_virtual_ext_batch`, `GpuProverSpec`
//! (`ceno_zkvm/src/scheme/gpu/mod.rs:2365-2402,379-410`), `cuda_hal.tower.create_proof` (`:343-353`); CPU semantics
//! `ceno_zkvm/src/scheme/utils.rs:402-659`, `scheme/cpu/mod.rs:346-554`.
use std::{ptr, sync::Arc};

use ceno_hip_sys as sys;

use crate::{
    error::Result,
    hal::{raw_stream, HipHal, HipStream},
    mle::HipMle,
    sumcheck::Sumcheck,
    ExtWords, FsTranscript,
};

/// built tower witness (`TowerProverSpec` / `GpuProverSpec`): layer l = 2 (product) or 4 (LogUp: p1, p2, q1, q2) limbs of 2^l
pub struct HipTower {
    hal: Arc<HipHal>,
    raw: *mut sys::ceno_hip_tower,
}
unsafe impl Send for HipTower {}
unsafe impl Sync for HipTower {}
impl HipTower {
    pub fn raw(&self) -> *mut sys::ceno_hip_tower {
        self.raw
    }
    /// interleave `records` over `num_instances` rows with `default` padding and build all product layers
    pub fn build_prod(hal: &Arc<HipHal>, records: &[&HipMle], num_instances: usize, default: ExtWords, stream: Option<&HipStream>) -> Result<Self> {
        let h: Vec<_> = records.iter().map(|m| m.raw()).collect();
        let mut t = ptr::null_mut();
        hal.check(unsafe { sys::ceno_hip_tower_build_prod(hal.ctx, h.as_ptr(), h.len() as i32, num_instances, default.as_ptr(), raw_stream(stream), &mut t) })?;
        Ok(Self { hal: hal.clone(), raw: t })
    }
    /// LogUp tower over denominators `q` and optional numerators `p` (None: all ones)
    pub fn build_logup(hal: &Arc<HipHal>, p: Option<&[&HipMle]>, q: &[&HipMle], num_instances: usize, default: ExtWords,
                       stream: Option<&HipStream>) -> Result<Self> {
        let qh: Vec<_> = q.iter().map(|m| m.raw()).collect();
        let ph: Option<Vec<_>> = p.map(|p| p.iter().map(|m| m.raw()).collect());
        let mut t = ptr::null_mut();
        hal.check(unsafe {
            sys::ceno_hip_tower_build_logup(hal.ctx, ph.as_ref().map_or(ptr::null(), |v| v.as_ptr()), qh.as_ptr(), qh.len() as i32, num_instances,
                                            default.as_ptr(), raw_stream(stream), &mut t)
        })?;
        Ok(Self { hal: hal.clone(), raw: t })
    }
    pub fn num_vars(&self) -> usize {
        unsafe { sys::ceno_hip_tower_num_vars(self.raw) as usize }
    }
    pub fn num_limbs(&self) -> usize {
        unsafe { sys::ceno_hip_tower_num_limbs(self.raw) as usize }
    }
    /// `GpuProverSpec::get_output_evals`: layer 0 of every limb
    pub fn out_evals(&self, stream: Option<&HipStream>) -> Result<Vec<ExtWords>> {
        let mut out = vec![[0u64; 2]; self.num_limbs()];
        self.hal.check(unsafe { sys::ceno_hip_tower_out_evals(self.hal.ctx, self.raw, out.as_mut_ptr() as *mut u64, raw_stream(stream)) })?;
        Ok(out)
    }
}
impl Drop for HipTower {
    fn drop(&mut self) {
        unsafe { sys::ceno_hip_tower_free(self.hal.ctx, self.raw) };
    }
}

/// `TowerProofs` (`ceno_zkvm/src/structs.rs:87-101`) in boundary words; inactive rounds of a spec are omitted like there
#[derive(Default, Clone)]
pub struct TowerProofWords {
    pub proofs: Vec<Vec<Vec<ExtWords>>>,          // tower round -> sumcheck round -> 3 evaluations
    pub prod_specs_eval: Vec<Vec<Vec<ExtWords>>>,  // spec -> active round -> [a, b]
    pub logup_specs_eval: Vec<Vec<Vec<ExtWords>>>, // spec -> active round -> [p1, p2, q1, q2]
}

/// `CpuTowerProver::create_proof(prod_specs, logup_specs, NUM_FANIN = 2, transcript)` (`scheme/cpu/mod.rs:346-554`):
/// alpha powers, `rt <- sample(b"product_sum")`, then per layer one degree-3 sumcheck over
/// `eq * [sum alpha_i a_i b_i + sum (alpha (p1 q2 + p2 q1) + alpha' q1 q2)]`, the per-spec evaluations into the transcript,
/// `r_merge <- sample(b"merge")`, new alpha powers.  Returns the final point and the proof.
pub fn create_proof(hal: &Arc<HipHal>, prod: &[&HipTower], logup: &[&HipTower], transcript: &mut impl FsTranscript,
                    stream: Option<&HipStream>) -> Result<(Vec<ExtWords>, TowerProofWords)> {
    let max_nv = prod.iter().chain(logup.iter()).map(|t| t.num_vars()).max().expect("tower: no specs");
    let n_alpha = prod.len() + 2 * logup.len();
    let ph: Vec<_> = prod.iter().map(|t| t.raw()).collect();
    let lh: Vec<_> = logup.iter().map(|t| t.raw()).collect();
    let mut alpha = transcript.challenge_pows(n_alpha);
    let mut out_rt = vec![transcript.challenge(b"product_sum")];
    let mut proof = TowerProofWords { proofs: vec![], prod_specs_eval: vec![vec![]; prod.len()], logup_specs_eval: vec![vec![]; logup.len()] };
    for round in 1..max_nv {
        let mut sc = ptr::null_mut();
        hal.check(unsafe {
            sys::ceno_hip_tower_layer_sumcheck_begin(hal.ctx, ph.as_ptr(), ph.len() as i32, lh.as_ptr(), lh.len() as i32, round as i32,
                                                     out_rt.as_ptr() as *const u64, alpha.as_ptr() as *const u64, raw_stream(stream), &mut sc)
        })?;
        // MLE order of the handle: [eq, active prod (a, b).., active logup (p1, p2, q1, q2)..]
        // (this crate moves boundary words and has no field arithmetic: the zkvm arm, which has E, states each layer's claim with
        // Sumcheck::set_claim before running it — INTEGRATION.md, TowerProver row; without it the proof is the same, round 0 a third dearer)
        let n_mles = 1 + 2 * prod.iter().filter(|t| t.num_vars() > round).count() + 4 * logup.iter().filter(|t| t.num_vars() > round).count();
        let (msgs, evals, point) = Sumcheck::from_raw(hal, sc, round, 3, n_mles).run(transcript)?;
        proof.proofs.push(msgs);
        let mut cursor = 1;
        for (i, t) in prod.iter().enumerate() {
            if t.num_vars() > round {
                let e = evals[cursor..cursor + 2].to_vec();
                e.iter().for_each(|x| transcript.append_ext(*x));
                proof.prod_specs_eval[i].push(e);
                cursor += 2;
            }
        }
        for (i, t) in logup.iter().enumerate() {
            if t.num_vars() > round {
                let e = evals[cursor..cursor + 4].to_vec();
                e.iter().for_each(|x| transcript.append_ext(*x));
                proof.logup_specs_eval[i].push(e);
                cursor += 4;
            }
        }
        let r_merge = transcript.challenge(b"merge");
        out_rt = point;
        out_rt.push(r_merge);
        alpha = transcript.challenge_pows(n_alpha);
    }
    Ok((out_rt, proof))
}
