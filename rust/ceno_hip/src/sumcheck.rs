//! Generic sumcheck with the transcript on the Rust side.
//! Reference operator: `cuda_hal.prove_generic_sumcheck_gpu(mles, mle_size_info, term_coefficients, mle_indices_per_term,
//! max_num_var, max_degree, Option<&CommonTermPlan>, &mut BasicTranscript, stream)`
//! (`gkr_iop/src/gkr/layer/gpu/mod.rs:259-271`, `ceno_zkvm/src/scheme/gpu/mod.rs:891-902,2968-2982`) = EXT
//! `IOPProverState::prove`.  A C ABI cannot take the Rust generic, so control is inverted: `round(challenge_{i-1}) -> message_i`.
use std::{ptr, sync::Arc};

use ceno_hip_sys as sys;

use crate::{
    error::Result,
    hal::{raw_stream, HipHal, HipStream},
    mle::HipMle,
    ExtWords, FsTranscript,
};

/// CSR of a list of index lists
pub fn csr(lists: &[Vec<usize>]) -> (Vec<u32>, Vec<u32>) {
    let mut off = vec![0u32];
    let mut idx = vec![];
    for l in lists {
        idx.extend(l.iter().map(|&x| x as u32));
        off.push(idx.len() as u32);
    }
    if idx.is_empty() {
        idx.push(0);
    }
    (off, idx)
}

/// `CommonTermPlan` (`gkr_iop/src/gkr/layer/gpu/utils.rs:69-119`): group g multiplies the sum of its terms by its common MLEs
#[derive(Default, Clone)]
pub struct CommonTermPlan {
    pub group_terms: Vec<Vec<usize>>,
    pub group_common_mles: Vec<Vec<usize>>,
}

/// A table of the plan that IS `eq(., point)` on the rows `[lo, hi)` and zero elsewhere — a `SelectorType::Whole` / `Prefix` selector
/// (`gkr_iop/src/selector.rs:131-245`).  Declared to `Sumcheck::begin_eq`, the rounds of the chips whose every group hangs on one such
/// table evaluate the quotient of the round polynomial by `eq(X, rt_i)` at one point fewer; the messages are the same words.
pub struct EqDeclaration<'a> {
    pub mle_index: usize,
    pub point: &'a [ExtWords],
    pub lo: usize,
    pub hi: usize,
}

/// (round messages `n x d`, final evaluations per MLE, challenges = opening point)
pub type SumcheckOutput = (Vec<Vec<ExtWords>>, Vec<ExtWords>, Vec<ExtWords>);

/// An in-flight sumcheck (`ceno_hip_sumcheck`), freed on drop.
pub struct Sumcheck {
    hal: Arc<HipHal>,
    raw: *mut sys::ceno_hip_sumcheck,
    pub num_vars: usize,
    pub degree: usize,
    pub num_mles: usize,
}
impl Sumcheck {
    #[allow(clippy::too_many_arguments)]
    pub fn begin(hal: &Arc<HipHal>, mles: &[&HipMle], term_coefficients: &[ExtWords], mle_indices_per_term: &[Vec<usize>], max_num_var: usize,
                 max_degree: usize, plan: Option<&CommonTermPlan>, stream: Option<&HipStream>) -> Result<Self> {
        let handles: Vec<*mut sys::ceno_hip_mle> = mles.iter().map(|m| m.raw()).collect();
        let (toff, tidx) = csr(mle_indices_per_term);
        let (goff, gidx) = plan.map_or((vec![0u32], vec![0u32]), |p| csr(&p.group_terms));
        let (coff, cidx) = plan.map_or((vec![0u32], vec![0u32]), |p| csr(&p.group_common_mles));
        let c_plan = sys::ceno_hip_sumcheck_plan {
            num_mles: handles.len() as i32,
            num_terms: mle_indices_per_term.len() as i32,
            term_coeffs: term_coefficients.as_ptr() as *const u64,
            term_offsets: toff.as_ptr(),
            term_mle_idx: tidx.as_ptr(),
            num_groups: plan.map_or(0, |p| p.group_terms.len() as i32),
            group_term_offsets: goff.as_ptr(),
            group_term_idx: gidx.as_ptr(),
            common_offsets: coff.as_ptr(),
            common_mle_idx: cidx.as_ptr(),
            max_num_vars: max_num_var as i32,
            max_degree: max_degree as i32,
        };
        let mut sc = ptr::null_mut();
        hal.check(unsafe { sys::ceno_hip_sumcheck_begin(hal.ctx, handles.as_ptr(), &c_plan, raw_stream(stream), &mut sc) })?;
        Ok(Self { hal: hal.clone(), raw: sc, num_vars: max_num_var, degree: max_degree, num_mles: handles.len() })
    }
    /// `begin` with selector declarations (`ceno_hip_sumcheck_begin_eq`): what `prove_batched_main_constraints` knows about its selectors
    #[allow(clippy::too_many_arguments)]
    pub fn begin_eq(hal: &Arc<HipHal>, mles: &[&HipMle], term_coefficients: &[ExtWords], mle_indices_per_term: &[Vec<usize>], max_num_var: usize,
                    max_degree: usize, plan: Option<&CommonTermPlan>, eq: &[EqDeclaration<'_>], stream: Option<&HipStream>) -> Result<Self> {
        let handles: Vec<*mut sys::ceno_hip_mle> = mles.iter().map(|m| m.raw()).collect();
        let (toff, tidx) = csr(mle_indices_per_term);
        let (goff, gidx) = plan.map_or((vec![0u32], vec![0u32]), |p| csr(&p.group_terms));
        let (coff, cidx) = plan.map_or((vec![0u32], vec![0u32]), |p| csr(&p.group_common_mles));
        let c_plan = sys::ceno_hip_sumcheck_plan {
            num_mles: handles.len() as i32,
            num_terms: mle_indices_per_term.len() as i32,
            term_coeffs: term_coefficients.as_ptr() as *const u64,
            term_offsets: toff.as_ptr(),
            term_mle_idx: tidx.as_ptr(),
            num_groups: plan.map_or(0, |p| p.group_terms.len() as i32),
            group_term_offsets: goff.as_ptr(),
            group_term_idx: gidx.as_ptr(),
            common_offsets: coff.as_ptr(),
            common_mle_idx: cidx.as_ptr(),
            max_num_vars: max_num_var as i32,
            max_degree: max_degree as i32,
        };
        let idx: Vec<i32> = eq.iter().map(|d| d.mle_index as i32).collect();
        let pts: Vec<*const u64> = eq.iter().map(|d| d.point.as_ptr() as *const u64).collect();
        let lo: Vec<usize> = eq.iter().map(|d| d.lo).collect();
        let hi: Vec<usize> = eq.iter().map(|d| d.hi).collect();
        let mut sc = ptr::null_mut();
        hal.check(unsafe {
            sys::ceno_hip_sumcheck_begin_eq(hal.ctx, handles.as_ptr(), &c_plan, eq.len() as i32, idx.as_ptr(), pts.as_ptr(), lo.as_ptr(), hi.as_ptr(),
                                            raw_stream(stream), &mut sc)
        })?;
        Ok(Self { hal: hal.clone(), raw: sc, num_vars: max_num_var, degree: max_degree, num_mles: handles.len() })
    }
    /// how many chips of the plan run in the eq-factored form (diagnostics)
    pub fn eq_components(&self) -> usize {
        unsafe { sys::ceno_hip_sumcheck_eq_components(self.raw) as usize }
    }
    /// adopt a handle made by another entry point (tower layers)
    pub(crate) fn from_raw(hal: &Arc<HipHal>, raw: *mut sys::ceno_hip_sumcheck, num_vars: usize, degree: usize, num_mles: usize) -> Self {
        Self { hal: hal.clone(), raw, num_vars, degree, num_mles }
    }
    /// promise to drive the rounds back to back: the round kernels are enqueued ahead and pick their challenges up from a mailbox
    pub fn set_pipelined(&mut self, on: bool) -> Result<()> {
        self.hal.check(unsafe { sys::ceno_hip_sumcheck_set_pipelined(self.hal.ctx, self.raw, on as i32) })
    }
    /// optional, before round 0: the sum this sumcheck proves.  Only the fused tower-layer rounds use it (two evaluation points in round 0
    /// instead of three); a tower prover has it from the layer before (`TowerVerify`'s expected evaluation, `scheme/verifier.rs:1587-1680`)
    pub fn set_claim(&mut self, claim: ExtWords) -> Result<()> {
        self.hal.check(unsafe { sys::ceno_hip_sumcheck_set_claim(self.hal.ctx, self.raw, claim.as_ptr()) })
    }
    /// how many leading rounds run on the fused tower-layer kernel (0 for every other handle): diagnostics
    pub fn fused_eq_rounds(&self) -> usize {
        unsafe { sys::ceno_hip_sumcheck_fused_eq_rounds(self.raw) as usize }
    }
    /// message of the next round: p(1..d); `challenge` = the previous round's (None for round 0)
    pub fn round(&mut self, challenge: Option<ExtWords>) -> Result<Vec<ExtWords>> {
        let mut out = vec![[0u64; 2]; self.degree];
        let ch = challenge.as_ref().map_or(ptr::null(), |c| c.as_ptr());
        self.hal.check(unsafe { sys::ceno_hip_sumcheck_round(self.hal.ctx, self.raw, ch, out.as_mut_ptr() as *mut u64) })?;
        Ok(out)
    }
    /// bind the last variable: `get_mle_flatten_final_evaluations`
    pub fn finish(&mut self, last_challenge: Option<ExtWords>) -> Result<Vec<ExtWords>> {
        let mut out = vec![[0u64; 2]; self.num_mles];
        let ch = last_challenge.as_ref().map_or(ptr::null(), |c| c.as_ptr());
        self.hal.check(unsafe { sys::ceno_hip_sumcheck_finish(self.hal.ctx, self.raw, ch, out.as_mut_ptr() as *mut u64) })?;
        Ok(out)
    }
    /// `IOPProverState::prove`: prologue `n`, `d` as `usize` LE bytes; per round the `d` evaluations, then the challenge under
    /// the label `b"Internal round"` (`ceno_recursion_v2/src/main/mod.rs:3503-3529`)
    pub fn run(mut self, transcript: &mut impl FsTranscript) -> Result<SumcheckOutput> {
        self.set_pipelined(true)?;
        transcript.append_bytes(&self.num_vars.to_le_bytes());
        transcript.append_bytes(&self.degree.to_le_bytes());
        let (mut msgs, mut point) = (Vec::with_capacity(self.num_vars), Vec::with_capacity(self.num_vars));
        let mut challenge = None;
        for _ in 0..self.num_vars {
            let msg = self.round(challenge)?;
            for e in &msg {
                transcript.append_ext(*e);
            }
            let r = transcript.challenge(b"Internal round");
            challenge = Some(r);
            point.push(r);
            msgs.push(msg);
        }
        let evals = self.finish(challenge)?;
        Ok((msgs, evals, point))
    }
}
impl Drop for Sumcheck {
    fn drop(&mut self) {
        unsafe { sys::ceno_hip_sumcheck_free(self.hal.ctx, self.raw) };
    }
}

/// the reference call in one piece
#[allow(clippy::too_many_arguments)]
pub fn prove(hal: &Arc<HipHal>, mles: &[&HipMle], term_coefficients: &[ExtWords], mle_indices_per_term: &[Vec<usize>], max_num_var: usize,
             max_degree: usize, plan: Option<&CommonTermPlan>, transcript: &mut impl FsTranscript, stream: Option<&HipStream>)
             -> Result<SumcheckOutput> {
    Sumcheck::begin(hal, mles, term_coefficients, mle_indices_per_term, max_num_var, max_degree, plan, stream)?.run(transcript)
}

/// `estimate_sumcheck_memory` (`ceno_zkvm/src/scheme/gpu/memory.rs:413-433`)
pub fn estimate_memory(max_num_vars: usize, max_degree: usize, mle_num_vars: &[usize], num_terms: usize) -> usize {
    let nv: Vec<i32> = mle_num_vars.iter().map(|&v| v as i32).collect();
    unsafe { sys::ceno_hip_sumcheck_estimate_memory(max_num_vars as i32, max_degree as i32, nv.as_ptr(), nv.len() as i32, num_terms as i32) }
}
