//! `ceno_zkvm::scheme::hip` — the ten `ProverDevice` traits (`scheme/hal.rs:19-35`) for `HipProver<HipBackend<E, PCS>>`
//! on top of the `ceno_hip` HAL; sibling of `scheme/cpu/mod.rs` (semantics) and `scheme/gpu/mod.rs` (shape).  Drop in as
//! `ceno_zkvm/src/scheme/hip/mod.rs` together with `rust/patches/0001-hip-backend.patch`.
//! NOT COMPILED in the image it was written in (no Rust toolchain there); every numeric path it calls is exercised by the
//! C++ mirror of the same control flow (`ceno_amd/host/`, `include/ceno_prover.h`) in the GPU test-suite.
//!
//! Commit / open: the device path produces Merkle roots and a flat proof (`include/ceno_prover.h`); turning them into the
//! `PCS::Commitment` / `PCS::Proof` of the reference's EXT `mpcs` crate is the job of [`HipPcsBridge`], the one place where
//! the PARITY-UNPINNED layout knowledge lives (`tools/goldens/` dumps what is needed to pin it).
use crate::{
    error::ZKVMError,
    scheme::{
        cpu::TowerRelationOutput,
        hal::{
            BatchedMainConstraintProver, BatchedMainConstraintResult, ChipInputPreparer, DeviceProvingKey, DeviceTransporter, EccQuarkProver,
            MainConstraintJob, MainConstraintResult, MainSumcheckEvals, MainSumcheckProver, OpeningProver, ProofInput, ProverDevice, RotationProver,
            RotationProverOutput, TowerProver, TowerProverSpec, TraceCommitter,
        },
        utils::first_layer_selector_contexts,
        MainConstraintProof,
    },
    structs::{ComposedConstrainSystem, EccQuarkProof, TowerProofs, ZKVMProvingKey},
};
use ceno_hip::{pcs::HipPcsData, prover::CTranscript, sumcheck::CommonTermPlan, tower::HipTower, ExtWords, HipMle};
use either::Either;
use ff_ext::ExtensionField;
use gkr_iop::{
    gkr::{layer::sumcheck_layer::SumcheckLayerProof, GKRCircuitWitness, GKRProof, GKRProverOutput, layer::LayerWitness},
    hal::ProverBackend,
    hip::{exts_words, ext_words, flatten_terms, get_hip_hal, get_thread_stream, words_ext, words_exts, HipBackend, HipProver, MultilinearExtensionHip, TranscriptAdapter},
    selector::SelectorType,
};
use itertools::{chain, Itertools};
use mpcs::{Point, PolynomialCommitmentScheme};
use multilinear_extensions::{mle::MultilinearExtension, util::ceil_log2, virtual_poly::eq_eval, Expression};
use std::{collections::BTreeMap, sync::Arc};
use sumcheck::{
    structs::{IOPProof, IOPProverMessage},
    util::{extrapolate_uni_poly, get_challenge_pows},
};
use transcript::Transcript;
use witness::next_pow2_instance_padding;

type PB<E, PCS> = HipBackend<E, PCS>;
type Mle<E> = Arc<MultilinearExtensionHip<'static, E>>;

/// What the device commit / open hand back, to be dressed as the PCS's own types.
pub trait HipPcsBridge<E: ExtensionField>: PolynomialCommitmentScheme<E> {
    const LOG_BLOWUP: usize;
    const NUM_QUERIES: usize;
    const POW_BITS: usize;
    /// THE Merkle root (4 base-field words) of the mixed-height commitment over all matrices, with their (num_vars, width)
    fn commitment_from_root(root: [u64; 4], shapes: &[(usize, usize)]) -> Self::Commitment;
    /// flat proof words, layout documented at `ceno_prover_basefold_proof_words` in `include/ceno_prover.h`
    fn proof_from_words(words: Vec<u64>, shapes: &[(usize, usize)]) -> Self::Proof;
}

fn hal_err(e: ceno_hip::HipError) -> ZKVMError {
    ZKVMError::BackendError(gkr_iop::error::BackendError::CircuitError(e.to_string().into_boxed_str()))
}
fn iop_proof<E: ExtensionField>(msgs: &[Vec<ExtWords>]) -> IOPProof<E> {
    IOPProof { proofs: msgs.iter().map(|m| IOPProverMessage { evaluations: words_exts(m) }).collect() }
}

// ------------------------------------------------------------------------------------------------
// TraceCommitter
// ------------------------------------------------------------------------------------------------
impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> TraceCommitter<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    fn commit_traces<'a>(&self, traces: BTreeMap<usize, witness::RowMajorMatrix<E::BaseField>>) -> (Vec<Mle<E>>, Arc<HipPcsData>, PCS::Commitment) {
        let hal = get_hip_hal().expect("HIP HAL");
        let stream = get_thread_stream().unwrap_or_else(|| Arc::new(hal.create_stream().expect("stream")));
        let max_poly_size_log2 = traces.values().map(|t| ceil_log2(next_pow2_instance_padding(t.num_instances()))).max().unwrap();
        assert!(max_poly_size_log2 <= self.backend.max_poly_size_log2, "max_poly_size_log2 {max_poly_size_log2} > backend {}", self.backend.max_poly_size_log2);
        // row-major words of every trace (canonical); rows are padded to next_pow2_instance_padding on the device
        let words: Vec<Vec<u64>> = traces.values().map(|t| t.values.iter().map(|v| p3::field::PrimeField64::as_canonical_u64(v)).collect()).collect();
        let mats: Vec<(&[u64], usize, usize)> = words.iter().zip(traces.values()).map(|(w, t)| (w.as_slice(), t.num_instances(), t.width())).collect();
        let pcs = Arc::new(HipPcsData::commit(&hal, &mats, PCS::LOG_BLOWUP, &stream).expect("commit_traces"));
        let shapes = (0..mats.len()).map(|m| (pcs.num_vars(m), mats[m].2)).collect_vec();
        let root = pcs.root(&stream).expect("root");
        // witness MLEs = borrowed views of the column-major device trace: nothing is copied or re-uploaded
        let mles = (0..mats.len()).flat_map(|m| (0..mats[m].2).map(move |c| (m, c))).map(|(m, c)| Arc::new(MultilinearExtensionHip::from_hip(pcs.witness_mle(m, c).expect("witness view")))).collect_vec();
        (mles, pcs, PCS::commitment_from_root(root, &shapes))
    }

    fn extract_witness_mles<'a, 'b>(&self, witness_mles: &'b mut Vec<Mle<E>>, _pcs_data: &'b Arc<HipPcsData>) -> Box<dyn Iterator<Item = Mle<E>> + 'b> {
        Box::new(witness_mles.drain(..))
    }
}

// ------------------------------------------------------------------------------------------------
// TowerProver
// ------------------------------------------------------------------------------------------------
struct BuiltTowers {
    prod: Vec<HipTower>,
    logup: Vec<HipTower>,
    out_evals: Vec<Vec<Vec<ExtWords>>>, // [r, w, lk] -> 0 or 1 groups -> evaluations
}

/// `build_tower_witness` (`scheme/cpu/mod.rs:608-757`): record slicing, `group_num_vars`, defaults ONE / challenges[0]
fn build_towers<E: ExtensionField>(cs: &ComposedConstrainSystem<E>, input_num_vars: usize, records: &[Mle<E>], challenges: &[E; 2]) -> Result<BuiltTowers, ZKVMError> {
    let hal = get_hip_hal().map_err(|e| ZKVMError::BackendError(gkr_iop::error::BackendError::CircuitError(e.into_boxed_str())))?;
    let stream = get_thread_stream();
    let c = &cs.zkvm_v1_css;
    let num_reads = c.r_expressions.len() + c.r_table_expressions.len();
    let num_writes = c.w_expressions.len() + c.w_table_expressions.len();
    let n_lk_n = c.lk_table_expressions.len();
    let n_lk_d = if n_lk_n > 0 { n_lk_n } else { c.lk_expressions.len() };
    let rec = |r: std::ops::Range<usize>| records[r].iter().map(|m| m.inner().as_ref()).collect_vec();
    let r_set = rec(0..num_reads);
    let w_set = rec(num_reads..num_reads + num_writes);
    let lk_n = rec(num_reads + num_writes..num_reads + num_writes + n_lk_n);
    let lk_d = rec(num_reads + num_writes + n_lk_n..num_reads + num_writes + n_lk_n + n_lk_d);
    let active_rows = 1usize << input_num_vars;
    let one: ExtWords = [1, 0];
    let alpha = ext_words(&challenges[0]);
    let mut b = BuiltTowers { prod: vec![], logup: vec![], out_evals: vec![vec![], vec![], vec![]] };
    for (slot, set) in [(0, &r_set), (1, &w_set)] {
        if !set.is_empty() {
            let t = HipTower::build_prod(&hal, set, active_rows, one, stream.as_deref()).map_err(hal_err)?;
            b.out_evals[slot].push(t.out_evals(stream.as_deref()).map_err(hal_err)?);
            b.prod.push(t);
        }
    }
    if !lk_d.is_empty() {
        let t = HipTower::build_logup(&hal, (!lk_n.is_empty()).then_some(lk_n.as_slice()), &lk_d, active_rows, alpha, stream.as_deref()).map_err(hal_err)?;
        b.out_evals[2].push(t.out_evals(stream.as_deref()).map_err(hal_err)?);
        b.logup.push(t);
    }
    Ok(b)
}

impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> TowerProver<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    fn build_tower_witness<'a, 'b, 'c>(&self, cs: &ComposedConstrainSystem<E>, input: &ProofInput<'a, PB<E, PCS>>, records: &'c [Mle<E>], challenges: &[E; 2])
        -> (Vec<Vec<Vec<E>>>, Vec<TowerProverSpec<'c, PB<E, PCS>>>, Vec<TowerProverSpec<'c, PB<E, PCS>>>)
    where
        'a: 'b,
        'b: 'c,
    {
        let nv = input.log2_num_instances() + cs.rotation_vars().unwrap_or(0);
        let b = build_towers(cs, nv, records, challenges).expect("build_tower_witness");
        let out = b.out_evals.iter().map(|g| g.iter().map(|e| words_exts::<E>(e)).collect_vec()).collect_vec();
        // the layered witness stays on the device inside the tower handles; `prove_tower_relation` consumes those directly.
        // The spec vectors of the trait are only needed by callers that walk the layers themselves: none in create_proof.
        (out, vec![], vec![])
    }

    fn prove_tower_relation<'a, 'b, 'c>(&self, cs: &ComposedConstrainSystem<E>, input: &ProofInput<'a, PB<E, PCS>>, records: &'c [Mle<E>], challenges: &[E; 2],
                                        transcript: &mut impl Transcript<E>) -> TowerRelationOutput<E>
    where
        'a: 'b,
        'b: 'c,
    {
        let hal = get_hip_hal().expect("HIP HAL");
        let stream = get_thread_stream();
        let nv = input.log2_num_instances() + cs.rotation_vars().unwrap_or(0);
        let b = build_towers(cs, nv, records, challenges).expect("build_tower_witness");
        let mut out = b.out_evals.iter().map(|g| g.iter().map(|e| words_exts::<E>(e)).collect_vec()).collect_vec();
        // bind read / write / lookup out-evals before deriving tower challenges (cpu/mod.rs:783-786)
        for e in out.iter().flat_map(|g| g.iter()).flatten() {
            transcript.append_field_element_ext(e);
        }
        let (rt, words) = ceno_hip::tower::create_proof(&hal, &b.prod.iter().collect_vec(), &b.logup.iter().collect_vec(), &mut TranscriptAdapter::new(transcript),
                                                       stream.as_deref()).expect("tower proof");
        let proofs = TowerProofs {
            proofs: words.proofs.iter().map(|layer| layer.iter().map(|m| IOPProverMessage { evaluations: words_exts(m) }).collect()).collect(),
            prod_specs_eval: words.prod_specs_eval.iter().map(|s| s.iter().map(|r| words_exts(r)).collect()).collect(),
            prod_specs_points: vec![],
            logup_specs_eval: words.logup_specs_eval.iter().map(|s| s.iter().map(|r| words_exts(r)).collect()).collect(),
            logup_specs_points: vec![],
        };
        let lk = out.pop().unwrap();
        let w = out.pop().unwrap();
        let r = out.pop().unwrap();
        (words_exts(&rt), proofs, lk, w, r)
    }
}

// ------------------------------------------------------------------------------------------------
// MainSumcheckProver: the per-chip GKR route (not reached from create_proof today, prover.rs:811-812)
// ------------------------------------------------------------------------------------------------
impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> MainSumcheckProver<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    #[allow(clippy::type_complexity)]
    fn prove_main_constraints<'a, 'b>(&self, rt_tower: Vec<E>, rotation: Option<RotationProverOutput<E>>, ecc_proof: Option<&EccQuarkProof<E>>,
                                      input: &'b ProofInput<'a, PB<E, PCS>>, cs: &ComposedConstrainSystem<E>, challenges: &[E; 2], transcript: &mut impl Transcript<E>)
        -> Result<(Point<E>, MainSumcheckEvals<E>, Option<Vec<IOPProverMessage<E>>>, Option<GKRProof<E>>), ZKVMError> {
        // the out-evaluation assembly (tower / rotation / ecc groups) is backend independent: shared with the CPU arm
        let (gkr_circuit, out_evals, selector_ctxs, num_var_with_rotation) =
            crate::scheme::cpu::assemble_main_out_evals(cs, input.num_instances, input.log2_num_instances(), &rt_tower, rotation.as_ref(), ecc_proof, transcript);
        let GKRProverOutput { gkr_proof, opening_evaluations, mut rt } = gkr_circuit.prove::<PB<E, PCS>, HipProver<_>>(
            1,
            num_var_with_rotation,
            GKRCircuitWitness { layers: vec![LayerWitness(chain!(&input.witness, &input.fixed, &input.structural_witness).cloned().collect_vec())] },
            &out_evals,
            &input.pi.iter().map(|v| v.map_either(E::from, |v| v).into_inner()).collect_vec(),
            challenges,
            transcript,
            &selector_ctxs,
        )?;
        assert_eq!(rt.len(), 1, "TODO support multi-layer gkr iop");
        let c = &cs.zkvm_v1_css;
        Ok((
            rt.remove(0),
            MainSumcheckEvals {
                wits_in_evals: opening_evaluations.iter().take(c.num_witin as usize).map(|e| e.value).collect(),
                fixed_in_evals: opening_evaluations.iter().skip(c.num_witin as usize).take(c.num_fixed).map(|e| e.value).collect(),
            },
            None,
            Some(gkr_proof),
        ))
    }
}

// ------------------------------------------------------------------------------------------------
// BatchedMainConstraintProver: ONE sumcheck over the first layer of every chip (scheme/cpu/mod.rs:1052-1390)
// ------------------------------------------------------------------------------------------------
impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> BatchedMainConstraintProver<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    fn prove_batched_main_constraints<'a>(&self, jobs: Vec<MainConstraintJob<'a, PB<E, PCS>>>, _pcs_data: &Arc<HipPcsData>, transcript: &mut impl Transcript<E>)
        -> BatchedMainConstraintResult<E> {
        if jobs.is_empty() {
            return Ok((MainConstraintProof { claimed_sum: E::ZERO, proof: SumcheckLayerProof { proof: IOPProof { proofs: vec![] }, evals: vec![] } }, vec![]));
        }
        let hal = get_hip_hal().expect("HIP HAL");
        let stream = get_thread_stream();
        struct Chip<'a, E: ExtensionField> {
            layer: &'a gkr_iop::gkr::layer::Layer<E>,
            nv: usize,
            mle_start: usize,
            num_mles: usize,
            alpha_start: usize,
            pi: Vec<Either<E::BaseField, E>>,
        }
        let mut chips = vec![];
        let mut owned_selectors: Vec<Vec<Option<MultilinearExtensionHip<'static, E>>>> = vec![];
        let (mut total_exprs, mut max_nv, mut max_degree) = (0usize, 0usize, 0usize);
        for job in &jobs {
            let gkr_circuit = job.cs.gkr_circuit.as_ref().expect("empty gkr circuit");
            let layer = gkr_circuit.layers.first().expect("empty gkr circuit layer");
            let nv = job.input.log2_num_instances() + job.cs.rotation_vars().unwrap_or(0);
            max_nv = max_nv.max(nv);
            max_degree = max_degree.max(layer.max_expr_degree + 1);
            // selector eq tables at the chip's tower point, first selector per structural witness id wins (cpu/mod.rs:1200-1234)
            let ctxs = first_layer_selector_contexts(job.cs, gkr_circuit, job.input.num_instances, nv);
            let mut by_id: Vec<Option<MultilinearExtensionHip<'static, E>>> = vec![None; layer.n_structural_witin];
            for ((sel_type, _), ctx) in layer.out_sel_and_eval_exprs.iter().zip(&ctxs) {
                let expr = match sel_type {
                    SelectorType::Whole(e) | SelectorType::Prefix(e) | SelectorType::OrderedSparse { expression: e, .. } | SelectorType::QuarkBinaryTreeLessThan(e) => e,
                    SelectorType::None => continue,
                };
                let Expression::StructuralWitIn(id, _) = expr else { panic!("selector expression must be StructuralWitIn") };
                if by_id[*id as usize].is_none() {
                    by_id[*id as usize] = Some(gkr_iop::gkr::layer::hip::build_eq_x_r_with_sel_hip(&job.rt_tower, ctx, sel_type));
                }
            }
            owned_selectors.push(by_id);
            chips.push(Chip { layer, nv, mle_start: 0, num_mles: layer.n_witin + layer.n_fixed + layer.n_structural_witin, alpha_start: total_exprs,
                              pi: job.input.pi.clone() });
            total_exprs += layer.exprs.len();
        }
        let alpha_pows = get_challenge_pows(total_exprs, transcript);
        // global MLE list (witness ++ fixed ++ structural with selectors substituted) and monomial terms with evaluated scalars
        let mut mles: Vec<&HipMle> = vec![];
        let mut mle_nv = vec![];
        for ((job, chip), sels) in jobs.iter().zip(chips.iter_mut()).zip(&owned_selectors) {
            chip.mle_start = mles.len();
            mles.extend(job.input.witness.iter().map(|m| m.inner().as_ref()));
            mles.extend(job.input.fixed.iter().map(|m| m.inner().as_ref()));
            for (sel, m) in sels.iter().zip(job.input.structural_witness.iter()) {
                mles.push(sel.as_ref().map_or_else(|| m.inner().as_ref(), |s| s.inner().as_ref()));
            }
            mle_nv.extend(std::iter::repeat(chip.nv).take(mles.len() - chip.mle_start));
            assert_eq!(mles.len() - chip.mle_start, chip.num_mles);
        }
        let (mut coeffs, mut terms): (Vec<E>, Vec<Vec<usize>>) = (vec![], vec![]);
        for chip in &chips {
            let ch = chain!(jobs[0].challenges.iter().copied(), alpha_pows[chip.alpha_start..chip.alpha_start + chip.layer.exprs.len()].iter().copied()).collect_vec();
            for t in chip.layer.main_sumcheck_expression_monomial_terms.as_ref().unwrap() {
                let scalar = gkr_iop::hip::term_scalar_with_instance(&t.scalar, &chip.pi, &ch);
                if scalar == E::ZERO {
                    continue;  // cpu/mod.rs:1308-1314
                }
                coeffs.push(scalar);
                terms.push(t.product.iter().map(|e| match e {
                    Expression::WitIn(id) => chip.mle_start + *id as usize,
                    _ => panic!("main monomial product must be converted to WitIn"),
                }).collect());
            }
        }
        // common-factor plan: terms grouped by their extension-field factors (the selectors), the role of CommonTermPlan in the
        // CUDA arm (scheme/gpu/mod.rs:2811-2962).  Any factoring yields the same messages.
        let mut groups: BTreeMap<Vec<usize>, Vec<usize>> = BTreeMap::new();
        let mut residual = vec![vec![]; terms.len()];
        for (t, fac) in terms.iter().enumerate() {
            let (ext, base): (Vec<usize>, Vec<usize>) = fac.iter().partition(|&&j| mles[j].is_ext());
            if ext.is_empty() || base.is_empty() {
                residual[t] = fac.clone();
            } else {
                residual[t] = base;
                groups.entry(ext.into_iter().sorted().collect()).or_default().push(t);
            }
        }
        let plan = CommonTermPlan { group_terms: groups.values().cloned().collect(), group_common_mles: groups.keys().cloned().collect() };
        let (msgs, evals, point) = ceno_hip::sumcheck::prove(&hal, &mles, &exts_words(&coeffs), &residual, max_nv, max_degree, Some(&plan),
                                                              &mut TranscriptAdapter::new(transcript), stream.as_deref()).map_err(hal_err)?;
        let proof: IOPProof<E> = iop_proof(&msgs);
        let global_rt: Vec<E> = words_exts(&point);
        let global_evals: Vec<E> = words_exts(&evals);
        // final claim by the front-load rule (scheme/verifier.rs:180-238), claimed sum recovered backwards (cpu/mod.rs:1393-1413)
        let final_claim = coeffs.iter().zip(&terms).map(|(c, fac)| {
            fac.iter().fold(*c, |acc, &j| acc * global_evals[j] * global_rt[mle_nv[j]..].iter().copied().product::<E>())
        }).sum::<E>();
        let claimed_sum = proof.proofs.iter().zip(&global_rt).rev().fold(final_claim, |expected, (m, r)| {
            let hidden = extrapolate_uni_poly(E::ONE, &vec![E::ZERO; m.evaluations.len()], *r);
            let without = extrapolate_uni_poly(-m.evaluations[0], &m.evaluations, *r);
            (expected - without) * hidden.inverse()
        });
        transcript.append_field_element_exts(&global_evals);
        let results = jobs.iter().zip(&chips).map(|(job, chip)| {
            let l = chip.layer;
            let ev = &global_evals[chip.mle_start..chip.mle_start + chip.num_mles];
            MainConstraintResult {
                circuit_idx: job.circuit_idx,
                input_opening_point: global_rt[..chip.nv].to_vec(),
                opening_evals: MainSumcheckEvals { wits_in_evals: ev[..l.n_witin].to_vec(), fixed_in_evals: ev[l.n_witin..l.n_witin + l.n_fixed].to_vec() },
            }
        }).collect();
        let _ = eq_eval::<E>;
        Ok((MainConstraintProof { claimed_sum, proof: SumcheckLayerProof { proof, evals: global_evals } }, results))
    }
}

// ------------------------------------------------------------------------------------------------
// rotation / ecc / opening / transport / input preparation
// ------------------------------------------------------------------------------------------------
impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> RotationProver<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    fn prove_rotation<'a>(&self, cs: &ComposedConstrainSystem<E>, input: &ProofInput<'a, PB<E, PCS>>, rt_tower: &Point<E>, challenges: &[E; 2],
                          transcript: &mut impl Transcript<E>) -> Result<Option<RotationProverOutput<E>>, ZKVMError> {
        let Some(gkr_circuit) = cs.gkr_circuit.as_ref() else { return Ok(None) };
        let layer = gkr_circuit.layers.first().expect("empty gkr circuit layer");
        if layer.rotation_exprs.1.is_empty() {
            return Ok(None);
        }
        let Some([subgroup_size, group_log2]) = layer.rotation_cyclic_params() else { return Ok(None) };
        let wit = LayerWitness(chain!(&input.witness, &input.fixed, &input.structural_witness).cloned().collect_vec());
        let (proof, points) = gkr_iop::gkr::layer::hip::prove_rotation_hip::<E, PCS>(rt_tower.len(), subgroup_size, group_log2, &wit, &layer.rotation_exprs.1,
                                                                                    layer.rotation_sumcheck_expression_monomial_terms.clone().unwrap(), rt_tower, challenges, transcript);
        Ok(Some(RotationProverOutput { proof, left_point: points.left, right_point: points.right, point: points.origin }))
    }
}

impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> EccQuarkProver<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    fn prove_ec_sum_quark<'a>(&self, _cs: &ComposedConstrainSystem<E>, input: &ProofInput<'a, PB<E, PCS>>, _transcript: &mut impl Transcript<E>)
        -> Result<Option<EccQuarkProof<E>>, ZKVMError> {
        // the shard-RAM ECC accumulation lives over the septic extension of BabyBear (reference F4/F5: Goldilocks e2e is
        // disabled for that circuit); the Goldilocks HIP arm proves chips without ECC ops only
        assert!(!input.has_ecc_ops, "ECC quark proofs are BabyBear-only in the reference");
        Ok(None)
    }
}

impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> OpeningProver<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    fn open(&self, witness_data: Arc<HipPcsData>, fixed_data: Option<Arc<Arc<HipPcsData>>>, points: Vec<Point<E>>, mut evals: Vec<Vec<Vec<E>>>,
            transcript: &mut (impl Transcript<E> + 'static)) -> PCS::Proof {
        let hal = get_hip_hal().expect("HIP HAL");
        let stream = get_thread_stream().unwrap_or_else(|| Arc::new(hal.create_stream().expect("stream")));
        // rounds = [(witness commitment, openings), (fixed commitment, openings)] (cpu/mod.rs:1426-1455): one (point, column
        // evaluations) per committed matrix, witness matrices first, then the fixed ones of the chips that have fixed columns
        let (mut pts, mut evs): (Vec<Vec<ExtWords>>, Vec<Vec<ExtWords>>) = evals.iter_mut().zip(&points)
            .filter_map(|(e, p)| { let w = e.remove(0); (!w.is_empty()).then(|| (exts_words(p), exts_words(&w))) }).unzip();
        let mut shapes = (0..pts.len()).map(|m| (witness_data.num_vars(m), witness_data.widths[m])).collect_vec();
        let fixed = fixed_data.as_ref().map(|f| f.as_ref().as_ref());
        if let Some(fd) = fixed {
            let (fp, fe): (Vec<Vec<ExtWords>>, Vec<Vec<ExtWords>>) = evals.iter_mut().zip(&points)
                .filter_map(|(e, p)| (!e.is_empty() && !e[0].is_empty()).then(|| (exts_words(p), exts_words(&e.remove(0))))).unzip();
            shapes.extend((0..fp.len()).map(|m| (fd.num_vars(m), fd.widths[m])));
            pts.extend(fp);
            evs.extend(fe);
        }
        let mut adapter = TranscriptAdapter::new(transcript);
        let mut ctr = CTranscript::new(&mut adapter);
        let words = witness_data.batch_open(fixed, &pts, &evs, PCS::NUM_QUERIES, PCS::POW_BITS, ctr.raw(), &stream).expect("batch_open");
        PCS::proof_from_words(words, &shapes)
    }
}

impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> DeviceTransporter<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    fn transport_proving_key(&self, _is_first_shard: bool, pk: Arc<ZKVMProvingKey<E, PCS>>) -> DeviceProvingKey<'static, PB<E, PCS>> {
        // fixed traces are committed on the device like witness traces; their columns become the fixed MLEs
        let (fixed_mles, pcs_data, _commit) = self.commit_traces(pk.fixed_traces().clone());
        DeviceProvingKey { fixed_mles, pcs_data: Arc::new(pcs_data) }
    }
    fn transport_mles<'a>(&self, mles: Vec<MultilinearExtension<'a, E>>) -> Vec<Mle<E>> {
        let hal = get_hip_hal().expect("HIP HAL");
        mles.iter().map(|m| Arc::new(MultilinearExtensionHip::from_ceno(&hal, m))).collect()
    }
}

impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> ChipInputPreparer<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    fn prepare_chip_input(&self, task: &mut crate::scheme::scheduler::ChipTask<'_, PB<E, PCS>>, _pcs_data: &Arc<HipPcsData>) {
        // witness columns are views of the committed device trace handed out by commit_traces; structural witnesses are uploaded
        if task.input.structural_witness.is_empty() {
            if let Some(rmm) = task.structural_rmm.as_ref() {
                task.input.structural_witness = self.transport_mles(rmm.to_mles());
            }
        }
    }
}

impl<E: ExtensionField, PCS: HipPcsBridge<E> + 'static> ProverDevice<PB<E, PCS>> for HipProver<PB<E, PCS>> {
    fn get_pb(&self) -> &PB<E, PCS> {
        self.backend.as_ref()
    }
}
