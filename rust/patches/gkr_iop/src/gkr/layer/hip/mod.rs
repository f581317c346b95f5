//! Layer provers of the HIP arm: `gkr_iop/src/gkr/layer/hip/mod.rs` (sibling of `layer/gpu/mod.rs`).
//!   `ZerocheckLayerProver::prove`  — selectors on the device, one generic sumcheck with the `CommonTermPlan`
//!                                    (template `layer/gpu/mod.rs:74-293`, CPU semantics `layer/cpu/mod.rs:102-238`)
//!   `SumcheckLayerProver::prove`   — plain sumcheck of the layer's single expression (`layer/cpu/mod.rs:72-96`; the CUDA arm panics)
//!   `LinearLayerProver::prove`     — no sumcheck: evaluate every witness at the out point (`layer/cpu/mod.rs:47-66`)
//!   `prove_rotation_hip`           — `prove_rotation_gpu` (`layer/gpu/mod.rs:305-462`)
//! NOT COMPILED in the image it was written in.
use crate::{
    gkr::{
        booleanhypercube::BooleanHypercube,
        layer::{
            hal::{LinearLayerProver, SumcheckLayerProver, ZerocheckLayerProver},
            sumcheck_layer::{LayerProof, SumcheckLayerProof},
            zerocheck_layer::RotationPoints,
            Layer, LayerWitness,
        },
    },
    hal::ProverBackend,
    hip::{exts_words, flatten_terms, get_hip_hal, get_thread_stream, words_ext, words_exts, HipBackend, HipProver, MultilinearExtensionHip, TranscriptAdapter},
    selector::{SelectorContext, SelectorType},
};
use ceno_hip::{sumcheck::CommonTermPlan, sys, ExtWords, HipMle};
use either::Either;
use ff_ext::ExtensionField;
use itertools::{chain, Itertools};
use mpcs::PolynomialCommitmentScheme;
use multilinear_extensions::{mle::Point, monomial::Term, Expression};
use sumcheck::{
    structs::{IOPProof, IOPProverMessage},
    util::get_challenge_pows,
};
use transcript::Transcript;

fn iop_proof<E: ExtensionField>(msgs: &[Vec<ExtWords>]) -> IOPProof<E> {
    IOPProof { proofs: msgs.iter().map(|m| IOPProverMessage { evaluations: words_exts(m) }).collect() }
}

/// `build_eq_x_r_with_sel_gpu` (`layer/gpu/utils.rs:121-190`): eq(x, point) masked by the selector, built on the device
pub fn build_eq_x_r_with_sel_hip<E: ExtensionField>(point: &Point<E>, ctx: &SelectorContext, selector: &SelectorType<E>) -> MultilinearExtensionHip<'static, E> {
    let hal = get_hip_hal().expect("HIP HAL");
    let stream = get_thread_stream();
    let p = exts_words(point);
    let m = match selector {
        SelectorType::None => panic!("SelectorType::None"),
        SelectorType::Whole(_) => HipMle::selector(&hal, sys::CENO_HIP_SEL_WHOLE, &p, 0, 1 << point.len(), &[], 0, stream.as_deref()),
        SelectorType::Prefix(_) => HipMle::selector(&hal, sys::CENO_HIP_SEL_PREFIX, &p, ctx.offset, ctx.num_instances, &[], 0, stream.as_deref()),
        SelectorType::OrderedSparse { indices, num_vars, .. } => {
            assert_eq!(ctx.offset, 0);
            let idx = indices.iter().map(|x| *x as u32).collect_vec();
            HipMle::selector(&hal, sys::CENO_HIP_SEL_ORDERED_SPARSE, &p, 0, ctx.num_instances, &idx, *num_vars, stream.as_deref())
        }
        // implemented here (the CUDA arm has `unimplemented!()`): `selector.rs:192-244`
        SelectorType::QuarkBinaryTreeLessThan(_) => HipMle::selector(&hal, sys::CENO_HIP_SEL_QUARK_LT, &p, 0, ctx.num_instances, &[], 0, stream.as_deref()),
    };
    MultilinearExtensionHip::from_hip(m.expect("selector build"))
}

impl<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>> LinearLayerProver<HipBackend<E, PCS>> for HipProver<HipBackend<E, PCS>> {
    fn prove(layer: &Layer<E>, wit: LayerWitness<HipBackend<E, PCS>>, out_point: &Point<E>, transcript: &mut impl Transcript<E>) -> LayerProof<E> {
        // a linear layer has no sumcheck: the proof is the witness evaluations at the out point (layer/cpu/mod.rs:47-66)
        let evals = wit.iter().take(layer.n_witin).map(|m| m.evaluate(out_point)).collect_vec();
        transcript.append_field_element_exts(&evals);
        LayerProof { main: SumcheckLayerProof { proof: IOPProof { proofs: vec![] }, evals } }
    }
}

impl<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>> SumcheckLayerProver<HipBackend<E, PCS>> for HipProver<HipBackend<E, PCS>> {
    fn prove(layer: &Layer<E>, _num_threads: usize, max_num_variables: usize, wit: LayerWitness<'_, HipBackend<E, PCS>>, challenges: &[E],
             transcript: &mut impl Transcript<E>) -> LayerProof<E> {
        let hal = get_hip_hal().expect("HIP HAL");
        let stream = get_thread_stream();
        let terms = layer.main_sumcheck_expression_monomial_terms.as_ref().expect("main sumcheck monomial terms must exist");
        let (coeffs, idx) = flatten_terms(terms, &[], challenges);
        let mles = wit.iter().map(|m| m.inner().as_ref()).collect_vec();
        let max_degree = idx.iter().map(|t| t.len()).max().unwrap_or(0);
        let (msgs, evals, _point) = ceno_hip::sumcheck::prove(&hal, &mles, &coeffs, &idx, max_num_variables, max_degree, None,
                                                               &mut TranscriptAdapter::new(transcript), stream.as_deref()).expect("sumcheck");
        let evals = words_exts::<E>(&evals);
        transcript.append_field_element_exts(&evals);
        LayerProof { main: SumcheckLayerProof { proof: iop_proof(&msgs), evals } }
    }
}

impl<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>> ZerocheckLayerProver<HipBackend<E, PCS>> for HipProver<HipBackend<E, PCS>> {
    #[allow(clippy::too_many_arguments)]
    fn prove(layer: &Layer<E>, _num_threads: usize, max_num_variables: usize, wit: LayerWitness<HipBackend<E, PCS>>, out_points: &[Point<E>],
             pub_io_evals: &[E], challenges: &[E], transcript: &mut impl Transcript<E>, selector_ctxs: &[SelectorContext]) -> (LayerProof<E>, Point<E>) {
        let hal = get_hip_hal().expect("HIP HAL");
        let stream = get_thread_stream();
        assert_eq!(challenges.len(), 2);
        assert_eq!(layer.out_sel_and_eval_exprs.len(), out_points.len());
        // alpha powers inside the layer (layer/cpu/mod.rs:138-142)
        let main_sumcheck_challenges = chain!(challenges.iter().copied(), get_challenge_pows(layer.exprs.len(), transcript)).collect_vec();
        // selector eq tables, first occurrence per structural witness id wins (layer/cpu/mod.rs:145-178)
        let mut selector_eq_by_wit_id: Vec<Option<MultilinearExtensionHip<'static, E>>> = vec![None; layer.n_structural_witin];
        for (((sel_type, _), point), ctx) in layer.out_sel_and_eval_exprs.iter().zip(out_points).zip(selector_ctxs) {
            let expr = match sel_type {
                SelectorType::Whole(e) | SelectorType::Prefix(e) | SelectorType::OrderedSparse { expression: e, .. } | SelectorType::QuarkBinaryTreeLessThan(e) => e,
                SelectorType::None => continue,
            };
            let Expression::StructuralWitIn(wit_id, _) = expr else { panic!("selector expression must be StructuralWitIn") };
            let wit_id = *wit_id as usize;
            assert!(wit_id < layer.n_structural_witin, "selector wit id out of range");
            if selector_eq_by_wit_id[wit_id].is_none() {
                selector_eq_by_wit_id[wit_id] = Some(build_eq_x_r_with_sel_hip(point, ctx, sel_type));
            }
        }
        // wit := witin ++ fixed ++ structural, selector slots replaced by the computed eq tables
        let base = layer.n_witin + layer.n_fixed;
        let all: Vec<&HipMle> = wit.iter().take(base).map(|m| m.inner().as_ref())
            .chain(selector_eq_by_wit_id.iter().zip(wit.iter().skip(base).take(layer.n_structural_witin))
                   .map(|(eq, m)| eq.as_ref().map_or_else(|| m.inner().as_ref(), |e| e.inner().as_ref())))
            .collect_vec();
        assert_eq!(all.len(), layer.n_witin + layer.n_fixed + layer.n_structural_witin);
        // residual monomials + common-factor plan (zerocheck_layer.rs:389-513)
        let plan = layer.main_sumcheck_expression_common_factored.as_ref();
        let monomial_terms = match (plan, layer.main_sumcheck_expression_monomial_terms_excluded_shared.as_ref()) {
            (Some(_), Some(residual)) => residual.clone(),
            (Some(_), None) => panic!("common factoring plan present without residual monomials"),
            (None, Some(terms)) => terms.clone(),
            (None, None) => layer.main_sumcheck_expression_monomial_terms.clone().expect("main sumcheck monomial terms must exist"),
        };
        let (coeffs, idx) = flatten_terms(&monomial_terms, &pub_io_evals.iter().map(|v| Either::Right(*v)).collect_vec(), &main_sumcheck_challenges);
        let hip_plan = plan.map(|p| CommonTermPlan {
            group_terms: p.groups.iter().map(|g| g.term_indices.clone()).collect(),
            group_common_mles: p.groups.iter().map(|g| g.witness_indices.clone()).collect(),
        });
        let max_degree = match plan {
            Some(p) => p.groups.iter().flat_map(|g| g.term_indices.iter().map(|t| g.shared_len + idx.get(*t).map_or(0, |v| v.len()))).max()
                .unwrap_or_else(|| idx.iter().map(|t| t.len()).max().unwrap_or(0)),
            None => idx.iter().map(|t| t.len()).max().unwrap_or(0),
        };
        let (msgs, evals, point) = ceno_hip::sumcheck::prove(&hal, &all, &coeffs, &idx, max_num_variables, max_degree, hip_plan.as_ref(),
                                                              &mut TranscriptAdapter::new(transcript), stream.as_deref()).expect("sumcheck");
        let evals = words_exts::<E>(&evals);
        transcript.append_field_element_exts(&evals);
        (LayerProof { main: SumcheckLayerProof { proof: iop_proof(&msgs), evals } }, words_exts(&point))
    }
}

/// `prove_rotation_gpu`: rotated copies + cyclic-subgroup selector on the device, degree-2 sumcheck, left evaluations by the
/// device evaluate kernel, right evaluations derived (`booleanhypercube.rs:170-186`)
#[allow(clippy::too_many_arguments)]
pub fn prove_rotation_hip<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>>(
    max_num_variables: usize, rotation_cyclic_subgroup_size: usize, rotation_cyclic_group_log2: usize, wit: &LayerWitness<HipBackend<E, PCS>>,
    raw_rotation_exprs: &[(Expression<E>, Expression<E>)], rotation_sumcheck_expression: Vec<Term<Expression<E>, Expression<E>>>, rt: &Point<E>,
    global_challenges: &[E], transcript: &mut impl Transcript<E>,
) -> (SumcheckLayerProof<E>, RotationPoints<E>) {
    let hal = get_hip_hal().expect("HIP HAL");
    let stream = get_thread_stream();
    let bh = BooleanHypercube::new(rotation_cyclic_group_log2);
    let wit_id = |e: &Expression<E>| match e {
        Expression::WitIn(id) => *id as usize,
        _ => panic!("rotation expressions must be WitIn"),
    };
    let rotated = raw_rotation_exprs.iter().map(|(src, _)| wit[wit_id(src)].inner().rotation_next_base(rotation_cyclic_group_log2, stream.as_deref()).expect("rotate")).collect_vec();
    let selector = HipMle::rotation_selector(&hal, &exts_words(rt), rotation_cyclic_subgroup_size, rotation_cyclic_group_log2, stream.as_deref()).expect("rotation selector");
    let rotation_challenges = chain!(global_challenges.iter().copied(), get_challenge_pows(raw_rotation_exprs.len(), transcript)).collect_vec();
    // mles: [rot_0, tgt_0, rot_1, tgt_1, .., selector]
    let mles: Vec<&HipMle> = rotated.iter().zip_eq(raw_rotation_exprs).flat_map(|(r, (_, tgt))| [r, wit[wit_id(tgt)].inner().as_ref()]).chain(std::iter::once(&selector)).collect_vec();
    let (coeffs, idx) = flatten_terms(&rotation_sumcheck_expression, &[], &rotation_challenges);
    let max_degree = idx.iter().map(|t| t.len()).max().unwrap_or(0);
    let (msgs, evals, point) = ceno_hip::sumcheck::prove(&hal, &mles, &coeffs, &idx, max_num_variables, max_degree, None,
                                                          &mut TranscriptAdapter::new(transcript), stream.as_deref()).expect("sumcheck");
    let mut evals_e = words_exts::<E>(&evals);
    evals_e.truncate(raw_rotation_exprs.len() * 2);  // the verifier derives the selector evaluation itself
    let origin = words_exts::<E>(&point);
    let (left_point, right_point) = bh.get_rotation_points(&origin);
    let evals = evals_e.chunks_exact(2).zip_eq(raw_rotation_exprs).flat_map(|(ev, (src, _))| {
        let left = wit[wit_id(src)].evaluate(&left_point);
        let right = bh.get_rotation_right_eval_from_left(ev[0], left, &origin);
        [left, right, ev[1]]
    }).collect_vec();
    transcript.append_field_element_exts(&evals);
    let _ = words_ext::<E>;
    (SumcheckLayerProof { proof: iop_proof(&msgs), evals }, RotationPoints { left: left_point, right: right_point, origin })
}
