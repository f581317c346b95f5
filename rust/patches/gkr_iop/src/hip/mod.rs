//! `gkr_iop::hip` — the MI355X arm of the GKR-IOP back end (feature `hip`), sibling of `gkr_iop::gpu` (the CUDA arm over the
//! external `ceno_gpu` crate).  Drop this file in as `gkr_iop/src/hip/mod.rs`; `rust/patches/0001-hip-backend.patch` adds the
//! `mod` lines, the cargo feature and the `create_backend` arm.  NOT COMPILED in the image it was written in (no Rust there).
//!
//! Field: `GoldilocksExt2` only (`north_star`); a table element crosses the boundary as canonical `u64` words.
use crate::{
    gkr::layer::Layer,
    hal::{MultilinearPolynomial, ProtocolWitnessGeneratorProver, ProverBackend, ProverDevice},
};
use ceno_hip::{sumcheck::csr, ExtWords, FsTranscript, HipMle};
pub use ceno_hip::{bind_thread_stream, get_hip_hal, get_thread_stream, HipHal, HipStream, ThreadStreamGuard};
use either::Either;
use ff_ext::{ExtensionField, GoldilocksExt2};
use itertools::Itertools;
use mpcs::{PolynomialCommitmentScheme, SecurityLevel};
use multilinear_extensions::{
    mle::{FieldType, MultilinearExtension, Point},
    monomial::Term,
    Expression,
};
use p3::{
    field::{FieldAlgebra, PrimeField64, TwoAdicField},
    goldilocks::Goldilocks,
};
use std::{marker::PhantomData, sync::Arc};
use transcript::Transcript;
use witness::RowMajorMatrix;

use crate::cpu::default_backend_config;

// ------------------------------------------------------------------------------------------------
// field elements <-> boundary words
// ------------------------------------------------------------------------------------------------
fn assert_goldilocks<E: ExtensionField>() {
    assert!(
        std::any::TypeId::of::<E>() == std::any::TypeId::of::<GoldilocksExt2>(),
        "HIP backend only supports GoldilocksExt2"
    );
}
/// canonical words of one extension element (p3 keeps a possibly non-canonical `u64`: always go through `as_canonical_u64`)
pub fn ext_words<E: ExtensionField>(e: &E) -> ExtWords {
    let c = e.as_bases();
    [c[0].to_canonical_u64(), c[1].to_canonical_u64()]
}
pub fn words_ext<E: ExtensionField>(w: &ExtWords) -> E {
    E::from_bases(&[E::BaseField::from_canonical_u64(w[0]), E::BaseField::from_canonical_u64(w[1])])
}
pub fn exts_words<E: ExtensionField>(v: &[E]) -> Vec<ExtWords> {
    v.iter().map(ext_words).collect()
}
pub fn words_exts<E: ExtensionField>(v: &[ExtWords]) -> Vec<E> {
    v.iter().map(words_ext).collect()
}

/// `impl Transcript<E>` seen through the HAL's `FsTranscript`
pub struct TranscriptAdapter<'a, E: ExtensionField, T: Transcript<E>>(pub &'a mut T, PhantomData<E>);
impl<'a, E: ExtensionField, T: Transcript<E>> TranscriptAdapter<'a, E, T> {
    pub fn new(t: &'a mut T) -> Self {
        Self(t, PhantomData)
    }
}
impl<'a, E: ExtensionField, T: Transcript<E>> FsTranscript for TranscriptAdapter<'a, E, T> {
    fn append_bytes(&mut self, bytes: &[u8]) {
        self.0.append_message(bytes);
    }
    fn append_ext(&mut self, e: ExtWords) {
        self.0.append_field_element_ext(&words_ext::<E>(&e));
    }
    fn append_base(&mut self, b: u64) {
        self.0.append_field_element(&E::BaseField::from_canonical_u64(b));
    }
    fn sample(&mut self) -> ExtWords {
        ext_words(&self.0.read_challenge().elements)
    }
    fn challenge_pows(&mut self, n: usize) -> Vec<ExtWords> {
        exts_words(&sumcheck::util::get_challenge_pows::<E>(n, self.0))
    }
    // `Transcript<E>: CanObserve<F> + CanSampleBits<usize> + GrindingChallenger` (ceno_recursion_v2/src/tower/tower.rs:85-101)
    fn sample_bits(&mut self, bits: usize) -> usize {
        p3::challenger::CanSampleBits::sample_bits(self.0, bits)
    }
    fn grind(&mut self, bits: usize) -> u64 {
        p3::challenger::GrindingChallenger::grind(self.0, bits).as_canonical_u64()
    }
}

// ------------------------------------------------------------------------------------------------
// device MLE
// ------------------------------------------------------------------------------------------------
/// Dense multilinear polynomial in device memory (`MultilinearExtensionGpu` of the CUDA arm).  Cloning shares the table.
#[derive(Clone, Debug, Default)]
pub struct MultilinearExtensionHip<'a, E: ExtensionField> {
    pub mle: Option<Arc<HipMle>>,
    _phantom: PhantomData<&'a E>,
}
impl<'a, E: ExtensionField> MultilinearExtensionHip<'a, E> {
    pub fn from_hip(mle: HipMle) -> Self {
        Self { mle: Some(Arc::new(mle)), _phantom: PhantomData }
    }
    pub fn inner(&self) -> &Arc<HipMle> {
        self.mle.as_ref().expect("Unreachable MultilinearExtensionHip")
    }
    /// upload a CPU polynomial (`from_ceno`)
    pub fn from_ceno(hal: &Arc<HipHal>, mle: &MultilinearExtension<'a, E>) -> MultilinearExtensionHip<'static, E> {
        assert_goldilocks::<E>();
        let stream = get_thread_stream();
        let (words, is_ext): (Vec<u64>, bool) = match &mle.evaluations {
            FieldType::Base(_) => (mle.get_base_field_vec().iter().map(|b| b.to_canonical_u64()).collect(), false),
            FieldType::Ext(_) => (mle.get_ext_field_vec().iter().flat_map(|e| ext_words(e)).collect(), true),
            FieldType::Unreachable => panic!("Unreachable FieldType"),
        };
        let m = HipMle::from_host(hal, &words, mle.num_vars(), is_ext, stream.as_deref()).expect("upload");
        MultilinearExtensionHip { mle: Some(Arc::new(m)), _phantom: PhantomData }
    }
    /// download (`inner_to_mle`)
    pub fn inner_to_mle(&self) -> MultilinearExtension<'a, E> {
        let stream = get_thread_stream();
        let m = self.inner();
        let words = m.to_host(stream.as_deref()).expect("download");
        if m.is_ext() {
            MultilinearExtension::from_evaluations_ext_vec_compact(m.num_vars(), words.chunks(2).map(|w| words_ext::<E>(&[w[0], w[1]])).collect())
        } else {
            MultilinearExtension::from_evaluations_vec_compact(m.num_vars(), words.iter().map(|&w| E::BaseField::from_canonical_u64(w)).collect())
        }
    }
    /// evaluate ON THE DEVICE (one read-only pass; the CUDA arm downloads and evaluates on the host)
    pub fn evaluate(&self, point: &[E]) -> E {
        let stream = get_thread_stream();
        words_ext(&self.inner().evaluate(&exts_words(point), stream.as_deref()).expect("evaluate"))
    }
}
impl<'a, E: ExtensionField> MultilinearPolynomial<E> for MultilinearExtensionHip<'a, E> {
    fn num_vars(&self) -> usize {
        self.inner().num_vars()
    }
    fn eval(&self, point: Point<E>) -> E {
        self.evaluate(&point)
    }
    fn evaluations_len(&self) -> usize {
        self.inner().evaluations_len()
    }
    fn bh_signature(&self) -> E {
        self.inner_to_mle().bh_signature()
    }
}

// ------------------------------------------------------------------------------------------------
// back end and prover
// ------------------------------------------------------------------------------------------------
pub struct HipBackend<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>> {
    pub pp: <PCS as PolynomialCommitmentScheme<E>>::ProverParam,
    pub vp: <PCS as PolynomialCommitmentScheme<E>>::VerifierParam,
    pub max_poly_size_log2: usize,
    pub security_level: SecurityLevel,
    _marker: PhantomData<E>,
}
impl<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>> Default for HipBackend<E, PCS> {
    fn default() -> Self {
        let (max_poly_size_log2, security_level) = default_backend_config();
        Self::new(max_poly_size_log2, security_level)
    }
}
impl<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>> HipBackend<E, PCS> {
    pub fn new(max_poly_size_log2: usize, security_level: SecurityLevel) -> Self {
        assert_goldilocks::<E>();
        let param = PCS::setup(1 << E::BaseField::TWO_ADICITY, security_level).unwrap();
        let (pp, vp) = PCS::trim(param, 1 << max_poly_size_log2).unwrap();
        Self { pp, vp, max_poly_size_log2, security_level, _marker: PhantomData }
    }
}
impl<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>> ProverBackend for HipBackend<E, PCS> {
    type E = E;
    type Pcs = PCS;
    type MultilinearPoly<'a> = MultilinearExtensionHip<'static, E>;
    type Matrix = RowMajorMatrix<E::BaseField>;
    type PcsData = Arc<ceno_hip::pcs::HipPcsData>;

    fn get_pp(&self) -> &<Self::Pcs as PolynomialCommitmentScheme<Self::E>>::ProverParam {
        &self.pp
    }
    fn get_vp(&self) -> &<Self::Pcs as PolynomialCommitmentScheme<Self::E>>::VerifierParam {
        &self.vp
    }
}

pub struct HipProver<PB: ProverBackend + 'static> {
    pub backend: Arc<PB>,
}
impl<PB: ProverBackend> HipProver<PB> {
    pub fn new(backend: Arc<PB>) -> Self {
        Self { backend }
    }
}
impl<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>> ProverDevice<HipBackend<E, PCS>> for HipProver<HipBackend<E, PCS>> {}

/// scalar of a monomial term at the given challenges / public values (`eval_by_expr_constant`), widened to `E`
pub(crate) fn term_scalar<E: ExtensionField>(scalar: &Expression<E>, pub_io_evals: &[Either<E::BaseField, E>], challenges: &[E]) -> E {
    match crate::utils::eval_by_expr_constant(pub_io_evals, challenges, scalar) {
        Either::Left(b) => E::from(b),
        Either::Right(e) => e,
    }
}
/// scalar of a main-sumcheck monomial with the chip's public-instance values (`eval_by_expr_with_instance(&[], &[], &[], &chip.pi,
/// &main_sumcheck_challenges, scalar_expr)`, `ceno_zkvm/src/scheme/cpu/mod.rs:1297-1304`), widened to `E`
pub fn term_scalar_with_instance<E: ExtensionField>(scalar: &Expression<E>, pi: &[Either<E::BaseField, E>], challenges: &[E]) -> E {
    match crate::utils::eval_by_expr_with_instance(&[], &[], &[], pi, challenges, scalar) {
        Either::Left(b) => E::from(b),
        Either::Right(e) => e,
    }
}
/// monomial terms -> (coefficients, factor lists): `extract_mle_relationships_from_monomial_terms` (`layer/gpu/utils.rs:24-67`)
pub(crate) fn flatten_terms<E: ExtensionField>(terms: &[Term<Expression<E>, Expression<E>>], pub_io_evals: &[Either<E::BaseField, E>], challenges: &[E])
                                               -> (Vec<ExtWords>, Vec<Vec<usize>>) {
    let mut coeffs = vec![];
    let mut idx = vec![];
    for t in terms {
        coeffs.push(ext_words(&term_scalar(&t.scalar, pub_io_evals, challenges)));
        idx.push(t.product.iter().map(|e| match e {
            Expression::WitIn(id) => *id as usize,
            e => panic!("Unsupported expression in product: {e:?}"),
        }).collect_vec());
    }
    (coeffs, idx)
}

impl<E: ExtensionField, PCS: PolynomialCommitmentScheme<E>> ProtocolWitnessGeneratorProver<HipBackend<E, PCS>> for HipProver<HipBackend<E, PCS>> {
    fn layer_witness<'a>(layer: &Layer<E>, layer_wits: &[Arc<MultilinearExtensionHip<'static, E>>], pub_io_evals: &[Either<E::BaseField, E>],
                         challenges: &[E]) -> Vec<Arc<MultilinearExtensionHip<'static, E>>> {
        Self::layer_witness_filtered(layer, layer_wits, pub_io_evals, challenges, None)
    }

    /// every output expression of the layer in ONE `wit_infer` launch (`wit_infer_by_monomial_expr`, `gkr_iop/src/gpu/mod.rs:483-640`);
    /// outputs masked out keep a default (unreachable) handle
    fn layer_witness_filtered<'a>(layer: &Layer<E>, layer_wits: &[Arc<MultilinearExtensionHip<'static, E>>], pub_io_evals: &[Either<E::BaseField, E>],
                                  challenges: &[E], output_mask: Option<&[bool]>) -> Vec<Arc<MultilinearExtensionHip<'static, E>>> {
        let hal = get_hip_hal().expect("HIP HAL");
        let stream = get_thread_stream();
        let num_vars = layer_wits.iter().find_map(|m| m.mle.as_ref().map(|m| m.num_vars())).expect("layer without witness");
        let present: Vec<(usize, &HipMle)> = layer_wits.iter().enumerate().filter_map(|(i, m)| m.mle.as_deref().map(|m| (i, m))).collect();
        let remap: std::collections::HashMap<usize, usize> = present.iter().enumerate().map(|(k, (i, _))| (*i, k)).collect();
        let (mut coeffs, mut terms, mut ranges, mut kept) = (vec![], vec![], vec![], vec![]);
        let out_exprs = layer.out_sel_and_eval_exprs.iter().flat_map(|(_, outs)| outs.iter()).zip(layer.exprs_with_selector_out_eval_monomial_form.iter());
        for (o, (_out_eval, monomial_terms)) in out_exprs.enumerate() {
            if output_mask.map_or(false, |m| !m[o]) {
                continue;
            }
            let (c, t) = flatten_terms(monomial_terms, pub_io_evals, challenges);
            let begin = terms.len();
            coeffs.extend(c);
            terms.extend(t.into_iter().map(|f| f.into_iter().map(|j| remap[&j]).collect_vec()));
            ranges.push(begin..terms.len());
            kept.push(o);
        }
        let outs = ceno_hip::mle::wit_infer(&hal, &present.iter().map(|p| p.1).collect_vec(), &coeffs, &terms, &ranges, num_vars, stream.as_deref()).expect("wit_infer");
        let n_out = layer.out_sel_and_eval_exprs.iter().map(|(_, outs)| outs.len()).sum::<usize>();
        let mut res = vec![Arc::new(MultilinearExtensionHip::default()); n_out];
        for (o, m) in kept.into_iter().zip(outs) {
            res[o] = Arc::new(MultilinearExtensionHip::from_hip(m));
        }
        let _ = csr; // (CSR helpers live in the HAL crate)
        res
    }
}
