//! `ceno_hip` — safe Rust HAL over `libceno_hip.so` / `libceno_prover.so`, the MI355X (gfx950) counterpart of the external
//! `ceno_gpu` CUDA HAL that the reference's GPU arm calls (`gkr_iop/src/gpu/mod.rs`, `gkr_iop/src/gkr/layer/gpu/`,
//! `ceno_zkvm/src/scheme/gpu/`).  The in-tree arms that implement the reference's traits on top of this crate are under
//! `rust/patches/` (they must live inside `gkr_iop` / `ceno_zkvm`: those crates select the back end in
//! `create_backend` / `create_prover`, `ceno_zkvm/src/scheme.rs:371-405`).
//!
//! Conventions (see `include/ceno_hip.h`): field elements cross as canonical `u64`, an extension element as two words
//! `[c0, c1]`; tables are dense, index bit k <-> variable k; every call takes an explicit stream; the Fiat-Shamir
//! transcript stays on the Rust side, so sumcheck is driven round by round ([`sumcheck::prove`]).
//!
//! NOT COMPILED in the image this repository was built in (no Rust toolchain there); `tests/test_rust_shim.py` checks the
//! FFI layer against the C headers mechanically.
pub mod error;
pub mod hal;
pub mod mle;
pub mod pcs;
pub mod prover;
pub mod sumcheck;
pub mod tower;
pub mod witgen;

pub use ceno_hip_sys as sys;
pub use error::{HipError, Result};
pub use hal::{bind_thread_stream, get_hip_hal, get_thread_stream, HipHal, HipStream, ThreadStreamGuard};
pub use mle::HipMle;

/// One extension element as it crosses the boundary.
pub type ExtWords = [u64; 2];

/// The transcript operations the round loops need; the in-tree arms adapt `impl transcript::Transcript<E>` to it
/// (`append_message`, `append_field_element_ext`, `sample_and_append_challenge`).
pub trait FsTranscript {
    /// `Transcript::append_message(bytes)`
    fn append_bytes(&mut self, bytes: &[u8]);
    /// `Transcript::append_field_element_ext(e)`
    fn append_ext(&mut self, e: ExtWords);
    /// `Transcript::append_field_element(b)` (one base-field element)
    fn append_base(&mut self, b: u64);
    /// `Transcript::read_challenge().elements`
    fn sample(&mut self) -> ExtWords;
    /// `Transcript::sample_and_append_challenge(label).elements` = `append_message(label)` then `read_challenge()`
    fn challenge(&mut self, label: &[u8]) -> ExtWords {
        self.append_bytes(label);
        self.sample()
    }
    /// `get_challenge_pows(n, transcript)`: label `b"combine subset evals"`, one sample, `[1, a, a^2, ..]` — the powers are
    /// formed by the caller's field type, so the adapter returns them ready-made
    fn challenge_pows(&mut self, n: usize) -> Vec<ExtWords>;
    /// `CanSampleBits::sample_bits(bits)` (a supertrait of `Transcript`): low bits of ONE base-field sample — the query indices
    /// and the proof-of-work check of the PCS (`ceno_recursion_v2/src/pcs/mod.rs:8125-8204,1252-1266`)
    fn sample_bits(&mut self, bits: usize) -> usize;
    /// `GrindingChallenger::grind(bits)` (a supertrait of `Transcript`): the transcript's own proof-of-work search; the witness
    /// has been observed and the bits sampled when it returns
    fn grind(&mut self, bits: usize) -> u64;
    /// Poseidon2 duplex state `[sponge 8][n_in][in 4][n_out][0][0]` for the device-side search (`ceno_hip_pow_grind_duplex`);
    /// `None` when the challenger's fields are not reachable — the library then calls [`FsTranscript::grind`]
    fn export_duplex_state(&self) -> Option<[u64; 16]> {
        None
    }
}
