//! PARITY-CLOSING KIT, reference side.  Drop this file into the reference as `ceno_zkvm/src/scheme/dump_goldens.rs`, add
//! `#[cfg(test)] mod dump_goldens;` to `ceno_zkvm/src/scheme.rs`, and run
//!
//!     cargo test -p ceno_zkvm --no-default-features --features goldilocks dump_hip_goldens -- --nocapture
//!
//! It writes `ref_goldens.json`; copy it to `tests/golden/ref_goldens.json` of the ceno_amd repository and run
//! `python -m pytest tests/test_ref_goldens.py`: everything marked PARITY UNPINNED in DESIGN.md section 5 (the extension's
//! W, the Poseidon2-Goldilocks table, label packing, the `BasicTranscript` byte stream, a sumcheck proof, the Basefold root
//! and opening) is then checked bit for bit against what the REAL reference produced.
//!
//! Written without a Rust toolchain at hand (none in the image of that repository): API names follow the call sites in
//! this tree (`ceno_zkvm/src/structs.rs:595`, `scheme/prover.rs:324,344-368,528-531,556-570`, `scheme/cpu/mod.rs:579,1456`,
//! `gkr_iop/src/gkr/layer/cpu/mod.rs:80-91`); adjust an import if one moved.
use ff_ext::{ExtensionField, GoldilocksExt2, PoseidonField, SmallField};
use itertools::Itertools;
use mpcs::{Basefold, BasefoldRSParams, PolynomialCommitmentScheme, SecurityLevel};
use multilinear_extensions::{
    mle::{IntoMLE, MultilinearExtension},
    virtual_polys::VirtualPolynomialsBuilder,
    Expression, ToExpr,
};
use p3::{
    field::{FieldAlgebra, PrimeField64},
    goldilocks::Goldilocks,
    symmetric::Permutation,
};
use serde_json::{json, Value};
use sumcheck::structs::IOPProverState;
use transcript::{BasicTranscript, Transcript};
use witness::RowMajorMatrix;

type E = GoldilocksExt2;
type F = Goldilocks;
type Pcs = Basefold<E, BasefoldRSParams>;

fn b(v: F) -> u64 {
    v.as_canonical_u64()
}
fn e(v: E) -> Vec<u64> {
    v.as_bases().iter().map(|x| b(*x)).collect()
}
fn es(v: &[E]) -> Vec<Vec<u64>> {
    v.iter().map(|x| e(*x)).collect()
}
/// deterministic field data (SplitMix64, the generator of ceno_amd's synthetic inputs)
fn splitmix(seed: u64, i: u64) -> u64 {
    let mut z = seed.wrapping_add((i + 1).wrapping_mul(0x9E3779B97F4A7C15));
    z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
    z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
    z ^= z >> 31;
    if z >= 0xFFFF_FFFF_0000_0001 { z - 0xFFFF_FFFF_0000_0001 } else { z }
}
fn fe(seed: u64, i: u64) -> F {
    F::from_canonical_u64(splitmix(seed, i))
}
fn ee(seed: u64, i: u64) -> E {
    E::from_bases(&[fe(seed, 2 * i), fe(seed, 2 * i + 1)])
}

#[test]
fn dump_hip_goldens() {
    let mut out = serde_json::Map::new();

    // ---- 1. the extension: one product pins W (X^2 = W) ----
    let (x, y) = (ee(1, 0), ee(1, 1));
    out.insert("ext_mul".into(), json!({ "a": e(x), "b": e(y), "ab": e(x * y), "a_inv": e(x.inverse()) }));

    // ---- 2. Poseidon2 over Goldilocks as the transcript / Merkle tree use it ----
    let perm = <F as PoseidonField>::get_default_perm();
    let rc = <F as PoseidonField>::get_default_perm_rc();  // the round-constant table the shard-RAM circuit embeds (tables/shard_ram.rs:227)
    let mut kats = vec![];
    for seed in 0..4u64 {
        // width is the permutation's own: the array length below must match it (8 for Poseidon2GoldilocksHL<8>)
        let input: [F; 8] = std::array::from_fn(|i| if seed == 0 { F::from_canonical_u64(i as u64) } else { fe(100 + seed, i as u64) });
        let output = perm.permute(input);
        kats.push(json!({ "in": input.iter().map(|v| b(*v)).collect_vec(), "out": output.iter().map(|v| b(*v)).collect_vec() }));
    }
    out.insert("poseidon2".into(), json!({
        "width": 8, "rate": 4,
        // serialise whatever shape the table has; tests/test_ref_goldens.py accepts {external:[[..8]x8], internal:[..22], diag:[..8]}
        // or a flat list in (external initial, internal, external terminal) order
        "round_constants": serde_json::to_value(&rc).unwrap_or(Value::Null),
        "kats": kats,
    }));

    // ---- 3. label packing ----
    let labels: [&[u8]; 8] = [b"riscv", b"fork", b"Internal round", b"combine subset evals", b"product_sum", b"merge", b"batch coeffs", b"query indices"];
    out.insert("bytes_to_field_elements".into(), Value::Array(labels.iter().map(|l| {
        json!({ "label": String::from_utf8_lossy(l), "elements": F::bytes_to_field_elements(l).iter().map(|v| b(*v)).collect_vec() })
    }).collect()));

    // ---- 4. BasicTranscript script (prover.rs:343-368,528-531,556-570) ----
    let mut t = BasicTranscript::<E>::new(b"riscv");
    let mut script = vec![];
    t.append_message(&26usize.to_le_bytes());
    t.append_message(&3usize.to_le_bytes());
    script.push(json!({ "op": "append_message", "bytes": 26usize.to_le_bytes().to_vec() }));
    script.push(json!({ "op": "append_message", "bytes": 3usize.to_le_bytes().to_vec() }));
    for i in 0..3 {
        t.append_field_element_ext(&ee(7, i));
        script.push(json!({ "op": "append_ext", "value": e(ee(7, i)) }));
    }
    let c1 = t.sample_and_append_challenge(b"Internal round").elements;
    script.push(json!({ "op": "challenge", "label": "Internal round", "value": e(c1) }));
    t.append_field_element(&fe(8, 0));
    script.push(json!({ "op": "append_base", "value": b(fe(8, 0)) }));
    let c2 = t.read_challenge().elements;
    script.push(json!({ "op": "read_challenge", "value": e(c2) }));
    let pows = sumcheck::util::get_challenge_pows::<E>(4, &mut t);
    script.push(json!({ "op": "challenge_pows", "n": 4, "value": es(&pows) }));
    let v = t.sample_and_append_vec(b"product_sum", 2);
    script.push(json!({ "op": "sample_and_append_vec", "label": "product_sum", "n": 2, "value": es(&v) }));
    let mut fork = BasicTranscript::<E>::new(b"fork");
    fork.append_field_element_ext(&c1);
    script.push(json!({ "op": "fork_sample", "appended": e(c1), "value": e(fork.read_challenge().elements) }));
    out.insert("transcript".into(), json!({ "label": "riscv", "script": script }));

    // ---- 5. one IOPProverState::prove: sum_x f g h, three extension MLEs of 4 variables ----
    let nv = 4usize;
    let tables = (0..3u64).map(|j| (0..1u64 << nv).map(|i| ee(0xCE10 + j, i)).collect_vec()).collect_vec();
    let mles: Vec<MultilinearExtension<E>> = tables.iter().map(|t| t.clone().into_mle()).collect();
    let mut builder = VirtualPolynomialsBuilder::new(1, nv);
    let exprs: Vec<Expression<E>> = mles.iter().map(|m| builder.lift(either::Either::Left(m))).collect();
    let mut tr = BasicTranscript::<E>::new(b"sumcheck");
    let (proof, state) = IOPProverState::prove(builder.to_virtual_polys(&[exprs.into_iter().product()], &[]), &mut tr);
    out.insert("sumcheck".into(), json!({
        "label": "sumcheck", "num_vars": nv, "degree": 3,
        "tables": tables.iter().map(|t| es(t)).collect_vec(),
        "messages": proof.proofs.iter().map(|m| es(&m.evaluations)).collect_vec(),
        "challenges": es(&state.collect_raw_challenges()),
        "final_evals": es(&state.get_mle_flatten_final_evaluations()),
    }));

    // ---- 6. Basefold: commit a 2^4 x 3 base matrix, open it at one point ----
    let (rows, width) = (16usize, 3usize);
    let values = (0..(rows * width) as u64).map(|i| fe(0xADD, i)).collect_vec();
    let rmm = RowMajorMatrix::<F>::new_by_values(values.clone(), width, witness::InstancePaddingStrategy::Default);
    let param = Pcs::setup(1 << 20, SecurityLevel::default()).unwrap();
    let (pp, _vp) = Pcs::trim(param, 1 << 10).unwrap();
    let comm_w = Pcs::batch_commit(&pp, vec![rmm]).unwrap();
    let comm = Pcs::get_pure_commitment(&comm_w);
    let point = (0..4u64).map(|i| ee(0x51, i)).collect_vec();
    let polys = Pcs::get_arc_mle_witness_from_commitment(&comm_w);
    let evals = polys.iter().map(|p| p.evaluate(&point)).collect_vec();
    let mut tr = BasicTranscript::<E>::new(b"basefold");
    Pcs::write_commitment(&comm, &mut tr).unwrap();
    let open = Pcs::batch_open(&pp, vec![(&comm_w, vec![(point.clone(), evals.clone())])], &mut tr).unwrap();
    out.insert("basefold".into(), json!({
        "label": "basefold", "rows": rows, "width": width, "values_row_major": values.iter().map(|v| b(*v)).collect_vec(),
        "point": es(&point), "evals": es(&evals),
        "commitment": serde_json::to_value(&comm).unwrap_or(Value::Null),
        "proof": serde_json::to_value(&open).unwrap_or(Value::Null),
        "params": { "note": "BasefoldRSParams: rate_log / num_queries / basecode_msg_size_log as compiled into mpcs" },
    }));

    std::fs::write("ref_goldens.json", serde_json::to_string_pretty(&Value::Object(out)).unwrap()).unwrap();
    println!("wrote ref_goldens.json");
}
