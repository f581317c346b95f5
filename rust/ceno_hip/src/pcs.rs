//! Basefold commit and batch open over the device.
//! Reference: `cuda_hal.basefold.{batch_commit, get_pure_commitment, get_trace, batch_open}`
//! (`ceno_zkvm/src/scheme/gpu/mod.rs:1642-1646,1717-1720,3401-3402`); traits `TraceCommitter::commit_traces`
//! (`scheme/hal.rs:137-156`) and `OpeningProver::open` (`:284-294`).  One commitment per `commit_traces` (mixed-height MMCS), p3
//! grinding and one-base-sample query indices as the in-tree verifier replays them (`ceno_recursion_v2/src/pcs/mod.rs`);
//! PARITY UNPINNED against the reference's `mpcs` crate for the Poseidon2 constants and label packing: DESIGN.md section 5.
use std::{ptr, sync::Arc};

use ceno_hip_sys as sys;

use crate::{error::Result, hal::HipHal, hal::HipStream, mle::HipMle, ExtWords};

/// committed traces: column-major traces and codewords per height class, ONE mixed-height Merkle tree (`PcsData`)
pub struct HipPcsData {
    hal: Arc<HipHal>,
    raw: *mut sys::ceno_pcs_data,
    pub widths: Vec<usize>,
}
unsafe impl Send for HipPcsData {}
unsafe impl Sync for HipPcsData {}

impl HipPcsData {
    /// `commit_traces`: row-major host matrices (`RowMajorMatrix::values`, `num_instances` rows each), padded to
    /// `next_pow2_instance_padding`, transposed, RS-encoded (blow-up `2^log_blowup`), row-hashed, Merkle-committed
    pub fn commit(hal: &Arc<HipHal>, matrices: &[(&[u64], usize, usize)], log_blowup: usize, stream: &HipStream) -> Result<Self> {
        let ptrs: Vec<*const u64> = matrices.iter().map(|m| m.0.as_ptr()).collect();
        let rows: Vec<usize> = matrices.iter().map(|m| m.1).collect();
        let widths: Vec<usize> = matrices.iter().map(|m| m.2).collect();
        let mut d = ptr::null_mut();
        hal.check_prover(unsafe {
            sys::ceno_prover_commit_traces(hal.ctx, ptrs.as_ptr(), rows.as_ptr(), widths.as_ptr(), matrices.len() as i32, log_blowup as i32, stream.raw(), &mut d)
        })?;
        Ok(Self { hal: hal.clone(), raw: d, widths })
    }
    /// same with device-resident row-major matrices (witness generated on the GPU: `device_backing_layout`)
    ///
    /// # Safety
    /// every pointer must reference `rows * width` device words that stay valid for the call.
    pub unsafe fn commit_device(hal: &Arc<HipHal>, matrices: &[(*const u64, usize, usize)], log_blowup: usize, stream: &HipStream) -> Result<Self> {
        let ptrs: Vec<*const u64> = matrices.iter().map(|m| m.0).collect();
        let rows: Vec<usize> = matrices.iter().map(|m| m.1).collect();
        let widths: Vec<usize> = matrices.iter().map(|m| m.2).collect();
        let mut d = ptr::null_mut();
        hal.check_prover(sys::ceno_prover_commit_traces_dev(hal.ctx, ptrs.as_ptr(), rows.as_ptr(), widths.as_ptr(), matrices.len() as i32,
                                                            log_blowup as i32, stream.raw(), &mut d))?;
        Ok(Self { hal: hal.clone(), raw: d, widths })
    }
    pub fn num_vars(&self, matrix: usize) -> usize {
        unsafe { sys::ceno_pcs_data_num_vars(self.raw, matrix as i32) as usize }
    }
    /// `get_pure_commitment`: THE root of the commitment (4 base-field words) — one for all matrices
    pub fn root(&self, stream: &HipStream) -> Result<[u64; 4]> {
        let mut r = [0u64; 4];
        self.hal.check_prover(unsafe { sys::ceno_pcs_data_root(self.hal.ctx, self.raw, r.as_mut_ptr(), stream.raw()) })?;
        Ok(r)
    }
    pub(crate) fn raw(&self) -> *mut sys::ceno_pcs_data {
        self.raw
    }
    /// `get_arc_mle_witness_from_commitment`: borrowed base-field view of one column (nothing is copied)
    pub fn witness_mle(self: &Arc<Self>, matrix: usize, col: usize) -> Result<HipMle> {
        let mut m = ptr::null_mut();
        self.hal.check_prover(unsafe { sys::ceno_pcs_data_witness_mle(self.hal.ctx, self.raw, matrix as i32, col, &mut m) })?;
        Ok(HipMle::from_raw(&self.hal, m))
    }
    /// `PCS::batch_open(rounds)`: `self` is the witness commitment, `fixed` the optional fixed one (`cpu/mod.rs:1418-1457`); every
    /// matrix at its own point with the claimed column evaluations, witness matrices first; flat proof words (layout:
    /// `include/ceno_prover.h`)
    pub fn batch_open(&self, fixed: Option<&HipPcsData>, points: &[Vec<ExtWords>], evals: &[Vec<ExtWords>], n_queries: usize, pow_bits: usize,
                      transcript: *mut sys::ceno_transcript, stream: &HipStream) -> Result<Vec<u64>> {
        let pp: Vec<*const u64> = points.iter().map(|p| p.as_ptr() as *const u64).collect();
        let ep: Vec<*const u64> = evals.iter().map(|e| e.as_ptr() as *const u64).collect();
        let commits: Vec<*mut sys::ceno_pcs_data> = std::iter::once(self.raw).chain(fixed.map(|f| f.raw())).collect();
        let words = unsafe { sys::ceno_prover_basefold_proof_words(commits.as_ptr(), commits.len() as i32, n_queries as i32) };
        let mut proof = vec![0u64; words];
        self.hal.check_prover(unsafe {
            sys::ceno_prover_basefold_open(self.hal.ctx, commits.as_ptr(), commits.len() as i32, pp.as_ptr(), ep.as_ptr(), n_queries as i32,
                                           pow_bits as i32, transcript, stream.raw(), proof.as_mut_ptr())
        })?;
        Ok(proof)
    }
}
impl Drop for HipPcsData {
    fn drop(&mut self) {
        unsafe { sys::ceno_pcs_data_free(self.hal.ctx, self.raw) };
    }
}

/// One rank's share of a commitment that spans the GPUs of a node (`ceno_dist_commit_traces`): this rank
/// RS-encodes its `widths[rank]` columns, ONE grouped `ncclSend` / `ncclRecv` all-to-all re-shards the codeword by rows, the
/// rank hashes the sub-tree over its rows and every rank finishes the top `log2(world)` levels from the gathered roots.
/// `comm` comes from `ceno_dist_comm_init` (RCCL) — the launcher distributes the 128-byte unique id.
pub struct ShardedCommitment {
    hal: Arc<HipHal>,
    pub root: [u64; 4],
    pub subtree_roots: Vec<[u64; 4]>,
    /// (sum of widths) x R / world codeword words, column-major: this rank's rows, kept for the openings
    pub rows: HipMle,
    subtree: *mut sys::ceno_hip_merkle,
}
unsafe impl Send for ShardedCommitment {}
impl ShardedCommitment {
    /// # Safety
    /// `local_cols_dev` must hold `widths[rank]` columns of `2^log_rows` base words each, column-major, on the device.
    pub unsafe fn commit(hal: &Arc<HipHal>, comm: *mut sys::ceno_dist_comm, world: usize, local_cols_dev: *const u64, widths: &[i32],
                         log_rows: usize, log_blowup: usize, stream: &HipStream) -> Result<Self> {
        assert_eq!(widths.len(), world);
        let w_total: usize = widths.iter().map(|&w| w as usize).sum();
        let local_words = (w_total << (log_rows + log_blowup)) / world;
        let rows = HipMle::alloc(hal, (usize::BITS - (local_words.max(2) - 1).leading_zeros()) as usize, false)?;
        let mut roots = vec![0u64; 4 * world];
        let mut root = [0u64; 4];
        let mut subtree = ptr::null_mut();
        hal.check_prover(sys::ceno_dist_commit_traces(hal.ctx, comm, local_cols_dev, widths.as_ptr(), log_rows as i32, log_blowup as i32,
                                                      stream.raw(), rows.device_ptr(), &mut subtree, roots.as_mut_ptr(), root.as_mut_ptr()))?;
        let subtree_roots = roots.chunks(4).map(|c| [c[0], c[1], c[2], c[3]]).collect();
        Ok(Self { hal: hal.clone(), root, subtree_roots, rows, subtree })
    }
}
/// The multi-rank form of `commit_traces`: SEVERAL matrices of several heights under ONE root across the GPUs of a node
/// (`ceno_dist_commit_traces_mmcs`); the root equals the single-device mixed-height commitment bit for bit.
pub struct ShardedMmcsCommitment {
    hal: Arc<HipHal>,
    pub root: [u64; 4],
    pub subtree_roots: Vec<[u64; 4]>,
    /// per matrix: this rank's codeword rows (all columns), column-major; the whole codeword for matrices shorter than the world
    pub rows: Vec<HipMle>,
    subtree: *mut sys::ceno_hip_merkle,
    top: *mut sys::ceno_hip_merkle,
}
unsafe impl Send for ShardedMmcsCommitment {}
impl ShardedMmcsCommitment {
    /// `widths[m * world + g]` = columns of matrix `m` on rank `g`; `local_cols_dev[m]` = this rank's columns of matrix `m`.
    ///
    /// # Safety
    /// every `local_cols_dev[m]` must hold `widths[m * world + rank]` columns of `2^log_rows[m]` base words, column-major, on the device.
    pub unsafe fn commit(hal: &Arc<HipHal>, comm: *mut sys::ceno_dist_comm, world: usize, log_rows: &[i32], widths: &[i32],
                         local_cols_dev: &[*const u64], log_blowup: usize, stream: &HipStream) -> Result<Self> {
        let n = log_rows.len();
        assert_eq!(widths.len(), n * world);
        let log_w = world.trailing_zeros() as usize;
        let mut rows = Vec::with_capacity(n);
        for m in 0..n {
            let w_total: usize = widths[m * world..(m + 1) * world].iter().map(|&w| w as usize).sum();
            let lr = log_rows[m] as usize + log_blowup;
            let local_words = if lr < log_w { w_total << lr } else { (w_total << lr) / world };
            rows.push(HipMle::alloc(hal, (usize::BITS - (local_words.max(2) - 1).leading_zeros()) as usize, false)?);
        }
        let outs: Vec<*mut u64> = rows.iter().map(|r| r.device_ptr()).collect();
        let mut roots = vec![0u64; 4 * world];
        let mut root = [0u64; 4];
        let (mut subtree, mut top) = (ptr::null_mut(), ptr::null_mut());
        hal.check_prover(sys::ceno_dist_commit_traces_mmcs(hal.ctx, comm, n as i32, log_rows.as_ptr(), widths.as_ptr(), local_cols_dev.as_ptr(),
                                                           log_blowup as i32, stream.raw(), outs.as_ptr(), &mut subtree, &mut top,
                                                           roots.as_mut_ptr(), root.as_mut_ptr()))?;
        let subtree_roots = roots.chunks(4).map(|c| [c[0], c[1], c[2], c[3]]).collect();
        Ok(Self { hal: hal.clone(), root, subtree_roots, rows, subtree, top })
    }
}
impl Drop for ShardedMmcsCommitment {
    fn drop(&mut self) {
        unsafe {
            sys::ceno_hip_merkle_free(self.hal.ctx, self.subtree);
            if !self.top.is_null() {
                sys::ceno_hip_merkle_free(self.hal.ctx, self.top);
            }
        }
    }
}
impl Drop for ShardedCommitment {
    fn drop(&mut self) {
        unsafe { sys::ceno_hip_merkle_free(self.hal.ctx, self.subtree) };
    }
}
