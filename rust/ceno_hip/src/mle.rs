//! Device multilinear polynomial: dense evaluation table, base (8 B/elem) or extension (16 B/elem), index bit k <->
//! variable k.  Reference: `MultilinearExtensionGpu` owning a `GpuPolynomial(Ext)` (`gkr_iop/src/gpu/mod.rs:157-370,437`).
use std::{ptr, sync::Arc};

use ceno_hip_sys as sys;

use crate::{
    error::Result,
    hal::{raw_stream, HipHal, HipStream},
    ExtWords,
};

/// Owning handle (`Drop` frees the pool block) or borrowed view (keeps its parent alive, `as_view_chunk`
/// `gkr_iop/src/gpu/mod.rs:244-253`).
pub struct HipMle {
    hal: Arc<HipHal>,
    raw: *mut sys::ceno_hip_mle,
    _parent: Option<Arc<HipMle>>,
}
unsafe impl Send for HipMle {}
unsafe impl Sync for HipMle {}

impl HipMle {
    pub(crate) fn from_raw(hal: &Arc<HipHal>, raw: *mut sys::ceno_hip_mle) -> Self {
        Self { hal: hal.clone(), raw, _parent: None }
    }
    pub fn raw(&self) -> *mut sys::ceno_hip_mle {
        self.raw
    }
    pub fn hal(&self) -> &Arc<HipHal> {
        &self.hal
    }
    /// uninitialised table (`alloc_elems_on_device` / `alloc_ext_elems_on_device`)
    pub fn alloc(hal: &Arc<HipHal>, num_vars: usize, is_ext: bool) -> Result<Self> {
        let mut m = ptr::null_mut();
        hal.check(unsafe { sys::ceno_hip_mle_alloc(hal.ctx, num_vars as i32, is_ext as i32, &mut m) })?;
        Ok(Self::from_raw(hal, m))
    }
    /// upload canonical words (`alloc_elems_from_host`); `words.len()` = 2^num_vars (x2 for extension tables)
    pub fn from_host(hal: &Arc<HipHal>, words: &[u64], num_vars: usize, is_ext: bool, stream: Option<&HipStream>) -> Result<Self> {
        assert_eq!(words.len(), (1usize << num_vars) * if is_ext { 2 } else { 1 });
        let mut m = ptr::null_mut();
        hal.check(unsafe { sys::ceno_hip_mle_upload(hal.ctx, words.as_ptr(), num_vars as i32, is_ext as i32, raw_stream(stream), &mut m) })?;
        Ok(Self::from_raw(hal, m))
    }
    /// borrow device memory owned by somebody else (e.g. a column of a committed trace)
    ///
    /// # Safety
    /// `device_ptr` must stay valid (and 16-byte aligned) for the life of the handle.
    pub unsafe fn wrap(hal: &Arc<HipHal>, device_ptr: *mut u64, num_vars: usize, is_ext: bool) -> Result<Self> {
        let mut m = ptr::null_mut();
        hal.check(sys::ceno_hip_mle_wrap(hal.ctx, device_ptr, num_vars as i32, is_ext as i32, &mut m))?;
        Ok(Self::from_raw(hal, m))
    }
    pub fn num_vars(&self) -> usize {
        unsafe { sys::ceno_hip_mle_num_vars(self.raw) as usize }
    }
    pub fn is_ext(&self) -> bool {
        unsafe { sys::ceno_hip_mle_is_ext(self.raw) != 0 }
    }
    pub fn evaluations_len(&self) -> usize {
        1 << self.num_vars()
    }
    pub fn device_ptr(&self) -> *mut u64 {
        unsafe { sys::ceno_hip_mle_device_ptr(self.raw) }
    }
    /// `Buffer::to_cpu_vec`
    pub fn to_host(&self, stream: Option<&HipStream>) -> Result<Vec<u64>> {
        let mut v = vec![0u64; self.evaluations_len() * if self.is_ext() { 2 } else { 1 }];
        self.hal.check(unsafe { sys::ceno_hip_mle_download(self.hal.ctx, self.raw, v.as_mut_ptr(), raw_stream(stream)) })?;
        Ok(v)
    }
    /// `MultilinearExtension::evaluate(point)` on the device (one read-only pass)
    pub fn evaluate(&self, point: &[ExtWords], stream: Option<&HipStream>) -> Result<ExtWords> {
        assert_eq!(point.len(), self.num_vars());
        let mut out = [0u64; 2];
        self.hal.check(unsafe { sys::ceno_hip_mle_evaluate(self.hal.ctx, self.raw, point.as_ptr() as *const u64, out.as_mut_ptr(), raw_stream(stream)) })?;
        Ok(out)
    }
    /// `fix_variables` (low variables first): always an extension table
    pub fn fix_variables(&self, point: &[ExtWords], stream: Option<&HipStream>) -> Result<Self> {
        let mut m = ptr::null_mut();
        self.hal.check(unsafe {
            sys::ceno_hip_mle_fix_variables(self.hal.ctx, self.raw, point.as_ptr() as *const u64, point.len() as i32, raw_stream(stream), &mut m)
        })?;
        Ok(Self::from_raw(&self.hal, m))
    }
    /// borrowed view of chunk `chunk` of `2^sub_vars` entries (`as_view_chunk`)
    pub fn view_chunk(self: &Arc<Self>, sub_vars: usize, chunk: usize) -> Result<Self> {
        let mut m = ptr::null_mut();
        self.hal.check(unsafe { sys::ceno_hip_mle_view_chunk(self.hal.ctx, self.raw, sub_vars as i32, chunk, &mut m) })?;
        Ok(Self { hal: self.hal.clone(), raw: m, _parent: Some(self.clone()) })
    }
    /// even / odd entries as a new table (`filter_mle_even_odd_batch`, `scheme/gpu/util.rs:186-266`)
    pub fn filter_even_odd(&self, odd: bool, stream: Option<&HipStream>) -> Result<Self> {
        let mut m = ptr::null_mut();
        self.hal.check(unsafe { sys::ceno_hip_mle_filter_even_odd(self.hal.ctx, self.raw, odd as i32, raw_stream(stream), &mut m) })?;
        Ok(Self::from_raw(&self.hal, m))
    }
    /// eq(x, point) * scalar (`build_eq_x_r_vec`)
    pub fn eq(hal: &Arc<HipHal>, point: &[ExtWords], scalar: Option<ExtWords>, stream: Option<&HipStream>) -> Result<Self> {
        let mut m = ptr::null_mut();
        let sc = scalar.as_ref().map_or(ptr::null(), |s| s.as_ptr());
        hal.check(unsafe { sys::ceno_hip_eq_build(hal.ctx, point.as_ptr() as *const u64, point.len() as i32, sc, raw_stream(stream), &mut m) })?;
        Ok(Self::from_raw(hal, m))
    }
    /// `SelectorType::compute` (`gkr_iop/src/selector.rs:131-245`): `kind` is a `CENO_HIP_SEL_*` constant
    #[allow(clippy::too_many_arguments)]
    pub fn selector(hal: &Arc<HipHal>, kind: i32, point: &[ExtWords], offset: usize, num_instances: usize, sparse_indices: &[u32],
                    sparse_num_vars: usize, stream: Option<&HipStream>) -> Result<Self> {
        let mut m = ptr::null_mut();
        hal.check(unsafe {
            sys::ceno_hip_selector_build(hal.ctx, kind, point.as_ptr() as *const u64, point.len() as i32, offset, num_instances,
                                         sparse_indices.as_ptr(), sparse_indices.len() as i32, sparse_num_vars as i32, raw_stream(stream), &mut m)
        })?;
        Ok(Self::from_raw(hal, m))
    }
    /// `rotation_next_base_mle_gpu` / `rotation_selector_gpu` (`gkr_iop/src/gkr/layer/gpu/utils.rs:231-336`)
    pub fn rotation_next_base(&self, cyclic_group_log2: usize, stream: Option<&HipStream>) -> Result<Self> {
        let mut m = ptr::null_mut();
        self.hal.check(unsafe { sys::ceno_hip_rotation_next_base_mle(self.hal.ctx, self.raw, cyclic_group_log2 as i32, raw_stream(stream), &mut m) })?;
        Ok(Self::from_raw(&self.hal, m))
    }
    pub fn rotation_selector(hal: &Arc<HipHal>, point: &[ExtWords], cyclic_subgroup_size: usize, cyclic_group_log2: usize,
                             stream: Option<&HipStream>) -> Result<Self> {
        let mut m = ptr::null_mut();
        hal.check(unsafe {
            sys::ceno_hip_rotation_selector_build(hal.ctx, point.as_ptr() as *const u64, point.len() as i32, cyclic_subgroup_size as i32,
                                                  cyclic_group_log2 as i32, raw_stream(stream), &mut m)
        })?;
        Ok(Self::from_raw(hal, m))
    }
}
impl Drop for HipMle {
    fn drop(&mut self) {
        // frees the pool block of an owning handle; a wrapped / view handle only releases the small host record
        unsafe { sys::ceno_hip_mle_free(self.hal.ctx, self.raw) };
    }
}
impl std::fmt::Debug for HipMle {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        f.debug_struct("HipMle").field("num_vars", &self.num_vars()).field("is_ext", &self.is_ext()).finish()
    }
}

/// `wit_infer_by_monomial_expr` (`gkr_iop/src/gpu/mod.rs:599-609`): outs[o][x] = sum_t coeff_t * prod_j mles[j][x]
pub fn wit_infer(hal: &Arc<HipHal>, mles: &[&HipMle], term_coeffs: &[ExtWords], terms: &[Vec<usize>], out_terms: &[std::ops::Range<usize>],
                 num_vars: usize, stream: Option<&HipStream>) -> Result<Vec<HipMle>> {
    let handles: Vec<*mut sys::ceno_hip_mle> = mles.iter().map(|m| m.raw()).collect();
    let (toff, tidx) = crate::sumcheck::csr(terms);
    let mut ooff = vec![0u32];
    for r in out_terms {
        assert_eq!(r.start as u32, *ooff.last().unwrap(), "terms must be listed output by output");
        ooff.push(r.end as u32);
    }
    let mut outs = vec![ptr::null_mut(); out_terms.len()];
    hal.check(unsafe {
        sys::ceno_hip_wit_infer(hal.ctx, handles.as_ptr(), handles.len() as i32, term_coeffs.as_ptr() as *const u64, toff.as_ptr(), tidx.as_ptr(),
                                terms.len() as i32, ooff.as_ptr(), out_terms.len() as i32, num_vars as i32, raw_stream(stream), outs.as_mut_ptr())
    })?;
    Ok(outs.into_iter().map(|m| HipMle::from_raw(hal, m)).collect())
}
