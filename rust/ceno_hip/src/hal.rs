//! Process-global HAL handle, streams and the thread-bound stream of the chip scheduler.
//! Reference: `CUDA_HAL` lazy global (`gkr_iop/src/gpu/mod.rs:53-66`), `bind_thread_stream` / `get_thread_stream`
//! (`:87-154`), lanes = streams of the scheduler (`ceno_zkvm/src/scheme/scheduler.rs:73-85`).
use std::{cell::RefCell, ptr, sync::Arc};

use ceno_hip_sys as sys;
use once_cell::sync::Lazy;

use crate::error::{from_status, last_error, last_prover_error, HipError, Result};

/// Owner of the device context (memory pool, default stream).  `Send + Sync`: the library is re-entrant across streams.
pub struct HipHal {
    pub(crate) ctx: *mut sys::ceno_hip_ctx,
    pub device: i32,
}
unsafe impl Send for HipHal {}
unsafe impl Sync for HipHal {}

impl HipHal {
    /// `ceno_hip_init(device, pool_bytes)`; `pool_bytes = 0` = unlimited pool
    pub fn new(device: i32, pool_bytes: usize) -> Result<Self> {
        let mut ctx = ptr::null_mut();
        let rc = unsafe { sys::ceno_hip_init(device, pool_bytes, &mut ctx) };
        if rc != 0 {
            return Err(from_status(rc, unsafe { last_error(ptr::null_mut()) }));
        }
        Ok(Self { ctx, device })
    }
    pub fn raw(&self) -> *mut sys::ceno_hip_ctx {
        self.ctx
    }
    /// map a device-library status
    pub fn check(&self, rc: i32) -> Result<()> {
        if rc == 0 { Ok(()) } else { Err(from_status(rc, unsafe { last_error(self.ctx) })) }
    }
    /// map a host-layer (`ceno_prover_*`) status
    pub fn check_prover(&self, rc: i32) -> Result<()> {
        if rc == 0 { Ok(()) } else { Err(from_status(rc, unsafe { last_prover_error() })) }
    }
    /// `ensure_context` (`gkr_iop/src/gpu/mod.rs:91-92`): a fresh thread starts on device 0
    pub fn ensure_context(&self) -> Result<()> {
        self.check(unsafe { sys::ceno_hip_make_current(self.ctx) })
    }
    pub fn create_stream(self: &Arc<Self>) -> Result<HipStream> {
        let mut s = ptr::null_mut();
        self.check(unsafe { sys::ceno_hip_stream_create(self.ctx, &mut s) })?;
        Ok(HipStream { hal: self.clone(), raw: s })
    }
    /// stream of proving lane `lane` (consecutive lanes land on different hardware queues)
    pub fn create_lane_stream(self: &Arc<Self>, lane: i32) -> Result<HipStream> {
        let mut s = ptr::null_mut();
        self.check(unsafe { sys::ceno_hip_stream_create_lane(self.ctx, lane, &mut s) })?;
        Ok(HipStream { hal: self.clone(), raw: s })
    }
    /// (free, total, pool_used, pool_cached) — `get_cuda_mem_info`
    pub fn mem_info(&self) -> Result<(usize, usize, usize, usize)> {
        let (mut f, mut t, mut u, mut c) = (0usize, 0usize, 0usize, 0usize);
        self.check(unsafe { sys::ceno_hip_mem_info(self.ctx, &mut f, &mut t, &mut u, &mut c) })?;
        Ok((f, t, u, c))
    }
    /// `trim_mem_pool` (`e2e.rs:3331-3334`)
    pub fn trim_mem_pool(&self) -> Result<()> {
        self.check(unsafe { sys::ceno_hip_mem_trim(self.ctx) })
    }
    /// `mem_pool.try_book_capacity` / `unbook_capacity` / `get_booked_total` (`scheduler.rs:342-347,390,622-652`)
    pub fn try_book_capacity(&self, bytes: usize) -> bool {
        unsafe { sys::ceno_hip_mem_book(self.ctx, bytes) == 0 }
    }
    pub fn unbook_capacity(&self, bytes: usize) {
        unsafe { sys::ceno_hip_mem_unbook(self.ctx, bytes) };
    }
    pub fn booked_total(&self) -> usize {
        unsafe { sys::ceno_hip_mem_booked(self.ctx) }
    }
    /// Poseidon2-Goldilocks parameter table for the commit / open path (placeholders until this is called: PARITY UNPINNED)
    pub fn poseidon2_set_constants(&self, external_rc: &[u64; 64], internal_rc: &[u64; 22], internal_diag: &[u64; 8]) -> Result<()> {
        self.check(unsafe { sys::ceno_hip_poseidon2_set_constants(self.ctx, external_rc.as_ptr(), internal_rc.as_ptr(), internal_diag.as_ptr()) })?;
        let rc = unsafe { sys::ceno_transcript_poseidon2_set_constants(external_rc.as_ptr(), internal_rc.as_ptr(), internal_diag.as_ptr()) };
        self.check_prover(rc)
    }
}
impl Drop for HipHal {
    fn drop(&mut self) {
        unsafe { sys::ceno_hip_destroy(self.ctx) }
    }
}

pub struct HipStream {
    hal: Arc<HipHal>,
    raw: sys::ceno_hip_stream,
}
unsafe impl Send for HipStream {}
unsafe impl Sync for HipStream {}
impl HipStream {
    pub fn raw(&self) -> sys::ceno_hip_stream {
        self.raw
    }
    pub fn synchronize(&self) -> Result<()> {
        self.hal.check(unsafe { sys::ceno_hip_stream_sync(self.hal.ctx, self.raw) })
    }
}
impl Drop for HipStream {
    fn drop(&mut self) {
        unsafe { sys::ceno_hip_stream_destroy(self.hal.ctx, self.raw) };
    }
}

/// device id: `CENO_GPU_DEVICE_ID` like the reference (`get_ceno_gpu_device_id(0)`), else `LOCAL_RANK`, else 0
fn device_id() -> i32 {
    for var in ["CENO_GPU_DEVICE_ID", "LOCAL_RANK"] {
        if let Ok(v) = std::env::var(var) {
            if let Ok(d) = v.parse() {
                return d;
            }
        }
    }
    0
}

static HIP_HAL: Lazy<std::result::Result<Arc<HipHal>, HipError>> = Lazy::new(|| HipHal::new(device_id(), 0).map(Arc::new));

pub fn get_hip_hal() -> std::result::Result<Arc<HipHal>, String> {
    HIP_HAL.as_ref().map(|h| h.clone()).map_err(|e| format!("HAL not available: {e}"))
}

thread_local! {
    static THREAD_STREAM: RefCell<Option<Arc<HipStream>>> = const { RefCell::new(None) };
}

/// Bind a stream to the current thread for all GPU work issued from it; also makes the device current for the thread.
pub fn bind_thread_stream(stream: Arc<HipStream>) -> ThreadStreamGuard {
    let hal = get_hip_hal().expect("Failed to get HIP HAL");
    hal.ensure_context().expect("hipSetDevice");
    // the pool tags freed blocks with the thread's stream and hands them to another lane only once that stream has drained
    // (include/ceno_hip.h "Memory"): tell the library which stream this thread works on before it allocates anything
    unsafe { sys::ceno_hip_stream_bind(hal.ctx, stream.raw()) };
    THREAD_STREAM.with(|c| *c.borrow_mut() = Some(stream));
    ThreadStreamGuard
}
pub struct ThreadStreamGuard;
impl Drop for ThreadStreamGuard {
    fn drop(&mut self) {
        THREAD_STREAM.with(|c| *c.borrow_mut() = None);
        // the C side keeps the last bound stream per thread as the tag of freed blocks: point it back at the context's default
        // stream, or a later free on this thread would be tagged with a stream that may since have been destroyed elsewhere
        if let Ok(hal) = get_hip_hal() {
            unsafe { sys::ceno_hip_stream_bind(hal.ctx, ptr::null_mut()) };
        }
    }
}
/// the thread's stream, or `None` = the context's default stream (a null `ceno_hip_stream`)
pub fn get_thread_stream() -> Option<Arc<HipStream>> {
    THREAD_STREAM.with(|c| c.borrow().clone())
}
pub(crate) fn raw_stream(s: Option<&HipStream>) -> sys::ceno_hip_stream {
    s.map_or(ptr::null_mut(), |s| s.raw())
}
