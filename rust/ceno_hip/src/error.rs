//! Status codes of the C ABI as a Rust error (reference: `HalError` mapped to `BackendError::CircuitError` /
//! `ZKVMError::BackendError`, `ceno_zkvm/src/scheme/gpu/mod.rs:348-352`, `scheme/gpu/util.rs` `hal_to_backend_error`).
use std::{ffi::CStr, fmt};

use ceno_hip_sys as sys;

#[derive(Debug, Clone)]
pub enum HipError {
    Invalid(String),
    Hip(String),
    OutOfMemory(String),
    State(String),
    Unsupported(String),
    Unknown(i32, String),
}

pub type Result<T> = std::result::Result<T, HipError>;

impl fmt::Display for HipError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            HipError::Invalid(m) => write!(f, "ceno_hip: invalid argument: {m}"),
            HipError::Hip(m) => write!(f, "ceno_hip: HIP runtime error: {m}"),
            HipError::OutOfMemory(m) => write!(f, "ceno_hip: out of device memory: {m}"),
            HipError::State(m) => write!(f, "ceno_hip: call out of order: {m}"),
            HipError::Unsupported(m) => write!(f, "ceno_hip: unsupported: {m}"),
            HipError::Unknown(c, m) => write!(f, "ceno_hip: status {c}: {m}"),
        }
    }
}
impl std::error::Error for HipError {}

pub(crate) fn from_status(code: i32, msg: String) -> HipError {
    match code {
        sys::CENO_HIP_ERR_INVALID => HipError::Invalid(msg),
        sys::CENO_HIP_ERR_HIP => HipError::Hip(msg),
        sys::CENO_HIP_ERR_OOM => HipError::OutOfMemory(msg),
        sys::CENO_HIP_ERR_STATE => HipError::State(msg),
        sys::CENO_HIP_ERR_UNSUPPORTED => HipError::Unsupported(msg),
        c => HipError::Unknown(c, msg),
    }
}

/// last error text of the device library (`ctx` may be null: last init error)
pub(crate) unsafe fn last_error(ctx: *mut sys::ceno_hip_ctx) -> String {
    let p = sys::ceno_hip_last_error(ctx);
    if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
}
/// last error text of the host layer (`libceno_prover.so`)
pub(crate) unsafe fn last_prover_error() -> String {
    let p = sys::ceno_prover_last_error();
    if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
}
