//! Status codes of the C ABI -> Rust.  The LA layer of the reference panics on shape / nbatch mismatches (its tests use `#[should_panic]`,
//! vector/mod.rs:723-726, :1102-1135); `LinearSolver::solve_in_place` returns `LaError::LinearSolverError` (error.rs:24-42).
use crate::ffi;
use diffsol_la::error::{LaError, LinearSolverError};
use std::ffi::CStr;
use std::os::raw::c_int;

/// Text of the thread-local last error of libdiffsol_hip.so (`dsh_last_error`, modelled on diffsol-c/src/error_c.rs:12-121).
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::dsh_last_error()) }.to_string_lossy().into_owned()
}

/// Panic on a negative status: shape errors are programming errors in the reference's LA layer as well.
#[track_caller]
pub fn check(rc: c_int, what: &str) {
    if rc < 0 {
        panic!("{what}: {} (status {rc})", last_error());
    }
}

/// Status -> `LaError` for the calls whose reference counterpart returns a `Result`.
pub fn to_la_error(rc: c_int) -> Result<(), LaError> {
    match rc {
        x if x >= 0 => Ok(()),
        ffi::DSH_E_SINGULAR => Err(LaError::from(LinearSolverError::LuSolveFailed)),
        ffi::DSH_E_NOT_SETUP => Err(LaError::from(LinearSolverError::LuNotInitialized)),
        _ => Err(LaError::Other(last_error())),
    }
}
