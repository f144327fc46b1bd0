//! `Context` (crates/diffsol-la/src/context/mod.rs:20-68): device + stream + `nbatch`, with the broadcast compatibility rule of the trait's
//! default `assert_compatible_nbatch`.  One `dsh_ctx` is shared (Rc) by every clone: all users issue work on its one in-order stream, which is
//! what makes the stream-ordered allocation cache of the library safe.
use crate::error::{check, last_error};
use crate::ffi;
use diffsol_la::error::LaError;
use diffsol_la::Context;
use std::os::raw::c_void;
use std::ptr;
use std::rc::Rc;

#[derive(Debug)]
pub(crate) struct CtxHandle(pub(crate) *mut ffi::dsh_ctx);
// ONE THREAD PER CONTEXT.  The C context is not thread-safe (record ring and sequence numbers, scratch buffers, the stream-ordered allocation cache, the
// last-error string): the handle is deliberately neither `Send` nor `Sync` (a raw pointer), so `HipContext`, and every `HipVec` / `HipMat` / `HipLU` that
// holds a clone of it, stay on the thread that created them.  diffsol's Context / Vector / Matrix / LinearSolver traits do not ask for `Send`.  Ensembles on
// several GPUs use one process (or one thread with its own context) per device — DESIGN.md 6.
impl Drop for CtxHandle {
    fn drop(&mut self) {
        unsafe { ffi::dsh_ctx_destroy(self.0) }
    }
}

#[derive(Clone, Debug)]
pub struct HipContext {
    pub(crate) raw: Rc<CtxHandle>,
    pub(crate) nbatch: usize,
}

impl HipContext {
    /// Context on HIP device `device` with its own non-blocking stream (replaces `CudaContext::new`, context/cuda.rs:48-68).
    pub fn new(device: i32) -> Result<Self, LaError> {
        Self::with_stream(device, ptr::null_mut())
    }
    /// Borrow an existing `hipStream_t` (e.g. the stream of the embedding application).
    pub fn with_stream(device: i32, stream: *mut c_void) -> Result<Self, LaError> {
        let mut h: *mut ffi::dsh_ctx = ptr::null_mut();
        let rc = unsafe { ffi::dsh_ctx_create(device, stream, &mut h) };
        if rc < 0 {
            return Err(LaError::Other(format!("dsh_ctx_create: {}", last_error())));
        }
        Ok(Self { raw: Rc::new(CtxHandle(h)), nbatch: 1 })
    }
    pub(crate) fn ptr(&self) -> *mut ffi::dsh_ctx {
        self.raw.0
    }
    /// Block until everything enqueued on the context's stream has finished.
    pub fn synchronize(&self) {
        check(unsafe { ffi::dsh_ctx_sync(self.ptr()) }, "dsh_ctx_sync");
    }
    /// Two contexts are the same device context if they share the handle (the `nbatch` may differ: broadcast operands).
    pub(crate) fn same_device_context(&self, other: &Self) -> bool {
        Rc::ptr_eq(&self.raw, &other.raw)
    }
}

impl Default for HipContext {
    /// Device 0 — what `Matrix::is_sparse()` and the builders reach for.  Panics without a HIP device: there is no CPU fallback.
    fn default() -> Self {
        thread_local! {
            static DEFAULT: HipContext = HipContext::new(0).expect("diffsol-hip needs a HIP device (no CPU fallback)");
        }
        DEFAULT.with(|c| c.clone())
    }
}

impl Context for HipContext {
    fn nbatch(&self) -> usize {
        self.nbatch
    }
    fn clone_with_nbatch(&self, nbatch: usize) -> Result<Self, LaError> {
        if nbatch == 0 {
            return Err(LaError::Other("nbatch must be at least 1".into()));
        }
        Ok(Self { raw: self.raw.clone(), nbatch })
    }
}
