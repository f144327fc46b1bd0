//! `Context` (crates/diffsol-la/src/context/mod.rs:20-68): device + stream + `nbatch`, with the broadcast compatibility rule of the trait's
//! default `assert_compatible_nbatch`.  One `dsh_ctx` is shared (`Arc`) by every clone: all users issue work on its one in-order stream, which is
//! what makes the stream-ordered allocation cache of the library safe.
//!
//! THREADING CONTRACT (include/diffsol_hip.h): `Vector` requires `Clone + Send` (diffsol-la/src/vector/mod.rs:163-177), `Matrix` requires
//! `Clone + Send + 'static` (matrix/mod.rs:169-170) and `OdeSolverState` requires `Send` (diffsol/src/ode_solver/state.rs:880), so `HipContext` — a field
//! of every `HipVec` / `HipMat` / `HipIndex` — must be `Send`: the handle is an `Arc<CtxHandle>` and `CtxHandle` / `DeviceBuf` / `HipLU` carry
//! `unsafe impl Send`.  Safe Rust can clone a context or a vector, move the clone to another thread and use both at once; the soundness of those impls
//! therefore cannot rest on a documented "one thread at a time" rule.  It rests on the C library: every `extern "C"` entry point that takes a `dsh_ctx`
//! (or a `dsh_lu` / `dsh_dist` made from one) holds the context's recursive mutex for the duration of the call (`DSH_ENTER`, csrc/dsh_internal.hpp), so the
//! record ring, the scratch buffers and the allocation cache are never touched by two threads at once, and concurrent callers are serialised call by
//! call onto the context's one in-order stream — the same position the reference's `Arc<CudaStream>` (context/cuda.rs:41-44) puts its `CudaVec`
//! operations in.  The current HIP device is per-thread state: the same guard re-binds it when the calling thread is not the one that used the context
//! last (keyed on the thread, inside the context object itself — a context destroyed and re-created at the same address starts with its creator's
//! thread id, so there is no stale pointer comparison on this side).  Ensembles on several GPUs use one process per device.
use crate::error::{check, last_error};
use crate::ffi;
use diffsol_la::error::LaError;
use diffsol_la::Context;
use std::os::raw::c_void;
use std::ptr;
use std::sync::Arc;

#[derive(Debug)]
pub(crate) struct CtxHandle(pub(crate) *mut ffi::dsh_ctx);
// SAFETY: see the threading contract above — every FFI call on the handle is serialised by the context's own lock inside the C library and re-binds the HIP
// device of the calling thread; the library holds no other thread-affine state in a context (its last-error string is thread-local).
unsafe impl Send for CtxHandle {}
unsafe impl Sync for CtxHandle {}
impl Drop for CtxHandle {
    fn drop(&mut self) {
        unsafe { ffi::dsh_ctx_destroy(self.0) }
    }
}

#[derive(Clone, Debug)]
pub struct HipContext {
    pub(crate) raw: Arc<CtxHandle>,
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
        Ok(Self { raw: Arc::new(CtxHandle(h)), nbatch: 1 })
    }
    /// The C handle for an FFI call.  The library locks the context and re-binds the calling thread's HIP device inside every entry point (threading contract above).
    pub(crate) fn ptr(&self) -> *mut ffi::dsh_ctx {
        self.raw.0
    }
    /// Block until everything enqueued on the context's stream has finished.
    pub fn synchronize(&self) {
        check(unsafe { ffi::dsh_ctx_sync(self.ptr()) }, "dsh_ctx_sync");
    }
    /// Two contexts are the same device context if they share the handle (the `nbatch` may differ: broadcast operands).
    pub(crate) fn same_device_context(&self, other: &Self) -> bool {
        Arc::ptr_eq(&self.raw, &other.raw)
    }
}

impl Default for HipContext {
    /// Device 0 — what `Matrix::is_sparse()` and the builders reach for.  Panics without a HIP device: there is no CPU fallback.  One default context per
    /// thread: two threads that each build a problem from `Default` never share a context, so they do not contend for one stream either.
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
