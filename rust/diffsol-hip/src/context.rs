//! `Context` (crates/diffsol-la/src/context/mod.rs:20-68): device + stream + `nbatch`, with the broadcast compatibility rule of the trait's
//! default `assert_compatible_nbatch`.  One `dsh_ctx` is shared (`Arc`) by every clone: all users issue work on its one in-order stream, which is
//! what makes the stream-ordered allocation cache of the library safe.
//!
//! THREADING CONTRACT (include/diffsol_hip.h): `Vector` requires `Clone + Send` (diffsol-la/src/vector/mod.rs:163-177), `Matrix` requires
//! `Clone + Send + 'static` (matrix/mod.rs:169-170) and `OdeSolverState` requires `Send` (diffsol/src/ode_solver/state.rs:880), so `HipContext` — a field
//! of every `HipVec` / `HipMat` / `HipIndex` — must be `Send`: the handle is an `Arc<CtxHandle>` and `CtxHandle` / `DeviceBuf` / `HipLU` carry
//! `unsafe impl Send`.  The invariant those impls rest on: a context and every object created from it may be MOVED to another thread (a solver built on
//! one thread and run on another, a problem handed to a worker), but they are used by ONE THREAD AT A TIME — the C context keeps its reduction records,
//! scratch buffers and allocation cache without locks.  This is the contract of the reference's CUDA backend too: `CudaContext` is an `Arc<CudaStream>`
//! (context/cuda.rs:41-44) and every `CudaVec` operation enqueues on that one stream without further synchronisation.  `Arc<T>: Send` needs `T: Sync`
//! as well; `CtxHandle` exposes no `&self` operation except through the FFI calls the invariant above serialises.  Ensembles on several GPUs use one
//! process per device.  The current HIP device is per-thread state: `ptr()` re-binds it (`dsh_ctx_bind_thread`) when the calling thread
//! differs from the one that used the context last.
use crate::error::{check, last_error};
use crate::ffi;
use diffsol_la::error::LaError;
use diffsol_la::Context;
use std::os::raw::c_void;
use std::ptr;
use std::sync::Arc;

#[derive(Debug)]
pub(crate) struct CtxHandle(pub(crate) *mut ffi::dsh_ctx);
// SAFETY: see the threading contract above — moved between threads, used by one thread at a time; the C library holds no thread-affine state in a context
// (its last-error string is thread-local, the HIP device binding is refreshed by `HipContext::ptr`).
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

thread_local! {
    /// the context this thread bound its HIP device for last (null: none yet)
    static BOUND: std::cell::Cell<*mut ffi::dsh_ctx> = const { std::cell::Cell::new(ptr::null_mut()) };
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
        BOUND.with(|b| b.set(h));
        Ok(Self { raw: Arc::new(CtxHandle(h)), nbatch: 1 })
    }
    /// The C handle for an FFI call.  A thread that has not used this context last re-binds its current HIP device first (the device is per-thread state).
    pub(crate) fn ptr(&self) -> *mut ffi::dsh_ctx {
        let h = self.raw.0;
        BOUND.with(|b| {
            if b.get() != h {
                check(unsafe { ffi::dsh_ctx_bind_thread(h) }, "dsh_ctx_bind_thread");
                b.set(h);
            }
        });
        h
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
    /// thread: two threads that each build a problem from `Default` never share a context, so the one-thread-at-a-time contract holds without the caller's help.
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
