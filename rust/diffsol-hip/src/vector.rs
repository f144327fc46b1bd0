//! `HipVec` / `HipVecRef` / `HipVecMut` / `HipIndex`: the `Vector`, `VectorView`, `VectorViewMut`, `VectorIndex`, `DefaultDenseMatrix` implementations
//! (trait definitions: crates/diffsol-la/src/vector/mod.rs:20-377; what they replace: vector/cuda.rs:127-1490 and the 26 kernels of
//! crates/diffsol-la/src/cuda_kernels/*.cu).  Every method is one `dsh_vec_*` call of include/diffsol_hip.h.
//!
//! Device layout: batch-fastest — element `i` of member `b` lives at `ptr[i * nbatch + b]` — so column `j` of a batched matrix is itself a
//! contiguous batched vector and a column view is just `(ptr + j * nrows * nbatch, nrows)`.  The host-facing layout (`from_vec`, `clone_as_vec`)
//! stays the reference's batch-major one; `dsh_vec_upload` / `dsh_vec_download` transpose.  A single-member view (`get_batch`) is the
//! stride-`nbatch` slice `ptr[i * nbatch + b]`; arithmetic on such a view goes through a contiguous temporary (`dsh_vec_extract_batch` /
//! `dsh_vec_insert_batch`), `get_index` reads it directly.
use crate::context::HipContext;
use crate::error::check;
use crate::ffi;
use crate::matrix::HipMat;
use diffsol_la::scalar::Scale;
use diffsol_la::{Context, DefaultDenseMatrix, IndexType, Vector, VectorCommon, VectorIndex, VectorView, VectorViewMut};
use std::marker::PhantomData;
use std::ops::{Add, AddAssign, Div, Mul, MulAssign, Sub, SubAssign};
use std::os::raw::c_void;
use std::ptr;

// ------------------------------------------------------------------ owned device memory
/// Device allocation owned by a vector / matrix / index (from the context's stream-ordered cache: `dsh_malloc` / `dsh_free`).
#[derive(Debug)]
pub struct DeviceBuf {
    pub(crate) ptr: *mut c_void,
    pub(crate) nbytes: usize,
    pub(crate) ctx: HipContext,
}
// SAFETY: the allocation belongs to the context it was made from; it moves with it under the contract of context.rs (every call on it is serialised by its context's lock).
unsafe impl Send for DeviceBuf {}
impl DeviceBuf {
    pub(crate) fn new(nbytes: usize, zero: bool, ctx: &HipContext) -> Self {
        let mut p: *mut c_void = ptr::null_mut();
        check(unsafe { ffi::dsh_malloc(ctx.ptr(), nbytes as i64, zero as i32, &mut p) }, "dsh_malloc");
        Self { ptr: p, nbytes, ctx: ctx.clone() }
    }
    pub(crate) fn f64(&self) -> *mut f64 {
        self.ptr as *mut f64
    }
}
impl Drop for DeviceBuf {
    fn drop(&mut self) {
        unsafe { ffi::dsh_free(self.ctx.ptr(), self.ptr) };
    }
}
impl Clone for DeviceBuf {
    fn clone(&self) -> Self {
        let out = DeviceBuf::new(self.nbytes, false, &self.ctx);
        check(unsafe { ffi::dsh_d2d(self.ctx.ptr(), out.ptr, self.ptr, self.nbytes as i64) }, "dsh_d2d");
        out
    }
}

// ------------------------------------------------------------------ types
/// Batched dense vector on the GPU: `nstates` values for each of `context.nbatch()` members.
#[derive(Debug, Clone)]
pub struct HipVec {
    pub(crate) buf: DeviceBuf,
    pub(crate) nstates: IndexType,
    pub(crate) context: HipContext,
}

/// Non-owning description of device data: a whole vector, a matrix column, or (with `member = Some((parent_nbatch, b))`) one member of a batched
/// vector.  `context.nbatch()` is the view's own batch count (1 for a member view).
#[derive(Debug, Clone)]
pub struct RawView {
    pub(crate) ptr: *mut f64,
    pub(crate) nstates: IndexType,
    pub(crate) context: HipContext,
    pub(crate) member: Option<(usize, usize)>,
}

/// Immutable view (`Vector::as_view`, `DenseMatrix::column`, `Vector::get_batch`).
#[derive(Debug)]
pub struct HipVecRef<'a> {
    pub(crate) raw: RawView,
    pub(crate) _life: PhantomData<&'a f64>,
}
/// Mutable view (`Vector::as_view_mut`, `DenseMatrix::column_mut`, `Vector::get_batch_mut`).
#[derive(Debug)]
pub struct HipVecMut<'a> {
    pub(crate) raw: RawView,
    pub(crate) _life: PhantomData<&'a mut f64>,
}

/// Indices (int32 on the device, `usize` on the host: vector/cuda.rs:135-138), shared by all members.
#[derive(Debug, Clone)]
pub struct HipIndex {
    pub(crate) buf: DeviceBuf,
    pub(crate) len: IndexType,
    pub(crate) context: HipContext,
}

impl DefaultDenseMatrix for HipVec {
    type M = HipMat;
}

// ------------------------------------------------------------------ operand plumbing
/// Anything that can appear on the right-hand side of a vector operation.
pub(crate) trait Operand {
    fn raw(&self) -> RawView;
}
impl Operand for HipVec {
    fn raw(&self) -> RawView {
        RawView { ptr: self.buf.f64(), nstates: self.nstates, context: self.context.clone(), member: None }
    }
}
impl Operand for &HipVec {
    fn raw(&self) -> RawView {
        (**self).raw()
    }
}
impl Operand for HipVecRef<'_> {
    fn raw(&self) -> RawView {
        self.raw.clone()
    }
}
impl Operand for &HipVecRef<'_> {
    fn raw(&self) -> RawView {
        self.raw.clone()
    }
}
impl Operand for HipVecMut<'_> {
    fn raw(&self) -> RawView {
        self.raw.clone()
    }
}
impl Operand for &HipVecMut<'_> {
    fn raw(&self) -> RawView {
        self.raw.clone()
    }
}

impl RawView {
    fn nb(&self) -> usize {
        self.context.nbatch()
    }
    /// Contiguous batch-fastest data for the kernels: the view itself, or a temporary holding the one member of a `get_batch` view.
    fn contiguous(&self) -> (*mut f64, Option<HipVec>) {
        match self.member {
            None => (self.ptr, None),
            Some((parent_nb, b)) => {
                let tmp = HipVec::uninit(self.nstates, self.context.clone());
                check(
                    unsafe { ffi::dsh_vec_extract_batch(self.context.ptr(), self.nstates as i64, parent_nb as i64, self.ptr, b as i64, tmp.buf.f64()) },
                    "dsh_vec_extract_batch",
                );
                (tmp.buf.f64(), Some(tmp))
            }
        }
    }
    /// Write a temporary produced by `contiguous` back into the member view.
    fn write_back(&self, tmp: Option<HipVec>) {
        if let (Some((parent_nb, b)), Some(t)) = (self.member, tmp) {
            check(
                unsafe { ffi::dsh_vec_insert_batch(self.context.ptr(), self.nstates as i64, parent_nb as i64, self.ptr, b as i64, t.buf.f64()) },
                "dsh_vec_insert_batch",
            );
        }
    }
}

fn check_len(a: &RawView, b: &RawView, op: &str) {
    assert_eq!(a.nstates, b.nstates, "Vector lengths do not match in {op}: {} != {}", a.nstates, b.nstates);
    a.context.assert_compatible_nbatch(b.nb(), op);
}

#[derive(Clone, Copy)]
pub(crate) enum Bin {
    Add,
    Sub,
}
#[derive(Clone, Copy)]
pub(crate) enum Asg {
    Add,
    Sub,
    Mul,
    Div,
}

/// `ret = lhs (+|-) rhs` with broadcasting of an `nbatch == 1` operand (vec_add.cu / vec_sub.cu semantics, vector/cuda.rs:439-488).
pub(crate) fn binary(kind: Bin, lhs: RawView, rhs: RawView) -> HipVec {
    check_len(&lhs, &rhs, "vector add/sub");
    let nb = lhs.nb().max(rhs.nb());
    let ctx = if lhs.nb() >= rhs.nb() { lhs.context.clone() } else { rhs.context.clone() };
    let ret = HipVec::uninit(lhs.nstates, ctx.clone());
    let (lp, _lt) = lhs.contiguous();
    let (rp, _rt) = rhs.contiguous();
    let (n, nbi) = (lhs.nstates as i64, nb as i64);
    let rc = unsafe {
        match kind {
            Bin::Add => ffi::dsh_vec_add(ctx.ptr(), n, nbi, lp, lhs.nb() as i64, rp, rhs.nb() as i64, ret.buf.f64()),
            Bin::Sub => ffi::dsh_vec_sub(ctx.ptr(), n, nbi, lp, lhs.nb() as i64, rp, rhs.nb() as i64, ret.buf.f64()),
        }
    };
    check(rc, "dsh_vec_add/sub");
    ret
}

/// `lhs (+|-|*|/)= rhs` in place; the right operand may broadcast, the left one may not shrink (vector/cuda.rs:396-437).
pub(crate) fn assign(kind: Asg, lhs: &RawView, rhs: RawView) {
    check_len(lhs, &rhs, "vector assign op");
    assert!(rhs.nb() == lhs.nb() || rhs.nb() == 1, "incompatible nbatch in assign op: lhs={}, rhs={}", lhs.nb(), rhs.nb());
    let (lp, lt) = lhs.contiguous();
    let (rp, _rt) = rhs.contiguous();
    let (c, n, nb, rnb) = (lhs.context.ptr(), lhs.nstates as i64, lhs.nb() as i64, rhs.nb() as i64);
    let rc = unsafe {
        match kind {
            Asg::Add => ffi::dsh_vec_add_assign(c, n, nb, lp, rp, rnb),
            Asg::Sub => ffi::dsh_vec_sub_assign(c, n, nb, lp, rp, rnb),
            Asg::Mul => ffi::dsh_vec_mul_assign(c, n, nb, lp, rp, rnb),
            Asg::Div => ffi::dsh_vec_div_assign(c, n, nb, lp, rp, rnb),
        }
    };
    check(rc, "dsh_vec_*_assign");
    lhs.write_back(lt);
}

pub(crate) fn scaled(x: RawView, s: f64) -> HipVec {
    let ret = HipVec::uninit(x.nstates, x.context.clone());
    let (xp, _t) = x.contiguous();
    check(unsafe { ffi::dsh_vec_mul_scalar(x.context.ptr(), x.nstates as i64, x.nb() as i64, xp, s, ret.buf.f64()) }, "dsh_vec_mul_scalar");
    ret
}
pub(crate) fn scale_in_place(x: &RawView, s: f64) {
    let (xp, t) = x.contiguous();
    check(unsafe { ffi::dsh_vec_mul_assign_scalar(x.context.ptr(), x.nstates as i64, x.nb() as i64, xp, s) }, "dsh_vec_mul_assign_scalar");
    x.write_back(t);
}
pub(crate) fn copy_into(dst: &RawView, src: RawView, op: &str) {
    check_len(dst, &src, op);
    assert!(src.nb() == dst.nb() || src.nb() == 1, "incompatible nbatch in {op}: lhs={}, rhs={}", dst.nb(), src.nb());
    let (dp, dt) = dst.contiguous();
    let (sp, _st) = src.contiguous();
    check(unsafe { ffi::dsh_vec_copy(dst.context.ptr(), dst.nstates as i64, dst.nb() as i64, sp, src.nb() as i64, dp) }, "dsh_vec_copy");
    dst.write_back(dt);
}
pub(crate) fn axpy_into(y: &RawView, alpha: f64, x: RawView, beta: f64) {
    check_len(y, &x, "axpy");
    assert!(x.nb() == y.nb() || x.nb() == 1, "incompatible nbatch in axpy: lhs={}, rhs={}", y.nb(), x.nb());
    let (yp, yt) = y.contiguous();
    let (xp, _xt) = x.contiguous();
    check(unsafe { ffi::dsh_vec_axpy(y.context.ptr(), y.nstates as i64, y.nb() as i64, alpha, xp, x.nb() as i64, beta, yp) }, "dsh_vec_axpy");
    y.write_back(yt);
}
pub(crate) fn squared_norm_of(x: RawView, y: RawView, atol: RawView, rtol: f64) -> f64 {
    assert!(x.nstates == y.nstates && x.nstates == atol.nstates, "Vector lengths do not match");
    x.context.assert_compatible_nbatch(y.nb(), "squared_norm");
    x.context.assert_compatible_nbatch(atol.nb(), "squared_norm");
    if x.nstates == 0 {
        return 0.0;
    }
    let (xp, _a) = x.contiguous();
    let (yp, _b) = y.contiguous();
    let (ap, _c) = atol.contiguous();
    let mut out = 0.0f64;
    // max over members of mean_i (x_i / (|y_i| rtol + atol_i))^2 (vec_squared_norm.cu:13-53 + the host max of vector/cuda.rs:1421-1432), one launch,
    // the result written straight into pinned host memory; per_batch_dev = NULL (the per-member values are what a batched integrator would use)
    check(
        unsafe {
            ffi::dsh_vec_squared_norm(x.context.ptr(), x.nstates as i64, x.nb() as i64, xp, yp, y.nb() as i64, ap, atol.nb() as i64, rtol, &mut out, ptr::null_mut())
        },
        "dsh_vec_squared_norm",
    );
    out
}

// ------------------------------------------------------------------ VectorCommon
impl VectorCommon for HipVec {
    type T = f64;
    type C = HipContext;
    type Inner = DeviceBuf;
    fn inner(&self) -> &Self::Inner {
        &self.buf
    }
}
impl VectorCommon for HipVecRef<'_> {
    type T = f64;
    type C = HipContext;
    type Inner = RawView;
    fn inner(&self) -> &Self::Inner {
        &self.raw
    }
}
impl VectorCommon for HipVecMut<'_> {
    type T = f64;
    type C = HipContext;
    type Inner = RawView;
    fn inner(&self) -> &Self::Inner {
        &self.raw
    }
}

// ------------------------------------------------------------------ operators (every combination the trait bounds ask for, vector/mod.rs:71-177)
macro_rules! impl_binary {
    ($lhs:ty, $rhs:ty) => {
        impl Add<$rhs> for $lhs {
            type Output = HipVec;
            fn add(self, rhs: $rhs) -> HipVec {
                binary(Bin::Add, Operand::raw(&self), Operand::raw(&rhs))
            }
        }
        impl Sub<$rhs> for $lhs {
            type Output = HipVec;
            fn sub(self, rhs: $rhs) -> HipVec {
                binary(Bin::Sub, Operand::raw(&self), Operand::raw(&rhs))
            }
        }
    };
}
// Vector: V op {V, &V, View, &View};  VectorRef<V>: &V op the same four;  VectorView: View op {View, V, &V, &View}
impl_binary!(HipVec, HipVec);
impl_binary!(HipVec, &HipVec);
impl_binary!(HipVec, HipVecRef<'_>);
impl_binary!(HipVec, &HipVecRef<'_>);
impl_binary!(&HipVec, HipVec);
impl_binary!(&HipVec, &HipVec);
impl_binary!(&HipVec, HipVecRef<'_>);
impl_binary!(&HipVec, &HipVecRef<'_>);
impl_binary!(HipVecRef<'_>, HipVecRef<'_>);
impl_binary!(HipVecRef<'_>, HipVec);
impl_binary!(HipVecRef<'_>, &HipVec);
impl_binary!(HipVecRef<'_>, &HipVecRef<'_>);

macro_rules! impl_assign {
    ($lhs:ty, $rhs:ty) => {
        impl AddAssign<$rhs> for $lhs {
            fn add_assign(&mut self, rhs: $rhs) {
                assign(Asg::Add, &Operand::raw(&*self), Operand::raw(&rhs));
            }
        }
        impl SubAssign<$rhs> for $lhs {
            fn sub_assign(&mut self, rhs: $rhs) {
                assign(Asg::Sub, &Operand::raw(&*self), Operand::raw(&rhs));
            }
        }
    };
}
impl_assign!(HipVec, HipVec);
impl_assign!(HipVec, &HipVec);
impl_assign!(HipVec, HipVecRef<'_>);
impl_assign!(HipVec, &HipVecRef<'_>);
impl_assign!(HipVecMut<'_>, HipVecRef<'_>);
impl_assign!(HipVecMut<'_>, HipVec);
impl_assign!(HipVecMut<'_>, &HipVecRef<'_>);
impl_assign!(HipVecMut<'_>, &HipVec);

macro_rules! impl_scale {
    ($lhs:ty) => {
        impl Mul<Scale<f64>> for $lhs {
            type Output = HipVec;
            fn mul(self, rhs: Scale<f64>) -> HipVec {
                scaled(Operand::raw(&self), rhs.value())
            }
        }
    };
}
impl_scale!(HipVec);
impl_scale!(&HipVec);
impl_scale!(HipVecRef<'_>);
impl Div<Scale<f64>> for HipVec {
    type Output = HipVec;
    fn div(self, rhs: Scale<f64>) -> HipVec {
        scaled(Operand::raw(&self), 1.0 / rhs.value())
    }
}
impl MulAssign<Scale<f64>> for HipVec {
    fn mul_assign(&mut self, rhs: Scale<f64>) {
        scale_in_place(&Operand::raw(&*self), rhs.value());
    }
}
impl MulAssign<Scale<f64>> for HipVecMut<'_> {
    fn mul_assign(&mut self, rhs: Scale<f64>) {
        scale_in_place(&self.raw, rhs.value());
    }
}

// ------------------------------------------------------------------ VectorIndex
impl VectorIndex for HipIndex {
    type C = HipContext;
    fn context(&self) -> &Self::C {
        &self.context
    }
    fn zeros(len: IndexType, ctx: Self::C) -> Self {
        Self { buf: DeviceBuf::new(4 * len.max(1), true, &ctx), len, context: ctx }
    }
    fn len(&self) -> IndexType {
        self.len
    }
    fn clone_as_vec(&self) -> Vec<IndexType> {
        let mut host = vec![0i32; self.len];
        if self.len > 0 {
            check(unsafe { ffi::dsh_d2h(self.context.ptr(), host.as_mut_ptr() as *mut c_void, self.buf.ptr, 4 * self.len as i64) }, "dsh_d2h");
        }
        host.into_iter().map(|x| x as IndexType).collect()
    }
    fn from_vec(v: Vec<IndexType>, ctx: Self::C) -> Self {
        let host: Vec<i32> = v.iter().map(|&x| i32::try_from(x).expect("index does not fit the device's int32")).collect();
        let out = Self { buf: DeviceBuf::new(4 * host.len().max(1), false, &ctx), len: host.len(), context: ctx };
        if !host.is_empty() {
            check(unsafe { ffi::dsh_h2d(out.context.ptr(), out.buf.ptr, host.as_ptr() as *const c_void, 4 * host.len() as i64) }, "dsh_h2d");
        }
        out
    }
}
impl HipIndex {
    pub(crate) fn i32(&self) -> *const i32 {
        self.buf.ptr as *const i32
    }
}

// ------------------------------------------------------------------ Vector
impl HipVec {
    pub(crate) fn uninit(nstates: IndexType, ctx: HipContext) -> Self {
        let nbytes = 8 * (nstates * ctx.nbatch()).max(1);
        Self { buf: DeviceBuf::new(nbytes, false, &ctx), nstates, context: ctx }
    }
    pub(crate) fn nb(&self) -> usize {
        self.context.nbatch()
    }
    pub(crate) fn ptr(&self) -> *mut f64 {
        self.buf.f64()
    }
    /// Per-member value of the weighted mean-square norm (one `f64` per member, on the device): what a per-member step controller consumes
    /// (`dsh_vec_squared_norm`'s `per_batch_dev` output).  Returns `(max over members, per-member vector)`.
    pub fn squared_norm_per_batch(&self, y: &Self, atol: &Self, rtol: f64) -> (f64, HipVec) {
        let per = HipVec::uninit(1, self.context.clone());
        let mut out = 0.0f64;
        check(
            unsafe {
                ffi::dsh_vec_squared_norm(
                    self.context.ptr(), self.nstates as i64, self.nb() as i64, self.ptr(), y.ptr(), y.nb() as i64, atol.ptr(), atol.nb() as i64, rtol, &mut out, per.ptr(),
                )
            },
            "dsh_vec_squared_norm",
        );
        (out, per)
    }
}

impl Vector for HipVec {
    type View<'a> = HipVecRef<'a>;
    type ViewMut<'a> = HipVecMut<'a>;
    type Index = HipIndex;

    fn context(&self) -> &Self::C {
        &self.context
    }
    fn inner_mut(&mut self) -> &mut Self::Inner {
        &mut self.buf
    }
    /// sets element `index` of EVERY member (vec_set_index.cu semantics)
    fn set_index(&mut self, index: IndexType, value: Self::T) {
        assert!(index < self.nstates, "Index out of bounds");
        check(unsafe { ffi::dsh_vec_set_index_all(self.context.ptr(), self.nb() as i64, self.ptr(), index as i64, value) }, "dsh_vec_set_index_all");
    }
    fn get_index(&self, index: IndexType) -> Self::T {
        assert!(self.nb() == 1, "get_index not supported for batched vectors; use get_batch(b).get_index(i)");
        assert!(index < self.nstates, "Index out of bounds");
        let mut out = 0.0;
        check(unsafe { ffi::dsh_vec_get_index(self.context.ptr(), 1, self.ptr(), index as i64, 0, &mut out) }, "dsh_vec_get_index");
        out
    }
    fn norm(&self, k: i32) -> Self::T {
        let mut out = 0.0;
        check(unsafe { ffi::dsh_vec_norm(self.context.ptr(), self.nstates as i64, self.nb() as i64, self.ptr(), k, &mut out) }, "dsh_vec_norm");
        out
    }
    fn squared_norm(&self, y: &Self, atol: &Self, rtol: Self::T) -> Self::T {
        squared_norm_of(Operand::raw(self), Operand::raw(y), Operand::raw(atol), rtol)
    }
    fn len(&self) -> IndexType {
        self.nstates
    }
    fn from_element(nstates: usize, value: Self::T, ctx: Self::C) -> Self {
        let mut v = Self::uninit(nstates, ctx);
        v.fill(value);
        v
    }
    fn fill(&mut self, value: Self::T) {
        check(unsafe { ffi::dsh_vec_fill(self.context.ptr(), self.nstates as i64, self.nb() as i64, self.ptr(), value) }, "dsh_vec_fill");
    }
    fn as_view(&self) -> Self::View<'_> {
        HipVecRef { raw: Operand::raw(self), _life: PhantomData }
    }
    fn as_view_mut(&mut self) -> Self::ViewMut<'_> {
        HipVecMut { raw: Operand::raw(&*self), _life: PhantomData }
    }
    fn get_batch(&self, batch: usize) -> Self::View<'_> {
        assert!(batch < self.nb(), "Batch index out of bounds");
        let ctx = self.context.clone_with_nbatch(1).unwrap();
        let member = if self.nb() == 1 { None } else { Some((self.nb(), batch)) };
        HipVecRef { raw: RawView { ptr: self.ptr(), nstates: self.nstates, context: ctx, member }, _life: PhantomData }
    }
    fn get_batch_mut(&mut self, batch: usize) -> Self::ViewMut<'_> {
        assert!(batch < self.nb(), "Batch index out of bounds");
        let ctx = self.context.clone_with_nbatch(1).unwrap();
        let member = if self.nb() == 1 { None } else { Some((self.nb(), batch)) };
        HipVecMut { raw: RawView { ptr: self.ptr(), nstates: self.nstates, context: ctx, member }, _life: PhantomData }
    }
    fn copy_from(&mut self, other: &Self) {
        copy_into(&Operand::raw(&*self), Operand::raw(other), "copy_from");
    }
    fn copy_from_view(&mut self, other: &Self::View<'_>) {
        copy_into(&Operand::raw(&*self), other.raw.clone(), "copy_from_view");
    }
    /// host data is batch-major (`[member 0 states, member 1 states, ...]`, vector/cuda.rs:119-125); transposed on the way in
    fn from_vec(vec: Vec<Self::T>, ctx: Self::C) -> Self {
        Self::from_slice(&vec, ctx)
    }
    fn from_slice(slice: &[Self::T], ctx: Self::C) -> Self {
        let nb = ctx.nbatch();
        assert!(slice.len() % nb == 0, "Vector length {} is not a multiple of nbatch {}", slice.len(), nb);
        let n = slice.len() / nb;
        let v = Self::uninit(n, ctx);
        if !slice.is_empty() {
            check(unsafe { ffi::dsh_vec_upload(v.context.ptr(), n as i64, nb as i64, slice.as_ptr(), v.ptr()) }, "dsh_vec_upload");
        }
        v
    }
    fn clone_as_vec(&self) -> Vec<Self::T> {
        let mut host = vec![0.0; self.nstates * self.nb()];
        if !host.is_empty() {
            check(unsafe { ffi::dsh_vec_download(self.context.ptr(), self.nstates as i64, self.nb() as i64, self.ptr(), host.as_mut_ptr()) }, "dsh_vec_download");
        }
        host
    }
    fn axpy(&mut self, alpha: Self::T, x: &Self, beta: Self::T) {
        axpy_into(&Operand::raw(&*self), alpha, Operand::raw(x), beta);
    }
    fn axpy_v(&mut self, alpha: Self::T, x: &Self::View<'_>, beta: Self::T) {
        axpy_into(&Operand::raw(&*self), alpha, x.raw.clone(), beta);
    }
    fn batched_axpy(&mut self, alpha: &[Self::T], x: &Self, beta: Self::T) {
        assert_eq!(alpha.len(), self.nb(), "batched_axpy needs one alpha per member");
        assert_eq!(self.nstates, x.nstates, "Vector lengths do not match");
        self.context.assert_compatible_nbatch(x.nb(), "batched_axpy");
        check(
            unsafe { ffi::dsh_vec_batched_axpy(self.context.ptr(), self.nstates as i64, self.nb() as i64, alpha.as_ptr(), x.ptr(), x.nb() as i64, beta, self.ptr()) },
            "dsh_vec_batched_axpy",
        );
    }
    fn component_mul_assign(&mut self, other: &Self) {
        assign(Asg::Mul, &Operand::raw(&*self), Operand::raw(other));
    }
    fn component_div_assign(&mut self, other: &Self) {
        assign(Asg::Div, &Operand::raw(&*self), Operand::raw(other));
    }
    /// `(found_zero, max |g1/(g1-g0)| over sign changes, its index)`; every member must agree on `(found, index)` — the reference panics otherwise
    /// (vector/cuda.rs:1166-1171), the C ABI returns DSH_E_BATCH_MISMATCH, which `check` turns into the same panic.
    fn root_finding(&self, g1: &Self) -> (bool, Self::T, i32) {
        assert_eq!(self.nstates, g1.nstates, "Vector lengths do not match");
        assert_eq!(self.nb(), g1.nb(), "root_finding needs equal nbatch");
        let (mut found, mut frac, mut idx) = (0i32, 0.0f64, -1i32);
        check(
            unsafe { ffi::dsh_vec_root_finding(self.context.ptr(), self.nstates as i64, self.nb() as i64, self.ptr(), g1.ptr(), &mut found, &mut frac, &mut idx) },
            "root_finding",
        );
        (found != 0, frac, idx)
    }
    fn assign_at_indices(&mut self, indices: &Self::Index, value: Self::T) {
        check(
            unsafe { ffi::dsh_vec_assign_at_indices(self.context.ptr(), self.nstates as i64, self.nb() as i64, indices.i32(), indices.len as i64, value, self.ptr()) },
            "dsh_vec_assign_at_indices",
        );
    }
    fn copy_from_indices(&mut self, other: &Self, indices: &Self::Index) {
        assert_eq!(self.nstates, other.nstates, "Vector lengths do not match");
        assert_eq!(self.nb(), other.nb(), "copy_from_indices needs equal nbatch");
        check(
            unsafe { ffi::dsh_vec_copy_from_indices(self.context.ptr(), self.nstates as i64, self.nb() as i64, other.ptr(), indices.i32(), indices.len as i64, self.ptr()) },
            "dsh_vec_copy_from_indices",
        );
    }
    fn gather(&mut self, other: &Self, indices: &Self::Index) {
        assert_eq!(self.nstates, indices.len, "gather: self.len() must equal indices.len()");
        assert_eq!(self.nb(), other.nb(), "gather needs equal nbatch");
        check(
            unsafe { ffi::dsh_vec_gather(self.context.ptr(), other.nstates as i64, self.nb() as i64, other.ptr(), indices.i32(), indices.len as i64, self.ptr()) },
            "dsh_vec_gather",
        );
    }
    fn scatter(&self, indices: &Self::Index, other: &mut Self) {
        assert_eq!(self.nstates, indices.len, "scatter: self.len() must equal indices.len()");
        assert_eq!(self.nb(), other.nb(), "scatter needs equal nbatch");
        check(
            unsafe { ffi::dsh_vec_scatter(self.context.ptr(), other.nstates as i64, self.nb() as i64, self.ptr(), indices.i32(), indices.len as i64, other.ptr()) },
            "dsh_vec_scatter",
        );
    }
}

// ------------------------------------------------------------------ views
impl<'a> VectorView<'a> for HipVecRef<'a> {
    type Owned = HipVec;
    fn get_index(&self, index: IndexType) -> Self::T {
        assert!(self.raw.nb() == 1, "get_index not supported for batched views");
        assert!(index < self.raw.nstates, "Index out of bounds");
        let (parent_nb, b) = self.raw.member.unwrap_or((1, 0));
        let mut out = 0.0;
        check(unsafe { ffi::dsh_vec_get_index(self.raw.context.ptr(), parent_nb as i64, self.raw.ptr, index as i64, b as i64, &mut out) }, "dsh_vec_get_index");
        out
    }
    fn squared_norm(&self, y: &Self::Owned, atol: &Self::Owned, rtol: Self::T) -> Self::T {
        squared_norm_of(self.raw.clone(), Operand::raw(y), Operand::raw(atol), rtol)
    }
    fn into_owned(self) -> Self::Owned {
        let out = HipVec::uninit(self.raw.nstates, self.raw.context.clone());
        copy_into(&Operand::raw(&out), self.raw.clone(), "into_owned");
        out
    }
}

impl<'a> VectorViewMut<'a> for HipVecMut<'a> {
    type Owned = HipVec;
    type View = HipVecRef<'a>;
    type Index = HipIndex;
    fn copy_from(&mut self, other: &Self::Owned) {
        copy_into(&self.raw, Operand::raw(other), "copy_from");
    }
    fn copy_from_view(&mut self, other: &Self::View) {
        copy_into(&self.raw, other.raw.clone(), "copy_from_view");
    }
    fn axpy(&mut self, alpha: Self::T, x: &Self::Owned, beta: Self::T) {
        axpy_into(&self.raw, alpha, Operand::raw(x), beta);
    }
    fn set_index(&mut self, index: IndexType, value: Self::T) {
        assert!(index < self.raw.nstates, "Index out of bounds");
        let c = self.raw.context.ptr();
        match self.raw.member {
            None => check(unsafe { ffi::dsh_vec_set_index_all(c, self.raw.nb() as i64, self.raw.ptr, index as i64, value) }, "dsh_vec_set_index_all"),
            Some((parent_nb, b)) => check(unsafe { ffi::dsh_vec_set_index(c, parent_nb as i64, self.raw.ptr, index as i64, b as i64, value) }, "dsh_vec_set_index"),
        }
    }
}
