//! `HipMat` / `HipMatRef` / `HipMatMut`: `Matrix`, `DenseMatrix`, `MatrixView`, `MatrixViewMut`, `DefaultSolver`
//! (trait definitions: crates/diffsol-la/src/matrix/mod.rs:35-410; what they replace: matrix/cuda.rs:32-1468 incl. its cuBLAS gemv / gemm call sites).
//!
//! Device layout: column-major per member, batch-fastest — entry `(i, j)` of member `b` at `ptr[(j * nrows + i) * nbatch + b]` — so
//! * the matrix data IS a batched vector of length `nrows * ncols` (element-wise operations reuse the `dsh_vec_*` entry points),
//! * column `j` is the contiguous batched vector at `ptr + j * nrows * nbatch`, a column range `[start, end)` the contiguous block behind it:
//!   views are pointer offsets, `gemv_o` / `gemm_vo` on column ranges (what `Bdf::_update_diff_for_step_size` and `Sdirk` use) need no copies.
//! The host-facing layout of `from_vec` / `triplet_iter` stays the reference's (member-major, column-major per member).
use crate::context::HipContext;
use crate::error::check;
use crate::ffi;
use crate::lu::HipLU;
use crate::vector::{binary, Asg, Bin, DeviceBuf, HipIndex, HipVec, HipVecMut, HipVecRef, Operand, RawView};
use diffsol_la::error::{LaError, MatrixError};
use diffsol_la::matrix::default_solver::DefaultSolver;
use diffsol_la::matrix::sparsity::{Dense, DenseRef};
use diffsol_la::matrix::{DenseMatrix, Matrix, MatrixCommon, MatrixView, MatrixViewMut};
use diffsol_la::scalar::Scale;
use diffsol_la::{Context, IndexType, Vector, VectorIndex};
use std::marker::PhantomData;
use std::ops::{Add, AddAssign, Mul, MulAssign, Sub, SubAssign};

#[derive(Debug, Clone)]
pub struct HipMat {
    pub(crate) buf: DeviceBuf,
    pub(crate) nrows: IndexType,
    pub(crate) ncols: IndexType,
    pub(crate) context: HipContext,
}

/// Non-owning description of a (column range of a) matrix.
#[derive(Debug, Clone)]
pub struct RawMat {
    pub(crate) ptr: *mut f64,
    pub(crate) nrows: IndexType,
    pub(crate) ncols: IndexType,
    pub(crate) context: HipContext,
}
#[derive(Debug)]
pub struct HipMatRef<'a> {
    pub(crate) raw: RawMat,
    pub(crate) _life: PhantomData<&'a f64>,
}
#[derive(Debug)]
pub struct HipMatMut<'a> {
    pub(crate) raw: RawMat,
    pub(crate) _life: PhantomData<&'a mut f64>,
}

impl DefaultSolver for HipMat {
    type LS = HipLU;
}

pub(crate) trait MatOperand {
    fn rawm(&self) -> RawMat;
}
impl MatOperand for HipMat {
    fn rawm(&self) -> RawMat {
        RawMat { ptr: self.buf.f64(), nrows: self.nrows, ncols: self.ncols, context: self.context.clone() }
    }
}
impl MatOperand for &HipMat {
    fn rawm(&self) -> RawMat {
        (**self).rawm()
    }
}
impl MatOperand for HipMatRef<'_> {
    fn rawm(&self) -> RawMat {
        self.raw.clone()
    }
}
impl MatOperand for &HipMatRef<'_> {
    fn rawm(&self) -> RawMat {
        self.raw.clone()
    }
}
impl MatOperand for HipMatMut<'_> {
    fn rawm(&self) -> RawMat {
        self.raw.clone()
    }
}
impl MatOperand for &HipMatMut<'_> {
    fn rawm(&self) -> RawMat {
        self.raw.clone()
    }
}

impl RawMat {
    fn nb(&self) -> usize {
        self.context.nbatch()
    }
    /// all entries as one batched vector (column after column)
    fn flat(&self) -> RawView {
        RawView { ptr: self.ptr, nstates: self.nrows * self.ncols, context: self.context.clone(), member: None }
    }
    fn same_shape(&self, o: &RawMat, op: &str) {
        assert!(self.nrows == o.nrows && self.ncols == o.ncols, "Matrix shapes do not match in {op}: {}x{} vs {}x{}", self.nrows, self.ncols, o.nrows, o.ncols);
    }
}

fn mat_from_flat(v: HipVec, nrows: IndexType, ncols: IndexType) -> HipMat {
    HipMat { buf: v.buf, nrows, ncols, context: v.context }
}
fn mat_binary(kind: Bin, a: RawMat, b: RawMat) -> HipMat {
    a.same_shape(&b, "matrix add/sub");
    mat_from_flat(binary(kind, a.flat(), b.flat()), a.nrows, a.ncols)
}
fn mat_assign(kind: Asg, a: &RawMat, b: RawMat) {
    a.same_shape(&b, "matrix assign op");
    crate::vector::assign(kind, &a.flat(), b.flat());
}
fn mat_scaled(a: RawMat, s: f64) -> HipMat {
    mat_from_flat(crate::vector::scaled(a.flat(), s), a.nrows, a.ncols)
}
/// `y = alpha A x + beta y` (one launch for all members; replaces the host loop over cublasDgemv, matrix/cuda.rs:65-101)
fn gemv_raw(a: &RawMat, alpha: f64, x: RawView, beta: f64, y: &mut HipVec) {
    assert!(x.nstates == a.ncols && y.len() == a.nrows, "gemv: shape mismatch");
    a.context.assert_compatible_nbatch(x.context.nbatch(), "gemv");
    y.context().assert_compatible_nbatch(a.nb(), "gemv");
    check(
        unsafe {
            ffi::dsh_mat_gemv(y.context().ptr(), a.nrows as i64, a.ncols as i64, y.context().nbatch() as i64, alpha, a.ptr, a.nb() as i64, x.ptr, x.context.nbatch() as i64, beta, y.ptr())
        },
        "dsh_mat_gemv",
    );
}
/// `C = alpha A B + beta C`; `A` / `B` may broadcast (`nbatch == 1`: the R·U matrix of `Bdf::_update_diff_for_step_size`, the tableau of `Sdirk`)
fn gemm_raw(c: &RawMat, alpha: f64, a: RawMat, b: RawMat, beta: f64) {
    assert!(a.nrows == c.nrows && b.ncols == c.ncols && a.ncols == b.nrows, "gemm: shape mismatch");
    c.context.assert_compatible_nbatch(a.nb(), "gemm");
    c.context.assert_compatible_nbatch(b.nb(), "gemm");
    check(
        unsafe { ffi::dsh_mat_gemm(c.context.ptr(), c.nrows as i64, c.ncols as i64, a.ncols as i64, c.nb() as i64, alpha, a.ptr, a.nb() as i64, b.ptr, b.nb() as i64, beta, c.ptr) },
        "dsh_mat_gemm",
    );
}

// ------------------------------------------------------------------ MatrixCommon
macro_rules! impl_common {
    ($t:ty, $inner:ty, $field:ident) => {
        impl MatrixCommon for $t {
            type V = HipVec;
            type T = f64;
            type C = HipContext;
            type Inner = $inner;
            fn nrows(&self) -> IndexType {
                self.rawm().nrows
            }
            fn ncols(&self) -> IndexType {
                self.rawm().ncols
            }
            fn inner(&self) -> &Self::Inner {
                &self.$field
            }
        }
    };
}
impl_common!(HipMat, DeviceBuf, buf);
impl_common!(HipMatRef<'_>, RawMat, raw);
impl_common!(HipMatMut<'_>, RawMat, raw);

// ------------------------------------------------------------------ operators (matrix/mod.rs:84-155, :335-349)
macro_rules! impl_mat_binary {
    ($lhs:ty, $rhs:ty) => {
        impl Add<$rhs> for $lhs {
            type Output = HipMat;
            fn add(self, rhs: $rhs) -> HipMat {
                mat_binary(Bin::Add, self.rawm(), rhs.rawm())
            }
        }
        impl Sub<$rhs> for $lhs {
            type Output = HipMat;
            fn sub(self, rhs: $rhs) -> HipMat {
                mat_binary(Bin::Sub, self.rawm(), rhs.rawm())
            }
        }
    };
}
impl_mat_binary!(HipMat, &HipMat); // DenseMatrix: MatrixOpsByValue<&Self, Self>
impl_mat_binary!(HipMat, &HipMatRef<'_>); // DenseMatrix: MatrixOpsByValue<&View, Self>
impl_mat_binary!(HipMatRef<'_>, &HipMat); // MatrixView: MatrixOpsByValue<&Owned, Owned>

macro_rules! impl_mat_assign {
    ($lhs:ty, $rhs:ty) => {
        impl AddAssign<$rhs> for $lhs {
            fn add_assign(&mut self, rhs: $rhs) {
                mat_assign(Asg::Add, &(&*self).rawm(), rhs.rawm());
            }
        }
        impl SubAssign<$rhs> for $lhs {
            fn sub_assign(&mut self, rhs: $rhs) {
                mat_assign(Asg::Sub, &(&*self).rawm(), rhs.rawm());
            }
        }
    };
}
impl_mat_assign!(HipMat, &HipMat);
impl_mat_assign!(HipMat, &HipMatRef<'_>);
impl_mat_assign!(HipMatMut<'_>, &HipMatMut<'_>);
impl_mat_assign!(HipMatMut<'_>, &HipMatRef<'_>);

macro_rules! impl_mat_scale {
    ($lhs:ty) => {
        impl Mul<Scale<f64>> for $lhs {
            type Output = HipMat;
            fn mul(self, rhs: Scale<f64>) -> HipMat {
                mat_scaled(self.rawm(), rhs.value())
            }
        }
    };
}
impl_mat_scale!(HipMat);
impl_mat_scale!(&HipMat); // MatrixRef<HipMat>
impl_mat_scale!(HipMatRef<'_>);
impl MulAssign<Scale<f64>> for HipMatMut<'_> {
    fn mul_assign(&mut self, rhs: Scale<f64>) {
        crate::vector::scale_in_place(&self.raw.flat(), rhs.value());
    }
}

// ------------------------------------------------------------------ views
impl<'a> MatrixView<'a> for HipMatRef<'a> {
    type Owned = HipMat;
    fn into_owned(self) -> Self::Owned {
        let out = HipMat::uninit(self.raw.nrows, self.raw.ncols, self.raw.context.clone());
        crate::vector::copy_into(&out.rawm().flat(), self.raw.flat(), "into_owned");
        out
    }
    fn gemv_v(&self, alpha: Self::T, x: &HipVecRef<'_>, beta: Self::T, y: &mut Self::V) {
        let xr = x.raw.clone();
        assert!(xr.member.is_none(), "gemv_v on a single-member view is not supported");
        gemv_raw(&self.raw, alpha, xr, beta, y);
    }
    fn gemv_o(&self, alpha: Self::T, x: &Self::V, beta: Self::T, y: &mut Self::V) {
        gemv_raw(&self.raw, alpha, Operand::raw(x), beta, y);
    }
}
impl<'a> MatrixViewMut<'a> for HipMatMut<'a> {
    type Owned = HipMat;
    type View = HipMatRef<'a>;
    fn into_owned(self) -> Self::Owned {
        let out = HipMat::uninit(self.raw.nrows, self.raw.ncols, self.raw.context.clone());
        crate::vector::copy_into(&out.rawm().flat(), self.raw.flat(), "into_owned");
        out
    }
    fn gemm_oo(&mut self, alpha: Self::T, a: &Self::Owned, b: &Self::Owned, beta: Self::T) {
        gemm_raw(&self.raw, alpha, a.rawm(), b.rawm(), beta);
    }
    fn gemm_vo(&mut self, alpha: Self::T, a: &Self::View, b: &Self::Owned, beta: Self::T) {
        gemm_raw(&self.raw, alpha, a.raw.clone(), b.rawm(), beta);
    }
}

// ------------------------------------------------------------------ Matrix
impl HipMat {
    pub(crate) fn uninit(nrows: IndexType, ncols: IndexType, ctx: HipContext) -> Self {
        let nbytes = 8 * (nrows * ncols * ctx.nbatch()).max(1);
        Self { buf: DeviceBuf::new(nbytes, false, &ctx), nrows, ncols, context: ctx }
    }
    pub(crate) fn ptr(&self) -> *mut f64 {
        self.buf.f64()
    }
    pub(crate) fn nb(&self) -> usize {
        self.context.nbatch()
    }
    fn diagonal(&self) -> HipVec {
        let n = self.nrows.min(self.ncols);
        let d = HipVec::uninit(n, self.context.clone());
        check(unsafe { ffi::dsh_mat_get_diagonal(self.context.ptr(), n as i64, self.nb() as i64, self.ptr(), d.ptr()) }, "dsh_mat_get_diagonal");
        d
    }
    /// batch-major, column-major-per-member host copy (the reference's flat layout, matrix/cuda.rs:20-31)
    fn download(&self) -> Vec<f64> {
        // as a batched vector of length nrows*ncols the download already produces [member][col-major entries]
        let flat = HipVec { buf: self.buf.clone(), nstates: self.nrows * self.ncols, context: self.context.clone() };
        flat.clone_as_vec()
    }
}

impl Matrix for HipMat {
    type Sparsity = Dense<Self>;
    type SparsityRef<'a> = DenseRef<'a, Self>;

    fn sparsity(&self) -> Option<Self::SparsityRef<'_>> {
        None
    }
    fn context(&self) -> &Self::C {
        &self.context
    }
    fn inner_mut(&mut self) -> &mut Self::Inner {
        &mut self.buf
    }
    /// decided on member 0's diagonal like the reference (matrix/cuda.rs:1367-1388: `diagonal[i]` for `i < nstates`)
    fn partition_indices_by_zero_diagonal(&self) -> (HipIndex, HipIndex) {
        let diag = self.diagonal().clone_as_vec();
        let (mut zero, mut nonzero) = (Vec::new(), Vec::new());
        for i in 0..self.nrows {
            if diag[i] == 0.0 {
                zero.push(i)
            } else {
                nonzero.push(i)
            }
        }
        (HipIndex::from_vec(zero, self.context.clone()), HipIndex::from_vec(nonzero, self.context.clone()))
    }
    fn gemv(&self, alpha: Self::T, x: &Self::V, beta: Self::T, y: &mut Self::V) {
        gemv_raw(&self.rawm(), alpha, Operand::raw(x), beta, y);
    }
    fn copy_from(&mut self, other: &Self) {
        self.rawm().same_shape(&other.rawm(), "copy_from");
        crate::vector::copy_into(&self.rawm().flat(), other.rawm().flat(), "copy_from");
    }
    fn zeros(nrows: IndexType, ncols: IndexType, ctx: Self::C) -> Self {
        let nbytes = 8 * (nrows * ncols * ctx.nbatch()).max(1);
        Self { buf: DeviceBuf::new(nbytes, true, &ctx), nrows, ncols, context: ctx }
    }
    fn new_from_sparsity(nrows: IndexType, ncols: IndexType, _sparsity: Option<Self::Sparsity>, ctx: Self::C) -> Self {
        Self::zeros(nrows, ncols, ctx)
    }
    fn from_diagonal(v: &Self::V) -> Self {
        let n = v.len();
        let m = Self::zeros(n, n, v.context().clone());
        check(unsafe { ffi::dsh_mat_from_diagonal(m.context.ptr(), n as i64, m.nb() as i64, v.ptr(), v.context().nbatch() as i64, m.ptr()) }, "dsh_mat_from_diagonal");
        m
    }
    fn set_column(&mut self, j: IndexType, v: &Self::V) {
        assert!(j < self.ncols && v.len() == self.nrows, "set_column: shape mismatch");
        self.context.assert_compatible_nbatch(v.context().nbatch(), "set_column");
        check(
            unsafe { ffi::dsh_mat_set_column(self.context.ptr(), self.nrows as i64, self.ncols as i64, self.nb() as i64, self.ptr(), j as i64, v.ptr(), v.context().nbatch() as i64) },
            "dsh_mat_set_column",
        );
    }
    /// `v += self[:, j]` (vec_add_assign on the column view; matrix/cuda.rs:1176-1208)
    fn add_column_to_vector(&self, j: IndexType, v: &mut Self::V) {
        assert!(j < self.ncols && v.len() == self.nrows, "add_column_to_vector: shape mismatch");
        *v += self.column(j);
    }
    fn set_data_with_indices(&mut self, dst_indices: &HipIndex, src_indices: &HipIndex, data: &Self::V) {
        assert_eq!(dst_indices.len(), src_indices.len(), "Destination and source indices must have the same length");
        assert_eq!(self.nb(), data.context().nbatch(), "set_data_with_indices needs equal nbatch");
        check(
            unsafe {
                ffi::dsh_mat_set_data_with_indices(
                    self.context.ptr(), (self.nrows * self.ncols) as i64, data.len() as i64, self.nb() as i64, self.ptr(), dst_indices.i32(), src_indices.i32(), dst_indices.len() as i64, data.ptr(),
                )
            },
            "dsh_mat_set_data_with_indices",
        );
    }
    /// `self.data[k] = other.data[indices[k]]` over the flat column-major entries (the block extraction of `Matrix::split`)
    fn gather(&mut self, other: &Self, indices: &HipIndex) {
        assert_eq!(self.nb(), other.nb(), "gather needs equal nbatch");
        if indices.len() == 0 {
            return;
        }
        check(
            unsafe { ffi::dsh_vec_gather(self.context.ptr(), (other.nrows * other.ncols) as i64, self.nb() as i64, other.ptr(), indices.i32(), indices.len() as i64, self.ptr()) },
            "dsh_vec_gather",
        );
    }
    /// `self = x + beta * y` — the `M - c J` assembly of `BdfCallable::jacobian_inplace` (mat_scale_add_assign.cu)
    fn scale_add_and_assign(&mut self, x: &Self, beta: Self::T, y: &Self) {
        self.rawm().same_shape(&x.rawm(), "scale_add_and_assign");
        self.rawm().same_shape(&y.rawm(), "scale_add_and_assign");
        check(
            unsafe { ffi::dsh_mat_scale_add_assign(self.context.ptr(), (self.nrows * self.ncols) as i64, self.nb() as i64, self.ptr(), x.ptr(), x.nb() as i64, beta, y.ptr(), y.nb() as i64) },
            "dsh_mat_scale_add_assign",
        );
    }
    fn triplet_iter(&self) -> (impl Iterator<Item = (IndexType, IndexType)> + '_, impl Iterator<Item = Self::T> + '_) {
        let (nrows, ncols) = (self.nrows, self.ncols);
        let indices = (0..ncols).flat_map(move |j| (0..nrows).map(move |i| (i, j)));
        (indices, self.download().into_iter())
    }
    fn try_from_triplets(nrows: IndexType, ncols: IndexType, indices: Vec<(IndexType, IndexType)>, values: Vec<Self::T>, ctx: Self::C) -> Result<Self, LaError> {
        let nbatch = ctx.nbatch();
        let nnz = indices.len();
        assert_eq!(values.len(), nnz * nbatch, "Expected {} values ({} triplets * {} batches), got {}", nnz * nbatch, nnz, nbatch, values.len());
        let mut m = vec![0.0; nrows * ncols * nbatch];
        for b in 0..nbatch {
            for (k, &(i, j)) in indices.iter().enumerate() {
                if i >= nrows || j >= ncols {
                    return Err(LaError::from(MatrixError::IndexOutOfBounds));
                }
                m[b * nrows * ncols + i + j * nrows] = values[b * nnz + k];
            }
        }
        Ok(<Self as DenseMatrix>::from_vec(nrows, ncols, m, ctx))
    }
}

// ------------------------------------------------------------------ DenseMatrix
impl DenseMatrix for HipMat {
    type View<'a> = HipMatRef<'a>;
    type ViewMut<'a> = HipMatMut<'a>;

    fn gemm(&mut self, alpha: Self::T, a: &Self, b: &Self, beta: Self::T) {
        gemm_raw(&self.rawm(), alpha, a.rawm(), b.rawm(), beta);
    }
    fn column_axpy(&mut self, alpha: Self::T, j: IndexType, i: IndexType) {
        assert!(i < self.ncols && j < self.ncols, "Column index out of bounds");
        check(unsafe { ffi::dsh_mat_column_axpy(self.context.ptr(), self.nrows as i64, self.nb() as i64, self.ptr(), alpha, j as i64, i as i64) }, "dsh_mat_column_axpy");
    }
    fn columns(&self, start: IndexType, end: IndexType) -> Self::View<'_> {
        assert!(start <= end && end <= self.ncols, "Column range out of bounds");
        let ptr = unsafe { self.ptr().add(start * self.nrows * self.nb()) };
        HipMatRef { raw: RawMat { ptr, nrows: self.nrows, ncols: end - start, context: self.context.clone() }, _life: PhantomData }
    }
    fn column(&self, i: IndexType) -> HipVecRef<'_> {
        assert!(i < self.ncols, "Column index out of bounds");
        let ptr = unsafe { self.ptr().add(i * self.nrows * self.nb()) };
        HipVecRef { raw: RawView { ptr, nstates: self.nrows, context: self.context.clone(), member: None }, _life: PhantomData }
    }
    fn columns_mut(&mut self, start: IndexType, end: IndexType) -> Self::ViewMut<'_> {
        assert!(start <= end && end <= self.ncols, "Column range out of bounds");
        let ptr = unsafe { self.ptr().add(start * self.nrows * self.nb()) };
        HipMatMut { raw: RawMat { ptr, nrows: self.nrows, ncols: end - start, context: self.context.clone() }, _life: PhantomData }
    }
    fn column_mut(&mut self, i: IndexType) -> HipVecMut<'_> {
        assert!(i < self.ncols, "Column index out of bounds");
        let ptr = unsafe { self.ptr().add(i * self.nrows * self.nb()) };
        HipVecMut { raw: RawView { ptr, nstates: self.nrows, context: self.context.clone(), member: None }, _life: PhantomData }
    }
    /// sets entry (i, j) of every member
    fn set_index(&mut self, i: IndexType, j: IndexType, value: Self::T) {
        assert!(i < self.nrows && j < self.ncols, "Index out of bounds");
        check(unsafe { ffi::dsh_vec_set_index_all(self.context.ptr(), self.nb() as i64, self.ptr(), (j * self.nrows + i) as i64, value) }, "dsh_vec_set_index_all");
    }
    fn get_index(&self, i: IndexType, j: IndexType) -> Self::T {
        assert!(self.nb() == 1, "get_index not supported for batched matrices");
        assert!(i < self.nrows && j < self.ncols, "Index out of bounds");
        let mut out = 0.0;
        check(unsafe { ffi::dsh_vec_get_index(self.context.ptr(), 1, self.ptr(), (j * self.nrows + i) as i64, 0, &mut out) }, "dsh_vec_get_index");
        out
    }
    /// keep the leading columns, zero-fill new ones; with the batch-fastest layout the kept columns are ONE contiguous block for all members
    fn resize_cols(&mut self, ncols: IndexType) {
        if ncols == self.ncols {
            return;
        }
        let new = HipMat::zeros(self.nrows, ncols, self.context.clone());
        let keep = ncols.min(self.ncols) * self.nrows * self.nb();
        if keep > 0 {
            check(unsafe { ffi::dsh_d2d(self.context.ptr(), new.buf.ptr, self.buf.ptr, 8 * keep as i64) }, "dsh_d2d");
        }
        *self = new;
    }
    /// host data: member-major, column-major per member (matrix/cuda.rs:897-914)
    fn from_vec(nrows: IndexType, ncols: IndexType, data: Vec<Self::T>, ctx: Self::C) -> Self {
        assert_eq!(data.len(), nrows * ncols * ctx.nbatch());
        let flat = HipVec::from_vec(data, ctx);
        mat_from_flat(flat, nrows, ncols)
    }
}
