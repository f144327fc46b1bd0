//! `HipLU`: `LinearSolver<HipMat>` (crates/diffsol-la/src/linear_solver/mod.rs:19-42) over the batched LU of libdiffsol_hip.so — one launch factors
//! (or solves) every member, where the reference's `CudaLU` loops over the members on the host with one cusolverDnDgetrf / Dgetrs call each
//! (linear_solver/cuda/lu.rs:59-191: the O(nbatch) serial launches per Newton iteration that dominate its batched path).
//! Partial pivoting (LAPACK getrf semantics, what nalgebra's LU and cuSOLVER do); a zero pivot is reported by `solve_in_place` as `LuSolveFailed`.
//! Operands that are banded inside their dense container are found by a probe pass and factored by the banded kernels — half-bandwidth <= 4 (1-D
//! finite-difference PDEs, the single-particle model) one lane per member, 5 .. 64 (2-D PDE discretisations: heat2d, foodweb) one wavefront per member —
//! same bits, O(n k^2) work; `set_structure(ffi::DSH_LU_STRUCTURE_DENSE)` keeps the dense kernels.
use crate::error::{check, to_la_error};
use crate::ffi;
use crate::matrix::HipMat;
use crate::vector::HipVec;
use diffsol_la::error::{LaError, LinearSolverError};
use diffsol_la::matrix::{Matrix, MatrixCommon};
use diffsol_la::{Context, LinearOp, LinearSolver, Vector};
use std::ptr;

pub struct HipLU {
    lu: *mut ffi::dsh_lu,
    matrix: Option<HipMat>,
    linearisation_set: bool,
    structure: i32,
}
// one host thread at a time per solver (like the reference); the factors live on the context's stream.
// SAFETY: the `dsh_lu` handle belongs to the context of `matrix`; it moves with the solver under the contract of context.rs.  (`LinearSolver` itself only asks for
// `Default`, linear_solver/mod.rs:19, but a solver object that owns its linear solver is moved as a whole.)
unsafe impl Send for HipLU {}

impl Default for HipLU {
    fn default() -> Self {
        Self { lu: ptr::null_mut(), matrix: None, linearisation_set: false, structure: ffi::DSH_LU_STRUCTURE_AUTO }
    }
}
impl Drop for HipLU {
    fn drop(&mut self) {
        if !self.lu.is_null() {
            unsafe { ffi::dsh_lu_destroy(self.lu) };
        }
    }
}
impl HipLU {
    /// `DSH_LU_STRUCTURE_AUTO` (default): banded operands use the banded kernels; `DSH_LU_STRUCTURE_DENSE`: always the dense kernels.
    pub fn set_structure(&mut self, structure: i32) {
        self.structure = structure;
        if !self.lu.is_null() {
            check(unsafe { ffi::dsh_lu_set_structure(self.lu, structure) }, "dsh_lu_set_structure");
        }
    }
    /// Number of members whose last factorisation met a zero pivot (cuSOLVER's `info`, linear_solver/cuda/lu.rs:21).
    pub fn n_singular(&self) -> i64 {
        let mut n = 0i64;
        if !self.lu.is_null() {
            check(unsafe { ffi::dsh_lu_info(self.lu, &mut n) }, "dsh_lu_info");
        }
        n
    }
    /// `nrhs` right-hand sides per member with one load of the factors (`b`: an `n x nrhs` batched matrix) — the linear algebra of forward
    /// sensitivities (`Bdf::sensitivity_solve`, bdf.rs:934-989: one solve per parameter with the state equations' factors).
    pub fn solve_multi_in_place(&self, b: &mut HipMat) -> Result<(), LaError> {
        if self.lu.is_null() || !self.linearisation_set {
            return Err(LaError::from(LinearSolverError::LinearSolverNotSetup));
        }
        to_la_error(unsafe { ffi::dsh_lu_solve_multi(self.lu, b.ptr(), b.ncols() as i64) })
    }
}

impl LinearSolver<HipMat> for HipLU {
    fn set_sparsity<C: LinearOp<T = f64, V = HipVec, M = HipMat, C = crate::HipContext>>(&mut self, op: &C) {
        let (nrows, ncols) = (op.nrows(), op.ncols());
        let ctx = op.context().clone();
        self.matrix = Some(HipMat::new_from_sparsity(nrows, ncols, op.sparsity(), ctx.clone()));
        if !self.lu.is_null() {
            unsafe { ffi::dsh_lu_destroy(self.lu) };
            self.lu = ptr::null_mut();
        }
        if nrows == ncols {
            check(unsafe { ffi::dsh_lu_create(ctx.ptr(), nrows as i64, ctx.nbatch() as i64, &mut self.lu) }, "dsh_lu_create");
            check(unsafe { ffi::dsh_lu_set_structure(self.lu, self.structure) }, "dsh_lu_set_structure");
        }
        self.linearisation_set = false;
    }
    fn set_linearisation<C: LinearOp<T = f64, V = HipVec, M = HipMat, C = crate::HipContext>>(&mut self, op: &C) {
        let matrix = self.matrix.as_mut().expect("Matrix not set");
        op.matrix_inplace(matrix);
        assert!(!self.lu.is_null(), "Linear solver matrix not square");
        // the operand is left untouched: the factors live in the solver (n <= 8: batch-fastest registers-per-lane layout, above: system-major)
        check(unsafe { ffi::dsh_lu_factor(self.lu, matrix.ptr()) }, "dsh_lu_factor");
        self.linearisation_set = true;
    }
    fn solve_in_place(&self, x: &mut HipVec) -> Result<(), LaError> {
        let matrix = self.matrix.as_ref().ok_or(LaError::from(LinearSolverError::LinearSolverNotSetup))?;
        if matrix.nrows() != matrix.ncols() {
            return Err(LaError::from(LinearSolverError::LinearSolverMatrixNotSquare));
        }
        if !self.linearisation_set {
            return Err(LaError::from(LinearSolverError::LinearSolverNotSetup));
        }
        if x.len() != matrix.nrows() || x.context().nbatch() != matrix.context().nbatch() {
            return Err(LaError::from(LinearSolverError::LinearSolverMatrixVectorNotCompatible));
        }
        to_la_error(unsafe { ffi::dsh_lu_solve(self.lu, x.ptr()) })
    }
}
