//! `diffsol-hip` — MI355X (gfx950) backend for diffsol's generic integrators.
//!
//! diffsol's `Bdf<'a, Eqn, LS, M>` / `Sdirk<..>` are generic over `M: DenseMatrix`, `Eqn::V: DefaultDenseMatrix`, `LS: LinearSolver<Eqn::M>`
//! (crates/diffsol/src/ode_solver/bdf.rs:992-1001).  This crate provides those types over the C ABI of `libdiffsol_hip.so`
//! (`include/diffsol_hip.h`; `src/ffi.rs` is generated from that header):
//!
//! | diffsol-la trait | type here | replaces (reference CUDA backend) |
//! |---|---|---|
//! | `Context` | [`HipContext`] | `CudaContext` context/cuda.rs:41-144 |
//! | `Vector` / `VectorView` / `VectorViewMut` / `VectorIndex` / `DefaultDenseMatrix` | [`HipVec`], [`HipVecRef`], [`HipVecMut`], [`HipIndex`] | vector/cuda.rs:741-1490 |
//! | `Matrix` / `DenseMatrix` / `MatrixView` / `MatrixViewMut` / `DefaultSolver` | [`HipMat`], [`HipMatRef`], [`HipMatMut`] | matrix/cuda.rs:848-1468 |
//! | `LinearSolver<HipMat>` | [`HipLU`] | linear_solver/cuda/lu.rs:59-191 |
//! | `OdeEquations` for device models | [`HipModelEquations`] | the batched closures of test_models/*.rs |
//!
//! Layout: the device stores batched data batch-FASTEST (element `i` of member `b` at `p[i * nbatch + b]`), the reference API is batch-major
//! (`from_vec` / `clone_as_vec`); the transposition happens in `dsh_vec_upload` / `dsh_vec_download`, so user code sees the reference layout.
//!
//! ```ignore
//! use diffsol::{OdeBuilder, OdeSolverMethod};
//! use diffsol_hip::{HipContext, HipLU, HipMat, HipModelEquations, Model};
//! let ctx = HipContext::new(0)?.clone_with_nbatch(100_000)?;
//! let eqn = HipModelEquations::registry(Model::RobertsonOde, 1, params, ctx.clone());
//! let problem = OdeBuilder::<HipMat>::new().rtol(1e-4).atol([1e-8, 1e-14, 1e-6]).context(ctx).build_from_eqn(eqn)?;
//! let mut solver = problem.bdf::<HipLU>()?;                       // diffsol's own Bdf, every Vector / Matrix / LU call on the GPU
//! let (ys, ts) = solver.solve_dense(&t_eval)?;
//! // or the whole ensemble solve in one launch (device-resident step control):
//! let out = diffsol_hip::ensemble::solve_dense_ensemble(&problem, Method::Bdf, &t_eval, EnsembleMode::Wavefront)?;
//! ```
pub mod context;
pub mod diffsl;
pub mod ensemble;
pub mod equations;
pub mod error;
#[allow(clippy::all)]
pub mod ffi;
pub mod lu;
pub mod matrix;
pub mod vector;

pub use context::HipContext;
pub use ensemble::{solve_dense_ensemble, solve_dense_sensitivities_ensemble, EnsembleMode, EnsembleSensSolution, EnsembleSolution, Method};
pub use equations::{HipModelEquations, Model};
pub use lu::HipLU;
pub use matrix::{HipMat, HipMatMut, HipMatRef};
pub use vector::{HipIndex, HipVec, HipVecMut, HipVecRef};
