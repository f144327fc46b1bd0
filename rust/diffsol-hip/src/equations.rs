//! Device-side `OdeEquations`: the model plug-in of the hot path (crates/diffsol/src/ode_equations/mod.rs:245-329).  A model is addressed by a
//! registry id of libdiffsol_hip.so — one of the built-in test / example models, or a run-time-compiled DiffSL model (`crate::diffsl`) — and every
//! operator is ONE launch over the whole ensemble (`dsh_model_rhs`, `dsh_model_jac_mul`, `dsh_model_jacobian`, `dsh_model_mass_gemv`, ...), where the
//! reference's batched closures loop over the members on the host (test_models/exponential_decay.rs:16-20).
//! Shape follows examples/custom-ode-equations/src/*.rs: an equations struct plus per-operator view structs implementing `Op` + the operator trait.
//! Host closures cannot run on the device: a user model goes through DiffSL (or is added to the registry).
use crate::context::HipContext;
use crate::error::check;
use crate::ffi;
use crate::matrix::HipMat;
use crate::vector::HipVec;
use diffsol::{ConstantOp, LinearOp, NonLinearOp, NonLinearOpJacobian, OdeEquations, OdeEquationsRef, Op, UnitCallable};
use diffsol_la::matrix::Matrix;
use diffsol_la::{Context, Vector};
use std::cell::Cell;

/// Built-in models of the registry (include/diffsol_hip.h DSH_MODEL_*: the reference's own test and example problems).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Model {
    ExponentialDecay = ffi::DSH_MODEL_EXPONENTIAL_DECAY as isize,
    ExponentialDecayAlgebraic = ffi::DSH_MODEL_EXPONENTIAL_DECAY_ALGEBRAIC as isize,
    ExponentialDecayAlgebraicBatched = ffi::DSH_MODEL_EXPONENTIAL_DECAY_ALGEBRAIC_BATCHED as isize,
    RobertsonOde = ffi::DSH_MODEL_ROBERTSON_ODE as isize,
    RobertsonDae = ffi::DSH_MODEL_ROBERTSON_DAE as isize,
    DydtY2 = ffi::DSH_MODEL_DYDT_Y2 as isize,
    GaussianDecay = ffi::DSH_MODEL_GAUSSIAN_DECAY as isize,
    Heat1d = ffi::DSH_MODEL_HEAT1D as isize,
    Rlc = ffi::DSH_MODEL_RLC as isize,
    ExponentialDecayRoot = ffi::DSH_MODEL_EXPONENTIAL_DECAY_ROOT as isize,
    Spm = ffi::DSH_MODEL_SPM as isize,
    /// 2-D heat equation (test_models/heat2d.rs), `size x size` grid, boundary rows algebraic, half-bandwidth `size`: factored by the general banded LU
    Heat2d = ffi::DSH_MODEL_HEAT2D as isize,
    /// predator-prey food web (test_models/foodweb.rs), `size x size` grid, predators algebraic, half-bandwidth `2 size`
    Foodweb = ffi::DSH_MODEL_FOODWEB as isize,
}

/// Operator call counters like the reference's `OpStatistics` (op/mod.rs:95-128).
#[derive(Default, Debug)]
pub struct ModelStatistics {
    pub number_of_calls: Cell<usize>,
    pub number_of_jac_muls: Cell<usize>,
    pub number_of_matrix_evals: Cell<usize>,
}

pub struct HipModelEquations {
    pub(crate) model: i32,
    pub(crate) size: i64,
    pub(crate) nstates: usize,
    pub(crate) nparams: usize,
    pub(crate) nroots: usize,
    pub(crate) has_mass: bool,
    pub(crate) p: HipVec,
    pub(crate) ctx: HipContext,
    pub statistics: ModelStatistics,
    /// a run-time-compiled model is released when its equations are dropped
    owns_model: bool,
}

impl HipModelEquations {
    /// A built-in model; `params` is batch-major, `nparams` values per member (like `OdeBuilder::p`), `ctx.nbatch()` members.
    pub fn registry(model: Model, size: i64, params: Vec<f64>, ctx: HipContext) -> Self {
        Self::from_id(model as i32, size, params, ctx, false)
    }
    /// A model id returned by `dsh_model_compile` (see `crate::diffsl::compile`).
    pub fn from_id(model: i32, size: i64, params: Vec<f64>, ctx: HipContext, owns_model: bool) -> Self {
        let (mut n, mut np, mut nroots, mut hm) = (0i64, 0i64, 0i64, 0i32);
        check(unsafe { ffi::dsh_model_info(model, size, &mut n, &mut np, &mut hm, &mut nroots) }, "dsh_model_info");
        assert_eq!(params.len(), np as usize * ctx.nbatch(), "expected {} parameters per member x {} members", np, ctx.nbatch());
        let p = if params.is_empty() { HipVec::zeros(0, ctx.clone()) } else { HipVec::from_vec(params, ctx.clone()) };
        Self { model, size, nstates: n as usize, nparams: np as usize, nroots: nroots as usize, has_mass: hm != 0, p, ctx, statistics: Default::default(), owns_model }
    }
    pub fn model_id(&self) -> i32 {
        self.model
    }
    pub fn model_size(&self) -> i64 {
        self.size
    }
    pub fn params(&self) -> &HipVec {
        &self.p
    }
    fn nb(&self) -> i64 {
        self.ctx.nbatch() as i64
    }
    /// Does the model provide parameter derivatives (forward sensitivities, `OdeEquationsImplicitSens`)?
    pub fn has_sens(&self) -> bool {
        unsafe { ffi::dsh_model_has_sens(self.model, self.size) != 0 }
    }
    /// `df/dp` at `(x, t)` as an `nstates x nparams` batched matrix in one launch (`NonLinearOpSens::sens_inplace`, op/nonlinear_op.rs:67-81).
    pub fn rhs_sens_inplace(&self, x: &HipVec, t: f64, sens: &mut HipMat) {
        check(unsafe { ffi::dsh_model_rhs_sens(self.ctx.ptr(), self.model, self.size, self.nb(), t, x.ptr(), self.p.ptr(), sens.ptr()) }, "dsh_model_rhs_sens");
    }
    /// `dy0/dp` as an `nstates x nparams` batched matrix (`SensInit`, ode_equations/sens_equations.rs:62-70).
    pub fn init_sens_inplace(&self, t: f64, sens0: &mut HipMat) {
        check(unsafe { ffi::dsh_model_init_sens(self.ctx.ptr(), self.model, self.size, self.nb(), t, self.p.ptr(), sens0.ptr()) }, "dsh_model_init_sens");
    }
}
impl Drop for HipModelEquations {
    fn drop(&mut self) {
        if self.owns_model {
            unsafe { ffi::dsh_model_release(self.model) };
        }
    }
}

macro_rules! impl_op {
    ($t:ty, $nout:expr) => {
        impl Op for $t {
            type T = f64;
            type V = HipVec;
            type M = HipMat;
            type C = HipContext;
            fn nstates(&self) -> usize {
                self.eqn().nstates
            }
            fn nout(&self) -> usize {
                let e = self.eqn();
                #[allow(clippy::redundant_closure_call)]
                ($nout)(e)
            }
            fn nparams(&self) -> usize {
                self.eqn().nparams
            }
            fn context(&self) -> &Self::C {
                &self.eqn().ctx
            }
        }
    };
}

impl HipModelEquations {
    fn eqn(&self) -> &HipModelEquations {
        self
    }
}
impl_op!(HipModelEquations, |e: &HipModelEquations| e.nstates);

pub struct ModelRhs<'a>(pub(crate) &'a HipModelEquations);
pub struct ModelMass<'a>(pub(crate) &'a HipModelEquations);
pub struct ModelInit<'a>(pub(crate) &'a HipModelEquations);
pub struct ModelRoot<'a>(pub(crate) &'a HipModelEquations);
macro_rules! impl_eqn {
    ($t:ident) => {
        impl $t<'_> {
            fn eqn(&self) -> &HipModelEquations {
                self.0
            }
        }
    };
}
impl_eqn!(ModelRhs);
impl_eqn!(ModelMass);
impl_eqn!(ModelInit);
impl_eqn!(ModelRoot);
impl_op!(ModelRhs<'_>, |e: &HipModelEquations| e.nstates);
impl_op!(ModelMass<'_>, |e: &HipModelEquations| e.nstates);
impl_op!(ModelInit<'_>, |e: &HipModelEquations| e.nstates);
impl_op!(ModelRoot<'_>, |e: &HipModelEquations| e.nroots);

impl NonLinearOp for ModelRhs<'_> {
    fn call_inplace(&self, x: &HipVec, t: f64, y: &mut HipVec) {
        let e = self.0;
        e.statistics.number_of_calls.set(e.statistics.number_of_calls.get() + 1);
        check(unsafe { ffi::dsh_model_rhs(e.ctx.ptr(), e.model, e.size, e.nb(), t, x.ptr(), e.p.ptr(), y.ptr()) }, "dsh_model_rhs");
    }
}
impl NonLinearOpJacobian for ModelRhs<'_> {
    fn jac_mul_inplace(&self, x: &HipVec, t: f64, v: &HipVec, y: &mut HipVec) {
        let e = self.0;
        e.statistics.number_of_jac_muls.set(e.statistics.number_of_jac_muls.get() + 1);
        check(unsafe { ffi::dsh_model_jac_mul(e.ctx.ptr(), e.model, e.size, e.nb(), t, x.ptr(), e.p.ptr(), v.ptr(), y.ptr()) }, "dsh_model_jac_mul");
    }
    /// the whole Jacobian in one launch (the default implementation would cost n jac_mul launches + n set_column, op/nonlinear_op.rs:211-219)
    fn jacobian_inplace(&self, x: &HipVec, t: f64, y: &mut HipMat) {
        let e = self.0;
        e.statistics.number_of_matrix_evals.set(e.statistics.number_of_matrix_evals.get() + 1);
        check(unsafe { ffi::dsh_model_jacobian(e.ctx.ptr(), e.model, e.size, e.nb(), t, x.ptr(), e.p.ptr(), y.ptr()) }, "dsh_model_jacobian");
    }
}
impl LinearOp for ModelMass<'_> {
    /// y = M x + beta y
    fn gemv_inplace(&self, x: &HipVec, t: f64, beta: f64, y: &mut HipVec) {
        let e = self.0;
        check(unsafe { ffi::dsh_model_mass_gemv(e.ctx.ptr(), e.model, e.size, e.nb(), t, x.ptr(), e.p.ptr(), beta, y.ptr()) }, "dsh_model_mass_gemv");
    }
    fn matrix_inplace(&self, t: f64, y: &mut HipMat) {
        let e = self.0;
        check(unsafe { ffi::dsh_model_mass_matrix(e.ctx.ptr(), e.model, e.size, e.nb(), t, e.p.ptr(), y.ptr()) }, "dsh_model_mass_matrix");
    }
}
impl ConstantOp for ModelInit<'_> {
    fn call_inplace(&self, t: f64, y: &mut HipVec) {
        let e = self.0;
        check(unsafe { ffi::dsh_model_init(e.ctx.ptr(), e.model, e.size, e.nb(), t, e.p.ptr(), y.ptr()) }, "dsh_model_init");
    }
}
impl NonLinearOp for ModelRoot<'_> {
    fn call_inplace(&self, x: &HipVec, t: f64, y: &mut HipVec) {
        let e = self.0;
        check(unsafe { ffi::dsh_model_root(e.ctx.ptr(), e.model, e.size, e.nb(), t, x.ptr(), e.p.ptr(), y.ptr()) }, "dsh_model_root");
    }
}

impl<'a> OdeEquationsRef<'a> for HipModelEquations {
    type Rhs = ModelRhs<'a>;
    type Mass = ModelMass<'a>;
    type Init = ModelInit<'a>;
    type Root = ModelRoot<'a>;
    type Out = UnitCallable<HipMat>;
    type Reset = UnitCallable<HipMat>;
}
impl OdeEquations for HipModelEquations {
    fn rhs(&self) -> ModelRhs<'_> {
        ModelRhs(self)
    }
    fn mass(&self) -> Option<ModelMass<'_>> {
        self.has_mass.then_some(ModelMass(self))
    }
    fn init(&self) -> ModelInit<'_> {
        ModelInit(self)
    }
    fn root(&self) -> Option<ModelRoot<'_>> {
        (self.nroots > 0).then_some(ModelRoot(self))
    }
    fn out(&self) -> Option<UnitCallable<HipMat>> {
        None
    }
    fn reset(&self) -> Option<UnitCallable<HipMat>> {
        None
    }
    fn set_params(&mut self, p: &HipVec) {
        self.p.copy_from(p);
    }
    fn get_params(&self, p: &mut HipVec) {
        p.copy_from(&self.p);
    }
}
