//! `solve_dense` of a whole ensemble in ONE launch: the device-resident integrators of libdiffsol_hip.so
//! (`dsh_bdf_solve_adaptive`, `dsh_sdirk_solve_resident`, `dsh_bdf_solve_wave_member`, `dsh_sdirk_solve_wave_member`).  diffsol's generic `Bdf` / `Sdirk` on `HipVec` / `HipMat` /
//! `HipLU` advance the ensemble in lock-step with one host round trip per reduction; here `Bdf::step` / `Sdirk::step`, `NewtonNonlinearSolver`,
//! `Convergence`, `JacobianUpdate`, `RootFinder` and `OdeSolverMethod::solve_dense` (method.rs:467-520) run per member — or per 64-member wavefront
//! group, the reference's batched semantics at nbatch = 64 — with the solver state in registers.  Results are bit-identical to the CPU restatement
//! of the reference algorithm (oracle/, tests/test_gpu_adaptive.py).
use crate::equations::HipModelEquations;
use crate::error::{check, last_error};
use crate::ffi;
use crate::matrix::HipMat;
use diffsol::OdeSolverProblem;
use diffsol_la::error::LaError;
use diffsol_la::matrix::Matrix;
use diffsol_la::{Context, Vector};
use std::os::raw::c_void;
use std::ptr;

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Method {
    Bdf = 0,
    TrBdf2 = 1,
    Esdirk34 = 2,
}
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum EnsembleMode {
    /// every member its own step-size / order history and its own event time: diffsol's CPU semantics for a sweep of independent IVPs
    PerMember = 1,
    /// the 64 members of a wavefront in lock-step (max-norms over the group): the reference's batched semantics with nbatch = 64
    Wavefront = 64,
}

/// Interpolated solution at `t_eval` plus per-member bookkeeping.
pub struct EnsembleSolution {
    /// `nstates x t_eval.len()` batched matrix: column k = the states at `t_eval[k]` (NaN after a member's own stop)
    pub ys: HipMat,
    /// 0 = ok, else the `OdeSolverError` ordinal of the member's failure (20: members of a lock-step group disagree on a root, 99: step guard)
    pub status: Vec<i32>,
    /// root time (NaN if none), root index (-1 if none) and number of valid columns per member
    pub t_root: Vec<f64>,
    pub root_index: Vec<i32>,
    pub ncols: Vec<i32>,
    /// [steps, Newton iterations, LU setups, error-test failures, Newton failures] per member
    pub stats: Vec<[i32; 5]>,
    /// the same counters summed over members + number of failed members
    pub totals: [i64; 6],
}

/// `OdeSolverOptions` / `InitialConditionSolverOptions` of the problem as the kernels' option block
fn adaptive_options(problem: &OdeSolverProblem<HipModelEquations>, mode: EnsembleMode) -> ffi::dsh_adaptive_options {
    let mut o = std::mem::MaybeUninit::<ffi::dsh_adaptive_options>::uninit();
    unsafe { ffi::dsh_adaptive_default_options(o.as_mut_ptr()) };
    let mut o = unsafe { o.assume_init() };
    let oo = &problem.ode_options;
    o.max_nonlinear_solver_iterations = oo.max_nonlinear_solver_iterations as i32;
    o.max_error_test_failures = oo.max_error_test_failures as i32;
    o.max_nonlinear_solver_failures = oo.max_nonlinear_solver_failures as i32;
    o.nonlinear_solver_tolerance = oo.nonlinear_solver_tolerance;
    o.min_timestep = oo.min_timestep;
    o.update_jacobian_after_steps = oo.update_jacobian_after_steps as i32;
    o.update_rhs_jacobian_after_steps = oo.update_rhs_jacobian_after_steps as i32;
    o.threshold_to_update_jacobian = oo.threshold_to_update_jacobian;
    o.threshold_to_update_rhs_jacobian = oo.threshold_to_update_rhs_jacobian;
    let ic = &problem.ic_options;
    o.ic_use_linesearch = ic.use_linesearch as i32;
    o.ic_max_linesearch_iterations = ic.max_linesearch_iterations as i32;
    o.ic_max_linear_solver_setups = ic.max_linear_solver_setups as i32;
    o.ic_max_newton_iterations = ic.max_newton_iterations as i32;
    o.ic_step_reduction_factor = ic.step_reduction_factor;
    o.ic_armijo_constant = ic.armijo_constant;
    o.group = mode as i32;
    o
}

/// `problem.bdf()/tr_bdf2()/esdirk34()` + `solve_dense(t_eval)` for every member, on the device.  Uses the problem's tolerances, options, t0 and h0.
pub fn solve_dense_ensemble(problem: &OdeSolverProblem<HipModelEquations>, method: Method, t_eval: &[f64], mode: EnsembleMode) -> Result<EnsembleSolution, LaError> {
    let eqn = &problem.eqn;
    let ctx = eqn.ctx.clone();
    let (nb, n, nt) = (ctx.nbatch(), eqn.nstates, t_eval.len());
    let o = adaptive_options(problem, mode);
    let ys = HipMat::zeros(n, nt, ctx.clone());
    let c = ctx.ptr();
    let alloc = |bytes: usize| -> *mut c_void {
        let mut p = ptr::null_mut();
        check(unsafe { ffi::dsh_malloc(c, bytes as i64, 0, &mut p) }, "dsh_malloc");
        p
    };
    let (stats_d, status_d, troot_d, ridx_d, ncols_d) = (alloc(20 * nb), alloc(4 * nb), alloc(8 * nb), alloc(4 * nb), alloc(4 * nb));
    let mut totals = [0i64; 6];
    let atol = &problem.atol;
    // which kernel: the banded lane-per-member twin of a run-time-sized model, the register-resident kernels (n <= 4), or one wavefront per member (n <= 64)
    let twin = unsafe { ffi::dsh_model_lane_twin(eqn.model, eqn.size) };
    let (model, size) = if twin >= 0 && unsafe { ffi::dsh_model_has_resident(method as i32, twin, 0) } != 0 { (twin, 0) } else { (eqn.model, eqn.size) };
    let rc = if unsafe { ffi::dsh_model_has_resident(method as i32, model, size) } != 0 {
        if method == Method::Bdf {
            unsafe {
                ffi::dsh_bdf_solve_adaptive(
                    c, model, size, nb as i64, eqn.p.ptr(), atol.ptr(), atol.context().nbatch() as i64, problem.rtol, problem.t0, problem.h0, &o, t_eval.as_ptr(), nt as i64, ys.ptr(),
                    stats_d as *mut i32, status_d as *mut i32, troot_d as *mut f64, ridx_d as *mut i32, ncols_d as *mut i32, totals.as_mut_ptr(),
                )
            }
        } else {
            unsafe {
                ffi::dsh_sdirk_solve_resident(
                    c, method as i32, model, size, nb as i64, eqn.p.ptr(), atol.ptr(), atol.context().nbatch() as i64, problem.rtol, problem.t0, problem.h0, &o, t_eval.as_ptr(), nt as i64,
                    ys.ptr(), stats_d as *mut i32, status_d as *mut i32, troot_d as *mut f64, ridx_d as *mut i32, ncols_d as *mut i32, totals.as_mut_ptr(),
                )
            }
        }
    } else if method == Method::Bdf && mode == EnsembleMode::PerMember && unsafe { ffi::dsh_model_has_wave_member(eqn.model, eqn.size) } != 0 {
        // one wavefront per member: dense run-time-sized models with n <= 64, DiffSL DAEs (n <= 48) included — made consistent on the device
        unsafe {
            ffi::dsh_bdf_solve_wave_member(
                c, eqn.model, eqn.size, nb as i64, eqn.p.ptr(), atol.ptr(), atol.context().nbatch() as i64, problem.rtol, problem.t0, problem.h0, &o, t_eval.as_ptr(), nt as i64, ys.ptr(),
                stats_d as *mut i32, status_d as *mut i32, troot_d as *mut f64, ridx_d as *mut i32, ncols_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    } else if method != Method::Bdf && mode == EnsembleMode::PerMember && unsafe { ffi::dsh_model_has_wave_member_sdirk(eqn.model, eqn.size) } != 0 {
        // the same for TR-BDF2 / ESDIRK34
        unsafe {
            ffi::dsh_sdirk_solve_wave_member(
                c, eqn.model, eqn.size, method as i32, nb as i64, eqn.p.ptr(), atol.ptr(), atol.context().nbatch() as i64, problem.rtol, problem.t0, problem.h0, &o, t_eval.as_ptr(), nt as i64,
                ys.ptr(), stats_d as *mut i32, status_d as *mut i32, troot_d as *mut f64, ridx_d as *mut i32, ncols_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    } else {
        ffi::DSH_E_UNSUPPORTED
    };
    let fetch = |dev: *mut c_void, host: *mut c_void, bytes: usize| check(unsafe { ffi::dsh_d2h(c, host, dev, bytes as i64) }, "dsh_d2h");
    let mut out = EnsembleSolution { ys, status: vec![0; nb], t_root: vec![0.0; nb], root_index: vec![0; nb], ncols: vec![0; nb], stats: vec![[0; 5]; nb], totals };
    if rc >= 0 {
        let mut flat = vec![0i32; 5 * nb];
        fetch(stats_d, flat.as_mut_ptr() as *mut c_void, 20 * nb);
        for b in 0..nb {
            for k in 0..5 {
                out.stats[b][k] = flat[k * nb + b];
            }
        }
        fetch(status_d, out.status.as_mut_ptr() as *mut c_void, 4 * nb);
        fetch(troot_d, out.t_root.as_mut_ptr() as *mut c_void, 8 * nb);
        fetch(ridx_d, out.root_index.as_mut_ptr() as *mut c_void, 4 * nb);
        fetch(ncols_d, out.ncols.as_mut_ptr() as *mut c_void, 4 * nb);
    }
    for p in [stats_d, status_d, troot_d, ridx_d, ncols_d] {
        unsafe { ffi::dsh_free(c, p) };
    }
    if rc < 0 {
        return Err(LaError::Other(if rc == ffi::DSH_E_UNSUPPORTED && last_error().is_empty() {
            "no device-resident kernel for this model / method (static models with n <= 4: BDF, TR-BDF2, ESDIRK34; run-time-sized ODE models with n <= 64: BDF)".into()
        } else {
            last_error()
        }));
    }
    Ok(out)
}

/// What `OdeSolverMethod::solve` returns, for every member of an ensemble: the state after every accepted step.
pub struct EnsembleStepsSolution {
    /// `nstates x max_cols` batched matrix: member b's solution is the first `ncols[b]` columns of its slice (column 0 = the initial state)
    pub ys: HipMat,
    /// `max_cols x nbatch` times, batch-fastest on the host as well: member b's time of column k at `ts[k * nbatch + b]`
    pub ts: Vec<f64>,
    /// columns member b PRODUCED (more than `max_cols`: not all of them were stored — call again with more room)
    pub ncols: Vec<i32>,
    pub status: Vec<i32>,
    pub t_root: Vec<f64>,
    pub root_index: Vec<i32>,
    pub totals: [i64; 6],
}

/// `problem.bdf()/tr_bdf2()/esdirk34()` + `solve(t_final)` (method.rs:227-258: the state after EVERY accepted step) for every member, in one launch.  Routed like
/// `dshs_solve_adaptive` (host/solver_c.cpp): the banded lane-per-member twin of a run-time-sized model or the register-resident kernels (static models, n <= 4) through
/// `dsh_bdf_solve_adaptive_steps` / `dsh_sdirk_solve_resident_steps`; dense run-time-sized models per member (n <= 320) through `dsh_bdf_solve_wave_member_steps` /
/// `dsh_sdirk_solve_wave_member_steps` (one wavefront or one workgroup per member: control is always per member there).  Uses the problem's tolerances, options, t0 and h0.
pub fn solve_ensemble(problem: &OdeSolverProblem<HipModelEquations>, method: Method, t_final: f64, max_cols: usize, mode: EnsembleMode) -> Result<EnsembleStepsSolution, LaError> {
    let eqn = &problem.eqn;
    let ctx = eqn.ctx.clone();
    let (nb, n) = (ctx.nbatch(), eqn.nstates);
    // which kernel family writes every step for this model and method (the order of solve_dense_ensemble above)
    let twin = unsafe { ffi::dsh_model_lane_twin(eqn.model, eqn.size) };
    let (model, size) = if twin >= 0 && unsafe { ffi::dsh_model_has_resident(method as i32, twin, 0) } != 0 { (twin, 0) } else { (eqn.model, eqn.size) };
    let resident = if method == Method::Bdf {
        unsafe { ffi::dsh_model_has_adaptive_steps(model, size) } != 0
    } else {
        unsafe { ffi::dsh_model_has_resident(method as i32, model, size) } != 0
    };
    let wave_member = !resident
        && mode == EnsembleMode::PerMember
        && (if method == Method::Bdf { unsafe { ffi::dsh_model_has_wave_member(eqn.model, eqn.size) } } else { unsafe { ffi::dsh_model_has_wave_member_sdirk(eqn.model, eqn.size) } }) != 0;
    if !resident && !wave_member {
        return Err(LaError::Other(
            "solve_ensemble: no device-resident integrator that writes every step for this model and method (register-resident static models, banded lane-per-member forms, \
             wavefront / workgroup per member with EnsembleMode::PerMember); Bdf::solve / Sdirk::solve on HipVec / HipMat / HipLU walk the same steps host-driven"
                .into(),
        ));
    }
    let o = adaptive_options(problem, mode);
    let ys = HipMat::zeros(n, max_cols, ctx.clone());
    let c = ctx.ptr();
    let alloc = |bytes: usize| -> *mut c_void {
        let mut p = ptr::null_mut();
        check(unsafe { ffi::dsh_malloc(c, bytes as i64, 0, &mut p) }, "dsh_malloc");
        p
    };
    let (ts_d, status_d, troot_d, ridx_d, ncols_d) = (alloc(8 * max_cols * nb), alloc(4 * nb), alloc(8 * nb), alloc(4 * nb), alloc(4 * nb));
    let mut totals = [0i64; 6];
    let atol = &problem.atol;
    let anb = atol.context().nbatch() as i64;
    let rc = if wave_member && method != Method::Bdf {
        unsafe {
            ffi::dsh_sdirk_solve_wave_member_steps(
                c, eqn.model, eqn.size, method as i32, nb as i64, eqn.p.ptr(), atol.ptr(), anb, problem.rtol, problem.t0, problem.h0, &o, t_final, max_cols as i64, ys.ptr(),
                ts_d as *mut f64, ptr::null_mut(), status_d as *mut i32, troot_d as *mut f64, ridx_d as *mut i32, ncols_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    } else if wave_member {
        unsafe {
            ffi::dsh_bdf_solve_wave_member_steps(
                c, eqn.model, eqn.size, nb as i64, eqn.p.ptr(), atol.ptr(), anb, problem.rtol, problem.t0, problem.h0, &o, t_final, max_cols as i64, ys.ptr(),
                ts_d as *mut f64, ptr::null_mut(), status_d as *mut i32, troot_d as *mut f64, ridx_d as *mut i32, ncols_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    } else if method != Method::Bdf {
        unsafe {
            ffi::dsh_sdirk_solve_resident_steps(
                c, method as i32, model, size, nb as i64, eqn.p.ptr(), atol.ptr(), anb, problem.rtol, problem.t0, problem.h0, &o, t_final, max_cols as i64, ys.ptr(),
                ts_d as *mut f64, ptr::null_mut(), status_d as *mut i32, troot_d as *mut f64, ridx_d as *mut i32, ncols_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    } else {
        unsafe {
            ffi::dsh_bdf_solve_adaptive_steps(
                c, model, size, nb as i64, eqn.p.ptr(), atol.ptr(), anb, problem.rtol, problem.t0, problem.h0, &o, t_final, max_cols as i64, ys.ptr(),
                ts_d as *mut f64, ptr::null_mut(), status_d as *mut i32, troot_d as *mut f64, ridx_d as *mut i32, ncols_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    };
    let fetch = |dev: *mut c_void, host: *mut c_void, bytes: usize| check(unsafe { ffi::dsh_d2h(c, host, dev, bytes as i64) }, "dsh_d2h");
    let mut out = EnsembleStepsSolution { ys, ts: vec![0.0; max_cols * nb], ncols: vec![0; nb], status: vec![0; nb], t_root: vec![0.0; nb], root_index: vec![0; nb], totals };
    if rc >= 0 {
        fetch(ts_d, out.ts.as_mut_ptr() as *mut c_void, 8 * max_cols * nb);
        fetch(ncols_d, out.ncols.as_mut_ptr() as *mut c_void, 4 * nb);
        fetch(status_d, out.status.as_mut_ptr() as *mut c_void, 4 * nb);
        fetch(troot_d, out.t_root.as_mut_ptr() as *mut c_void, 8 * nb);
        fetch(ridx_d, out.root_index.as_mut_ptr() as *mut c_void, 4 * nb);
    }
    for p in [ts_d, status_d, troot_d, ridx_d, ncols_d] {
        unsafe { ffi::dsh_free(c, p) };
    }
    if rc < 0 {
        return Err(LaError::Other(last_error()));
    }
    Ok(out)
}

/// States and forward sensitivities of a whole ensemble at `t_eval` from one launch.
pub struct EnsembleSensSolution {
    /// `nstates x t_eval.len()` batched matrix of the states
    pub ys: HipMat,
    /// one `nstates x t_eval.len()` batched matrix per parameter: `dy/dp_j` at `t_eval[k]` in column k (the `Vec<M>` of `solve_dense_sensitivities`)
    pub sens: Vec<HipMat>,
    pub status: Vec<i32>,
    pub stats: Vec<[i32; 5]>,
    pub totals: [i64; 6],
}

/// `problem.bdf_sens()/tr_bdf2_sens()/esdirk34_sens()` + `SensitivitiesOdeSolverMethod::solve_dense_sensitivities(t_eval)` (sensitivities.rs:114-260) for every member,
/// on the device: `dsh_bdf_solve_adaptive_sens` / `dsh_sdirk_solve_resident_sens` integrate `s_j = dy/dp_j` of every parameter alongside the states in the same launch
/// (`Bdf::sensitivity_solve` bdf.rs:934-989, the sensitivity half of `do_stage_sdirk` runge_kutta.rs:691-748).  Models in the register-resident form with parameter
/// derivatives, n <= 4, no mass matrix, no root functions (`dsh_model_has_adaptive_sens`); everything else integrates its sensitivities through the trait seam.
/// `sens_tol = None` is `turn_off_sensitivities_error_control`.
pub fn solve_dense_sensitivities_ensemble(
    problem: &OdeSolverProblem<HipModelEquations>, method: Method, t_eval: &[f64], mode: EnsembleMode, sens_tol: Option<(f64, &[f64])>,
) -> Result<EnsembleSensSolution, LaError> {
    let eqn = &problem.eqn;
    let ctx = eqn.ctx.clone();
    let (nb, n, np, nt) = (ctx.nbatch(), eqn.nstates, eqn.nparams, t_eval.len());
    // which kernel: the banded lane-per-member twin of a run-time-sized model, the register-resident kernels (n <= 4), or — dense run-time-compiled models with
    // n <= 64, BDF per member — one wavefront per member
    let twin = unsafe { ffi::dsh_model_lane_twin(eqn.model, eqn.size) };
    let (model, size) = if twin >= 0 && unsafe { ffi::dsh_model_has_adaptive_sens(twin, 0) } != 0 { (twin, 0) } else { (eqn.model, eqn.size) };
    let wave_member = unsafe { ffi::dsh_model_has_adaptive_sens(model, size) } == 0
        && mode == EnsembleMode::PerMember
        && unsafe { ffi::dsh_model_has_wave_member_sens(eqn.model, eqn.size) } != 0;
    if !wave_member && unsafe { ffi::dsh_model_has_adaptive_sens(model, size) } == 0 {
        return Err(LaError::Other("no device-resident integrator with forward sensitivities for this model (ODE model with parameter derivatives and no root functions: register-resident n <= 4, banded lane form, or dense n <= 140 per member)".into()));
    }
    let o = adaptive_options(problem, mode);
    let ys = HipMat::zeros(n, nt, ctx.clone());
    let c = ctx.ptr();
    let alloc = |bytes: usize| -> *mut c_void {
        let mut p = ptr::null_mut();
        check(unsafe { ffi::dsh_malloc(c, bytes as i64, 0, &mut p) }, "dsh_malloc");
        p
    };
    // the kernels write [save point][parameter][state][member]; one n x nt matrix per parameter is gathered from it below
    let sens_d = alloc(8 * nt * np * n * nb);
    let (stats_d, status_d) = (alloc(20 * nb), alloc(4 * nb));
    let mut totals = [0i64; 6];
    let atol = &problem.atol;
    let (srtol, satol): (f64, &[f64]) = sens_tol.unwrap_or((0.0, &[]));
    let rc = if wave_member && method != Method::Bdf {
        unsafe {
            ffi::dsh_sdirk_solve_wave_member_sens(
                c, eqn.model, eqn.size, method as i32, nb as i64, eqn.p.ptr(), atol.ptr(), atol.context().nbatch() as i64, problem.rtol, problem.t0, problem.h0, &o, t_eval.as_ptr(),
                nt as i64, srtol, satol.as_ptr(), satol.len() as i64, ys.ptr(), sens_d as *mut f64, stats_d as *mut i32, status_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    } else if wave_member {
        unsafe {
            ffi::dsh_bdf_solve_wave_member_sens(
                c, eqn.model, eqn.size, nb as i64, eqn.p.ptr(), atol.ptr(), atol.context().nbatch() as i64, problem.rtol, problem.t0, problem.h0, &o, t_eval.as_ptr(), nt as i64, srtol,
                satol.as_ptr(), satol.len() as i64, ys.ptr(), sens_d as *mut f64, stats_d as *mut i32, status_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    } else if method == Method::Bdf {
        unsafe {
            ffi::dsh_bdf_solve_adaptive_sens(
                c, model, size, nb as i64, eqn.p.ptr(), atol.ptr(), atol.context().nbatch() as i64, problem.rtol, problem.t0, problem.h0, &o, t_eval.as_ptr(), nt as i64, srtol,
                satol.as_ptr(), satol.len() as i64, ys.ptr(), sens_d as *mut f64, stats_d as *mut i32, status_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    } else {
        unsafe {
            ffi::dsh_sdirk_solve_resident_sens(
                c, method as i32, model, size, nb as i64, eqn.p.ptr(), atol.ptr(), atol.context().nbatch() as i64, problem.rtol, problem.t0, problem.h0, &o, t_eval.as_ptr(),
                nt as i64, srtol, satol.as_ptr(), satol.len() as i64, ys.ptr(), sens_d as *mut f64, stats_d as *mut i32, status_d as *mut i32, totals.as_mut_ptr(),
            )
        }
    };
    let mut sens = Vec::with_capacity(np);
    if rc >= 0 {
        for j in 0..np {
            let m = HipMat::zeros(n, nt, ctx.clone());
            for k in 0..nt {
                // column k of parameter j: n x nb doubles, contiguous in both layouts
                let src = unsafe { (sens_d as *const f64).add((k * np + j) * n * nb) };
                let dst = unsafe { m.ptr().add(k * n * nb) };
                check(unsafe { ffi::dsh_d2d(c, dst as *mut c_void, src as *const c_void, (8 * n * nb) as i64) }, "dsh_d2d");
            }
            sens.push(m);
        }
    }
    let fetch = |dev: *mut c_void, host: *mut c_void, bytes: usize| check(unsafe { ffi::dsh_d2h(c, host, dev, bytes as i64) }, "dsh_d2h");
    let mut out = EnsembleSensSolution { ys, sens, status: vec![0; nb], stats: vec![[0; 5]; nb], totals };
    if rc >= 0 {
        let mut flat = vec![0i32; 5 * nb];
        fetch(stats_d, flat.as_mut_ptr() as *mut c_void, 20 * nb);
        for b in 0..nb {
            for k in 0..5 {
                out.stats[b][k] = flat[k * nb + b];
            }
        }
        fetch(status_d, out.status.as_mut_ptr() as *mut c_void, 4 * nb);
    }
    for p in [sens_d, stats_d, status_d] {
        unsafe { ffi::dsh_free(c, p) };
    }
    if rc < 0 {
        return Err(LaError::Other(last_error()));
    }
    Ok(out)
}
