/*
 * diffsol_hip_solver.h — C ABI of libdiffsol_hip_host.so: the host-side integrators (OdeBuilder -> problem -> .bdf() / .tr_bdf2() /
 * .esdirk34() -> OdeSolverMethod) running on the HIP backend of diffsol_hip.h.
 *
 * In a deployment with a Rust toolchain this layer is diffsol itself (its generic Bdf / Sdirk instantiated with HipMat / HipLU, see
 * INTEGRATION.md); this environment has no rustc, so the same control flow is provided in C++ (the headers under diffsol_amd/host mirror
 * crates/diffsol/src/ode_solver/{bdf,sdirk,runge_kutta,state,method}.rs and crates/diffsol-nl) and exported here in the style of the
 * reference's own C wrapper (crates/diffsol-c/src/ode_c.rs, error convention crates/diffsol-c/src/error_c.rs:12-121).
 * All host arrays are batch-major ([b][state]) like the reference API.  Return 0 on success, negative on error (dshs_last_error()).
 */
#ifndef DIFFSOL_HIP_SOLVER_H
#define DIFFSOL_HIP_SOLVER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dshs_solver dshs_solver;
typedef struct dsh_ctx dsh_ctx; /* include/diffsol_hip.h */

#define DSHS_METHOD_BDF 0      /* OdeSolverProblem::bdf       crates/diffsol/src/ode_solver/problem.rs:649-655 */
#define DSHS_METHOD_TR_BDF2 1  /* OdeSolverProblem::tr_bdf2   problem.rs:839-861 (sdirk_solver_from_tableau!) */
#define DSHS_METHOD_ESDIRK34 2 /* OdeSolverProblem::esdirk34 */

#define DSHS_STOP_INTERNAL_TIMESTEP 0 /* OdeSolverStopReason, crates/diffsol/src/ode_solver/method.rs:22-40 */
#define DSHS_STOP_ROOT_FOUND 1
#define DSHS_STOP_TSTOP_REACHED 2

/* OdeSolverOptions + InitialConditionSolverOptions (crates/diffsol/src/ode_solver/problem.rs:15-152).  Negative / NaN = keep default. */
typedef struct {
  int max_nonlinear_solver_iterations; /* 10 */
  int max_error_test_failures;         /* 40 */
  int max_nonlinear_solver_failures;   /* 50 */
  double nonlinear_solver_tolerance;   /* 0.2 */
  double min_timestep;                 /* 1e-13 */
  int update_jacobian_after_steps;     /* 20 */
  int update_rhs_jacobian_after_steps; /* 50 */
  double threshold_to_update_jacobian;     /* 0.3 */
  double threshold_to_update_rhs_jacobian; /* 0.2 */
  int ic_use_linesearch;               /* 1 */
  int use_fused_kernels;               /* 1: fused device kernels where the model provides them; 0: 1:1 trait ops only */
  int block_threads;                   /* 0 = default (64) */
  int ic_max_linesearch_iterations;    /* 10   InitialConditionSolverOptions (problem.rs:15-45) */
  int ic_max_linear_solver_setups;     /* 4 */
  int ic_max_newton_iterations;        /* 10 */
  double ic_step_reduction_factor;     /* 0.5 */
  double ic_armijo_constant;           /* 1e-4 */
} dshs_options;

const char* dshs_last_error(void);
void dshs_default_options(dshs_options* opts);

/* OdeBuilder::new().t0().h0().rtol().atol().p().context(ctx.with_nbatch(nbatch)).build() + problem.<method>()  (builder.rs:1784-1893).
 * params: batch-major, nparams per batch member.  atol: 1 or nstates entries.  stream: NULL or a hipStream_t to run on. */
int dshs_create(int device, void* stream, int model, int64_t model_size, int64_t nbatch, const double* params, int64_t nparams_total, double rtol,
                const double* atol, int64_t natol, double t0, double h0, int method, const dshs_options* opts, dshs_solver** out);
/* problem.bdf_sens() (problem.rs:819-832): the same solver with the forward sensitivities s_j = dy/dp_j of every parameter integrated
 * alongside the states — Bdf::sensitivity_solve (bdf.rs:934-989: one Newton solve per parameter and step with the LU factors of the state equations),
 * SensEquations (ode_equations/sens_equations.rs), sensitivity difference arrays rescaled / updated with the states'.  nsens_atol > 0: the
 * sensitivities take part in the error control with sens_rtol / sens_atol (length 1 or nstates, the same for every parameter); nsens_atol = 0:
 * turn_off_sensitivities_error_control.  BDF, ODE models with parameter derivatives (dsh_model_has_sens); host-driven lock-step over the ensemble. */
int dshs_create_sens(int device, void* stream, int model, int64_t model_size, int64_t nbatch, const double* params, int64_t nparams_total, double rtol,
                     const double* atol, int64_t natol, double t0, double h0, int method, const dshs_options* opts, int sens, double sens_rtol,
                     const double* sens_atol, int64_t nsens_atol, dshs_solver** out);
void dshs_destroy(dshs_solver* s);

/* Re-create the solver state from the problem (parameters stay resident on the device): OdeSolverProblem::bdf()/... again —
 * initial state, consistent initialisation, initial step size, first Jacobian + LU.  Lets a benchmark time whole solves back to back. */
int dshs_reset(dshs_solver* s);
/* forwarders to dsh_ctx_set_timing / dsh_ctx_get_timing of the solver's context */
/* the solver's device context (its stream is where every launch of this solver goes): e.g. for dsh_dist_init — a gather issued on it waits for the solver's work */
dsh_ctx* dshs_context(dshs_solver* s);
/* dsh_ctx_set_solve_mode of the solver's context: DSH_SOLVE_EXACT (default, bit-identical to the CPU path) | DSH_SOLVE_REORDERED (opt-in, tolerance-level differences) */
int dshs_set_linear_solve_mode(dshs_solver* s, int mode);
int dshs_set_kernel_timing(dshs_solver* s, int enable);
int dshs_set_kernel_timing_target(dshs_solver* s, int target); /* DSH_TIMING_* of diffsol_hip.h */
int dshs_get_kernel_timing(dshs_solver* s, int64_t* launches, double* total_ms);
int dshs_get_kernel_timing_overhead(dshs_solver* s, double* empty_bracket_ms, double* device_clock_total_ms);

int64_t dshs_nstates(const dshs_solver* s);
int64_t dshs_nbatch(const dshs_solver* s);
int dshs_is_fused(const dshs_solver* s);

/* OdeSolverMethod::step / set_stop_time / interpolate / state  (method.rs:42-198) */
int dshs_step(dshs_solver* s, int* stop_reason);
int dshs_set_stop_time(dshs_solver* s, double tstop);
int dshs_interpolate(dshs_solver* s, double t, double* y_host);
int dshs_get_state(dshs_solver* s, double* t, double* h, int* order, double* y_host, double* dy_host);
int dshs_root_info(dshs_solver* s, double* t_root, int* root_index);
/* OdeSolverMethod::interpolate_sens (bdf.rs:1162-1215): s_host [nparams][b][state]; t = NaN returns state.s (the sensitivities at the current time) */
int64_t dshs_nparams(const dshs_solver* s);
int dshs_interpolate_sens(dshs_solver* s, double t, double* s_host);
/* BDF only: difference array D as [b][col(8)][row] */
int dshs_bdf_get_diff(dshs_solver* s, double* diff_host);
/* out[0..10) OdeSolverStatistics in declaration order (ode_solver/mod.rs:28-69); out[10..13) rhs OpStatistics calls / jac_muls / matrix_evals */
int dshs_stats(dshs_solver* s, int64_t* out);

/* The reference's known-answer harness (crates/diffsol/src/ode_solver/mod.rs:104-194, use_tstop = false): for each point step until
 * |t| >= |t_point| and interpolate there.  y_host: [npoints][b][state].  Returns 1 if a root stopped the solve. */
int dshs_solve_to_points(dshs_solver* s, const double* t_points, int64_t npoints, double* y_host);
/* OdeSolverMethod::solve (method.rs:227-258): integrate to final_time; writes the final state.y (host, batch-major) and the number of
 * output columns; the trajectory matrix is kept on the device when keep_trajectory != 0 (fetch with dshs_trajectory). */
int dshs_solve(dshs_solver* s, double final_time, int keep_trajectory, double* y_final_host, int64_t* ncols, int* stop_reason);
int dshs_trajectory(dshs_solver* s, double* t_host, double* y_host /* [col][b][state] */);
/* OdeSolverMethod::solve_dense (method.rs:467-520): interpolated output at t_eval.  y_host ([nt][b][state]) and/or y_dev (DEVICE pointer,
 * [nt][state][b] batch-fastest — the buffer the multi-GPU gather concatenates along the batch axis) may be NULL. */
int dshs_solve_dense(dshs_solver* s, const double* t_eval, int64_t nt, double* y_host, double* y_dev, int* stop_reason);
/* Which integrator dshs_solve_dense runs for an ensemble (nbatch members):
 *   DSHS_ENSEMBLE_AUTO (default; environment DSH_ENSEMBLE_MODE=lockstep|member|wave overrides it): the device-resident integrators whenever the
 *     model/method has one (dsh_model_has_resident / dsh_model_has_wave_member) — the whole solve_dense in ONE launch, solver state in registers:
 *     wavefront-sized lock-step groups for models without root functions (the reference's batched semantics with nbatch = 64 per group; an
 *     ensemble of <= 64 members is one group, i.e. exactly the lock-step ensemble, bit for bit), one step-size/order history per member for
 *     ENSEMBLES (nbatch > 1) of models with root functions (every member stops at its own event: its output columns behind the event are NaN,
 *     dshs_root_info reports the earliest member event, dshs_solve_dense_adaptive returns every member's); a single IVP (nbatch == 1) with root
 *     functions keeps the reference's contract to the letter — output truncated at the root, solver state moved to it — on the host-driven path;
 *     otherwise DSHS_ENSEMBLE_LOCKSTEP.
 *   DSHS_ENSEMBLE_LOCKSTEP: host-driven, one (t, h, order) sequence for the whole ensemble over the Vector/Matrix/LinearSolver operations of
 *     diffsol_hip.h — what the reference's generic Bdf/Sdirk do on a batched context (method.rs:467-520 over bdf.rs:1277-1589).
 *   DSHS_ENSEMBLE_PER_MEMBER / DSHS_ENSEMBLE_WAVEFRONT: force one of the device-resident granularities (error if the model has no such kernel).
 * In the device-resident modes the host-side solver state (dshs_get_state, dshs_stats) is not advanced — the problem is integrated from
 * (t0, y0) on every call; dshs_last_solve_info returns the counters.  A member that fails makes dshs_solve_dense fail like the reference's
 * solve_dense does (-100 - OdeSolverError ordinal of the first failing member); dshs_solve_dense_adaptive returns per-member status instead.
 * stop_reason: DSHS_STOP_ROOT_FOUND if every member stopped at a root, else DSHS_STOP_TSTOP_REACHED. */
#define DSHS_ENSEMBLE_AUTO (-1)
#define DSHS_ENSEMBLE_LOCKSTEP 0
#define DSHS_ENSEMBLE_PER_MEMBER 1
#define DSHS_ENSEMBLE_WAVEFRONT 64
int dshs_set_ensemble_mode(dshs_solver* s, int mode);
/* Process-wide: pow() of the HOST-driven integrators (step-size controller, convergence rate, initial step).  0 (default) = libm, the reference's
 * arithmetic; 1 = include/diffsol_detpow.h, the pow of the device-resident integrators — then DSHS_ENSEMBLE_LOCKSTEP and the device-resident
 * wavefront mode give the same bits for ensembles of <= 64 members (one group). */
int dshs_set_deterministic_pow(int on);
/* Process-wide: arithmetic of the device-resident BDF launched by dshs_solve_dense in its non-lock-step modes (the explicit *_adaptive entry points take theirs as
 * an argument).  DSHS_ARITH_FAST (default; environment DSH_RESIDENT_ARITH=fast): the fast-arithmetic build where one exists — static models with n <= 4, BDF, no
 * forward sensitivities: contracted multiply-adds, reciprocal-math division, ocml pow (dsh_adaptive_options.deterministic_pow = 2).  Same algorithm and, on
 * BASELINE config 2 at full size, the same step / order / refactorisation decisions for every member; states within 1e-9 relative of the exact kernel
 * (tests/test_gpu_adaptive.py).  DSHS_ARITH_EXACT (DSH_RESIDENT_ARITH=exact): always the exact kernel, bit-identical to the CPU oracle (deterministic_pow = 1). */
#define DSHS_ARITH_EXACT 1
#define DSHS_ARITH_FAST 2
int dshs_set_resident_arithmetic(int mode);
int dshs_get_resident_arithmetic(void);
int dshs_get_ensemble_mode(const dshs_solver* s, int* requested, int* resolved);
/* mode the last solve_dense ran in and its counters summed over members: totals[6] = steps, Newton iterations, LU setups, error-test failures,
 * Newton failures, failed members (lock-step: the solver's counters x nbatch). */
int dshs_last_solve_info(const dshs_solver* s, int* mode, int64_t* totals);
/* solve_dense entirely on the device, the whole ensemble in one launch (dsh_bdf_solve_adaptive / dsh_sdirk_solve_resident).
 * group = 1: every member is integrated as the independent IVP it is on diffsol's CPU path, with its own step sizes, orders and EVENT TIMES;
 * group = 64: wavefront-sized lock-step groups (the reference's batched semantics with nbatch = 64).  Static models with n <= 4 and banded run-time-sized models (built-in: n <= 512, from DiffSL: n <= 64; BDF, TR-BDF2,
 * ESDIRK34), mass matrices and root functions included.
 * stats_host: [5][b] int32 (steps, Newton iterations, LU setups, error-test failures, Newton failures); status_host: [b] (0 ok, else OdeSolverError
 * ordinal, 20 root batch mismatch, 99 runaway guard); t_root_host / root_idx_host / ncols_host: [b] root time (NaN if none), root index (-1),
 * number of valid output columns; any of them may be NULL.  totals[6]: counters summed over members + number of failed members. */
int dshs_solve_dense_adaptive(dshs_solver* s, const double* t_eval, int64_t nt, int group /* 1 | 64 */, int deterministic_pow /* diffsol_detpow.h */,
                              double* y_host, double* y_dev, int32_t* stats_host,
                              int32_t* status_host, double* t_root_host, int32_t* root_idx_host, int32_t* ncols_host, int64_t* totals);

/* OdeSolverMethod::solve (method.rs:227-258: the state after every accepted step, not at save points) entirely on the device: every member's steps out of ONE
 * launch of the register-resident BDF (dsh_bdf_solve_adaptive_steps; static models with n <= 4, built-in or from DiffSL — DSH_E_UNSUPPORTED otherwise: dshs_solve
 * returns every step of the host-driven lock-step solver).  group = 1: every member its own times; group = 64: shared per wavefront.  y_host: [max_cols][b][state],
 * t_host: [max_cols][b]; ncols_host[b]: columns member b produced — column 0 = (t0, y0), the last one at t_final or at the member's event (t_root_host /
 * root_idx_host) — ncols > max_cols: not all of them were stored, call again with more room.  stats_host / status_host as dshs_solve_dense_adaptive. */
int dshs_solve_adaptive(dshs_solver* s, double t_final, int64_t max_cols, int group /* 1 | 64 */, int deterministic_pow, double* y_host, double* t_host,
                        int32_t* ncols_host, int32_t* stats_host, int32_t* status_host, double* t_root_host, int32_t* root_idx_host, int64_t* totals);

/* dshs_solve_dense_adaptive for a solver created with forward sensitivities (dshs_create_sens): states and sensitivities at t_eval from ONE launch
 * (dsh_bdf_solve_adaptive_sens; BDF, static ODE models with parameter derivatives, n <= 4, no root functions — DSH_E_UNSUPPORTED otherwise: such problems
 * integrate their sensitivities host-driven, dshs_solve_dense + dshs_interpolate_sens).  The reference's solve_dense_sensitivities (sensitivities.rs:114-260)
 * per member (group = 1) or per 64-member lock-step group (group = 64).  y_host: [nt][b][state] or NULL; sens_host: [nparams][nt][b][state]. */
int dshs_solve_dense_adaptive_sens(dshs_solver* s, const double* t_eval, int64_t nt, int group /* 1 | 64 */, int deterministic_pow, double* y_host,
                                   double* sens_host, int32_t* stats_host, int32_t* status_host, int64_t* totals);

/* ---- DiffSL front end: what OdeBuilder::build_from_diffsl does with the external `diffsl` compiler
 * (crates/diffsol/src/ode_solver/builder.rs, crates/diffsol/src/ode_equations/diffsl.rs): DiffSL text -> source code of the model.
 * target DSHS_DIFFSL_HIP_STATIC: `struct dsh::JitModel` (n <= 8, register-resident; feeds dsh_model_compile);
 *        DSHS_DIFFSL_HIP_DYNAMIC: per-component device functions for run-time-sized models (feeds dsh_model_compile);
 *        DSHS_DIFFSL_HOST_C: an extern "C" CPU model (dsl_dims, dsl_rhs, dsl_jac_mul, dsl_mass_gemv, dsl_init, dsl_root, dsl_out) — the shape of the
 *        reference's external-model ABI (crates/diffsol-c/tests/external-dynamic-logistic/src/lib.rs:123-161).
 * *source_out is malloc'ed (free with dshs_free_string).  dims[10] = n, nparams, nroots, nout, has_mass, declared-no-inputs (1 = the single
 * parameter is an unused placeholder), then the structural bandwidths jac_kl, jac_ku, mass_kl, mass_ku of f_y and of the mass matrix (for
 * dsh_model_set_band: banded models are assembled and factored on the band only).  defaults_out (may be NULL) receives nparams default values. */
#define DSHS_DIFFSL_HIP_STATIC 0
#define DSHS_DIFFSL_HIP_DYNAMIC 1
#define DSHS_DIFFSL_HOST_C 2
int dshs_diffsl_generate(const char* code, int target, char** source_out, int64_t* dims, double* defaults_out, int64_t defaults_cap);
void dshs_free_string(char* s);
/* The reference's DiffSL model index (DiffSlContext::model_index, ode_equations/diffsl.rs:52,115,406-411; the scalar `N` of a DiffSL text, 0 unless
 * set_params_and_model changes it).  It is a compile-time constant of the generated model: another index is another compiled model.
 * dshs_diffsl_generate_indexed passes it explicitly.  dshs_diffsl_set_model_index arms it ONE-SHOT for the next text this thread compiles through an entry
 * that has no index argument (dshs_diffsl_generate, diffsol_ode_new_jit): that call consumes it, every later call compiles with index 0 again.
 * A caller that runs dshs_diffsl_generate TWICE for one model (dimensions first, then the source with its defaults) must therefore use
 * dshs_diffsl_generate_indexed for both calls — an armed index would serve the first call only and the second would describe model 0. */
int dshs_diffsl_generate_indexed(const char* code, int target, int model_index, char** source_out, int64_t* dims, double* defaults_out, int64_t defaults_cap);
int dshs_diffsl_set_model_index(int model_index);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSOL_HIP_SOLVER_H */
