/*
 * diffsol_c_hip.h — the reference's runtime-typed C API (crates/diffsol-c, the layer pydiffsol binds) for the HIP backend.
 * Exported by libdiffsol_hip_host.so.
 *
 * Same entry-point names, argument order, ownership and error conventions as crates/diffsol-c/src/<name>_c.rs, so a binding written against the
 * reference's C API drives the GPU backend by (a) choosing the backend's enum values below and (b) optionally passing nbatch > 1 parameter sets:
 *   - status codes DIFFSOL_OK / DIFFSOL_ERR / DIFFSOL_BAD_ARG                         c_api_utils.rs:3-5
 *   - thread-local last error with message, file and line                             error_c.rs:12-121
 *   - HostArray handles: data pointer, ndim, dim, stride (BYTES), dtype               host_array_c.rs, host_array.rs:157-230
 *   - runtime enums with *_count / *_is_valid / *_name                                matrix_type_c.rs, linear_solver_type_c.rs, ode_solver_type_c.rs,
 *                                                                                     scalar_type_c.rs, jit_c.rs
 *   - OdeWrapper: new_jit, free, y0, rhs, rhs_jac_mul, solve, solve_dense,            ode_c.rs
 *     get/set rtol, atol, t0, h0, ode_solver, linear_solver, matrix_type, options
 *   - OdeSolverOptions / InitialConditionSolverOptions get_/set_ per field            ode_options_c.rs, initial_condition_options_c.rs
 *   - SolutionWrapper: get_ys, get_ts, free                                           solution_wrapper_c.rs
 * The enum values of the reference keep their numbers; the HIP variants are appended: matrix type 3 = "hip_dense", JIT back end 2 = "hiprtc".
 * *_is_valid() reports what THIS library can run (hip_dense; default/lu; bdf, esdirk34, tr_bdf2; f64; hiprtc).
 *
 * Ensembles.  The reference's `params` is one parameter set.  Here params_len may be nbatch x nparams (batch-major: member 0's parameters first);
 * nbatch is inferred.  Arrays then carry a trailing batch axis:
 *   y0 / rhs / rhs_jac_mul   1-D, nbatch x n values, batch-major (member 0's vector first) — for nbatch = 1 exactly the reference's vector;
 *                            y / v may hold one vector (used for every member) or nbatch vectors
 *   ys                       ndim 2 for nbatch = 1: (nrows, ncols) column-major as in the reference (solution.rs:39-43);
 *                            ndim 3 for nbatch > 1: (nrows, ncols, nbatch), the batch axis fastest (strides in bytes: 8 nbatch, 8 nbatch nrows, 8)
 *   ts                       1-D, ncols
 * nrows = number of out_i components if the model defines out_i, else the number of states (DiffSl::out; ode_solver/method.rs write_state_out).
 * Ensemble mode (diffsol_ode_set_ensemble_mode): DIFFSOL_ENSEMBLE_AUTO (default) runs solve_dense on the device-resident integrators whenever the
 * model has such a kernel (wavefront groups of 64 without root functions — for nbatch <= 64 that is the lock-step ensemble, bit for bit —, per member
 * with them), else lock-step; `solve` is always lock-step.  DIFFSOL_ENSEMBLE_LOCKSTEP integrates all members with one (t, h, order) sequence — the
 * reference's batched-vector semantics, and `solve` returns every accepted step; DIFFSOL_ENSEMBLE_PER_MEMBER / _WAVEFRONT run solve_dense
 * entirely on the device (dsh_bdf_solve_adaptive / dsh_sdirk_solve_resident: every member its own steps and event time / 64-member groups);
 * columns after a member's own root stop are NaN and diffsol_solution_wrapper_get_member_info returns per-member status, root time and column count.
 */
#ifndef DIFFSOL_C_HIP_H
#define DIFFSOL_C_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIFFSOL_OK 0
#define DIFFSOL_ERR (-1)
#define DIFFSOL_BAD_ARG (-2)

/* MatrixType (matrix_type.rs:9-13) + the HIP variant */
#define DIFFSOL_MATRIX_NALGEBRA_DENSE 0
#define DIFFSOL_MATRIX_FAER_DENSE 1
#define DIFFSOL_MATRIX_FAER_SPARSE 2
#define DIFFSOL_MATRIX_HIP_DENSE 3
/* LinearSolverType (linear_solver_type.rs:15-19) */
#define DIFFSOL_LINEAR_SOLVER_DEFAULT 0
#define DIFFSOL_LINEAR_SOLVER_LU 1
#define DIFFSOL_LINEAR_SOLVER_KLU 2
/* OdeSolverType (ode_solver_type.rs:34-39) */
#define DIFFSOL_ODE_SOLVER_BDF 0
#define DIFFSOL_ODE_SOLVER_ESDIRK34 1
#define DIFFSOL_ODE_SOLVER_TR_BDF2 2
#define DIFFSOL_ODE_SOLVER_TSIT45 3
/* ScalarType (scalar_type.rs:11-14) */
#define DIFFSOL_SCALAR_F32 0
#define DIFFSOL_SCALAR_F64 1
/* JitBackendType (jit.rs:6-11) + the HIP variant */
#define DIFFSOL_JIT_CRANELIFT 0
#define DIFFSOL_JIT_LLVM 1
#define DIFFSOL_JIT_HIPRTC 2

#define DIFFSOL_ENSEMBLE_AUTO (-1)
#define DIFFSOL_ENSEMBLE_LOCKSTEP 0
#define DIFFSOL_ENSEMBLE_PER_MEMBER 1
#define DIFFSOL_ENSEMBLE_WAVEFRONT 64

typedef struct diffsol_ode_wrapper OdeWrapper;
typedef struct diffsol_host_array HostArray;
typedef struct diffsol_solution_wrapper SolutionWrapper;
typedef struct diffsol_ode_solver_options OdeSolverOptions;
typedef struct diffsol_ic_solver_options InitialConditionSolverOptions;

/* ---- error_c.rs */
int32_t diffsol_error_code(void);              /* 1 if an error is recorded for this thread */
const char* diffsol_error(void);               /* message or NULL */
const char* diffsol_last_error_message(void);
const char* diffsol_last_error_file(void);
uint32_t diffsol_last_error_line(void);
void diffsol_clear_last_error(void);

/* ---- host_array_c.rs */
HostArray* diffsol_host_array_alloc_vector(size_t len, int32_t dtype);
void diffsol_host_array_free(HostArray* array);
const uint8_t* diffsol_host_array_ptr(const HostArray* array);
size_t diffsol_host_array_ndim(const HostArray* array);
size_t diffsol_host_array_dim(const HostArray* array, size_t index);
size_t diffsol_host_array_stride(const HostArray* array, size_t index); /* bytes */
int32_t diffsol_host_array_dtype(const HostArray* array);

/* ---- runtime enums */
size_t diffsol_matrix_type_count(void);
int32_t diffsol_matrix_type_is_valid(int32_t value);
const char* diffsol_matrix_type_name(int32_t value);
size_t diffsol_linear_solver_type_count(void);
int32_t diffsol_linear_solver_type_is_valid(int32_t value);
const char* diffsol_linear_solver_type_name(int32_t value);
size_t diffsol_ode_solver_type_count(void);
int32_t diffsol_ode_solver_type_is_valid(int32_t value);
const char* diffsol_ode_solver_type_name(int32_t value);
size_t diffsol_scalar_type_count(void);
int32_t diffsol_scalar_type_is_valid(int32_t value);
const char* diffsol_scalar_type_name(int32_t value);
size_t diffsol_jit_backend_type_count(void);
int32_t diffsol_jit_backend_type_is_valid(int32_t value);
const char* diffsol_jit_backend_type_name(int32_t value);

/* ---- ode_c.rs */
/* diffsol_ode_new_jit (ode_c.rs:263-317): DiffSL text -> model (front end + hiprtc).  NULL on error. */
OdeWrapper* diffsol_ode_new_jit(const char* code, int32_t jit_backend, int32_t matrix_type, int32_t linear_solver, int32_t ode_solver);
/* ode_c.rs:46-49 */
typedef struct DiffsolDepPair { size_t row, col; } DiffsolDepPair;
/* diffsol_ode_new_external (ode_c.rs:181-230): models linked into the library.  Nothing is linked into this one (device code is compiled at run time):
 * always NULL with an error message that points to diffsol_ode_new_external_dynamic. */
OdeWrapper* diffsol_ode_new_external(int32_t matrix_type, int32_t linear_solver, int32_t ode_solver, const DiffsolDepPair* rhs_state_deps_ptr, size_t rhs_state_deps_len,
                                     const DiffsolDepPair* rhs_input_deps_ptr, size_t rhs_input_deps_len, const DiffsolDepPair* mass_state_deps_ptr,
                                     size_t mass_state_deps_len);
/* diffsol_ode_new_external_dynamic (ode_c.rs:232-281).  `path`: a HIP SOURCE file that defines the reference's external model functions (same names and
 * argument orders as crates/diffsol-c/tests/external-dynamic-logistic/src/lib.rs) as `DIFFSOL_DEVICE void name(...)` device functions — set_inputs,
 * set_u0, rhs, rhs_grad; mass / calc_stop / calc_out when declared — and what get_dims returns as macros DIFFSOL_EXTERNAL_STATES, _INPUTS, _OUTPUTS, _DATA,
 * _STOP, _HAS_MASS.  At most 8 states and one stop condition (register-resident form).  The dependency lists are accepted and not needed. */
OdeWrapper* diffsol_ode_new_external_dynamic(const char* path, int32_t matrix_type, int32_t linear_solver, int32_t ode_solver, const DiffsolDepPair* rhs_state_deps_ptr,
                                             size_t rhs_state_deps_len, const DiffsolDepPair* rhs_input_deps_ptr, size_t rhs_input_deps_len,
                                             const DiffsolDepPair* mass_state_deps_ptr, size_t mass_state_deps_len);
void diffsol_ode_free(OdeWrapper* ode);
int32_t diffsol_ode_get_options(const OdeWrapper* ode, OdeSolverOptions** out_options);           /* shares state with the ode; free with _options_free */
int32_t diffsol_ode_get_ic_options(const OdeWrapper* ode, InitialConditionSolverOptions** out_options);
int32_t diffsol_ode_y0(OdeWrapper* ode, const double* params_ptr, size_t params_len, HostArray** out_array);
int32_t diffsol_ode_rhs(OdeWrapper* ode, const double* params_ptr, size_t params_len, double t, const double* y_ptr, size_t y_len, HostArray** out_array);
int32_t diffsol_ode_rhs_jac_mul(OdeWrapper* ode, const double* params_ptr, size_t params_len, double t, const double* y_ptr, size_t y_len, const double* v_ptr,
                                size_t v_len, HostArray** out_array);
int32_t diffsol_ode_solve(OdeWrapper* ode, const double* params_ptr, size_t params_len, double final_time, SolutionWrapper** out_solution);
int32_t diffsol_ode_solve_dense(OdeWrapper* ode, const double* params_ptr, size_t params_len, const double* t_eval_ptr, size_t t_eval_len,
                                SolutionWrapper** out_solution);
/* ode_c.rs:586-617: states and forward sensitivities at t_eval (OdeSolverMethod::solve_dense_sensitivities); the sensitivities come back through
 * diffsol_solution_wrapper_get_sens.  sens_rtol / sens_atol (ode_c.rs:949-1040) put them into the error control when both are set. */
int32_t diffsol_ode_solve_fwd_sens(OdeWrapper* ode, const double* params_ptr, size_t params_len, const double* t_eval_ptr, size_t t_eval_len,
                                   SolutionWrapper** out_solution);
int32_t diffsol_ode_get_sens_rtol(const OdeWrapper* ode, int32_t* out_is_some, double* out_value);
int32_t diffsol_ode_set_sens_rtol(OdeWrapper* ode, int32_t value_is_some, double value);
int32_t diffsol_ode_get_sens_atol(const OdeWrapper* ode, int32_t* out_is_some, double* out_value);
int32_t diffsol_ode_set_sens_atol(OdeWrapper* ode, int32_t value_is_some, double value);
/* ode_c.rs:893-1190: stored and returned; a solve with integrate_out set fails (not implemented by this backend) */
int32_t diffsol_ode_get_integrate_out(const OdeWrapper* ode, int32_t* out_value);
int32_t diffsol_ode_set_integrate_out(OdeWrapper* ode, int32_t value);
int32_t diffsol_ode_get_out_rtol(const OdeWrapper* ode, int32_t* out_is_some, double* out_value);
int32_t diffsol_ode_set_out_rtol(OdeWrapper* ode, int32_t value_is_some, double value);
int32_t diffsol_ode_get_out_atol(const OdeWrapper* ode, int32_t* out_is_some, double* out_value);
int32_t diffsol_ode_set_out_atol(OdeWrapper* ode, int32_t value_is_some, double value);
int32_t diffsol_ode_get_param_rtol(const OdeWrapper* ode, int32_t* out_is_some, double* out_value);
int32_t diffsol_ode_set_param_rtol(OdeWrapper* ode, int32_t value_is_some, double value);
int32_t diffsol_ode_get_param_atol(const OdeWrapper* ode, int32_t* out_is_some, double* out_value);
int32_t diffsol_ode_set_param_atol(OdeWrapper* ode, int32_t value_is_some, double value);
/* string_c.rs:11-78 */
char* diffsol_alloc_string(size_t size);
void diffsol_free_string(char* ptr, size_t size);
uint8_t* diffsol_alloc(size_t size, size_t align);
void diffsol_free(uint8_t* ptr, size_t size, size_t align);
int32_t diffsol_ode_get_matrix_type(const OdeWrapper* ode);
int32_t diffsol_ode_get_ode_solver(const OdeWrapper* ode);
int32_t diffsol_ode_set_ode_solver(OdeWrapper* ode, int32_t value);
int32_t diffsol_ode_get_linear_solver(const OdeWrapper* ode);
int32_t diffsol_ode_set_linear_solver(OdeWrapper* ode, int32_t value);
int32_t diffsol_ode_get_rtol(const OdeWrapper* ode, double* out_value);
int32_t diffsol_ode_set_rtol(OdeWrapper* ode, double value);
int32_t diffsol_ode_get_atol(const OdeWrapper* ode, double* out_value);
int32_t diffsol_ode_set_atol(OdeWrapper* ode, double value);
int32_t diffsol_ode_get_t0(const OdeWrapper* ode, double* out_value);
int32_t diffsol_ode_set_t0(OdeWrapper* ode, double value);
int32_t diffsol_ode_get_h0(const OdeWrapper* ode, double* out_value);
int32_t diffsol_ode_set_h0(OdeWrapper* ode, double value);
/* additions of this backend */
int32_t diffsol_ode_get_ensemble_mode(const OdeWrapper* ode);
int32_t diffsol_ode_set_ensemble_mode(OdeWrapper* ode, int32_t mode);
int32_t diffsol_ode_get_dims(const OdeWrapper* ode, size_t* nstates, size_t* nparams, size_t* nout, size_t* nroots);
int32_t diffsol_ode_set_atol_vector(OdeWrapper* ode, const double* atol_ptr, size_t atol_len); /* per-state tolerances (OdeBuilder::atol) */

/* ---- ode_options_c.rs / initial_condition_options_c.rs */
void diffsol_ode_options_free(OdeSolverOptions* options);
void diffsol_ic_options_free(InitialConditionSolverOptions* options);
#define DIFFSOL_DECLARE_OPTION(prefix, type, ctype, field)                       \
  int32_t prefix##_get_##field(const type* options, ctype* out_value);           \
  int32_t prefix##_set_##field(type* options, ctype value);
DIFFSOL_DECLARE_OPTION(diffsol_ode_options, OdeSolverOptions, size_t, max_nonlinear_solver_iterations)
DIFFSOL_DECLARE_OPTION(diffsol_ode_options, OdeSolverOptions, size_t, max_error_test_failures)
DIFFSOL_DECLARE_OPTION(diffsol_ode_options, OdeSolverOptions, size_t, update_jacobian_after_steps)
DIFFSOL_DECLARE_OPTION(diffsol_ode_options, OdeSolverOptions, size_t, update_rhs_jacobian_after_steps)
DIFFSOL_DECLARE_OPTION(diffsol_ode_options, OdeSolverOptions, double, threshold_to_update_jacobian)
DIFFSOL_DECLARE_OPTION(diffsol_ode_options, OdeSolverOptions, double, threshold_to_update_rhs_jacobian)
DIFFSOL_DECLARE_OPTION(diffsol_ode_options, OdeSolverOptions, double, min_timestep)
DIFFSOL_DECLARE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, int32_t, use_linesearch)
DIFFSOL_DECLARE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, size_t, max_linesearch_iterations)
DIFFSOL_DECLARE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, size_t, max_newton_iterations)
DIFFSOL_DECLARE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, size_t, max_linear_solver_setups)
DIFFSOL_DECLARE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, double, step_reduction_factor)
DIFFSOL_DECLARE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, double, armijo_constant)
#undef DIFFSOL_DECLARE_OPTION

/* ---- solution_wrapper_c.rs */
void diffsol_solution_wrapper_free(SolutionWrapper* solution);
int32_t diffsol_solution_wrapper_get_ys(const SolutionWrapper* solution, HostArray** out_array);
int32_t diffsol_solution_wrapper_get_ts(const SolutionWrapper* solution, HostArray** out_array);
/* solution_wrapper_c.rs:101-125: one HostArray per parameter, shaped like ys; free every array with diffsol_host_array_free and the list with
 * diffsol_host_array_list_free (ode_c.rs:163-171) */
int32_t diffsol_solution_wrapper_get_sens(const SolutionWrapper* solution, HostArray*** out_sens, size_t* out_sens_len);
void diffsol_host_array_list_free(HostArray** list, size_t len);
/* addition of this backend: per-member outcome of an ensemble solve.  status (0 ok, else OdeSolverError ordinal), t_root (NaN if the member hit no
 * stop condition), root_index (-1), ncols (valid output columns); each array has nbatch entries and may be NULL.  Returns nbatch. */
int64_t diffsol_solution_wrapper_get_member_info(const SolutionWrapper* solution, int32_t* status, double* t_root, int32_t* root_index, int32_t* ncols);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSOL_C_HIP_H */
