// Host-side mirror of the parts of the `diffsol` crate that sit on the hot path's caller side (crates/diffsol/src):
// OdeEquations plug-in boundary, OdeBuilder / OdeSolverProblem, options, statistics, JacobianUpdate, InitOp + consistent
// initialisation, initial step size, RootFinder, OdeSolverMethod (solve / solve_dense).  Scalar control logic runs here on the
// host (north_star); everything that touches ensemble data goes through HipVec / HipMat / HipLU or the fused dsh_* kernels.
//   ode_equations/mod.rs:204-329   ode_solver/builder.rs:22-146,1784-1893   ode_solver/problem.rs:15-193
//   ode_solver/mod.rs:28-69        ode_solver/jacobian_update.rs:12-79      op/init.rs:14-135
//   ode_solver/state.rs:84-162,969-997,1086-1124,1209-1277                 nonlinear_solver/root.rs:12-222
//   ode_solver/method.rs:42-198,227-258,467-520,721-1040
#pragma once
#include <cstdlib>
#include <string>
#include <functional>
#include <array>
#include <cmath>
#include <optional>

#include "nonlinear.hpp"

namespace diffsol_hip {

// diffsol/src/error.rs:40-122
enum class OdeSolverError {
  Ok = 0, StepSizeTooSmall, TooManyErrorTestFailures, TooManyNonlinearSolverFailures, InitialConditionDidNotConverge, StopTimeBeforeCurrentTime,
  StopTimeAtCurrentTime, InterpolationTimeAfterCurrentTime, InterpolationTimeOutsideCurrentStep, InterpolationVectorWrongSize, InvalidTEval,
  StateProblemMismatch, LinearSolveFailed
};
struct DiffsolError : std::runtime_error {
  OdeSolverError kind;
  DiffsolError(OdeSolverError k, const std::string& msg) : std::runtime_error(msg), kind(k) {}
};
#define DSH_ODE_ERR(kind) ::diffsol_hip::DiffsolError(::diffsol_hip::OdeSolverError::kind, #kind)

enum class OdeSolverStopReason { InternalTimestep = 0, RootFound = 1, TstopReached = 2 };

struct OpStatistics { long number_of_calls = 0, number_of_jac_muls = 0, number_of_matrix_evals = 0; };  // op/mod.rs:95-128

// ode_solver/mod.rs:28-69
struct OdeSolverStatistics {
  long number_of_linear_solver_setups = 0, number_of_steps = 0, number_of_error_test_failures = 0, number_of_nonlinear_solver_iterations = 0,
       number_of_nonlinear_solver_fails = 0, number_of_linear_solver_setups_from_checkpoint = 0,
       number_of_linear_solver_setups_from_first_convergence_fail = 0, number_of_linear_solver_setups_from_second_convergence_fail = 0,
       number_of_linear_solver_setups_from_error_test_fail = 0, number_of_linear_solver_setups_from_step_success = 0;
};
enum class SolverState { StepSuccess, FirstConvergenceFail, SecondConvergenceFail, ErrorTestFail, Checkpoint };
inline void record_linear_solver_setup(OdeSolverStatistics& s, SolverState st) {
  s.number_of_linear_solver_setups++;
  switch (st) {
    case SolverState::Checkpoint: s.number_of_linear_solver_setups_from_checkpoint++; break;
    case SolverState::FirstConvergenceFail: s.number_of_linear_solver_setups_from_first_convergence_fail++; break;
    case SolverState::SecondConvergenceFail: s.number_of_linear_solver_setups_from_second_convergence_fail++; break;
    case SolverState::ErrorTestFail: s.number_of_linear_solver_setups_from_error_test_fail++; break;
    case SolverState::StepSuccess: s.number_of_linear_solver_setups_from_step_success++; break;
  }
}

// The OdeEquations plug-in boundary (ode_equations/mod.rs:245-329): rhs (NonLinearOpJacobian), optional mass (LinearOp), init
// (ConstantOp), optional root.  A device model is addressed by registry id; `fused_model()` tells the integrators that the
// register-resident fused kernels exist for it.
class OdeEquations {
 public:
  virtual ~OdeEquations() = default;
  virtual int64_t nstates() const = 0;
  virtual int64_t nparams() const = 0;
  virtual int64_t nroots() const = 0;
  virtual bool has_mass() const = 0;
  virtual const HipContext& context() const = 0;
  virtual void rhs_call_inplace(const HipVec& x, double t, HipVec& y) const = 0;
  virtual void rhs_jac_mul_inplace(const HipVec& x, double t, const HipVec& v, HipVec& y) const = 0;
  virtual void rhs_jacobian_inplace(const HipVec& x, double t, HipMat& y) const = 0;
  virtual void mass_gemv_inplace(const HipVec& x, double t, double beta, HipVec& y) const = 0;
  virtual void mass_matrix_inplace(double t, HipMat& y) const = 0;
  virtual void init_call_inplace(double t, HipVec& y) const = 0;
  virtual void root_call_inplace(const HipVec& x, double t, HipVec& g) const = 0;
  // forward sensitivities (OdeEquationsImplicitSens): df/dp at (x, t) and dy0/dp as n x nparams matrices (NonLinearOpSens::sens_inplace,
  // op/nonlinear_op.rs:67-81; SensInit, ode_equations/sens_equations.rs:62-70)
  virtual bool has_sens() const { return false; }
  // OdeEquations::reset (ode_equations/mod.rs): the state after an event, y = reset(x, t); hybrid models only
  virtual bool has_reset() const { return false; }
  virtual void reset_call_inplace(const HipVec&, double, HipVec&) const { throw LaError(DSH_E_UNSUPPORTED, "the equations have no reset operator"); }
  virtual void rhs_sens_inplace(const HipVec&, double, HipMat&) const { throw LaError(DSH_E_UNSUPPORTED, "the equations have no parameter sensitivities"); }
  virtual void init_sens_inplace(double, HipMat&) const { throw LaError(DSH_E_UNSUPPORTED, "the equations have no parameter sensitivities"); }
  virtual bool fused_model(int* model, int64_t* size) const { (void)model; (void)size; return false; }
  // the equations declare a narrow band for f_y and have no mass matrix: the integrators keep Jacobian, mass and M - cJ in band containers
  // (HipMat::zeros_banded: (kl + ku + 1) n entries per member instead of n^2) and the LU keeps banded factors
  virtual bool packed_band(int* kl, int* ku) const { (void)kl; (void)ku; return false; }
  // registry id of the model when it is one of libdiffsol_hip's built-in device models (fused or not)
  virtual bool registry_model(int*, int64_t*) const { return false; }
  virtual const HipVec& params() const = 0;
  mutable OpStatistics rhs_statistics;
};

// Equations backed by the device model registry (dsh_model_*): Robertson, exponential decay, heat, RLC, ...
class HipKernelEquations : public OdeEquations {
 public:
  // p is batch-major host data: nparams entries per batch member (exponential_decay.rs:300-311)
  HipKernelEquations(int model, int64_t size, const std::vector<double>& p, const HipContext& ctx) : model_(model), size_(size), ctx_(ctx) {
    int hm = 0;
    check(dsh_model_info(model, size, &n_, &np_, &hm, &nroots_), "HipKernelEquations");
    has_mass_ = hm != 0;
    if ((int64_t)p.size() != np_ * ctx.nbatch()) throw LaError(DSH_E_INVALID, "parameter vector must have nparams*nbatch entries");
    p_ = np_ > 0 ? HipVec::from_vec(p, ctx) : HipVec::zeros(0, ctx);
    fused_ = dsh_model_has_fused(model, size) != 0;
    check(dsh_model_band(model, size, &jac_kl_, &jac_ku_, &mass_kl_, &mass_ku_), "HipKernelEquations (band)");
    const char* e = std::getenv("DSH_JAC_BAND");  // =0: always evaluate the whole dense container
    band_eval_ = dsh_model_has_band_jacobian(model, size) != 0 && !(e && e[0] == '0');
  }
  int64_t nstates() const override { return n_; }
  int64_t nparams() const override { return np_; }
  int64_t nroots() const override { return nroots_; }
  bool has_mass() const override { return has_mass_; }
  const HipContext& context() const override { return ctx_; }
  void rhs_call_inplace(const HipVec& x, double t, HipVec& y) const override {
    rhs_statistics.number_of_calls++;
    check(dsh_model_rhs(ctx_.raw(), model_, size_, ctx_.nbatch(), t, x.ptr(), p_.ptr(), y.ptr()), "rhs");
  }
  void rhs_jac_mul_inplace(const HipVec& x, double t, const HipVec& v, HipVec& y) const override {
    rhs_statistics.number_of_jac_muls++;
    check(dsh_model_jac_mul(ctx_.raw(), model_, size_, ctx_.nbatch(), t, x.ptr(), p_.ptr(), v.ptr(), y.ptr()), "jac_mul");
  }
  // The registry assembles the dense Jacobian in one launch with the arithmetic of n jac_mul calls on unit vectors
  // (the reference's default assembly, op/nonlinear_op.rs:211-219); the counters record that equivalent work.
  void rhs_jacobian_inplace(const HipVec& x, double t, HipMat& y) const override {
    rhs_statistics.number_of_matrix_evals++;
    rhs_statistics.number_of_jac_muls += n_;
    if (y.packed()) {  // band container: the same entries, (kl + ku + 1) n of them
      check(dsh_model_jacobian_band_packed(ctx_.raw(), model_, size_, ctx_.nbatch(), t, x.ptr(), p_.ptr(), y.band_kl(), y.band_ku(), y.band_ptr()), "jacobian (band container)");
      return;
    }
    // a container that is known to be zero outside the declared band (freshly zeroed, or last written by this very function) is evaluated on the band only
    const bool on_band = band_eval_ && jac_kl_ >= 0 && jac_ku_ >= 0 && y.has_band() && y.band_kl() <= jac_kl_ && y.band_ku() <= jac_ku_ && n_ >= 16 &&
                         (int64_t)(jac_kl_ + jac_ku_ + 1) * 2 <= n_;
    if (on_band) check(dsh_model_jacobian_band(ctx_.raw(), model_, size_, ctx_.nbatch(), t, x.ptr(), p_.ptr(), jac_kl_, jac_ku_, y.ptr()), "jacobian (band)");
    else check(dsh_model_jacobian(ctx_.raw(), model_, size_, ctx_.nbatch(), t, x.ptr(), p_.ptr(), y.ptr()), "jacobian");
    if (jac_kl_ >= 0 && jac_ku_ >= 0) y.set_band(jac_kl_, jac_ku_);  // the model declares the structure of f_y (dsh_model_band)
  }
  void mass_gemv_inplace(const HipVec& x, double t, double beta, HipVec& y) const override {
    check(dsh_model_mass_gemv(ctx_.raw(), model_, size_, ctx_.nbatch(), t, x.ptr(), p_.ptr(), beta, y.ptr()), "mass_gemv");
  }
  void mass_matrix_inplace(double t, HipMat& y) const override {
    check(dsh_model_mass_matrix(ctx_.raw(), model_, size_, ctx_.nbatch(), t, p_.ptr(), y.ptr()), "mass_matrix");
    if (mass_kl_ >= 0 && mass_ku_ >= 0) y.set_band(mass_kl_, mass_ku_);
  }
  void init_call_inplace(double t, HipVec& y) const override { check(dsh_model_init(ctx_.raw(), model_, size_, ctx_.nbatch(), t, p_.ptr(), y.ptr()), "init"); }
  void root_call_inplace(const HipVec& x, double t, HipVec& g) const override {
    check(dsh_model_root(ctx_.raw(), model_, size_, ctx_.nbatch(), t, x.ptr(), p_.ptr(), g.ptr()), "root");
  }
  bool has_sens() const override { return dsh_model_has_sens(model_, size_) != 0; }
  bool has_reset() const override { return dsh_model_has_reset(model_, size_) != 0; }
  void reset_call_inplace(const HipVec& x, double t, HipVec& y) const override {
    check(dsh_model_reset(ctx_.raw(), model_, size_, ctx_.nbatch(), t, x.ptr(), p_.ptr(), y.ptr()), "reset");
  }
  void rhs_sens_inplace(const HipVec& x, double t, HipMat& S) const override {
    check(dsh_model_rhs_sens(ctx_.raw(), model_, size_, ctx_.nbatch(), t, x.ptr(), p_.ptr(), S.ptr()), "rhs_sens");
  }
  void init_sens_inplace(double t, HipMat& S0) const override {
    check(dsh_model_init_sens(ctx_.raw(), model_, size_, ctx_.nbatch(), t, p_.ptr(), S0.ptr()), "init_sens");
  }
  bool fused_model(int* model, int64_t* size) const override { if (!fused_) return false; *model = model_; *size = size_; return true; }
  bool packed_band(int* kl, int* ku) const override {
    // DSH_BAND_CONTAINER=0 keeps the dense containers with a structure tag (round 2); DSH_LU_STRUCTURE=dense asks for dense factorisations, which need them
    const char* e = std::getenv("DSH_BAND_CONTAINER");
    const char* st = std::getenv("DSH_LU_STRUCTURE");
    if ((e && e[0] == '0') || (st && std::string(st) == "dense")) return false;
    if (!band_eval_ || has_mass_ || jac_kl_ < 0 || jac_ku_ < 0 || n_ < 16 || std::max(jac_kl_, jac_ku_) > 4 || (int64_t)(jac_kl_ + jac_ku_ + 1) * 2 > n_) return false;
    *kl = jac_kl_; *ku = jac_ku_;
    return true;
  }
  bool registry_model(int* model, int64_t* size) const override { *model = model_; *size = size_; return true; }
  const HipVec& params() const override { return p_; }
  void count_fused_rhs_call() const { rhs_statistics.number_of_calls++; }

 private:
  int model_;
  int64_t size_, n_ = 0, np_ = 0, nroots_ = 0;
  bool has_mass_ = false, fused_ = false;
  int jac_kl_ = -1, jac_ku_ = -1, mass_kl_ = -1, mass_ku_ = -1;
  bool band_eval_ = false;
  HipContext ctx_;
  HipVec p_;
};

// ode_solver/problem.rs:15-45
struct InitialConditionSolverOptions {
  bool use_linesearch = true;
  int max_linesearch_iterations = 10, max_linear_solver_setups = 4, max_newton_iterations = 10;
  double step_reduction_factor = 0.5, armijo_constant = 1e-4;
};
// ode_solver/problem.rs:96-152
struct OdeSolverOptions {
  int max_nonlinear_solver_iterations = 10, max_error_test_failures = 40, max_nonlinear_solver_failures = 50;
  double nonlinear_solver_tolerance = 0.2, min_timestep = 1e-13;
  std::optional<double> max_timestep_growth, min_timestep_growth, max_timestep_shrink, min_timestep_shrink;
  int update_jacobian_after_steps = 20, update_rhs_jacobian_after_steps = 50;
  double threshold_to_update_jacobian = 0.3, threshold_to_update_rhs_jacobian = 0.2;
  double pi_control_proportional = 0.0, pi_control_integral = 0.5;
};

// OdeSolverProblem (ode_solver/problem.rs:154-193)
struct OdeSolverProblem {
  std::shared_ptr<OdeEquations> eqn;
  double rtol = 1e-6;
  HipVec atol;  // nstates entries, nbatch 1 (broadcast)
  double t0 = 0.0, h0 = 1.0;
  InitialConditionSolverOptions ic_options;
  OdeSolverOptions ode_options;
  bool use_fused_kernels = true;  // build-time choice of this backend: fused device kernels where the model provides them
  // forward sensitivities (problem.bdf_sens(), problem.rs:819-832): s_j = dy/dp_j integrated alongside the states; with sens_rtol / sens_atol they take
  // part in the error control (sens_equations.rs:283-285), without (builder.rs:1501-1505) they do not
  bool sens = false;
  bool sens_error_control = false;
  double sens_rtol = 0.0;
  HipVec sens_atol;  // nstates entries, nbatch 1; the same for every parameter (builder.rs build_atols with unit parameter scales)
  const HipContext& context() const { return eqn->context(); }
};

// OdeBuilder (ode_solver/builder.rs:22-146; defaults :112-139)
class OdeBuilder {
 public:
  OdeBuilder& t0(double v) { t0_ = v; return *this; }
  OdeBuilder& h0(double v) { h0_ = v; return *this; }
  OdeBuilder& rtol(double v) { rtol_ = v; return *this; }
  OdeBuilder& atol(const std::vector<double>& v) { atol_ = v; return *this; }
  OdeBuilder& context(const HipContext& c) { ctx_ = c; return *this; }
  OdeBuilder& use_fused_kernels(bool v) { fused_ = v; return *this; }
  OdeBuilder& ode_options(const OdeSolverOptions& o) { ode_options_ = o; return *this; }
  OdeBuilder& ic_options(const InitialConditionSolverOptions& o) { ic_options_ = o; return *this; }
  // sensitivities(true) = bdf_sens(); sens_tolerances(rtol, atol) = .sens_rtol().sens_atol() (builder.rs:1453-1477), none = turn_off_sensitivities_error_control
  OdeBuilder& sensitivities(bool v) { sens_ = v; return *this; }
  OdeBuilder& sens_tolerances(double rtol, const std::vector<double>& atol) { sens_rtol_ = rtol; sens_atol_ = atol; return *this; }
  // build_from_eqn (builder.rs:1933-1981): atol of length 1 is broadcast to all states
  OdeSolverProblem build_from_eqn(std::shared_ptr<OdeEquations> eqn) const {
    OdeSolverProblem p;
    int64_t n = eqn->nstates();
    std::vector<double> a;
    if (atol_.size() == 1) a.assign((size_t)n, atol_[0]);
    else if ((int64_t)atol_.size() == n) a = atol_;
    else throw LaError(DSH_E_INVALID, "atol must have length 1 or nstates");
    p.atol = HipVec::from_vec(a, eqn->context().clone_with_nbatch(1));
    p.eqn = std::move(eqn);
    p.rtol = rtol_; p.t0 = t0_; p.h0 = h0_;
    p.ode_options = ode_options_; p.ic_options = ic_options_;
    p.use_fused_kernels = fused_;
    if (sens_) {
      if (!p.eqn->has_sens()) throw LaError(DSH_E_UNSUPPORTED, "forward sensitivities: the model has no parameter derivatives (dsh_model_has_sens)");
      p.sens = true;
      p.sens_error_control = !sens_atol_.empty();
      p.sens_rtol = sens_rtol_;
      std::vector<double> sa((size_t)n, 0.0);
      if (sens_atol_.size() == 1) sa.assign((size_t)n, sens_atol_[0]);
      else if ((int64_t)sens_atol_.size() == n) sa = sens_atol_;
      else if (!sens_atol_.empty()) throw LaError(DSH_E_INVALID, "sens_atol must have length 1 or nstates");
      p.sens_atol = HipVec::from_vec(sa, p.eqn->context().clone_with_nbatch(1));
    }
    return p;
  }
  OdeSolverProblem build_model(int model, int64_t size, const std::vector<double>& params) const {
    return build_from_eqn(std::make_shared<HipKernelEquations>(model, size, params, ctx_));
  }

 private:
  double t0_ = 0.0, h0_ = 1.0, rtol_ = 1e-6;
  std::vector<double> atol_{1e-6};
  HipContext ctx_;
  bool fused_ = true;
  bool sens_ = false;
  double sens_rtol_ = 0.0;
  std::vector<double> sens_atol_;
  OdeSolverOptions ode_options_;
  InitialConditionSolverOptions ic_options_;
};

// jacobian_update.rs:12-79
class JacobianUpdate {
 public:
  explicit JacobianUpdate(const OdeSolverOptions& o)
      : threshold_to_update_jacobian_(o.threshold_to_update_jacobian), threshold_to_update_rhs_jacobian_(o.threshold_to_update_rhs_jacobian),
        update_jacobian_after_steps_(o.update_jacobian_after_steps), update_rhs_jacobian_after_steps_(o.update_rhs_jacobian_after_steps) {}
  void update_jacobian(double h) { steps_since_jacobian_eval_ = 0; h_at_last_jacobian_update_ = h; }
  void update_rhs_jacobian(double h) { steps_since_rhs_jacobian_eval_ = 0; steps_since_jacobian_eval_ = 0; h_at_last_jacobian_update_ = h; }
  void step() { steps_since_jacobian_eval_++; steps_since_rhs_jacobian_eval_++; }
  bool check_jacobian_update(double h, SolverState st) const {
    if (st == SolverState::StepSuccess)
      return steps_since_jacobian_eval_ >= update_jacobian_after_steps_ || std::fabs(h / h_at_last_jacobian_update_ - 1.0) > threshold_to_update_jacobian_;
    return true;
  }
  bool check_rhs_jacobian_update(double h, SolverState st) const {
    switch (st) {
      case SolverState::StepSuccess: return steps_since_rhs_jacobian_eval_ >= update_rhs_jacobian_after_steps_;
      case SolverState::FirstConvergenceFail: return std::fabs(h / h_at_last_jacobian_update_ - 1.0) < threshold_to_update_rhs_jacobian_;
      case SolverState::SecondConvergenceFail: return steps_since_rhs_jacobian_eval_ > 0;
      case SolverState::ErrorTestFail: return false;
      case SolverState::Checkpoint: return true;
    }
    return false;
  }

 private:
  int steps_since_jacobian_eval_ = 0, steps_since_rhs_jacobian_eval_ = 0;
  double h_at_last_jacobian_update_ = 1.0;
  double threshold_to_update_jacobian_, threshold_to_update_rhs_jacobian_;
  int update_jacobian_after_steps_, update_rhs_jacobian_after_steps_;
};

// runge_kutta.rs:1313-1336
inline double pi_controller_raw(double error_norm, std::optional<double> prev_error_norm, double pi_integral, double pi_proportional, int eff_order) {
  double order_f = (double)eff_order, ki = pi_integral / order_f;
  if (pi_proportional == 0.0) return hpow(error_norm, -ki);
  if (prev_error_norm) {
    double kp = pi_proportional / order_f;
    return hpow(error_norm, -(ki + kp)) * hpow(*prev_error_norm, kp);
  }
  return hpow(error_norm, -ki);
}

// StateCommon (ode_solver/state.rs) restricted to the main equations
struct StateCommon { HipVec y, dy; double t = 0.0, h = 0.0; };

// op/init.rs:14-135
class InitOp : public NonLinearOpRef {
 public:
  // rhs_fn / jac_fn: the right-hand side and its Jacobian when the equations are built ON `eqn` rather than being it — InitOp::new(augmented_eqn, ..)
  // in set_consistent_augmented (state.rs:209-214): SensRhs, whose Jacobian is the state equations' at the linearisation point (sens_equations.rs:185-187)
  InitOp(const OdeEquations& eqn, double t0, const HipVec& y0, const std::vector<int>& alg, std::function<void(const HipVec&, double, HipVec&)> rhs_fn = nullptr,
         std::function<void(double, HipMat&)> jac_fn = nullptr)
      : eqn_(eqn), y0_(y0.clone()), alg_host_(alg), alg_(alg, eqn.context()), rhs_fn_(std::move(rhs_fn)) {
    const int64_t n = eqn.nstates();
    const HipContext& ctx = eqn.context();
    HipMat rhs_jac = HipMat::zeros(n, n, ctx), mass = HipMat::zeros(n, n, ctx);
    if (jac_fn) jac_fn(t0, rhs_jac);
    else eqn.rhs_jacobian_inplace(y0, t0, rhs_jac);
    eqn.mass_matrix_inplace(t0, mass);
    // jac = (-M_u, df/dv; 0, dg/dv), neg_mass = (-M_u, 0; 0, 0) in the original ordering (Matrix::split / combine, matrix/mod.rs:261-303)
    jac_ = HipMat::zeros(n, n, ctx);
    neg_mass_ = HipMat::zeros(n, n, ctx);
    std::vector<char> is_alg((size_t)n, 0);
    for (int i : alg) is_alg[(size_t)i] = 1;
    HipVec col = HipVec::zeros(n, ctx);
    for (int64_t j = 0; j < n; ++j) {
      if (!is_alg[(size_t)j]) {
        col.copy_from_view(mass.column(j));
        col.mul_assign(scale(-1.0));
        col.assign_at_indices(alg_, 0.0);
        jac_.set_column(j, col);
        neg_mass_.set_column(j, col);
      } else {
        col.copy_from_view(rhs_jac.column(j));
        jac_.set_column(j, col);
      }
    }
  }
  int64_t nstates() const override { return eqn_.nstates(); }
  const HipContext& context() const override { return eqn_.context(); }
  void call_inplace(const HipVec& x, double t, HipVec& y) override {  // :103-115
    y0_.copy_from_indices(x, alg_);
    if (rhs_fn_) rhs_fn_(y0_, t, y);
    else eqn_.rhs_call_inplace(y0_, t, y);
    neg_mass_.gemv(1.0, x, 1.0, y);
  }
  void jacobian_inplace(const HipVec&, double, HipMat& y) override { y.copy_from(jac_); }  // :125-127
  void scatter_soln(const HipVec& soln, HipVec& y, HipVec& dy) const {  // :76-81
    HipVec tmp = dy.clone();
    dy.copy_from(soln);
    dy.copy_from_indices(tmp, alg_);
    y.copy_from_indices(soln, alg_);
  }
  const HipIndex& algebraic_indices() const { return alg_; }

 private:
  const OdeEquations& eqn_;
  HipMat jac_, neg_mass_;
  HipVec y0_;
  std::vector<int> alg_host_;
  HipIndex alg_;
  std::function<void(const HipVec&, double, HipVec&)> rhs_fn_;
};

// state.rs:1086-1124
inline StateCommon new_without_initialise(const OdeSolverProblem& pr) {
  StateCommon s;
  s.t = pr.t0; s.h = pr.h0;
  s.y = HipVec::zeros(pr.eqn->nstates(), pr.context());
  s.dy = HipVec::zeros(pr.eqn->nstates(), pr.context());
  pr.eqn->init_call_inplace(s.t, s.y);
  pr.eqn->rhs_call_inplace(s.y, s.t, s.dy);
  return s;
}

// state.rs:84-162.  no_linesearch: the root solver apply_reset_with_mass builds (state.rs:299: NewtonNonlinearSolver::new(LS::default(), NoLineSearch))
inline void set_consistent(StateCommon& s, const OdeSolverProblem& pr, bool no_linesearch = false) {
  const OdeEquations& eqn = *pr.eqn;
  if (!eqn.has_mass()) return;
  const int64_t n = eqn.nstates();
  HipMat mass = HipMat::zeros(n, n, pr.context());
  eqn.mass_matrix_inplace(pr.t0, mass);
  // partition_indices_by_zero_diagonal: zero diagonal entries of M(t0), read from batch member 0
  std::vector<double> diag = mass.diagonal().clone_as_vec();
  std::vector<int> alg;
  for (int64_t i = 0; i < n; ++i) if (diag[(size_t)i] == 0.0) alg.push_back((int)i);
  if (alg.empty()) return;
  InitOp f(eqn, pr.t0, s.y, alg);
  NewtonNonlinearSolver root_solver;
  root_solver.set_problem(f);
  HipVec y_tmp = s.dy.clone();
  y_tmp.copy_from_indices(s.y, f.algebraic_indices());
  HipVec yerr = y_tmp.clone();
  Convergence conv(pr.rtol, &pr.atol, pr.ode_options.nonlinear_solver_tolerance);
  conv.set_max_iter(pr.ic_options.max_newton_iterations);
  std::unique_ptr<LineSearch> ls;
  if (pr.ic_options.use_linesearch && !no_linesearch) {
    auto b = std::make_unique<BacktrackingLineSearch>();
    b->c = pr.ic_options.armijo_constant; b->max_iter = pr.ic_options.max_linesearch_iterations; b->tau = pr.ic_options.step_reduction_factor;
    ls = std::move(b);
  } else ls = std::make_unique<NoLineSearch>();
  NlError result = NlError::Ok;
  for (int k = 0; k < pr.ic_options.max_linear_solver_setups; ++k) {
    root_solver.reset_jacobian(f, y_tmp, s.t);
    result = root_solver.solve_in_place(f, y_tmp, s.t, yerr, conv, *ls);
    if (result == NlError::Ok) break;
    if (result != NlError::NewtonMaxIterations) throw DSH_ODE_ERR(InitialConditionDidNotConverge);
    yerr.copy_from(y_tmp);
  }
  if (result != NlError::Ok) throw DSH_ODE_ERR(InitialConditionDidNotConverge);
  f.scatter_soln(y_tmp, s.y, s.dy);
  s.dy.assign_at_indices(f.algebraic_indices(), 0.0);
}

// The DAE half of set_consistent_augmented (state.rs:187-238), shared by Bdf and Sdirk: per parameter one Newton solve on InitOp over the sensitivity
// equations for (ds_j on the differential, s_j on the algebraic components), with the tolerances of the STATE equations and the consistent-initialisation
// options.  sens_rhs(j, x, t, y) = SensRhs::call_inplace for parameter j about the linearisation point sens_y.
template <class SensRhsFn>
inline void sens_set_consistent_augmented(const OdeSolverProblem& pr, double t0, const HipVec& sens_y, std::vector<HipVec>& s, std::vector<HipVec>& ds, SensRhsFn sens_rhs) {
  const OdeEquations& eqn = *pr.eqn;
  if (!eqn.has_mass()) return;
  const int64_t n0 = eqn.nstates();
  HipMat mass = HipMat::zeros(n0, n0, pr.context());
  eqn.mass_matrix_inplace(pr.t0, mass);
  std::vector<double> diag = mass.diagonal().clone_as_vec();
  std::vector<int> alg;
  for (int64_t i = 0; i < n0; ++i) if (diag[(size_t)i] == 0.0) alg.push_back((int)i);
  if (alg.empty()) return;
  Convergence conv(pr.rtol, &pr.atol, pr.ode_options.nonlinear_solver_tolerance);
  conv.set_max_iter(pr.ic_options.max_newton_iterations);
  std::unique_ptr<LineSearch> ls;
  if (pr.ic_options.use_linesearch) {
    auto b = std::make_unique<BacktrackingLineSearch>();
    b->c = pr.ic_options.armijo_constant; b->max_iter = pr.ic_options.max_linesearch_iterations; b->tau = pr.ic_options.step_reduction_factor;
    ls = std::move(b);
  } else ls = std::make_unique<NoLineSearch>();
  for (size_t j = 0; j < s.size(); ++j) {
    InitOp f(eqn, t0, s[j], alg, [&sens_rhs, j](const HipVec& x, double t, HipVec& y) { sens_rhs((int)j, x, t, y); },
             [&eqn, &sens_y](double t, HipMat& out) { eqn.rhs_jacobian_inplace(sens_y, t, out); });
    NewtonNonlinearSolver root_solver;
    root_solver.set_problem(f);
    HipVec y_tmp = ds[j].clone();
    y_tmp.copy_from_indices(s[j], f.algebraic_indices());
    HipVec yerr = y_tmp.clone();
    NlError result = NlError::Ok;
    for (int k = 0; k < pr.ic_options.max_linear_solver_setups; ++k) {
      root_solver.reset_jacobian(f, y_tmp, t0);
      result = root_solver.solve_in_place(f, y_tmp, t0, yerr, conv, *ls);
      if (result == NlError::Ok) break;
      if (result != NlError::NewtonMaxIterations) throw DSH_ODE_ERR(InitialConditionDidNotConverge);
      yerr.copy_from(y_tmp);
    }
    if (result != NlError::Ok) throw DSH_ODE_ERR(InitialConditionDidNotConverge);
    f.scatter_soln(y_tmp, s[j], ds[j]);
  }
}

// state.rs:1209-1277
inline void set_step_size(StateCommon& s, double h0_in, const HipVec& atol, double rtol, const OdeEquations& eqn, int solver_order) {
  const bool is_neg_h = h0_in < 0.0;
  const HipVec& y0 = s.y;
  const HipVec& f0 = s.dy;
  const double t0 = s.t;
  double d0 = std::sqrt(y0.squared_norm(y0, atol, rtol));
  double d1 = std::sqrt(f0.squared_norm(y0, atol, rtol));
  double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
  // y1 = f0*(+-h0) + y0
  HipVec y1 = f0.mul(scale(is_neg_h ? -h0 : h0));
  y1.add_assign(y0);
  HipVec f1 = HipVec::zeros(y0.len(), y0.context());
  eqn.rhs_call_inplace(y1, is_neg_h ? t0 - h0 : t0 + h0, f1);
  HipVec df = f1.sub(f0);
  double d2 = std::sqrt(df.squared_norm(y0, atol, rtol)) / std::fabs(h0);
  double max_d = d2;
  if (max_d < d1) max_d = d1;
  double h1;
  if (max_d < 1e-15) { h1 = h0 * 1e-3; if (h1 < 1e-6) h1 = 1e-6; }
  else h1 = hpow(0.01 / max_d, 1.0 / (1.0 + (double)solver_order));
  s.h = 100.0 * h0;
  if (s.h > h1) s.h = h1;
  if (is_neg_h) s.h = -s.h;
}

// state.rs:969-997
inline StateCommon new_and_consistent(const OdeSolverProblem& pr, int solver_order) {
  StateCommon s = new_without_initialise(pr);
  set_consistent(s, pr);
  set_step_size(s, pr.h0, pr.atol, pr.rtol, *pr.eqn, solver_order);
  return s;
}

// nonlinear_solver/root.rs:12-222.  Values are read from batch member 0 (:46, :103-104); Vector::root_finding throws if batches disagree.
class RootFinder {
 public:
  RootFinder(int64_t nroots, int64_t nstates, const HipContext& ctx)
      : g0_(HipVec::zeros(nroots, ctx)), g1_(HipVec::zeros(nroots, ctx)), gmid_(HipVec::zeros(nroots, ctx)), ymid_(HipVec::zeros(nstates, ctx)) {}
  void init(const OdeEquations& eqn, const HipVec& y, double t) { eqn.root_call_inplace(y, t, g0_); t0_ = t; }
  template <class Interp>
  std::optional<std::pair<double, int>> check_root(const Interp& interpolate_inplace, const OdeEquations& eqn, const HipVec& y, double t) {
    eqn.root_call_inplace(y, t, g1_);
    bool rootfnd; double frac; int imax_i;
    g0_.root_finding(g1_, rootfnd, frac, imax_i);
    if (imax_i < 0) {
      std::swap(g0_, g1_);
      t0_ = t;
      if (rootfnd) return std::make_pair(t, find_zero_index(g0_));
      return std::nullopt;
    }
    int imax = imax_i;
    double alpha = 1.0;
    bool sign_change[2] = {false, true};
    int i = 0;
    double t1 = t, t0 = t0_;
    const double eps = std::numeric_limits<double>::epsilon();
    double tol = 100.0 * eps * (std::fabs(t1) + std::fabs(t1 - t0));
    while (std::fabs(t1 - t0) > tol) {
      double g1v = batch0(g1_, imax), g0v = batch0(g0_, imax);
      double t_mid = t1 - (t1 - t0) * g1v / (g1v - alpha * g0v);
      if (std::fabs(t_mid - t0) < 0.5 * tol) {
        double fracint = std::fabs(t1 - t0) / tol;
        double fracsub = fracint > 5.0 ? 0.1 : 0.5 / fracint;
        t_mid = t0 + fracsub * (t1 - t0);
      }
      if (std::fabs(t1 - t_mid) < 0.5 * tol) {
        double fracint = std::fabs(t1 - t0) / tol;
        double fracsub = fracint > 5.0 ? 0.1 : 0.5 / fracint;
        t_mid = t1 - fracsub * (t1 - t0);
      }
      interpolate_inplace(t_mid, ymid_);
      eqn.root_call_inplace(ymid_, t_mid, gmid_);
      bool rf; double fr; int im;
      g0_.root_finding(gmid_, rf, fr, im);
      bool lower = im >= 0;
      if (lower) { t1 = t_mid; imax = im; std::swap(g1_, gmid_); }
      else if (rf) { eqn.root_call_inplace(y, t, g0_); return std::make_pair(t_mid, imax); }
      else { t0 = t_mid; std::swap(g0_, gmid_); }
      sign_change[i % 2] = lower;
      if (i >= 2) alpha = (sign_change[0] != sign_change[1]) ? 1.0 : (sign_change[0] ? 0.5 * alpha : 2.0 * alpha);
      i += 1;
    }
    eqn.root_call_inplace(y, t, g0_);
    return std::make_pair(t1, imax);
  }

 private:
  static double batch0(const HipVec& g, int64_t i) {
    double out = 0.0;
    check(dsh_vec_get_index(g.context().raw(), g.nb(), g.ptr(), i, 0, &out), "root value");
    return out;
  }
  static int find_zero_index(const HipVec& g) {
    int mi = 0;
    double mv = std::fabs(batch0(g, 0));
    for (int64_t i = 1; i < g.len(); ++i) { double v = std::fabs(batch0(g, i)); if (v < mv) { mv = v; mi = (int)i; } }
    return mi;
  }
  double t0_ = 0.0;
  HipVec g0_, g1_, gmid_, ymid_;
};

// OdeSolverMethod (ode_solver/method.rs:42-198) restricted to the main equations
class OdeSolverMethod {
 public:
  virtual ~OdeSolverMethod() = default;
  virtual OdeSolverStopReason step() = 0;
  virtual void set_stop_time(double tstop) = 0;
  virtual void interpolate_inplace(double t, HipVec& y) const = 0;
  virtual const HipVec& y() const = 0;
  virtual const HipVec& dy() const = 0;
  virtual double t() const = 0;
  virtual double h() const = 0;
  virtual int order() const = 0;
  virtual const OdeSolverStatistics& get_statistics() const = 0;
  virtual const OdeSolverProblem& problem() const = 0;
  virtual void state_mut_back(double t) = 0;
  // Bdf / Sdirk::apply_reset (bdf.rs:1017-1020, sdirk.rs:368-374) over StateRefMut::apply_reset_with_mass (state.rs:279-306): y <- reset(y, t), then dy <- f(y, t)
  // or, with a mass matrix, set_consistent with a Newton solver without line search; state marked as modified
  virtual void apply_reset() = 0;
  double root_time = 0.0;
  int root_index = -1;

  HipVec interpolate(double t) const {
    HipVec y = HipVec::zeros(problem().eqn->nstates(), problem().context());
    interpolate_inplace(t, y);
    return y;
  }

  // solve (method.rs:227-258 + :881-964): one output column per accepted step (plus the initial state)
  OdeSolverStopReason solve(double final_time, HipMat& ret_y, std::vector<double>& ret_t) {
    const int64_t n = problem().eqn->nstates();
    ret_y = HipMat::zeros(n, 10, problem().context());
    ret_t.clear();
    auto write_out = [&]() {
      ret_t.push_back(t());
      int64_t i = (int64_t)ret_t.size() - 1;
      if (i >= ret_y.ncols()) ret_y.resize_cols(2 * ret_y.ncols());
      ret_y.column_mut(i).copy_from(y());
    };
    write_out();
    set_stop_time(final_time);
    OdeSolverStopReason reason;
    while (true) {
      reason = step();
      if (reason == OdeSolverStopReason::InternalTimestep) { write_out(); continue; }
      if (reason == OdeSolverStopReason::TstopReached) { write_out(); break; }
      state_mut_back(root_time);
      if (problem().eqn->has_reset()) {  // method.rs:926-944: the reset state is written out at the root time, then the solve continues
        apply_reset();
        write_out();
        if (t() < final_time) { set_stop_time(final_time); continue; }
        reason = OdeSolverStopReason::TstopReached;
        break;
      }
      write_out();
      break;
    }
    ret_y.resize_cols((int64_t)ret_t.size());
    return reason;
  }

  // solve_dense (method.rs:467-520 + :721-848): interpolated output at t_eval
  OdeSolverStopReason solve_dense(const std::vector<double>& t_eval, HipMat& ret) {
    const int64_t n = problem().eqn->nstates();
    if (t_eval.empty()) throw DSH_ODE_ERR(InvalidTEval);
    for (size_t k = 0; k + 1 < t_eval.size(); ++k) if (t_eval[k] > t_eval[k + 1]) throw DSH_ODE_ERR(InvalidTEval);
    if (t_eval[0] < t()) throw DSH_ODE_ERR(InvalidTEval);
    ret = HipMat::zeros(n, (int64_t)t_eval.size(), problem().context());
    HipVec tmp = HipVec::zeros(n, problem().context());
    set_stop_time(t_eval.back());
    size_t col = 0;
    OdeSolverStopReason reason;
    auto drain = [&](double upto) {
      while (col < t_eval.size() && t_eval[col] <= upto) {
        interpolate_inplace(t_eval[col], tmp);
        ret.column_mut((int64_t)col).copy_from(tmp);
        col++;
      }
    };
    while (true) {
      reason = step();
      if (reason == OdeSolverStopReason::InternalTimestep) { drain(t()); continue; }
      if (reason == OdeSolverStopReason::TstopReached) { drain(t()); break; }
      drain(root_time);
      state_mut_back(root_time);
      if (problem().eqn->has_reset()) {  // a reset operator is configured (method.rs:774-797): apply it at the root and continue to the last evaluation time
        apply_reset();
        if (t() < t_eval.back()) { set_stop_time(t_eval.back()); continue; }
        reason = OdeSolverStopReason::TstopReached;
        break;
      }
      if (col < t_eval.size()) {
        ret.column_mut((int64_t)col).copy_from(y());
        if ((int64_t)col + 1 < ret.ncols()) ret.resize_cols((int64_t)col + 1);
      }
      break;
    }
    return reason;
  }
};

}  // namespace diffsol_hip
