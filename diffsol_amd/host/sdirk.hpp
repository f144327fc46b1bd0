// Host-side (E)SDIRK integrators over the HIP backend: mirror of crates/diffsol/src/ode_solver/sdirk.rs (Sdirk), runge_kutta.rs (Rk
// core), tableau.rs (Tableau::{tr_bdf2, esdirk34}), op/sdirk.rs (SdirkCallable), sdirk_state.rs (RkState).  Lock-step over the ensemble.
// Stage Newton iterations use the fused device kernel (dsh_sdirk_newton_iter) for register-resident models; everything else goes
// through the 1:1 trait ops.
//   tableau.rs:41-160   op/sdirk.rs:18-300   runge_kutta.rs:110-190, :466-495, :505-516, :610-689, :752-800, :841-960, :962-1127
//   sdirk.rs:172-215 _new, :260-303 jacobian_updates, :409-543 step
#pragma once
#include <functional>
#include <memory>
#include "ode.hpp"

namespace diffsol_hip {

struct Tableau {
  int s = 0, order_ = 0;
  std::vector<double> a;  // s x s column-major
  std::vector<double> b, c, d;
  bool has_beta = false;
  int beta_rows = 0, beta_cols = 0;
  std::vector<double> beta;  // column-major
  double A(int i, int j) const { return a[(size_t)(j * s + i)]; }
  int order() const { return order_; }
  static Tableau tr_bdf2() {  // tableau.rs:41-98
    Tableau t;
    t.s = 3; t.order_ = 2;
    const double gamma = 2.0 - std::sqrt(2.0), d = gamma / 2.0, w = std::sqrt(2.0) / 4.0;
    t.a = {0.0, d, w, 0.0, d, w, 0.0, 0.0, d};
    t.b = {w, w, d};
    const std::vector<double> b_hat = {(1.0 - w) / 3.0, (3.0 * w + 1.0) / 3.0, d / 3.0};
    t.d.resize(3);
    for (int i = 0; i < 3; ++i) t.d[(size_t)i] = t.b[(size_t)i] - b_hat[(size_t)i];
    t.has_beta = true; t.beta_rows = 3; t.beta_cols = 2;
    t.beta = {2.0 * w, 2.0 * w, gamma - 1.0, -w, -w, 2.0 * w};
    t.c = {0.0, gamma, 1.0};
    return t;
  }
  static Tableau esdirk34() {  // tableau.rs:101-159
    Tableau t;
    t.s = 4; t.order_ = 3;
    const double gamma = 0.435866521508459;
    t.a = {0.0, gamma, 0.1407377747247062, 0.102399400619911, 0.0, gamma, -0.1083655513813208, -0.3768784522555561,
           0.0, 0.0, gamma, 0.8386125301271861, 0.0, 0.0, 0.0, gamma};
    t.b = {t.A(3, 0), t.A(3, 1), t.A(3, 2), t.A(3, 3)};
    t.c = {0.0, 0.871733043016918, 0.4682387448518444, 1.0};
    t.d = {-0.05462549724041394, -0.49420889362599496, 0.22193449973506466, 0.32689989113134427};
    return t;
  }
};

// op/sdirk.rs:18-300
class SdirkCallable : public NonLinearOpRef {
 public:
  SdirkCallable(const OdeEquations& eqn, double c)
      : eqn_(eqn), c_(c), phi_(HipVec::zeros(eqn.nstates(), eqn.context())), tmp_(HipVec::zeros(eqn.nstates(), eqn.context())),
        rhs_jac_(HipMat::zeros(eqn.nstates(), eqn.nstates(), eqn.context())) {
    const int64_t n = eqn.nstates();
    int pkl = 0, pku = 0;
    if (eqn.packed_band(&pkl, &pku)) {  // declared narrow band, identity mass: band containers for f_y, M and (through packed_band below) M - cJ and its factors
      packed_ = true; pkl_ = pkl; pku_ = pku;
      rhs_jac_ = HipMat::zeros_banded(n, pkl, pku, eqn.context());
      mass_jac_ = HipMat::from_diagonal_banded(HipVec::from_element(n, 1.0, eqn.context()), pkl, pku);
    } else
    if (!eqn.has_mass()) mass_jac_ = HipMat::from_diagonal(HipVec::from_element(n, 1.0, eqn.context()));
    else mass_jac_ = HipMat::zeros(n, n, eqn.context());
  }
  int64_t nstates() const override { return eqn_.nstates(); }
  const HipContext& context() const override { return eqn_.context(); }
  bool packed_band(int* kl, int* ku) const override { if (!packed_) return false; *kl = pkl_; *ku = pku_; return true; }
  void set_h(double h) { h_ = h; }
  double h() const { return h_; }
  double c() const { return c_; }
  // phi = y0 + diff[:,0..ncols) * a   (set_phi with h = 1, :174-184)
  void set_phi(const HipMatView& diff, const HipVec& y0, const HipVec& a) {
    if (y0.nb() == phi_.nb()) { diff.gemv_from(1.0, a, 1.0, y0, phi_); return; }  // the copy and the gemv in one pass (dsh_mat_gemv_from)
    phi_.copy_from(y0);
    diff.gemv_o(1.0, a, 1.0, phi_);
  }
  void set_tmp(const HipVec& x) { tmp_.copy_from(phi_); tmp_.axpy(c_, x, 1.0); }                               // :186-195
  void get_f_eval(const HipVec& x, HipVec& f_eval) const { f_eval.copy_from(phi_); f_eval.axpy(c_, x, 1.0); }  // :197-203
  // get_f_eval and the copy of the stage increment into its column of diff (runge_kutta.rs:672-676) in one pass
  void get_f_eval_and_store(const HipVec& x, HipVec& f_eval, double* column) const { f_eval.assign_axpby(c_, x.ptr(), 1.0, phi_.ptr(), column); }
  void set_jacobian_is_stale() { jacobian_is_stale_ = true; }
  bool jacobian_is_stale() const { return jacobian_is_stale_; }
  void clear_jacobian_is_stale() { jacobian_is_stale_ = false; }
  // SdirkCallable over the sensitivity equations (s_op): the right-hand side is SensRhs for the current parameter
  std::function<void(const HipVec&, double, HipVec&)> rhs_override;
  void call_inplace(const HipVec& x, double t, HipVec& y) override {  // :229-244
    set_tmp(x);
    if (rhs_override) rhs_override(tmp_, t, y);
    else eqn_.rhs_call_inplace(tmp_, t, y);
    const double beta = -h_;
    if (eqn_.has_mass()) eqn_.mass_gemv_inplace(x, t, beta, y);
    else y.axpy(1.0, x, beta);
  }
  void jacobian_inplace(const HipVec& x, double t, HipMat& y) override {  // :266-296
    if (jacobian_is_stale_) {
      set_tmp(x);
      eqn_.rhs_jacobian_inplace(tmp_, t, rhs_jac_);
      if (eqn_.has_mass()) eqn_.mass_matrix_inplace(t, mass_jac_);
      y.scale_add_and_assign(mass_jac_, -(c_ * h_), rhs_jac_);
      jacobian_is_stale_ = false;
    } else {
      y.scale_add_and_assign(mass_jac_, -(c_ * h_), rhs_jac_);
    }
  }
  HipVec& phi() { return phi_; }
  HipVec& tmp() { return tmp_; }
  HipMat& rhs_jac() { return rhs_jac_; }
  HipMat& mass_jac() { return mass_jac_; }
  const HipMat& current_mass() const { return mass_jac_; }

 private:
  const OdeEquations& eqn_;
  double c_, h_ = 0.0;
  HipVec phi_, tmp_;
  HipMat rhs_jac_, mass_jac_;
  bool packed_ = false;
  int pkl_ = 0, pku_ = 0;
  bool jacobian_is_stale_ = true;
};

class Sdirk : public OdeSolverMethod {
 public:
  // problem.tr_bdf2::<LS>() / esdirk34::<LS>() (problem.rs:839-861): consistent RkState with solver_order = tableau.order()
  Sdirk(const OdeSolverProblem& problem, Tableau tableau)
      : pr_(problem), tab_(std::move(tableau)), convergence_(problem.rtol, &problem.atol, problem.ode_options.nonlinear_solver_tolerance),
        op_(*problem.eqn, tab_.A(1, 1)), jacobian_update_(problem.ode_options) {
    const OdeSolverOptions& o = problem.ode_options;
    minimum_timestep_ = o.min_timestep;
    maximum_error_test_failures_ = o.max_error_test_failures;
    maximum_newton_fails_ = o.max_nonlinear_solver_failures;
    maximum_timestep_growth_ = o.max_timestep_growth.value_or(2.0);
    minimum_timestep_growth_ = o.min_timestep_growth.value_or(2.0);
    maximum_timestep_shrink_ = o.max_timestep_shrink.value_or(0.9);
    minimum_timestep_shrink_ = o.min_timestep_shrink.value_or(0.5);
    // forward sensitivities run on the trait operations (the fused stage kernels integrate the state equations only)
    fused_ = problem.use_fused_kernels && !problem.sens && problem.eqn->fused_model(&model_, &model_size_);
    // run-time-sized registry models: the Newton iteration in its staged form (three launches, one wait: dsh_sdirk_newton_iter); everything else stays generic
    staged_ = !fused_ && problem.use_fused_kernels && !problem.sens && problem.eqn->registry_model(&model_, &model_size_) && dsh_model_has_staged_newton(model_, model_size_) != 0;
    state_ = new_and_consistent(problem, tab_.order());
    const int64_t n = problem.eqn->nstates();
    const HipContext& ctx = problem.context();
    const HipContext ctx1 = ctx.clone_with_nbatch(1);
    for (int i = 0; i < tab_.s; ++i) {  // a_rows (runge_kutta.rs:127-138)
      std::vector<double> row;
      for (int j = 0; j < i; ++j) row.push_back(tab_.A(i, j));
      a_rows_.push_back(row.empty() ? HipVec::zeros(0, ctx1) : HipVec::from_vec(row, ctx1));
    }
    d_vec_ = HipVec::from_vec(tab_.d, ctx1);
    if (problem.eqn->nroots() > 0) { root_finder_.emplace(problem.eqn->nroots(), n, ctx); root_finder_->init(*problem.eqn, state_.y, state_.t); }
    diff_ = HipMat::zeros(n, tab_.s, ctx);
    if (problem.sens) {
      // RkState::new_with_sensitivities_and_consistent (state.rs:1032-1083): s_j = SensInit(t0), ds_j = SensRhs(s_j) about (y0, t0), DAEs through InitOp
      const int64_t npar = problem.eqn->nparams();
      sens_mat_ = HipMat::zeros(n, npar, ctx);
      sens_y_ = HipVec::zeros(n, ctx);
      HipMat s0 = HipMat::zeros(n, npar, ctx);
      problem.eqn->init_sens_inplace(state_.t, s0);
      sens_update_state(state_.y, state_.t);
      for (int64_t j = 0; j < npar; ++j) {
        HipVec sj = HipVec::zeros(n, ctx), dsj = HipVec::zeros(n, ctx);
        sj.copy_from_view(s0.column(j));
        sens_rhs_call((int)j, sj, state_.t, dsj);
        s_.push_back(std::move(sj));
        ds_.push_back(std::move(dsj));
      }
      sens_set_consistent_augmented(problem, state_.t, sens_y_, s_, ds_, [this](int j, const HipVec& x, double t, HipVec& y) { sens_rhs_call(j, x, t, y); });
      for (int64_t j = 0; j < npar; ++j) {
        old_s_.push_back(s_[(size_t)j].clone());
        old_ds_.push_back(ds_[(size_t)j].clone());
        sdiff_.push_back(HipMat::zeros(n, tab_.s, ctx));
      }
      sens_error_ = HipVec::zeros(n, ctx);
      s_op_ = std::make_unique<SdirkCallable>(*problem.eqn, tab_.A(1, 1));
      s_op_->rhs_override = [this](const HipVec& x, double t, HipVec& y) { sens_rhs_call(sens_index_, x, t, y); };
    }
    old_state_.y = state_.y.clone(); old_state_.dy = state_.dy.clone(); old_state_.t = state_.t; old_state_.h = state_.h;
    error_ = HipVec::zeros(n, ctx);
    error_tmp_ = HipVec::zeros(n, ctx);
    // Sdirk::_new (sdirk.rs:172-215)
    jacobian_update_.update_jacobian(state_.h);
    jacobian_update_.update_rhs_jacobian(state_.h);
    convergence_.set_max_iter(o.max_nonlinear_solver_iterations);
    set_op_h(state_.h);
    nonlinear_solver_.set_problem(op_);
    // Sdirk::new_augmented ends with jacobian_updates(h, Checkpoint) (sdirk.rs:251): with sensitivities the first linearisation is made here, at t0 and —
    // op.phi still being zero — about gamma * y0, not lazily inside the first stage (the reference's behaviour; its step counts only reproduce with it)
    if (s_op_) jacobian_updates(state_.h, SolverState::Checkpoint);
  }
  // OdeSolverMethod::interpolate_sens (runge_kutta.rs:1237-1330): the state's interpolant applied to (old_state.s, state.s, sdiff)
  void interpolate_sens_inplace(double t, std::vector<HipVec>& out) const {
    if (!pr_.sens) throw LaError(DSH_E_UNSUPPORTED, "the solver was created without forward sensitivities");
    const bool is_forward = state_.h > 0.0;
    if ((is_forward && (t > state_.t || t < old_state_.t)) || (!is_forward && (t < state_.t || t > old_state_.t))) throw DSH_ODE_ERR(InterpolationTimeOutsideCurrentStep);
    const double dt = state_.t - old_state_.t;
    const double theta = dt == 0.0 ? 1.0 : (t - old_state_.t) / dt;
    out.clear();
    for (size_t j = 0; j < s_.size(); ++j) out.push_back(HipVec::zeros(n(), ctx()));
    if (tab_.has_beta) {
      std::vector<double> thetav{theta};
      for (int i = 1; i < tab_.beta_cols; ++i) thetav.push_back(theta * thetav[(size_t)i - 1]);
      std::vector<double> beta_f((size_t)tab_.beta_rows);
      for (int i = 0; i < tab_.beta_rows; ++i) {
        double acc = 1.0 * tab_.beta[(size_t)i] * thetav[0];
        for (int j = 1; j < tab_.beta_cols; ++j) acc = 1.0 * tab_.beta[(size_t)(j * tab_.beta_rows + i)] * thetav[(size_t)j] + acc;
        beta_f[(size_t)i] = acc;
      }
      HipVec bf = HipVec::from_vec(beta_f, pr_.context().clone_with_nbatch(1));
      for (size_t j = 0; j < s_.size(); ++j) {
        out[j].copy_from(old_s_[j]);
        sdiff_[j].gemv(1.0, bf, 1.0, out[j]);
      }
    } else {
      for (size_t j = 0; j < s_.size(); ++j) {
        HipVec& ret = out[j];
        ret.copy_from(s_[j]);
        ret.sub_assign(old_s_[j]);
        ret.axpy_v(1.0 * (theta - 1.0), sdiff_[j].column(0), 1.0 - 2.0 * theta);
        ret.axpy_v(1.0 * theta, sdiff_[j].column(tab_.s - 1), 1.0);
        ret.axpy(1.0 - theta, old_s_[j], theta * (theta - 1.0));
        ret.axpy(theta, s_[j], 1.0);
      }
    }
  }
  const std::vector<HipVec>& sens() const { return s_; }

  OdeSolverStopReason step() override {  // sdirk.rs:409-543
    if (is_state_mutated_) {  // Rk::start_step (runge_kutta.rs:444-464)
      if (root_finder_) root_finder_->init(*pr_.eqn, state_.y, state_.t);
      if (tstop_) set_stop_time(*tstop_);
      is_state_mutated_ = false;
    }
    double h = state_.h;
    if (std::fabs(h) < minimum_timestep_) throw DSH_ODE_ERR(StepSizeTooSmall);
    set_op_h(h);
    int nattempts = 0;
    bool updated_jacobian = false;
    const int start = skip_first_stage() ? 1 : 0;
    double fac = 1.0, error_norm = 0.0;
    while (true) {
      prepared_stage_ = -1;
      error_ready_ = false;
      if (skip_first_stage()) {  // start_step_attempt (runge_kutta.rs:505-534), "sensitivities too"
        if (single_pass_stages()) {
          // diff[:,0] = h dy, and with it what stage 1 starts with — phi = y + diff[:,0] a10 (set_phi) and its predictor k = diff[:,0] — in one pass
          check(dsh_sdirk_begin_attempt(ctx().raw(), n(), nb(), h, tab_.A(1, 0), state_.dy.ptr(), state_.y.ptr(), diff_.column_mut(0).p, op_.phi().ptr(), old_state_.dy.ptr()),
                "dsh_sdirk_begin_attempt");
          prepared_stage_ = 1;
        } else
        diff_.column_mut(0).axpy(h, state_.dy, 0.0);
        for (size_t j = 0; j < sdiff_.size(); ++j) sdiff_[j].column_mut(0).axpy(h, ds_[j], 0.0);
      }
      bool failed = false;
      for (int i = start; i < tab_.s; ++i) {
        if (do_stage_sdirk(i, h) != NlError::Ok) {
          if (!updated_jacobian) {
            updated_jacobian = true;
            jacobian_updates(h, SolverState::FirstConvergenceFail);
          } else {
            h *= 0.3;
            convergence_.reset_eta_timestep_change();
            set_op_h(h);
            jacobian_updates(h, SolverState::SecondConvergenceFail);
          }
          prev_error_norm_.reset();
          statistics_.number_of_nonlinear_solver_fails += 1;  // solve_fail (runge_kutta.rs:868-892)
          if (statistics_.number_of_nonlinear_solver_fails > maximum_newton_fails_) throw DSH_ODE_ERR(TooManyNonlinearSolverFailures);
          if (std::fabs(h) < minimum_timestep_) throw DSH_ODE_ERR(StepSizeTooSmall);
          failed = true;
          break;
        }
      }
      if (failed) continue;
      // error_norm (runge_kutta.rs:783-800) filtered through one more LU solve (sdirk.rs:474-495)
      if (!error_ready_) diff_.gemv(1.0, d_vec_, 0.0, error_);  // (the last stage's single pass has formed it already)
      if (pr_.eqn->has_mass()) {
        error_tmp_.copy_from(error_);
        op_.current_mass().gemv(1.0, error_tmp_, 0.0, error_);
      }
      {
        double en = 0.0;  // the solve and the norm with one wait (dsh_lu_solve_squared_norm)
        if (!nonlinear_solver_.linear_solver().solve_in_place_and_norm(error_, state_.y, pr_.atol, pr_.rtol, &en)) throw DSH_ODE_ERR(LinearSolveFailed);
        error_norm = std::fmax(0.0, en);
      }
      if (pr_.sens && pr_.sens_error_control)  // runge_kutta.rs:812-822 — no linear solve on the sensitivity error estimates
        for (size_t j = 0; j < sdiff_.size(); ++j) {
          sdiff_[j].gemv(1.0, d_vec_, 0.0, sens_error_);
          error_norm = std::fmax(error_norm, sens_error_.squared_norm(s_[j], pr_.sens_atol, pr_.sens_rtol));
        }
      const double maxiter = (double)convergence_.max_iter(), niter = (double)convergence_.niter();
      const double safety_factor = (2.0 * maxiter + 1.0) / (2.0 * maxiter + niter);
      fac = factor(error_norm, safety_factor);
      if (error_norm < 1.0) break;
      h *= fac;
      convergence_.reset_eta_timestep_change();
      set_op_h(h);
      jacobian_updates(h, SolverState::ErrorTestFail);
      nattempts += 1;
      prev_error_norm_.reset();
      statistics_.number_of_error_test_failures += 1;  // error_test_fail (runge_kutta.rs:841-866)
      if (nattempts >= maximum_error_test_failures_) throw DSH_ODE_ERR(TooManyErrorTestFailures);
      if (std::fabs(h) < minimum_timestep_) throw DSH_ODE_ERR(StepSizeTooSmall);
    }
    const double new_h = h * fac;
    if (fac != 1.0) convergence_.reset_eta_timestep_change();
    set_op_h(new_h);
    jacobian_updates(new_h, SolverState::StepSuccess);
    jacobian_update_.step();
    prev_error_norm_ = error_norm;
    // step_accepted(h, new_h, true) (runge_kutta.rs:894-960)
    old_state_.t = state_.t + h;
    old_state_.h = new_h;
    old_state_.dy.mul_assign(scale(1.0 / h));
    for (HipVec& d : old_ds_) d.mul_assign(scale(1.0 / h));
    std::swap(old_state_, state_);
    std::swap(old_s_, s_);
    std::swap(old_ds_, ds_);
    statistics_.number_of_steps += 1;
    if (root_finder_) {
      auto interp = [&](double tt, HipVec& yy) { interpolate_inplace(tt, yy); };
      auto ret = root_finder_->check_root(interp, *pr_.eqn, state_.y, state_.t);
      if (ret) { root_time = ret->first; root_index = ret->second; return OdeSolverStopReason::RootFound; }
    }
    if (tstop_) {
      if (handle_tstop(*tstop_)) { tstop_.reset(); return OdeSolverStopReason::TstopReached; }
    }
    return OdeSolverStopReason::InternalTimestep;
  }

  void set_stop_time(double tstop) override {  // runge_kutta.rs:436-447
    tstop_ = tstop;
    bool reached;
    try { reached = handle_tstop(tstop); } catch (...) { tstop_.reset(); throw; }
    if (reached) { tstop_.reset(); throw DSH_ODE_ERR(StopTimeAtCurrentTime); }
  }

  void interpolate_inplace(double t, HipVec& ret) const override {  // runge_kutta.rs:1080-1127
    if (ret.len() != state_.y.len()) throw DSH_ODE_ERR(InterpolationVectorWrongSize);
    if (is_state_mutated_) {
      if (t == state_.t) { ret.copy_from(state_.y); return; }
      throw DSH_ODE_ERR(InterpolationTimeOutsideCurrentStep);
    }
    const bool is_forward = state_.h > 0.0;
    if ((is_forward && (t > state_.t || t < old_state_.t)) || (!is_forward && (t < state_.t || t > old_state_.t))) throw DSH_ODE_ERR(InterpolationTimeOutsideCurrentStep);
    const double dt = state_.t - old_state_.t;
    const double theta = dt == 0.0 ? 1.0 : (t - old_state_.t) / dt;
    if (tab_.has_beta) {
      // interpolate_beta_function (runge_kutta.rs:969-983): beta_f = beta * [theta, theta^2, ...], nalgebra gemv order
      std::vector<double> thetav{theta};
      for (int i = 1; i < tab_.beta_cols; ++i) thetav.push_back(theta * thetav[(size_t)i - 1]);
      std::vector<double> beta_f((size_t)tab_.beta_rows);
      for (int i = 0; i < tab_.beta_rows; ++i) {
        double acc = 1.0 * tab_.beta[(size_t)i] * thetav[0];
        for (int j = 1; j < tab_.beta_cols; ++j) acc = 1.0 * tab_.beta[(size_t)(j * tab_.beta_rows + i)] * thetav[(size_t)j] + acc;
        beta_f[(size_t)i] = acc;
      }
      HipVec bf = HipVec::from_vec(beta_f, pr_.context().clone_with_nbatch(1));
      ret.copy_from(old_state_.y);
      diff_.gemv(1.0, bf, 1.0, ret);
    } else {  // interpolate_hermite (runge_kutta.rs:1016-1035), scale_diff = 1
      ret.copy_from(state_.y);
      ret.sub_assign(old_state_.y);
      ret.axpy_v(1.0 * (theta - 1.0), diff_.column(0), 1.0 - 2.0 * theta);
      ret.axpy_v(1.0 * theta, diff_.column(diff_.ncols() - 1), 1.0);
      ret.axpy(1.0 - theta, old_state_.y, theta * (theta - 1.0));
      ret.axpy(theta, state_.y, 1.0);
    }
  }
  void interpolate_dy_inplace(double t, HipVec& dy) const {  // runge_kutta.rs:1129-1181
    if (dy.len() != state_.y.len()) throw DSH_ODE_ERR(InterpolationVectorWrongSize);
    if (is_state_mutated_) {
      if (t == state_.t) { dy.copy_from(state_.dy); return; }
      throw DSH_ODE_ERR(InterpolationTimeOutsideCurrentStep);
    }
    const bool is_forward = state_.h > 0.0;
    if ((is_forward && (t > state_.t || t < old_state_.t)) || (!is_forward && (t < state_.t || t > old_state_.t))) throw DSH_ODE_ERR(InterpolationTimeOutsideCurrentStep);
    const double dt = state_.t - old_state_.t;
    if (dt == 0.0) { dy.copy_from(state_.dy); return; }
    const double theta = (t - old_state_.t) / dt;
    const double scale_diff = 1.0;
    if (tab_.has_beta) {
      // interpolate_beta_function_deriv (runge_kutta.rs:985-1002): d_beta_f = beta * [1, 2 theta, 3 theta^2, ...], nalgebra gemv order
      std::vector<double> d_thetav{1.0};
      double theta_pow = theta;
      for (int i = 1; i < tab_.beta_cols; ++i) { d_thetav.push_back(((double)i + 1.0) * theta_pow); theta_pow *= theta; }
      std::vector<double> d_beta_f((size_t)tab_.beta_rows);
      for (int i = 0; i < tab_.beta_rows; ++i) {
        double acc = 1.0 * tab_.beta[(size_t)i] * d_thetav[0];
        for (int j = 1; j < tab_.beta_cols; ++j) acc = 1.0 * tab_.beta[(size_t)(j * tab_.beta_rows + i)] * d_thetav[(size_t)j] + acc;
        d_beta_f[(size_t)i] = acc;
      }
      HipVec bf = HipVec::from_vec(d_beta_f, pr_.context().clone_with_nbatch(1));
      diff_.gemv(scale_diff / dt, bf, 0.0, dy);
    } else {  // interpolate_hermite_deriv (runge_kutta.rs:1037-1078)
      HipVec q = HipVec::zeros(dy.len(), pr_.context());
      q.copy_from(state_.y);
      q.sub_assign(old_state_.y);
      q.axpy_v(scale_diff * (theta - 1.0), diff_.column(0), 1.0 - 2.0 * theta);
      q.axpy_v(scale_diff * theta, diff_.column(diff_.ncols() - 1), 1.0);
      dy.copy_from(state_.y);
      dy.sub_assign(old_state_.y);
      dy.axpy((2.0 * theta - 1.0) / dt, q, 1.0 / dt);
      q.copy_from(old_state_.y);
      q.sub_assign(state_.y);
      q.axpy_v(scale_diff, diff_.column(0), 2.0);
      q.axpy_v(scale_diff, diff_.column(diff_.ncols() - 1), 1.0);
      dy.axpy(theta * (theta - 1.0) / dt, q, 1.0);
    }
  }
  void apply_reset() override {
    if (pr_.sens) throw LaError(DSH_E_UNSUPPORTED, "apply_reset with forward sensitivities is not supported by the HIP backend");
    HipVec y_out = HipVec::zeros(state_.y.len(), pr_.context());
    pr_.eqn->reset_call_inplace(state_.y, state_.t, y_out);
    state_.y.copy_from(y_out);
    is_state_mutated_ = true;  // state_mut()
    if (pr_.eqn->has_mass()) { set_consistent(state_, pr_, true); return; }  // state.rs:297-300
    pr_.eqn->rhs_call_inplace(state_.y, state_.t, y_out);
    state_.dy.copy_from(y_out);
    is_state_mutated_ = true;  // state_mut()
  }
  void state_mut_back(double t) override {  // runge_kutta.rs:396-434: y and dy from the step's interpolants
    if (pr_.sens) throw LaError(DSH_E_UNSUPPORTED, "state_mut_back with forward sensitivities is not supported by the HIP backend");
    HipVec ynew = HipVec::zeros(state_.y.len(), pr_.context()), dynew = HipVec::zeros(state_.y.len(), pr_.context());
    interpolate_inplace(t, ynew);
    interpolate_dy_inplace(t, dynew);
    state_.y.copy_from(ynew);
    state_.dy.copy_from(dynew);
    state_.t = t;
    is_state_mutated_ = true;
  }

  const HipVec& y() const override { return state_.y; }
  const HipVec& dy() const override { return state_.dy; }
  double t() const override { return state_.t; }
  double h() const override { return state_.h; }
  int order() const override { return tab_.order(); }
  const OdeSolverStatistics& get_statistics() const override { return statistics_; }
  const OdeSolverProblem& problem() const override { return pr_; }
  bool is_fused() const { return fused_; }

 private:
  int64_t n() const { return pr_.eqn->nstates(); }
  int64_t nb() const { return pr_.context().nbatch(); }
  const HipContext& ctx() const { return pr_.context(); }
  bool skip_first_stage() const { return tab_.A(0, 0) == 0.0; }
  // the stage bookkeeping between two Newton solves as single passes (dsh_sdirk_begin_attempt / _next_stage / _finish_error): state equations only, every operand a
  // full n x nbatch vector.  DSH_SDIRK_SINGLE_PASS=0 keeps the trait operations (same bits; tests compare).
  bool single_pass_stages() const {
    static const bool on = [] { const char* e = std::getenv("DSH_SDIRK_SINGLE_PASS"); return !(e && e[0] == '0'); }();
    return on && pr_.use_fused_kernels && !s_op_ && tab_.s >= 2 && tab_.s <= 8 && state_.y.nb() == nb() && state_.dy.nb() == nb() && old_state_.y.nb() == nb() &&
           old_state_.dy.nb() == nb() && error_.nb() == nb();
  }

  // reset_jacobian(op, x, t) with x := state.y, linearised at phi + c*x (the reference's quirk, op/sdirk.rs:186-195, :266-276)
  void reset_jacobian(double t) {
    if (fused_) {
      const bool recompute = op_.jacobian_is_stale();
      if (recompute) { op_.set_tmp(state_.y); pr_.eqn->rhs_statistics.number_of_matrix_evals++; pr_.eqn->rhs_statistics.number_of_jac_muls += n(); }
      check(dsh_jac_factor(ctx().raw(), model_, model_size_, nb(), t, op_.c() * op_.h(), op_.tmp().ptr(), pr_.eqn->params().ptr(), recompute ? 1 : 0,
                           op_.rhs_jac().ptr(), op_.mass_jac().ptr(), nonlinear_solver_.linear_solver().raw()), "dsh_jac_factor");
      op_.clear_jacobian_is_stale();
      nonlinear_solver_.mark_jacobian_set();
    } else {
      nonlinear_solver_.reset_jacobian(op_, state_.y, t);
    }
  }

  void jacobian_updates(double h, SolverState st) {  // sdirk.rs:260-303
    bool did_update = false;
    if (jacobian_update_.check_rhs_jacobian_update(h, st)) {
      op_.set_jacobian_is_stale();
      reset_jacobian(state_.t);
      jacobian_update_.update_rhs_jacobian(h);
      jacobian_update_.update_jacobian(h);
      convergence_.reset_eta();
      did_update = true;
    } else if (jacobian_update_.check_jacobian_update(h, st)) {
      reset_jacobian(state_.t);
      jacobian_update_.update_jacobian(h);
      convergence_.reset_eta();
      did_update = true;
    }
    if (did_update) record_linear_solver_setup(statistics_, st);
  }

  bool handle_tstop(double tstop) {  // runge_kutta.rs:752-781
    const double eps = std::numeric_limits<double>::epsilon();
    const double troundoff = 100.0 * eps * (std::fabs(state_.t) + std::fabs(state_.h));
    if (std::fabs(state_.t - tstop) <= troundoff) return true;
    if ((state_.h > 0.0 && tstop < state_.t - troundoff) || (state_.h < 0.0 && tstop > state_.t + troundoff)) throw DSH_ODE_ERR(StopTimeBeforeCurrentTime);
    if ((state_.h > 0.0 && state_.t + state_.h > tstop + troundoff) || (state_.h < 0.0 && state_.t + state_.h < tstop - troundoff)) {
      const double factor = (tstop - state_.t) / state_.h;
      state_.h *= factor;
    }
    return false;
  }

  void predict_stage_sdirk(int i, double h, const HipVec& dy0, HipVec& hdy) const { predict_stage_sdirk(i, h, dy0, diff_, hdy); }
  void predict_stage_sdirk(int i, double h, const HipVec& dy0, const HipMat& df, HipVec& hdy) const {  // runge_kutta.rs:610-629
    if (i == 0) hdy.axpy(h, dy0, 0.0);
    else if (i == 1) hdy.copy_from_view(df.column(0));
    else {
      const double c = (tab_.c[(size_t)i] - tab_.c[(size_t)i - 2]) / (tab_.c[(size_t)i - 1] - tab_.c[(size_t)i - 2]);
      if (df.nb() == hdy.nb()) { hdy.assign_axpby(-c, df.column(i - 2).p, 1.0 + c, df.column(i - 1).p); return; }  // the copy and the axpy in one pass
      hdy.copy_from_view(df.column(i - 1));
      hdy.axpy_v(-c, df.column(i - 2), 1.0 + c);
    }
  }

  NlError do_stage_sdirk(int i, double h) {  // runge_kutta.rs:631-689
    const double t = state_.t + tab_.c[(size_t)i] * h;
    if (prepared_stage_ != i) {  // (else the previous stage's single pass has stored phi and the predictor)
      op_.set_phi(diff_.columns(0, i), state_.y, a_rows_[(size_t)i]);
      predict_stage_sdirk(i, h, state_.dy, old_state_.dy);
    }
    prepared_stage_ = -1;
    if (!nonlinear_solver_.is_jacobian_set()) {
      reset_jacobian(t);
      record_linear_solver_setup(statistics_, SolverState::Checkpoint);
    }
    NlError r = (fused_ || staged_) ? newton_fused(t) : nonlinear_solver_.solve_in_place(op_, old_state_.dy, t, state_.y, convergence_, line_search_);
    statistics_.number_of_nonlinear_solver_iterations += convergence_.niter();
    if (r != NlError::Ok) return r;
    if (single_pass_stages() && i >= 1 && i + 1 < tab_.s) {
      // get_f_eval + the copy into diff[:,i] + set_phi and the predictor of stage i + 1 (runge_kutta.rs:672-676, 610-629; op/sdirk.rs:174-203) in one pass
      double a_next[8] = {0.0};
      for (int j = 0; j <= i; ++j) a_next[j] = tab_.A(i + 1, j);
      const double cp = (tab_.c[(size_t)i + 1] - tab_.c[(size_t)i - 1]) / (tab_.c[(size_t)i] - tab_.c[(size_t)i - 1]);
      check(dsh_sdirk_next_stage(ctx().raw(), n(), nb(), i, op_.c(), old_state_.dy.ptr(), op_.phi().ptr(), state_.y.ptr(), old_state_.y.ptr(), diff_.ptr(), a_next, -cp, 1.0 + cp),
            "dsh_sdirk_next_stage");
      prepared_stage_ = i + 1;
    } else if (single_pass_stages() && i >= 1 && i + 1 == tab_.s) {
      // the last stage: get_f_eval + the copy + the error estimate diff d (runge_kutta.rs:783-800) in one pass
      check(dsh_sdirk_finish_error(ctx().raw(), n(), nb(), tab_.s, op_.c(), old_state_.dy.ptr(), op_.phi().ptr(), old_state_.y.ptr(), diff_.ptr(), tab_.d.data(), error_.ptr()),
            "dsh_sdirk_finish_error");
      error_ready_ = true;
    } else
    if (old_state_.dy.nb() == nb() && old_state_.y.nb() == nb()) op_.get_f_eval_and_store(old_state_.dy, old_state_.y, diff_.column_mut(i).p);
    else {
      op_.get_f_eval(old_state_.dy, old_state_.y);
      diff_.column_mut(i).copy_from(old_state_.dy);
    }
    if (s_op_) {  // the sensitivity half of do_stage_sdirk (runge_kutta.rs:691-748)
      sens_update_state(old_state_.y, t);  // update_rhs_out_state(old_state.y, old_state.dy, t)
      for (size_t j = 0; j < sdiff_.size(); ++j) {
        s_op_->set_phi(sdiff_[j].columns(0, i), s_[j], a_rows_[(size_t)i]);
        sens_index_ = (int)j;
        predict_stage_sdirk(i, h, ds_[j], sdiff_[j], old_ds_[j]);
        NlError rs = nonlinear_solver_.solve_in_place(*s_op_, old_ds_[j], t, s_[j], convergence_, line_search_);
        statistics_.number_of_nonlinear_solver_iterations += convergence_.niter();  // counted before the `?` here
        if (rs != NlError::Ok) return rs;
        s_op_->get_f_eval(old_ds_[j], old_s_[j]);
        sdiff_[j].column_mut(i).copy_from(old_ds_[j]);
      }
    }
    return NlError::Ok;
  }
  void set_op_h(double h) { op_.set_h(h); if (s_op_) s_op_->set_h(h); }  // update_op_step_size (sdirk.rs:305-313)
  // SensRhs (ode_equations/sens_equations.rs:87-190): df/dp and the linearisation point; J(sens_y) x + (df/dp)[:, index]
  void sens_update_state(const HipVec& y, double t) {
    pr_.eqn->rhs_sens_inplace(y, t, sens_mat_);
    sens_y_.copy_from(y);
  }
  void sens_rhs_call(int index, const HipVec& x, double t, HipVec& y) const {
    pr_.eqn->rhs_jac_mul_inplace(sens_y_, t, x, y);
    y.add_assign(sens_mat_.column(index));
  }

  NlError newton_fused(double t) {
    convergence_.reset();
    for (int it = 0; it < convergence_.max_iter(); ++it) {
      double out[3] = {0.0, 0.0, 0.0};
      pr_.eqn->rhs_statistics.number_of_calls++;
      check(dsh_sdirk_newton_iter(ctx().raw(), model_, model_size_, nb(), t, op_.h(), op_.c(), old_state_.dy.ptr(), old_state_.dy.ptr(), op_.phi().ptr(), pr_.eqn->params().ptr(),
                                  nonlinear_solver_.linear_solver().raw(), state_.y.ptr(), pr_.atol.ptr(), pr_.atol.nb(), pr_.rtol, out), "dsh_sdirk_newton_iter");
      if (out[2] != 0.0) return NlError::LuSolveFailed;
      ConvergenceStatus st = convergence_.check_new_iteration(std::sqrt(out[0]));
      if (st == ConvergenceStatus::Converged) return NlError::Ok;
      if (st == ConvergenceStatus::Diverged) return NlError::NewtonDiverged;
    }
    return NlError::NewtonMaxIterations;
  }

  double factor(double error_norm, double safety_factor) const {  // runge_kutta.rs:466-495
    const double safety = 0.9 * safety_factor;
    const double raw = pi_controller_raw(error_norm, prev_error_norm_, pr_.ode_options.pi_control_integral, pr_.ode_options.pi_control_proportional, tab_.order() + 1);
    double f = safety * raw;
    if (f > maximum_timestep_shrink_ && f < minimum_timestep_growth_) f = 1.0;
    if (f < minimum_timestep_shrink_) f = minimum_timestep_shrink_;
    if (f > maximum_timestep_growth_) f = maximum_timestep_growth_;
    return f;
  }

  const OdeSolverProblem& pr_;
  Tableau tab_;
  NewtonNonlinearSolver nonlinear_solver_;
  NoLineSearch line_search_;
  Convergence convergence_;
  SdirkCallable op_;
  JacobianUpdate jacobian_update_;
  OdeSolverStatistics statistics_;
  std::vector<HipVec> a_rows_;
  HipVec d_vec_;
  HipMat diff_;
  HipVec error_, error_tmp_;
  StateCommon state_, old_state_;
  std::optional<double> tstop_;
  std::optional<RootFinder> root_finder_;
  std::optional<double> prev_error_norm_;
  bool is_state_mutated_ = false;
  double minimum_timestep_, maximum_timestep_growth_, minimum_timestep_growth_, maximum_timestep_shrink_, minimum_timestep_shrink_;
  int maximum_error_test_failures_, maximum_newton_fails_;
  bool fused_ = false, staged_ = false;
  int prepared_stage_ = -1;   // stage whose phi and predictor the previous single pass has already stored (-1: none)
  bool error_ready_ = false;  // error_ = diff d formed by the last stage's single pass
  int model_ = -1;
  int64_t model_size_ = 0;
  // forward sensitivities (problem.tr_bdf2_sens() / esdirk34_sens())
  std::vector<HipVec> s_, ds_, old_s_, old_ds_;
  std::vector<HipMat> sdiff_;
  std::unique_ptr<SdirkCallable> s_op_;
  HipMat sens_mat_;
  HipVec sens_y_, sens_error_;
  int sens_index_ = 0;
};

}  // namespace diffsol_hip
