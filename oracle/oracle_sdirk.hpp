// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of diffsol's (E)SDIRK integrators: Tableau::{tr_bdf2, esdirk34}, SdirkCallable, Rk core, Sdirk::step.
// Follows (relative to /root/reference/crates/diffsol/src):
//   ode_solver/tableau.rs:41-160          tableaus (column-major `a`)
//   op/sdirk.rs:18-300                    SdirkCallable (residual M k - h f(phi + c k), Jacobian M - c h J, set_phi, get_f_eval)
//   ode_solver/runge_kutta.rs:110-190     Rk::_new (a_rows, diff n x s)
//                            :466-495     factor;  :505-516 start_step_attempt;  :610-629 predict_stage_sdirk;
//                            :631-750     do_stage_sdirk;  :752-781 handle_tstop;  :783-800 error_norm;
//                            :841-892     error_test_fail / solve_fail;  :894-960 step_accepted;  :962-990, :1016-1035 interpolation
//   ode_solver/sdirk.rs:172-215 _new, :260-303 jacobian_updates, :409-543 step
#pragma once
#include <functional>
#include <memory>
#include "oracle_ode.hpp"

namespace orc {

struct Tableau {
  int s = 0, order = 0;
  M a;                     // s x s (nb = 1)
  std::vector<double> b, c, d;
  bool has_beta = false;
  M beta;                  // s x poly_order
  double A(int i, int j) const { return a.at(0, i, j); }
  static Tableau tr_bdf2() {  // tableau.rs:41-98
    Tableau t;
    t.s = 3; t.order = 2;
    double gamma = 2.0 - std::sqrt(2.0);
    double d = gamma / 2.0;
    double w = std::sqrt(2.0) / 4.0;
    t.a = M(3, 3, 1);
    t.a.d = {0.0, d, w, 0.0, d, w, 0.0, 0.0, d};
    t.b = {w, w, d};
    std::vector<double> b_hat = {(1.0 - w) / 3.0, (3.0 * w + 1.0) / 3.0, d / 3.0};
    t.d.resize(3);
    for (int i = 0; i < 3; ++i) t.d[i] = t.b[i] - b_hat[i];
    t.has_beta = true;
    t.beta = M(3, 2, 1);
    t.beta.d = {2.0 * w, 2.0 * w, gamma - 1.0, -w, -w, 2.0 * w};
    t.c = {0.0, gamma, 1.0};
    return t;
  }
  static Tableau esdirk34() {  // tableau.rs:101-159
    Tableau t;
    t.s = 4; t.order = 3;
    double gamma = 0.435866521508459;
    t.a = M(4, 4, 1);
    t.a.d = {0.0, gamma, 0.1407377747247062, 0.102399400619911,
             0.0, gamma, -0.1083655513813208, -0.3768784522555561,
             0.0, 0.0, gamma, 0.8386125301271861,
             0.0, 0.0, 0.0, gamma};
    t.b = {t.A(3, 0), t.A(3, 1), t.A(3, 2), t.A(3, 3)};
    t.c = {0.0, 0.871733043016918, 0.4682387448518444, 1.0};
    t.d = {-0.05462549724041394, -0.49420889362599496, 0.22193449973506466, 0.32689989113134427};
    return t;
  }
};

// y = alpha*A[:,0..k)*x + beta*y with nalgebra's gemv ordering (first column carries beta; zero columns => y *= beta)
inline void gemv_cols(const M& a, int k, double alpha, const double* x, double beta, V& y) {
  if (k == 0) {
    if (beta == 0.0) fill(y, 0.0); else mul_assign(y, beta);
    return;
  }
  for (int b = 0; b < y.nb; ++b)
    for (int i = 0; i < a.nr; ++i) {
      double acc = beta == 0.0 ? alpha * a.at(b, i, 0) * x[0] : alpha * a.at(b, i, 0) * x[0] + beta * y.at(b, i);
      for (int j = 1; j < k; ++j) acc = alpha * a.at(b, i, j) * x[j] + acc;
      y.at(b, i) = acc;
    }
}

// op/sdirk.rs
struct SdirkCallable {
  const Eqn* eqn;
  double c, h = 0.0;
  V phi, tmp;
  M rhs_jac, mass_jac;
  bool jacobian_is_stale = true;
  SdirkCallable(const Eqn* e, double c_) : eqn(e), c(c_), phi(e->n(), e->nb), tmp(e->n(), e->nb), rhs_jac(e->n(), e->n(), e->nb) {
    mass_jac = e->has_mass() ? M(e->n(), e->n(), e->nb) : M::identity(e->n(), e->nb);
  }
  void set_h(double h_) { h = h_; }
  void set_phi(const M& diff, int ncols, const V& y0, const std::vector<double>& a_row) {  // :174-184 with h = 1
    copy_from(phi, y0);
    gemv_cols(diff, ncols, 1.0, a_row.data(), 1.0, phi);
  }
  void set_tmp(const V& x) { copy_from(tmp, phi); axpy(tmp, c, x, 1.0); }           // :186-195
  void get_f_eval(const V& x, V& f_eval) const { copy_from(f_eval, phi); axpy(f_eval, c, x, 1.0); }  // :197-203
  void set_jacobian_is_stale() { jacobian_is_stale = true; }
  // SdirkCallable over the sensitivity equations (s_op): the right-hand side is SensRhs for the current parameter index
  std::function<void(const V&, double, V&)> rhs_override;
  void call_inplace(const V& x, double t, V& y) {  // :229-244
    set_tmp(x);
    if (rhs_override) rhs_override(tmp, t, y);
    else eqn->rhs(tmp, t, y);
    double beta = -h;
    if (eqn->has_mass()) eqn->mass_gemv(x, t, beta, y);
    else axpy(y, 1.0, x, beta);
  }
  void jacobian_inplace(const V& x, double t, M& y) {  // :266-296
    if (jacobian_is_stale) {
      set_tmp(x);
      eqn->jacobian(tmp, t, rhs_jac);
      if (eqn->has_mass()) eqn->mass_matrix(t, mass_jac);
      scale_add_and_assign(y, mass_jac, -(c * h), rhs_jac);
      jacobian_is_stale = false;
    } else {
      scale_add_and_assign(y, mass_jac, -(c * h), rhs_jac);
    }
  }
};

struct Sdirk : SolverBase {
  const Problem* pr;
  Tableau tab;
  NewtonSolver nonlinear_solver;
  NoLineSearch line_search;
  Convergence convergence;
  SdirkCallable op;
  JacobianUpdate jacobian_update;
  Stats statistics;
  // Rk
  std::vector<std::vector<double>> a_rows;
  M diff;  // n x s
  V error;
  StateCommon state, old_state;
  std::optional<double> tstop;
  std::optional<RootFinder> root_finder;
  std::optional<double> prev_error_norm;
  // forward sensitivities (problem.tr_bdf2_sens() / esdirk34_sens(); runge_kutta.rs:196-232 new_augmented, :691-748 the sensitivity half of do_stage_sdirk,
  // :812-822 sensitivities in the error norm, :1237-1300 interpolate_sens_inplace; SensRhs ode_equations/sens_equations.rs:87-190)
  std::vector<V> s_, ds_, old_s_, old_ds_;  // state.s / state.ds and their old_state partners
  std::vector<M> sdiff;
  std::unique_ptr<SdirkCallable> s_op;
  M sens_mat;
  V sens_y, sens_error;
  int sens_index = 0;
  int naug() const { return pr->sens ? pr->eqn->model->np : 0; }
  void sens_update_state(const V& y, double t) { pr->eqn->rhs_sens(y, t, sens_mat); copy_from(sens_y, y); }
  void sens_rhs_call(int index, const V& x, double t, V& y) const { pr->eqn->jac_mul(sens_y, t, x, y); add_assign(y, sens_mat.column(index)); }
  // config.rs:76-109
  double minimum_timestep, maximum_timestep_growth, minimum_timestep_growth, maximum_timestep_shrink, minimum_timestep_shrink;
  int maximum_error_test_failures, maximum_newton_fails;
  OdeErr init_error = OdeErr::Ok;

  Sdirk(const Problem* p, Tableau t)
      : pr(p), tab(std::move(t)), convergence(p->rtol, &p->atol, p->ode_options.nonlinear_solver_tolerance),
        op(p->eqn.get(), tab.A(1, 1)), jacobian_update(p->ode_options) {
    const OdeSolverOptions& o = p->ode_options;
    minimum_timestep = o.min_timestep;
    maximum_error_test_failures = o.max_error_test_failures;
    maximum_newton_fails = o.max_nonlinear_solver_failures;
    maximum_timestep_growth = o.max_timestep_growth.value_or(2.0);
    minimum_timestep_growth = o.min_timestep_growth.value_or(2.0);
    maximum_timestep_shrink = o.max_timestep_shrink.value_or(0.9);
    minimum_timestep_shrink = o.min_timestep_shrink.value_or(0.5);
    // problem.tr_bdf2()/esdirk34(): RkState::new_and_consistent(problem, tableau.order())  (problem.rs:850-861)
    init_error = new_and_consistent(*p, tab.order, state);
    if (init_error != OdeErr::Ok) return;
    int n = p->n(), nb = p->nb();
    for (int i = 0; i < tab.s; ++i) { std::vector<double> row; for (int j = 0; j < i; ++j) row.push_back(tab.A(i, j)); a_rows.push_back(row); }
    if (p->eqn->model->nroots > 0) { root_finder.emplace(p->eqn->model->nroots, n, nb); root_finder->init(*p->eqn, state.y, state.t); }
    diff = M(n, tab.s, nb);
    if (p->sens) {
      // RkState::new_with_sensitivities_and_consistent (state.rs:1032-1083): s_j = SensInit(t0), ds_j = SensRhs(s_j) about (y0, t0), DAEs through InitOp
      const int npar = naug();
      sens_mat = M(n, npar, nb);
      sens_y = V(n, nb);
      s_.assign((size_t)npar, V(n, nb)); ds_.assign((size_t)npar, V(n, nb));
      for (int j = 0; j < npar; ++j) p->eqn->init_sens(state.t, j, s_[(size_t)j]);
      sens_update_state(state.y, state.t);
      for (int j = 0; j < npar; ++j) sens_rhs_call(j, s_[(size_t)j], state.t, ds_[(size_t)j]);
      init_error = sens_set_consistent(*p, state.t, sens_y, s_, ds_, [this](int j, const V& x, double t, V& y) { sens_rhs_call(j, x, t, y); });
      if (init_error != OdeErr::Ok) return;
      old_s_ = s_; old_ds_ = ds_;
      sdiff.assign((size_t)npar, M(n, tab.s, nb));
      sens_error = V(n, nb);
      s_op = std::make_unique<SdirkCallable>(p->eqn.get(), tab.A(1, 1));
      s_op->rhs_override = [this](const V& x, double t, V& y) { sens_rhs_call(sens_index, x, t, y); };
    }
    old_state = state;
    error = V(n, nb);
    // Sdirk::_new
    jacobian_update.update_jacobian(state.h);
    jacobian_update.update_rhs_jacobian(state.h);
    convergence.max_iter = o.max_nonlinear_solver_iterations;
    op.set_h(state.h);
    if (s_op) s_op->set_h(state.h);
    nonlinear_solver.set_problem(n, nb);
    // Sdirk::new_augmented ends with jacobian_updates(h, Checkpoint) (sdirk.rs:251): the first linearisation of a solver WITH sensitivities is made here, at
    // t0 and — op.phi still being zero — about gamma * y0 (SdirkCallable::jacobian_inplace evaluates at phi + c x), not lazily inside the first stage
    if (s_op) jacobian_updates(state.h, SolverState::Checkpoint);
  }
  void set_op_h(double h) { op.set_h(h); if (s_op) s_op->set_h(h); }  // update_op_step_size (sdirk.rs)

  bool skip_first_stage() const { return tab.A(0, 0) == 0.0; }

  void jacobian_updates(double h, SolverState st) {  // sdirk.rs:260-303
    bool did_update = false;
    if (jacobian_update.check_rhs_jacobian_update(h, st)) {
      op.set_jacobian_is_stale();
      nonlinear_solver.reset_jacobian(op, state.y, state.t);
      jacobian_update.update_rhs_jacobian(h);
      jacobian_update.update_jacobian(h);
      convergence.reset_eta();
      did_update = true;
    } else if (jacobian_update.check_jacobian_update(h, st)) {
      nonlinear_solver.reset_jacobian(op, state.y, state.t);
      jacobian_update.update_jacobian(h);
      convergence.reset_eta();
      did_update = true;
    }
    if (did_update) record_linear_solver_setup(statistics, st);
  }

  OdeErr handle_tstop(double ts, std::optional<StopReason>& out) {  // runge_kutta.rs:752-781
    out.reset();
    const double eps = std::numeric_limits<double>::epsilon();
    double troundoff = 100.0 * eps * (std::fabs(state.t) + std::fabs(state.h));
    if (std::fabs(state.t - ts) <= troundoff) { out = StopReason::TstopReached; return OdeErr::Ok; }
    if ((state.h > 0.0 && ts < state.t - troundoff) || (state.h < 0.0 && ts > state.t + troundoff)) return OdeErr::StopTimeBeforeCurrentTime;
    if ((state.h > 0.0 && state.t + state.h > ts + troundoff) || (state.h < 0.0 && state.t + state.h < ts - troundoff)) {
      double factor = (ts - state.t) / state.h;
      state.h *= factor;
    }
    return OdeErr::Ok;
  }
  OdeErr set_stop_time(double ts) override {  // runge_kutta.rs:436-447
    tstop = ts;
    std::optional<StopReason> r;
    OdeErr e = handle_tstop(ts, r);
    if (e != OdeErr::Ok) { tstop.reset(); return e; }
    if (r && *r == StopReason::TstopReached) { tstop.reset(); return OdeErr::StopTimeAtCurrentTime; }
    return OdeErr::Ok;
  }

  void predict_stage_sdirk(int i, double h, const V& dy0, V& hdy) const { predict_stage_sdirk(i, h, dy0, diff, hdy); }
  void predict_stage_sdirk(int i, double h, const V& dy0, const M& df, V& hdy) const {  // runge_kutta.rs:610-629
    if (i == 0) axpy(hdy, h, dy0, 0.0);
    else if (i == 1) hdy = df.column(0);
    else {
      double c = (tab.c[i] - tab.c[i - 2]) / (tab.c[i - 1] - tab.c[i - 2]);
      hdy = df.column(i - 1);
      axpy(hdy, -c, df.column(i - 2), 1.0 + c);
    }
  }

  NlErr do_stage_sdirk(int i, double h) {  // runge_kutta.rs:631-689
    double t = state.t + tab.c[i] * h;
    op.set_phi(diff, i, state.y, a_rows[i]);
    predict_stage_sdirk(i, h, state.dy, old_state.dy);
    if (!nonlinear_solver.is_jacobian_set) {
      nonlinear_solver.reset_jacobian(op, state.y, t);
      record_linear_solver_setup(statistics, SolverState::Checkpoint);
    }
    NlErr r = nonlinear_solver.solve_in_place(op, old_state.dy, t, state.y, convergence, line_search);
    statistics.number_of_nonlinear_solver_iterations += convergence.niter;
    if (r != NlErr::Ok) return r;
    op.get_f_eval(old_state.dy, old_state.y);
    diff.set_column(i, old_state.dy);
    if (s_op) {  // :691-748
      sens_update_state(old_state.y, t);  // update_rhs_out_state(old_state.y, old_state.dy, t)
      for (int j = 0; j < naug(); ++j) {
        s_op->set_phi(sdiff[(size_t)j], i, s_[(size_t)j], a_rows[i]);
        sens_index = j;
        predict_stage_sdirk(i, h, ds_[(size_t)j], sdiff[(size_t)j], old_ds_[(size_t)j]);
        NlErr rs = nonlinear_solver.solve_in_place(*s_op, old_ds_[(size_t)j], t, s_[(size_t)j], convergence, line_search);
        statistics.number_of_nonlinear_solver_iterations += convergence.niter;  // here the count is added before the `?`
        if (rs != NlErr::Ok) return rs;
        s_op->get_f_eval(old_ds_[(size_t)j], old_s_[(size_t)j]);
        sdiff[(size_t)j].set_column(i, old_ds_[(size_t)j]);
      }
    }
    return NlErr::Ok;
  }

  double factor(double error_norm, double safety_factor) const {  // runge_kutta.rs:466-495
    double safety = 0.9 * safety_factor;
    double raw = pi_controller_raw(error_norm, prev_error_norm, pr->ode_options.pi_control_integral, pr->ode_options.pi_control_proportional, tab.order + 1);
    double f = safety * raw;
    if (f > maximum_timestep_shrink && f < minimum_timestep_growth) f = 1.0;
    if (f < minimum_timestep_shrink) f = minimum_timestep_shrink;
    if (f > maximum_timestep_growth) f = maximum_timestep_growth;
    return f;
  }

  bool is_state_mutated = false;
  OdeErr state_mut_back(double t) override {  // runge_kutta.rs:396-434
    if (pr->sens) return OdeErr::InterpolationTimeOutsideCurrentStep;  // not restated with sensitivities
    V ynew(state.y.n, state.y.nb), dynew(state.y.n, state.y.nb);
    OdeErr e = interpolate_inplace(t, ynew);
    if (e != OdeErr::Ok) return e;
    e = interpolate_dy_inplace(t, dynew);
    if (e != OdeErr::Ok) return e;
    copy_from(state.y, ynew);
    copy_from(state.dy, dynew);
    state.t = t;
    is_state_mutated = true;
    return OdeErr::Ok;
  }
  OdeErr apply_reset() override {  // sdirk.rs:368-374 over state.rs:279-306, through state_mut()
    if (!pr->eqn->model->has_reset) return OdeErr::InterpolationTimeOutsideCurrentStep;
    V y_out(state.y.n, state.y.nb);
    pr->eqn->reset(state.y, state.t, y_out);
    copy_from(state.y, y_out);
    is_state_mutated = true;
    if (pr->eqn->has_mass()) return set_consistent(state, *pr, true);  // apply_reset_with_mass (state.rs:297-300)
    pr->eqn->rhs(state.y, state.t, y_out);
    copy_from(state.dy, y_out);
    is_state_mutated = true;
    return OdeErr::Ok;
  }
  OdeErr step(StopReason& reason) override {  // sdirk.rs:409-543
    if (is_state_mutated) {  // Rk::start_step (runge_kutta.rs:444-464)
      if (root_finder) root_finder->init(*pr->eqn, state.y, state.t);
      if (tstop) { OdeErr e = set_stop_time(*tstop); if (e != OdeErr::Ok) return e; }
      is_state_mutated = false;
    }
    double h = state.h;  // rk.start_step()
    if (std::fabs(h) < minimum_timestep) return OdeErr::StepSizeTooSmall;
    set_op_h(h);
    int nattempts = 0;
    bool updated_jacobian = false;
    int start = skip_first_stage() ? 1 : 0;
    double fac = 1.0, error_norm = 0.0;
    while (true) {
      // start_step_attempt (runge_kutta.rs:505-516)
      if (skip_first_stage()) {
        V c0(state.dy.n, state.dy.nb); axpy(c0, h, state.dy, 0.0); diff.set_column(0, c0);
        for (int j = 0; j < naug(); ++j) { axpy(c0, h, ds_[(size_t)j], 0.0); sdiff[(size_t)j].set_column(0, c0); }  // "sensitivities too" (:518-523)
      }
      bool failed = false;
      for (int i = start; i < tab.s; ++i) {
        if (do_stage_sdirk(i, h) != NlErr::Ok) {
          if (!updated_jacobian) {
            updated_jacobian = true;
            jacobian_updates(h, SolverState::FirstConvergenceFail);
          } else {
            h *= 0.3;
            convergence.reset_eta_timestep_change();
            set_op_h(h);
            jacobian_updates(h, SolverState::SecondConvergenceFail);
          }
          prev_error_norm.reset();
          // solve_fail (runge_kutta.rs:868-892)
          statistics.number_of_nonlinear_solver_fails += 1;
          if (statistics.number_of_nonlinear_solver_fails > maximum_newton_fails) return OdeErr::TooManyNonlinearSolverFailures;
          if (std::fabs(h) < minimum_timestep) return OdeErr::StepSizeTooSmall;
          failed = true;
          break;
        }
      }
      if (failed) continue;
      // error_norm (runge_kutta.rs:783-800 + sdirk.rs:474-495)
      gemv_cols(diff, tab.s, 1.0, tab.d.data(), 0.0, error);
      if (pr->eqn->has_mass()) {
        V new_x = error;
        // mass.gemv(1, new_x, 0, x) with the dense current mass matrix
        for (int b = 0; b < error.nb; ++b)
          for (int i = 0; i < error.n; ++i) {
            double acc = 1.0 * op.mass_jac.at(b, i, 0) * new_x.at(b, 0);
            for (int j = 1; j < error.n; ++j) acc = 1.0 * op.mass_jac.at(b, i, j) * new_x.at(b, j) + acc;
            error.at(b, i) = acc;
          }
      }
      if (!nonlinear_solver.solve_linearised_in_place(error)) return OdeErr::TooManyNonlinearSolverFailures;  // `?` on LuSolveFailed
      error_norm = std::fmax(0.0, squared_norm(error, state.y, pr->atol, pr->rtol));
      if (pr->sens && pr->sens_error_control)  // :812-822 — no linear solve on the sensitivity error estimates
        for (int j = 0; j < naug(); ++j) {
          gemv_cols(sdiff[(size_t)j], tab.s, 1.0, tab.d.data(), 0.0, sens_error);
          error_norm = std::fmax(error_norm, squared_norm(sens_error, s_[(size_t)j], pr->sens_atol, pr->sens_rtol));
        }
      double maxiter = (double)convergence.max_iter, niter = (double)convergence.niter;
      double safety_factor = (2.0 * maxiter + 1.0) / (2.0 * maxiter + niter);
      fac = factor(error_norm, safety_factor);
      if (error_norm < 1.0) break;
      h *= fac;
      convergence.reset_eta_timestep_change();
      set_op_h(h);
      jacobian_updates(h, SolverState::ErrorTestFail);
      nattempts += 1;
      prev_error_norm.reset();
      // error_test_fail (runge_kutta.rs:841-866)
      statistics.number_of_error_test_failures += 1;
      if (nattempts >= maximum_error_test_failures) return OdeErr::TooManyErrorTestFailures;
      if (std::fabs(h) < minimum_timestep) return OdeErr::StepSizeTooSmall;
    }
    double new_h = h * fac;
    if (fac != 1.0) convergence.reset_eta_timestep_change();
    set_op_h(new_h);
    jacobian_updates(new_h, SolverState::StepSuccess);
    jacobian_update.step();
    prev_error_norm = error_norm;
    // step_accepted(h, new_h, true) (runge_kutta.rs:894-960)
    old_state.t = state.t + h;
    old_state.h = new_h;
    mul_assign(old_state.dy, 1.0 / h);
    for (V& d : old_ds_) mul_assign(d, 1.0 / h);
    std::swap(old_state, state);
    std::swap(old_s_, s_);
    std::swap(old_ds_, ds_);
    statistics.number_of_steps += 1;
    if (root_finder) {
      auto interp = [&](double tt, V& yy) { (void)interpolate_inplace(tt, yy); };
      auto ret = root_finder->check_root(interp, *pr->eqn, state.y, state.t);
      if (root_finder->mismatch) return OdeErr::RootBatchMismatch;
      if (ret) { root_time = ret->first; root_index = ret->second; reason = StopReason::RootFound; return OdeErr::Ok; }
    }
    if (tstop) {
      std::optional<StopReason> r;
      OdeErr e = handle_tstop(*tstop, r);
      if (e != OdeErr::Ok) return e;
      if (r && *r == StopReason::TstopReached) { tstop.reset(); reason = StopReason::TstopReached; return OdeErr::Ok; }
    }
    reason = StopReason::InternalTimestep;
    return OdeErr::Ok;
  }

  OdeErr interpolate_inplace(double t, V& ret) const override {  // runge_kutta.rs:1080-1127
    if (is_state_mutated) { if (t != state.t) return OdeErr::InterpolationTimeOutsideCurrentStep; copy_from(ret, state.y); return OdeErr::Ok; }
    bool is_forward = state.h > 0.0;
    if ((is_forward && (t > state.t || t < old_state.t)) || (!is_forward && (t < state.t || t > old_state.t)))
      return OdeErr::InterpolationTimeOutsideCurrentStep;
    double dt = state.t - old_state.t;
    double theta = dt == 0.0 ? 1.0 : (t - old_state.t) / dt;
    if (tab.has_beta) {
      int poly_order = tab.beta.nc, s_star = tab.beta.nr;
      std::vector<double> thetav{theta};
      for (int i = 1; i < poly_order; ++i) thetav.push_back(theta * thetav[i - 1]);
      V beta_f(s_star, 1);
      gemv_cols(tab.beta, poly_order, 1.0, thetav.data(), 0.0, beta_f);
      copy_from(ret, old_state.y);
      gemv_cols(diff, s_star, 1.0, beta_f.d.data(), 1.0, ret);
    } else {  // interpolate_hermite (runge_kutta.rs:1016-1035) with scale_diff = 1
      V f0 = diff.column(0), f1 = diff.column(diff.nc - 1);
      copy_from(ret, state.y);
      sub_assign(ret, old_state.y);
      axpy(ret, 1.0 * (theta - 1.0), f0, 1.0 - 2.0 * theta);
      axpy(ret, 1.0 * theta, f1, 1.0);
      axpy(ret, 1.0 - theta, old_state.y, theta * (theta - 1.0));
      axpy(ret, theta, state.y, 1.0);
    }
    return OdeErr::Ok;
  }
  // interpolate_sens_inplace (runge_kutta.rs:1237-1330): the state's interpolant applied to (old_state.s, state.s, sdiff)
  OdeErr interpolate_sens(double t, std::vector<V>& out) const {
    bool is_forward = state.h > 0.0;
    if ((is_forward && (t > state.t || t < old_state.t)) || (!is_forward && (t < state.t || t > old_state.t)))
      return OdeErr::InterpolationTimeOutsideCurrentStep;
    double dt = state.t - old_state.t;
    double theta = dt == 0.0 ? 1.0 : (t - old_state.t) / dt;
    out.assign(s_.size(), V(state.y.n, state.y.nb));
    if (tab.has_beta) {
      int poly_order = tab.beta.nc, s_star = tab.beta.nr;
      std::vector<double> thetav{theta};
      for (int i = 1; i < poly_order; ++i) thetav.push_back(theta * thetav[i - 1]);
      V beta_f(s_star, 1);
      gemv_cols(tab.beta, poly_order, 1.0, thetav.data(), 0.0, beta_f);
      for (size_t j = 0; j < s_.size(); ++j) {
        copy_from(out[j], old_s_[j]);
        gemv_cols(sdiff[j], s_star, 1.0, beta_f.d.data(), 1.0, out[j]);
      }
    } else {
      for (size_t j = 0; j < s_.size(); ++j) {
        V f0 = sdiff[j].column(0), f1 = sdiff[j].column(sdiff[j].nc - 1);
        V& ret = out[j];
        copy_from(ret, s_[j]);
        sub_assign(ret, old_s_[j]);
        axpy(ret, 1.0 * (theta - 1.0), f0, 1.0 - 2.0 * theta);
        axpy(ret, 1.0 * theta, f1, 1.0);
        axpy(ret, 1.0 - theta, old_s_[j], theta * (theta - 1.0));
        axpy(ret, theta, s_[j], 1.0);
      }
    }
    return OdeErr::Ok;
  }
  OdeErr interpolate_dy_inplace(double t, V& dy) const override {  // runge_kutta.rs:1129-1181
    if (is_state_mutated) { if (t != state.t) return OdeErr::InterpolationTimeOutsideCurrentStep; copy_from(dy, state.dy); return OdeErr::Ok; }
    bool is_forward = state.h > 0.0;
    if ((is_forward && (t > state.t || t < old_state.t)) || (!is_forward && (t < state.t || t > old_state.t)))
      return OdeErr::InterpolationTimeOutsideCurrentStep;
    double dt = state.t - old_state.t;
    if (dt == 0.0) { copy_from(dy, state.dy); return OdeErr::Ok; }
    double theta = (t - old_state.t) / dt;
    const double scale_diff = 1.0;
    if (tab.has_beta) {  // interpolate_beta_function_deriv (:985-1002): d_thetav = [1, 2 theta, 3 theta^2, ...]
      int poly_order = tab.beta.nc, s_star = tab.beta.nr;
      std::vector<double> d_thetav{1.0};
      double theta_pow = theta;
      for (int i = 1; i < poly_order; ++i) { d_thetav.push_back(((double)i + 1.0) * theta_pow); theta_pow *= theta; }
      V d_beta_f(s_star, 1);
      gemv_cols(tab.beta, poly_order, 1.0, d_thetav.data(), 0.0, d_beta_f);
      gemv_cols(diff, s_star, scale_diff / dt, d_beta_f.d.data(), 0.0, dy);
    } else {  // interpolate_hermite_deriv (:1037-1078)
      V f0 = diff.column(0), f1 = diff.column(diff.nc - 1);
      V q = state.y;
      sub_assign(q, old_state.y);
      axpy(q, scale_diff * (theta - 1.0), f0, 1.0 - 2.0 * theta);
      axpy(q, scale_diff * theta, f1, 1.0);
      copy_from(dy, state.y);
      sub_assign(dy, old_state.y);
      axpy(dy, (2.0 * theta - 1.0) / dt, q, 1.0 / dt);
      copy_from(q, old_state.y);
      sub_assign(q, state.y);
      axpy(q, scale_diff, f0, 2.0);
      axpy(q, scale_diff, f1, 1.0);
      axpy(dy, theta * (theta - 1.0) / dt, q, 1.0);
    }
    return OdeErr::Ok;
  }
  const V& y() const override { return state.y; }
  const V& dy() const override { return state.dy; }
  double t() const override { return state.t; }
  double h() const override { return state.h; }
  int order() const override { return tab.order; }
  const Stats& stats() const override { return statistics; }
  const Problem& problem() const override { return *pr; }
};

}  // namespace orc
