// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of diffsol's implicit time-integration hot path (BDF part + shared pieces):
// OdeSolverOptions / problem, JacobianUpdate, BdfCallable, InitOp + set_consistent, set_step_size,
// RootFinder, Bdf.  Follows (relative to /root/reference/crates/diffsol/src):
//   ode_solver/problem.rs:15-152        options + defaults
//   ode_solver/jacobian_update.rs:12-79 refactor / re-evaluate policy
//   op/bdf.rs:15-300                    BdfCallable (residual, Jacobian, psi)
//   op/init.rs:14-135                   InitOp
//   ode_solver/state.rs:84-162          set_consistent;  :969-997 new_and_consistent;  :1086-1124 new_without_initialise;
//                                       :1209-1277 set_step_size
//   nonlinear_solver/root.rs:12-222     RootFinder
//   ode_solver/bdf.rs:244-368 _new, :433-463 _compute_r, :465-506 _jacobian_updates, :508-577 _update_step_size,
//                     :646-692 _update_diff/_predict, :694-731 handle_tstop, :767-782 interpolate, :812-932 error control,
//                     :1277-1589 step, :1591-1600 set_stop_time
//   ode_solver/runge_kutta.rs:1313-1336 pi_controller_raw
// Parity pin: the integer work counters of the reference's insta snapshots (bdf.rs:1740-1757, :2090-2107,
// :2179-2196, :2303-2320, :2353-2370, :2403-2420) and its known-answer tables are checked in tests/test_oracle_golden.py.
#pragma once
#include "oracle_models.hpp"
#include "oracle_nl.hpp"
#include <array>
#include <optional>

namespace orc {

// problem.rs:132-152
struct OdeSolverOptions {
  int max_nonlinear_solver_iterations = 10;
  int max_error_test_failures = 40;
  int max_nonlinear_solver_failures = 50;
  double nonlinear_solver_tolerance = 0.2;
  double min_timestep = 1e-13;
  std::optional<double> max_timestep_growth, min_timestep_growth, max_timestep_shrink, min_timestep_shrink;
  int update_jacobian_after_steps = 20;
  int update_rhs_jacobian_after_steps = 50;
  double threshold_to_update_jacobian = 0.3;
  double threshold_to_update_rhs_jacobian = 0.2;
  double pi_control_proportional = 0.0;
  double pi_control_integral = 0.5;
};
// problem.rs:15-45
struct InitialConditionSolverOptions {
  bool use_linesearch = true;
  int max_linesearch_iterations = 10;
  int max_linear_solver_setups = 4;
  int max_newton_iterations = 10;
  double step_reduction_factor = 0.5;
  double armijo_constant = 1e-4;
};

struct Problem {
  std::unique_ptr<Eqn> eqn;
  double rtol = 1e-6;
  V atol;  // n x 1 (broadcast over batches)
  double t0 = 0.0, h0 = 1.0;
  OdeSolverOptions ode_options;
  InitialConditionSolverOptions ic_options;
  // forward sensitivities (problem.bdf_sens(), problem.rs:819-832): integrate s_j = dy/dp_j alongside the states; with sens_rtol / sens_atol they
  // take part in the error control (SensEquations::include_in_error_control, sens_equations.rs:283-285), without
  // (turn_off_sensitivities_error_control, builder.rs:1501-1505) they do not
  bool sens = false;
  bool sens_error_control = false;
  double sens_rtol = 0.0;
  V sens_atol;  // n x 1, the same for every parameter (builder.rs build_atols)
  int n() const { return eqn->n(); }
  int nb() const { return eqn->nb; }
};

// ode_solver/mod.rs:28-69
struct Stats {
  long number_of_linear_solver_setups = 0, number_of_steps = 0, number_of_error_test_failures = 0,
       number_of_nonlinear_solver_iterations = 0, number_of_nonlinear_solver_fails = 0,
       setups_from_checkpoint = 0, setups_from_first_convergence_fail = 0, setups_from_second_convergence_fail = 0,
       setups_from_error_test_fail = 0, setups_from_step_success = 0;
};

enum class SolverState { StepSuccess, FirstConvergenceFail, SecondConvergenceFail, ErrorTestFail, Checkpoint };

inline void record_linear_solver_setup(Stats& s, SolverState st) {  // ode_solver/mod.rs:44-68
  s.number_of_linear_solver_setups++;
  switch (st) {
    case SolverState::Checkpoint: s.setups_from_checkpoint++; break;
    case SolverState::FirstConvergenceFail: s.setups_from_first_convergence_fail++; break;
    case SolverState::SecondConvergenceFail: s.setups_from_second_convergence_fail++; break;
    case SolverState::ErrorTestFail: s.setups_from_error_test_fail++; break;
    case SolverState::StepSuccess: s.setups_from_step_success++; break;
  }
}

// jacobian_update.rs:12-79
struct JacobianUpdate {
  int steps_since_jacobian_eval = 0, steps_since_rhs_jacobian_eval = 0;
  double h_at_last_jacobian_update = 1.0;
  double threshold_to_update_jacobian, threshold_to_update_rhs_jacobian;
  int update_jacobian_after_steps, update_rhs_jacobian_after_steps;
  explicit JacobianUpdate(const OdeSolverOptions& o)
      : threshold_to_update_jacobian(o.threshold_to_update_jacobian), threshold_to_update_rhs_jacobian(o.threshold_to_update_rhs_jacobian),
        update_jacobian_after_steps(o.update_jacobian_after_steps), update_rhs_jacobian_after_steps(o.update_rhs_jacobian_after_steps) {}
  void update_jacobian(double h) { steps_since_jacobian_eval = 0; h_at_last_jacobian_update = h; }
  void update_rhs_jacobian(double h) { steps_since_rhs_jacobian_eval = 0; steps_since_jacobian_eval = 0; h_at_last_jacobian_update = h; }
  void step() { steps_since_jacobian_eval++; steps_since_rhs_jacobian_eval++; }
  bool check_jacobian_update(double h, SolverState st) const {
    if (st == SolverState::StepSuccess)
      return steps_since_jacobian_eval >= update_jacobian_after_steps || std::fabs(h / h_at_last_jacobian_update - 1.0) > threshold_to_update_jacobian;
    return true;
  }
  bool check_rhs_jacobian_update(double h, SolverState st) const {
    switch (st) {
      case SolverState::StepSuccess: return steps_since_rhs_jacobian_eval >= update_rhs_jacobian_after_steps;
      case SolverState::FirstConvergenceFail: return std::fabs(h / h_at_last_jacobian_update - 1.0) < threshold_to_update_rhs_jacobian;
      case SolverState::SecondConvergenceFail: return steps_since_rhs_jacobian_eval > 0;
      case SolverState::ErrorTestFail: return false;
      case SolverState::Checkpoint: return true;
    }
    return false;
  }
};

// runge_kutta.rs:1313-1336
inline double pi_controller_raw(double error_norm, std::optional<double> prev_error_norm, double pi_integral, double pi_proportional, int eff_order) {
  double order_f = (double)eff_order;
  double ki = pi_integral / order_f;
  if (pi_proportional == 0.0) return rpow(error_norm, -ki);
  if (prev_error_norm) {
    double kp = pi_proportional / order_f;
    return rpow(error_norm, -(ki + kp)) * rpow(*prev_error_norm, kp);
  }
  return rpow(error_norm, -ki);
}

// Newton solver owning the LU of the current linearisation (diffsol-nl/src/newton.rs:88-180 + nalgebra/lu.rs)
struct NewtonSolver {
  DenseLU lu;
  M matrix;
  bool is_jacobian_set = false;
  V tmp;
  void set_problem(int n, int nb) { matrix = M(n, n, nb); tmp = V(n, nb); is_jacobian_set = false; }
  template <class Op> void reset_jacobian(Op& op, const V& x, double t) { op.jacobian_inplace(x, t, matrix); lu.factor(matrix); is_jacobian_set = true; }
  bool solve_linearised_in_place(V& x) const { return lu.solve(x); }
  template <class Op> NlErr solve_in_place(Op& op, V& xn, double t, const V& error_y, Convergence& conv, LineSearch& ls) {
    if (!is_jacobian_set) return NlErr::JacobianNotReset;
    FunT fun = [&](const V& x, V& y) { op.call_inplace(x, t, y); };
    LinSolveT solve = [&](V& x) { return lu.solve(x); };
    return newton_iteration(xn, tmp, error_y, fun, solve, conv, ls);
  }
};

// op/bdf.rs:15-300
struct BdfCallable {
  const Eqn* eqn;
  V psi_neg_y0, tmp;
  double c = 0.0;
  M rhs_jac, mass_jac;
  bool jacobian_is_stale = true;
  explicit BdfCallable(const Eqn* e) : eqn(e), psi_neg_y0(e->n(), e->nb), tmp(e->n(), e->nb), rhs_jac(e->n(), e->n(), e->nb) {
    mass_jac = e->has_mass() ? M(e->n(), e->n(), e->nb) : M::identity(e->n(), e->nb);  // :138-150
  }
  void set_c(double h, double alpha) { c = h * alpha; }  // :179-181
  void set_psi(const M& diff, const std::vector<double>& gamma, const std::vector<double>& alpha, int order, V& psi) const {  // :182-196
    axpy(psi, gamma[1], diff.column(1), 0.0);
    for (int i = 2; i <= order; ++i) axpy(psi, gamma[i], diff.column(i), 1.0);
    mul_assign(psi, alpha[order]);
  }
  void set_psi_and_y0(const M& diff, const std::vector<double>& gamma, const std::vector<double>& alpha, int order, const V& y0) {  // :197-210
    set_psi(diff, gamma, alpha, order, psi_neg_y0);
    sub_assign(psi_neg_y0, y0);
  }
  void set_jacobian_is_stale() { jacobian_is_stale = true; }
  // F(y) = M (y - y0 + psi) - c f(y)   :240-256
  void call_inplace(const V& x, double t, V& y) {
    eqn->rhs(x, t, y);
    copy_from(tmp, x);
    add_assign(tmp, psi_neg_y0);
    if (eqn->has_mass()) eqn->mass_gemv(tmp, t, -c, y);
    else axpy(y, 1.0, tmp, -c);
  }
  // M - c f'(y)   :273-300
  void jacobian_inplace(const V& x, double t, M& y) {
    if (jacobian_is_stale) {
      eqn->jacobian(x, t, rhs_jac);
      if (eqn->has_mass()) eqn->mass_matrix(t, mass_jac);
      scale_add_and_assign(y, mass_jac, -c, rhs_jac);
      jacobian_is_stale = false;
    } else {
      scale_add_and_assign(y, mass_jac, -c, rhs_jac);
    }
  }
};

// op/init.rs:14-135
struct InitOp {
  const Eqn* eqn;
  M jac, neg_mass;
  V y0;
  std::vector<int> algebraic_indices;
  // the right-hand side and its Jacobian when the equations are not `eqn` itself but equations built on it (InitOp::new(augmented_eqn, ..) in
  // set_consistent_augmented, state.rs:209-214: SensRhs, whose Jacobian is the state equations' at the linearisation point, sens_equations.rs:185-187)
  std::function<void(const V&, double, V&)> rhs_override;
  InitOp(const Eqn* e, double t0, const V& y0_, const std::vector<int>& alg, std::function<void(const V&, double, V&)> rhs_fn = nullptr,
         std::function<void(double, M&)> jac_fn = nullptr)
      : eqn(e), y0(y0_), algebraic_indices(alg), rhs_override(std::move(rhs_fn)) {
    int n = e->n(), nb = e->nb;
    M rhs_jac(n, n, nb), mass(n, n, nb);
    if (jac_fn) jac_fn(t0, rhs_jac);
    else e->jacobian(y0_, t0, rhs_jac);
    e->mass_matrix(t0, mass);
    std::vector<char> is_alg(n, 0);
    for (int i : alg) is_alg[i] = 1;
    jac = M(n, n, nb);
    neg_mass = M(n, n, nb);
    // jac = (-M_u, df/dv; 0, dg/dv) and neg_mass = (-M_u, 0; 0, 0) in the original index ordering (Matrix::split/combine)
    for (int b = 0; b < nb; ++b)
      for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
          if (!is_alg[j]) {
            double v = is_alg[i] ? 0.0 : mass.at(b, i, j) * (-1.0);
            jac.at(b, i, j) = v;
            neg_mass.at(b, i, j) = v;
          } else {
            jac.at(b, i, j) = rhs_jac.at(b, i, j);
            neg_mass.at(b, i, j) = 0.0;
          }
        }
  }
  static void copy_from_indices(V& dst, const V& src, const std::vector<int>& idx) {
    for (int b = 0; b < dst.nb; ++b) for (int i : idx) dst.at(b, i) = src.at(b, i);
  }
  // y = alpha*A*x + beta*y, nalgebra gemv order (blas.rs gemv: first column with beta, then += remaining columns)
  static void gemv(const M& a, double alpha, const V& x, double beta, V& y) {
    for (int b = 0; b < y.nb; ++b)
      for (int i = 0; i < a.nr; ++i) {
        double acc = alpha * a.at(b, i, 0) * x.at(b, 0) + beta * y.at(b, i);
        for (int j = 1; j < a.nc; ++j) acc = alpha * a.at(b, i, j) * x.at(b, j) + acc;
        y.at(b, i) = acc;
      }
  }
  void call_inplace(const V& x, double t, V& y) {  // :103-115
    copy_from_indices(y0, x, algebraic_indices);
    if (rhs_override) rhs_override(y0, t, y);
    else eqn->rhs(y0, t, y);
    gemv(neg_mass, 1.0, x, 1.0, y);
  }
  void jacobian_inplace(const V&, double, M& y) const { y = jac; }  // :125-127
  void scatter_soln(const V& soln, V& y, V& dy) const {  // :76-81
    V tmp = dy;
    copy_from(dy, soln);
    copy_from_indices(dy, tmp, algebraic_indices);
    copy_from_indices(y, soln, algebraic_indices);
  }
};

enum class OdeErr { Ok = 0, StepSizeTooSmall, TooManyErrorTestFailures, TooManyNonlinearSolverFailures, InitialConditionDidNotConverge,
                    StopTimeBeforeCurrentTime, StopTimeAtCurrentTime, InterpolationTimeAfterCurrentTime, InterpolationTimeOutsideCurrentStep, RootBatchMismatch };
enum class StopReason { InternalTimestep = 0, RootFound = 1, TstopReached = 2 };

struct StateCommon { V y, dy; double t = 0.0, h = 0.0; };

// state.rs:1086-1124
inline StateCommon new_without_initialise(const Problem& pr) {
  StateCommon s;
  s.t = pr.t0; s.h = pr.h0;
  s.y = V(pr.n(), pr.nb());
  s.dy = V(pr.n(), pr.nb());
  pr.eqn->init(s.t, s.y);
  pr.eqn->rhs(s.y, s.t, s.dy);
  return s;
}

// state.rs:84-162
// no_linesearch: the root solver apply_reset_with_mass builds (state.rs:299: NewtonNonlinearSolver::new(LS::default(), NoLineSearch)), whatever ic_options say
inline OdeErr set_consistent(StateCommon& s, const Problem& pr, bool no_linesearch = false) {
  const Eqn& eqn = *pr.eqn;
  if (!eqn.has_mass()) return OdeErr::Ok;
  int n = pr.n(), nb = pr.nb();
  M mass(n, n, nb);
  eqn.mass_matrix(pr.t0, mass);
  std::vector<int> alg;
  for (int i = 0; i < n; ++i) if (mass.at(0, i, i) == 0.0) alg.push_back(i);  // partition_indices_by_zero_diagonal
  if (alg.empty()) return OdeErr::Ok;
  InitOp f(&eqn, pr.t0, s.y, alg);
  NewtonSolver root_solver;
  root_solver.set_problem(n, nb);
  V y_tmp = s.dy;
  InitOp::copy_from_indices(y_tmp, s.y, alg);
  V yerr = y_tmp;
  Convergence conv(pr.rtol, &pr.atol, pr.ode_options.nonlinear_solver_tolerance);
  conv.max_iter = pr.ic_options.max_newton_iterations;
  std::unique_ptr<LineSearch> ls;
  if (pr.ic_options.use_linesearch && !no_linesearch) {
    auto b = std::make_unique<BacktrackingLineSearch>();
    b->c = pr.ic_options.armijo_constant; b->max_iter = pr.ic_options.max_linesearch_iterations; b->tau = pr.ic_options.step_reduction_factor;
    ls = std::move(b);
  } else ls = std::make_unique<NoLineSearch>();
  NlErr result = NlErr::Ok;
  for (int k = 0; k < pr.ic_options.max_linear_solver_setups; ++k) {
    root_solver.reset_jacobian(f, y_tmp, s.t);
    result = root_solver.solve_in_place(f, y_tmp, s.t, yerr, conv, *ls);
    if (result == NlErr::Ok) break;
    if (result != NlErr::NewtonMaxIterations) return OdeErr::InitialConditionDidNotConverge;  // `e => e.clone()?`
    copy_from(yerr, y_tmp);
  }
  if (result != NlErr::Ok) return OdeErr::InitialConditionDidNotConverge;
  f.scatter_soln(y_tmp, s.y, s.dy);
  for (int b = 0; b < nb; ++b) for (int i : alg) s.dy.at(b, i) = 0.0;
  return OdeErr::Ok;
}

// state.rs:1209-1277
inline void set_step_size(StateCommon& s, double h0_in, const V& atol, double rtol, const Eqn& eqn, int solver_order) {
  bool is_neg_h = h0_in < 0.0;
  const V& y0 = s.y; const V& f0 = s.dy; double t0 = s.t;
  double d0 = std::sqrt(squared_norm(y0, y0, atol, rtol));
  double d1 = std::sqrt(squared_norm(f0, y0, atol, rtol));
  double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
  V y1(y0.n, y0.nb), f1(y0.n, y0.nb);
  double hh = is_neg_h ? -h0 : h0;
  for (size_t i = 0; i < y1.d.size(); ++i) y1.d[i] = f0.d[i] * hh + y0.d[i];
  eqn.rhs(y1, is_neg_h ? t0 - h0 : t0 + h0, f1);
  V df(y0.n, y0.nb);
  for (size_t i = 0; i < df.d.size(); ++i) df.d[i] = f1.d[i] - f0.d[i];
  double d2 = std::sqrt(squared_norm(df, y0, atol, rtol)) / std::fabs(h0);
  double max_d = d2;
  if (max_d < d1) max_d = d1;
  double h1;
  if (max_d < 1e-15) { h1 = h0 * 1e-3; if (h1 < 1e-6) h1 = 1e-6; }
  else h1 = rpow(0.01 / max_d, 1.0 / (1.0 + (double)solver_order));
  s.h = 100.0 * h0;
  if (s.h > h1) s.h = h1;
  if (is_neg_h) s.h = -s.h;
}

// state.rs:969-997
inline OdeErr new_and_consistent(const Problem& pr, int solver_order, StateCommon& out) {
  out = new_without_initialise(pr);
  OdeErr e = set_consistent(out, pr);
  if (e != OdeErr::Ok) return e;
  set_step_size(out, pr.h0, pr.atol, pr.rtol, *pr.eqn, solver_order);
  return OdeErr::Ok;
}

// The DAE half of set_consistent_augmented (state.rs:187-238), shared by the BDF and SDIRK restatements: per parameter one Newton solve on InitOp over the
// sensitivity equations for (ds_j on the differential, s_j on the algebraic components); tolerances of the STATE equations, the consistent-initialisation
// options.  `sens_rhs(j, x, t, y)` = SensRhs::call_inplace for parameter j about the linearisation point sens_y.
template <class SensRhsFn>
inline OdeErr sens_set_consistent(const Problem& pr, double t0, const V& sens_y, std::vector<V>& s, std::vector<V>& ds, SensRhsFn&& sens_rhs) {
  const Eqn& eqn = *pr.eqn;
  if (!eqn.has_mass()) return OdeErr::Ok;
  const int n0 = pr.n(), nb0 = pr.nb();
  M mass(n0, n0, nb0);
  eqn.mass_matrix(pr.t0, mass);
  std::vector<int> alg;
  for (int i = 0; i < n0; ++i) if (mass.at(0, i, i) == 0.0) alg.push_back(i);
  if (alg.empty()) return OdeErr::Ok;
  Convergence conv(pr.rtol, &pr.atol, pr.ode_options.nonlinear_solver_tolerance);
  conv.max_iter = pr.ic_options.max_newton_iterations;
  std::unique_ptr<LineSearch> ls;
  if (pr.ic_options.use_linesearch) {
    auto b = std::make_unique<BacktrackingLineSearch>();
    b->c = pr.ic_options.armijo_constant; b->max_iter = pr.ic_options.max_linesearch_iterations; b->tau = pr.ic_options.step_reduction_factor;
    ls = std::move(b);
  } else ls = std::make_unique<NoLineSearch>();
  NewtonSolver root_solver;
  for (size_t j = 0; j < s.size(); ++j) {
    InitOp f(&eqn, t0, s[j], alg, [&sens_rhs, j](const V& x, double t, V& y) { sens_rhs((int)j, x, t, y); },
             [&eqn, &sens_y](double t, M& out) { eqn.jacobian(sens_y, t, out); });
    root_solver.set_problem(n0, nb0);
    V y_tmp = ds[j];
    InitOp::copy_from_indices(y_tmp, s[j], alg);
    V yerr = y_tmp;
    NlErr result = NlErr::Ok;
    for (int k = 0; k < pr.ic_options.max_linear_solver_setups; ++k) {
      root_solver.reset_jacobian(f, y_tmp, t0);
      result = root_solver.solve_in_place(f, y_tmp, t0, yerr, conv, *ls);
      if (result == NlErr::Ok) break;
      if (result != NlErr::NewtonMaxIterations) return OdeErr::InitialConditionDidNotConverge;
      copy_from(yerr, y_tmp);
    }
    if (result != NlErr::Ok) return OdeErr::InitialConditionDidNotConverge;
    f.scatter_soln(y_tmp, s[j], ds[j]);
  }
  return OdeErr::Ok;
}

// vector root_finding, batch semantics of crates/diffsol-la/src/vector/cuda.rs:1153-1177 (all batches must agree)
struct RootFindingResult { bool found; double frac; int idx; bool mismatch; };
inline RootFindingResult root_finding(const V& g0, const V& g1) {
  RootFindingResult first{false, 0.0, -1, false};
  for (int b = 0; b < g0.nb; ++b) {
    bool found = false; double mx = 0.0; int mi = -1;
    for (int i = 0; i < g0.n; ++i) {
      double a = g0.at(b, i), c = g1.at(b, i);
      if (c == 0.0) found = true;
      if (a * c < 0.0) { double frac = std::fabs(c / (c - a)); if (frac > mx) { mx = frac; mi = i; } }
    }
    if (b == 0) first = {found, mx, mi, false};
    else if (first.found != found || first.idx != mi) first.mismatch = true;
  }
  return first;
}

// nonlinear_solver/root.rs:12-222
struct RootFinder {
  double t0 = 0.0;
  V g0, g1, gmid, ymid;
  bool mismatch = false;
  RootFinder(int nroots, int nstates, int nb) : g0(nroots, nb), g1(nroots, nb), gmid(nroots, nb), ymid(nstates, nb) {}
  void init(const Eqn& eqn, const V& y, double t) { eqn.root(y, t, g0); t0 = t; }
  static int find_zero_index(const V& g) {
    int mi = 0; double mv = std::fabs(g.at(0, 0));
    for (int i = 1; i < g.n; ++i) { double v = std::fabs(g.at(0, i)); if (v < mv) { mv = v; mi = i; } }
    return mi;
  }
  template <class Interp>
  std::optional<std::pair<double, int>> check_root(const Interp& interpolate_inplace, const Eqn& eqn, const V& y, double t) {
    eqn.root(y, t, g1);
    auto r = root_finding(g0, g1);
    mismatch = mismatch || r.mismatch;
    if (r.idx < 0) {
      std::swap(g0, g1);
      t0 = t;
      if (r.found) return std::make_pair(t, find_zero_index(g0));
      return std::nullopt;
    }
    int imax = r.idx;
    double alpha = 1.0;
    bool sign_change[2] = {false, true};
    int i = 0;
    double t1 = t, t0l = t0;
    const double eps = std::numeric_limits<double>::epsilon();
    double tol = 100.0 * eps * (std::fabs(t1) + std::fabs(t1 - t0l));
    while (std::fabs(t1 - t0l) > tol) {
      double g1v = g1.at(0, imax), g0v = g0.at(0, imax);
      double t_mid = t1 - (t1 - t0l) * g1v / (g1v - alpha * g0v);
      if (std::fabs(t_mid - t0l) < 0.5 * tol) {
        double fracint = std::fabs(t1 - t0l) / tol;
        double fracsub = fracint > 5.0 ? 0.1 : 0.5 / fracint;
        t_mid = t0l + fracsub * (t1 - t0l);
      }
      if (std::fabs(t1 - t_mid) < 0.5 * tol) {
        double fracint = std::fabs(t1 - t0l) / tol;
        double fracsub = fracint > 5.0 ? 0.1 : 0.5 / fracint;
        t_mid = t1 - fracsub * (t1 - t0l);
      }
      interpolate_inplace(t_mid, ymid);
      eqn.root(ymid, t_mid, gmid);
      auto rr = root_finding(g0, gmid);
      mismatch = mismatch || rr.mismatch;
      bool lower = rr.idx >= 0;
      if (lower) { t1 = t_mid; imax = rr.idx; std::swap(g1, gmid); }
      else if (rr.found) { eqn.root(y, t, g0); return std::make_pair(t_mid, imax); }
      else { t0l = t_mid; std::swap(g0, gmid); }
      sign_change[i % 2] = lower;
      if (i >= 2) alpha = (sign_change[0] != sign_change[1]) ? 1.0 : (sign_change[0] ? 0.5 * alpha : 2.0 * alpha);
      i += 1;
    }
    eqn.root(y, t, g0);
    return std::make_pair(t1, imax);
  }
};

// Common solver interface used by the C ABI / tests
struct SolverBase {
  virtual ~SolverBase() = default;
  virtual OdeErr step(StopReason& reason) = 0;
  virtual OdeErr set_stop_time(double tstop) = 0;
  virtual OdeErr interpolate_inplace(double t, V& y) const = 0;
  virtual OdeErr interpolate_dy_inplace(double t, V& dy) const = 0;  // what state_mut_back stores in state.dy after a root stop
  virtual const V& y() const = 0;
  virtual const V& dy() const = 0;
  virtual double t() const = 0;
  virtual double h() const = 0;
  virtual int order() const = 0;
  virtual const Stats& stats() const = 0;
  virtual const Problem& problem() const = 0;
  // state_mut_back (bdf.rs:1232-1262, runge_kutta.rs:396-434): move the state to an interpolated time inside the last step; the next step restarts from it
  virtual OdeErr state_mut_back(double t) = 0;
  // Bdf / Sdirk::apply_reset (bdf.rs:1017-1020, sdirk.rs:368-374) over StateRefMut::apply_reset_with_mass (state.rs:279-306): y <- reset(y, t), then
  // dy <- f(y, t), or set_consistent when there is a mass matrix
  virtual OdeErr apply_reset() = 0;
  double root_time = 0.0;
  int root_index = -1;
};

// ode_solver/bdf.rs
struct Bdf : SolverBase {
  static constexpr int MAX_ORDER = 5;
  const Problem* pr;
  NewtonSolver nonlinear_solver;
  NoLineSearch line_search;
  Convergence convergence;
  BdfCallable op;
  int n_equal_steps = 0;
  V y_delta, y_predict;
  double t_predict = 0.0;
  M diff, diff_tmp, u;
  std::vector<double> alpha, gamma, error_const2;
  Stats statistics;
  // state
  int order_ = 1;
  V y_, dy_;
  double t_ = 0.0, h_ = 0.0;
  std::optional<double> tstop;
  std::optional<RootFinder> root_finder;
  JacobianUpdate jacobian_update;
  // config.rs:53-74
  double minimum_timestep, maximum_timestep_growth, minimum_timestep_growth, maximum_timestep_shrink, minimum_timestep_shrink;
  int maximum_error_test_failures, maximum_newton_fails;
  std::optional<double> prev_error_norm;
  OdeErr init_error = OdeErr::Ok;
  // forward sensitivities (bdf.rs:370-432 new_augmented, :934-989 sensitivity_solve; SensRhs / SensEquations ode_equations/sens_equations.rs:72-180)
  std::vector<V> s_, ds_, s_deltas;
  std::vector<M> sdiff;
  V s_predict, s_psi_neg_y0, s_tmp, sens_y;  // sens_y: the state SensRhs linearises about (update_rhs_out_state)
  M sens_mat;                                // df/dp at sens_y, n x np
  // BdfCallable::c of the sensitivity operator.  new_augmented builds it with BdfCallable::new_no_jacobian (bdf.rs:403) and never calls set_c on it
  // (c = 0, op/bdf.rs:61): until the first _update_step_size (:551-553) the sensitivity residual is F(s) = s - s0 + psi.  Restated as it is — the
  // reference's counters (14 setups / 56 steps / 1 error-test failure on exponential decay) only reproduce with it.
  double s_c = 0.0;
  int naug() const { return pr->sens ? pr->eqn->model->np : 0; }

  static M compute_r(int order, double factor) {  // :433-463
    int nrows = order + 1, ncols = order + 1;
    M r(nrows, ncols, 1);
    for (int j = 0; j < ncols; ++j) r.d[(size_t)j * nrows] = 1.0;
    for (int j = 1; j < ncols; ++j)
      for (int i = 1; i < nrows; ++i) {
        size_t idx = (size_t)j * nrows + i;
        r.d[idx] = r.d[idx - 1] * ((double)i - 1.0 - factor * (double)j) / (double)i;
      }
    return r;
  }

  explicit Bdf(const Problem* p)
      : pr(p), convergence(p->rtol, &p->atol, p->ode_options.nonlinear_solver_tolerance), op(p->eqn.get()), jacobian_update(p->ode_options) {
    const OdeSolverOptions& o = p->ode_options;
    minimum_timestep = o.min_timestep;
    maximum_error_test_failures = o.max_error_test_failures;
    maximum_newton_fails = o.max_nonlinear_solver_failures;
    maximum_timestep_growth = o.max_timestep_growth.value_or(2.0);
    minimum_timestep_growth = o.min_timestep_growth.value_or(2.0);
    maximum_timestep_shrink = o.max_timestep_shrink.value_or(0.9);
    minimum_timestep_shrink = o.min_timestep_shrink.value_or(0.5);
    // problem.bdf(): BdfState::new_and_consistent(problem, 1)  (problem.rs:597-602, :649-655)
    StateCommon sc;
    init_error = new_and_consistent(*p, 1, sc);
    y_ = sc.y; dy_ = sc.dy; t_ = sc.t; h_ = sc.h;
    if (init_error != OdeErr::Ok) return;
    if (p->sens) {
      // bdf_state_sens -> new_with_sensitivities_and_consistent (state.rs:1032-1083): initialise_augmented_state (:1157-1205: s_j = SensInit(t0), ds_j = 0),
      // [set_consistent — done above, before set_step_size in both orders the result is the same: it only touches y, dy],
      // set_consistent_augmented (:167-240: ds_j = SensRhs(s_j) about (y0, t0); with a singular mass matrix InitOp on the sensitivity equations, below)
      const int n0 = p->n(), nb0 = p->nb(), npar = p->eqn->model->np;
      sens_mat = M(n0, npar, nb0);
      sens_y = V(n0, nb0);
      s_.assign((size_t)npar, V(n0, nb0)); ds_.assign((size_t)npar, V(n0, nb0));
      for (int j = 0; j < npar; ++j) p->eqn->init_sens(t_, j, s_[(size_t)j]);
      sens_update_state(y_, t_);
      for (int j = 0; j < npar; ++j) sens_rhs_call(j, s_[(size_t)j], t_, ds_[(size_t)j]);
      init_error = sens_set_consistent(*p, t_, sens_y, s_, ds_, [this](int j, const V& x, double t, V& y) { sens_rhs_call(j, x, t, y); });
      if (init_error != OdeErr::Ok) return;
    }
    // _new :244-368
    const double kappa[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
    alpha = {0.0}; gamma = {0.0}; error_const2 = {1.0};
    for (int i = 1; i <= MAX_ORDER; ++i) {
      double i_t = (double)i;
      double one_over_i = 1.0 / i_t;
      double one_over_i_plus_one = 1.0 / (i_t + 1.0);
      gamma.push_back(gamma[i - 1] + one_over_i);
      alpha.push_back(1.0 / ((1.0 - kappa[i]) * gamma[i]));
      double e = kappa[i] * gamma[i] + one_over_i_plus_one;
      error_const2.push_back(e * e);
    }
    convergence.max_iter = o.max_nonlinear_solver_iterations;
    int n = p->n(), nb = p->nb();
    op.set_c(h_, alpha[order_]);
    nonlinear_solver.set_problem(n, nb);
    nonlinear_solver.reset_jacobian(op, y_, t_);
    diff = M(n, MAX_ORDER + 3, nb);
    initialise_diff_to_first_order();  // state.set_problem -> bdf_state.rs:72-78
    if (p->eqn->model->nroots > 0) { root_finder.emplace(p->eqn->model->nroots, n, nb); root_finder->init(*p->eqn, y_, t_); }
    diff_tmp = M(n, MAX_ORDER + 3, nb);
    y_delta = V(n, nb);
    y_predict = V(n, nb);
    u = compute_r(order_, 1.0);
    statistics.number_of_linear_solver_setups = 1;
    statistics.setups_from_checkpoint = 1;
    if (p->sens) {  // new_augmented (bdf.rs:384-432): set_augmented_problem -> initialise_sdiff_to_first_order (bdf_state.rs:80-91); s_op = BdfCallable::new_no_jacobian
      const int npar = naug();
      sdiff.assign((size_t)npar, M(n, MAX_ORDER + 3, nb));
      for (int j = 0; j < npar; ++j) {
        sdiff[(size_t)j].set_column(0, s_[(size_t)j]);
        V c1 = ds_[(size_t)j];
        mul_assign(c1, h_);
        sdiff[(size_t)j].set_column(1, c1);
      }
      s_deltas.assign((size_t)npar, V(n, nb));
      s_predict = V(n, nb);
      s_psi_neg_y0 = V(n, nb);
      s_tmp = V(n, nb);
    }
  }

  // SensRhs::update_state (sens_equations.rs:119-125): df/dp and the linearisation point
  void sens_update_state(const V& y, double t) {
    pr->eqn->rhs_sens(y, t, sens_mat);
    copy_from(sens_y, y);
  }
  // SensRhs::call_inplace (:155-161): J(sens_y) x + (df/dp)[:, index]
  void sens_rhs_call(int index, const V& x, double t, V& y) const {
    pr->eqn->jac_mul(sens_y, t, x, y);
    add_assign(y, sens_mat.column(index));
  }
  // BdfCallable::call_inplace of the sensitivity operator (op/bdf.rs:240-256): F(s) = M (s - s0 + psi) - c SensRhs(s)
  void s_op_call(int index, const V& x, double t, V& y) {
    sens_rhs_call(index, x, t, y);
    copy_from(s_tmp, x);
    add_assign(s_tmp, s_psi_neg_y0);
    if (pr->eqn->has_mass()) pr->eqn->mass_gemv(s_tmp, t, -s_c, y);  // SensEquations::mass is the state equations' mass (sens_equations.rs:305-307)
    else axpy(y, 1.0, s_tmp, -s_c);
  }
  // sensitivity_solve (bdf.rs:934-989): one Newton solve per parameter with the factors of the state equations
  bool sensitivity_solve(double t_new) {
    const int order = order_;
    sens_update_state(y_predict, t_new);  // `y_new = &self.y_predict` (:941)
    for (int j = 0; j < naug(); ++j) {
      predict_using_diff(s_predict, sdiff[(size_t)j], order);
      axpy(s_psi_neg_y0, gamma[1], sdiff[(size_t)j].column(1), 0.0);  // set_psi_and_y0 (op/bdf.rs:182-210)
      for (int i = 2; i <= order; ++i) axpy(s_psi_neg_y0, gamma[(size_t)i], sdiff[(size_t)j].column(i), 1.0);
      mul_assign(s_psi_neg_y0, alpha[(size_t)order]);
      sub_assign(s_psi_neg_y0, s_predict);
      V& s_new = s_[(size_t)j];
      copy_from(s_new, s_predict);
      FunT fun = [&](const V& x, V& y) { s_op_call(j, x, t_new, y); };
      LinSolveT solve = [&](V& x) { return nonlinear_solver.lu.solve(x); };
      NlErr e = newton_iteration(s_new, nonlinear_solver.tmp, s_predict, fun, solve, convergence, line_search);
      if (e != NlErr::Ok) return false;  // `?` before the iteration count is added
      statistics.number_of_nonlinear_solver_iterations += convergence.niter;
      copy_from(s_deltas[(size_t)j], s_new);
      sub_assign(s_deltas[(size_t)j], s_predict);
    }
    return true;
  }

  void initialise_diff_to_first_order() {  // bdf_state.rs:72-78
    order_ = 1;
    diff.set_column(0, y_);
    V c1 = dy_;
    mul_assign(c1, h_);
    diff.set_column(1, c1);
  }

  void jacobian_updates(double c, SolverState state) {  // :465-506
    bool did_update = false;
    if (jacobian_update.check_rhs_jacobian_update(c, state)) {
      op.set_jacobian_is_stale();
      nonlinear_solver.reset_jacobian(op, y_, t_);
      jacobian_update.update_rhs_jacobian(c);
      jacobian_update.update_jacobian(c);
      convergence.reset_eta();
      did_update = true;
    } else if (jacobian_update.check_jacobian_update(c, state)) {
      nonlinear_solver.reset_jacobian(op, y_, t_);
      jacobian_update.update_jacobian(c);
      convergence.reset_eta();
      did_update = true;
    }
    if (did_update) record_linear_solver_setup(statistics, state);
  }

  OdeErr update_step_size(double factor, double* new_h_out = nullptr) {  // :508-566
    event_counts().step_size_updates++;
    double new_h = factor * h_;
    n_equal_steps = 0;
    int order = order_;
    M r = compute_r(order, factor);
    M ru = mat_mul_small(r, u);
    // _update_diff_for_step_size :568-577 : diff_tmp[:,0..order+1] = diff[:,0..order+1]*RU ; swap(diff, diff_tmp)
    gemm_cols(diff_tmp, diff, order + 1, ru);
    std::swap(diff, diff_tmp);
    for (M& sd : sdiff) {  // :546-548: every sdiff through the SAME scratch matrix (the swap chain hands the previous matrix's columns on)
      gemm_cols(diff_tmp, sd, order + 1, ru);
      std::swap(sd, diff_tmp);
    }
    op.set_c(new_h, alpha[order]);
    if (pr->sens) s_c = new_h * alpha[order];
    h_ = new_h;
    convergence.reset_eta_timestep_change();
    if (new_h_out) *new_h_out = new_h;
    if (std::fabs(h_) < minimum_timestep) return OdeErr::StepSizeTooSmall;
    return OdeErr::Ok;
  }

  static void update_diff(int order, const V& d, M& diff) {  // :646-664
    V dm = d;
    sub_assign(dm, diff.column(order + 1));
    diff.set_column(order + 2, dm);
    diff.set_column(order + 1, d);
    for (int i = order; i >= 0; --i) diff.column_axpy(1.0, i + 1, i);
  }
  static void predict_using_diff(V& yp, const M& diff, int order) {  // :667-672
    fill(yp, 0.0);
    for (int i = 0; i <= order; ++i) add_assign(yp, diff.column(i));
  }
  void predict_forward() {  // :674-692
    predict_using_diff(y_predict, diff, order_);
    op.set_psi_and_y0(diff, gamma, alpha, order_, y_predict);
    t_predict = t_ + h_;
  }

  OdeErr handle_tstop(double ts, std::optional<StopReason>& out) {  // :694-731
    out.reset();
    const double eps = std::numeric_limits<double>::epsilon();
    double troundoff = 100.0 * eps * (std::fabs(t_) + std::fabs(h_));
    if (std::fabs(t_ - ts) <= troundoff) { tstop.reset(); out = StopReason::TstopReached; return OdeErr::Ok; }
    if ((h_ > 0.0 && ts < t_ - troundoff) || (h_ < 0.0 && ts > t_ + troundoff)) { tstop.reset(); return OdeErr::StopTimeBeforeCurrentTime; }
    if ((h_ > 0.0 && t_ + h_ > ts + troundoff) || (h_ < 0.0 && t_ + h_ < ts - troundoff)) {
      double factor = (ts - t_) / h_;
      (void)update_step_size(factor);  // ignoring "step size too small"
    }
    return OdeErr::Ok;
  }

  static void interpolate_from_diff(double t, const M& diff, double t1, double h, int order, V& y) {  // :767-782
    double time_factor = 1.0;
    y = diff.column(0);
    for (int i = 0; i < order; ++i) {
      double i_t = (double)i;
      time_factor *= (t - (t1 - h * i_t)) / (h * (1.0 + i_t));
      axpy(y, time_factor, diff.column(i + 1), 1.0);
    }
  }

  static void interpolate_derivative_from_diff(double t, const M& diff, double t1, double h, int order, V& dy) {  // :784-811
    double pi = 1.0, d_pi = 0.0;
    fill(dy, 0.0);
    for (int i = 0; i < order; ++i) {
      double i_t = (double)i;
      double denom = h * (1.0 + i_t);
      double w = (t - (t1 - h * i_t)) / denom;
      double dw = 1.0 / denom;
      double new_d_pi = d_pi * w + pi * dw;
      pi *= w;
      d_pi = new_d_pi;
      axpy(dy, d_pi, diff.column(i + 1), 1.0);
    }
  }

  double error_control() const {  // :812-843 (main equations only)
    double err = squared_norm(y_delta, y_, pr->atol, pr->rtol) * error_const2[order_ - 1];
    double error_norm = std::fmax(0.0, err);  // `error_norm.max(err)` starting from zero (f64::max drops NaN like fmax)
    if (pr->sens && pr->sens_error_control)  // :844-858 — note error_const2[order], not [order - 1]
      for (size_t j = 0; j < sdiff.size(); ++j)
        error_norm = std::fmax(error_norm, squared_norm(s_deltas[j], s_[j], pr->sens_atol, pr->sens_rtol) * error_const2[order_]);
    return error_norm;
  }
  double predict_error_control(int order) const {  // :871-932
    double error_norm = squared_norm(diff.column(order + 1), y_, pr->atol, pr->rtol) * error_const2[order];
    if (pr->sens) error_norm = std::fmax(0.0, error_norm);  // with an augmented system the reference's `error_norm.max(err)` chain is visible (NaN handling)
    if (pr->sens && pr->sens_error_control)
      for (size_t j = 0; j < sdiff.size(); ++j)
        error_norm = std::fmax(error_norm, squared_norm(sdiff[j].column(order + 1), s_[j], pr->sens_atol, pr->sens_rtol) * error_const2[order]);
    return error_norm;
  }

  bool is_state_modified = false;
  OdeErr state_mut_back(double t) override {  // :1232-1262
    if (pr->sens) return OdeErr::InterpolationTimeOutsideCurrentStep;  // not restated with sensitivities
    if (is_state_modified) return t == t_ ? OdeErr::Ok : OdeErr::InterpolationTimeOutsideCurrentStep;
    V ynew(pr->n(), pr->nb()), dynew(pr->n(), pr->nb());
    OdeErr e = interpolate_inplace(t, ynew);
    if (e != OdeErr::Ok) return e;
    e = interpolate_dy_inplace(t, dynew);
    if (e != OdeErr::Ok) return e;
    copy_from(y_, ynew);
    copy_from(dy_, dynew);
    t_ = t;
    is_state_modified = true;
    return OdeErr::Ok;
  }
  OdeErr apply_reset() override {
    if (!pr->eqn->model->has_reset) return OdeErr::InterpolationTimeOutsideCurrentStep;
    V y_out(pr->n(), pr->nb());
    pr->eqn->reset(y_, t_, y_out);
    copy_from(y_, y_out);
    is_state_modified = true;  // already set by state_mut_back (bdf.rs:1260)
    if (pr->eqn->has_mass()) {  // apply_reset_with_mass (state.rs:279-306, bdf.rs:1017-1020): consistent (y, dy) by a Newton solve on InitOp, no line search
      StateCommon sc; sc.y = std::move(y_); sc.dy = std::move(dy_); sc.t = t_; sc.h = h_;
      OdeErr e = set_consistent(sc, *pr, true);
      y_ = std::move(sc.y); dy_ = std::move(sc.dy);
      return e;
    }
    pr->eqn->rhs(y_, t_, y_out);
    copy_from(dy_, y_out);
    return OdeErr::Ok;
  }
  OdeErr step(StopReason& reason) override {  // :1277-1589
    double safety = 0.0, error_norm = 0.0;
    long old_num_error_test_failures = statistics.number_of_error_test_failures;
    bool convergence_fail = false;
    if (is_state_modified) {  // :1290-1318: restart from first order at the modified state
      if (root_finder) root_finder->init(*pr->eqn, y_, t_);
      n_equal_steps = 0;
      initialise_diff_to_first_order();
      u = compute_r(1, 1.0);
      is_state_modified = false;
      const double c = h_ * alpha[(size_t)order_];
      op.set_c(h_, alpha[(size_t)order_]);
      jacobian_updates(c, SolverState::StepSuccess);
      prev_error_norm.reset();
      if (tstop) { OdeErr e = set_stop_time(*tstop); if (e != OdeErr::Ok) return e; }
    }
    predict_forward();
    while (true) {
      int order = order_;
      copy_from(y_delta, y_predict);
      NlErr solve_result = nonlinear_solver.solve_in_place(op, y_delta, t_predict, y_predict, convergence, line_search);
      statistics.number_of_nonlinear_solver_iterations += convergence.niter;
      if (solve_result == NlErr::Ok) sub_assign(y_delta, y_predict);
      if (solve_result == NlErr::Ok && pr->sens && !sensitivity_solve(t_predict)) solve_result = NlErr::NewtonDiverged;  // SensitivitySolveFailed (:1355-1360)
      if (solve_result != NlErr::Ok) {
        statistics.number_of_nonlinear_solver_fails += 1;
        if (statistics.number_of_nonlinear_solver_fails > maximum_newton_fails) return OdeErr::TooManyNonlinearSolverFailures;
        if (convergence_fail) {
          prev_error_norm.reset();
          double new_h = 0.0;
          OdeErr e = update_step_size(0.3, &new_h);
          if (e != OdeErr::Ok) return e;
          jacobian_updates(new_h * alpha[order], SolverState::SecondConvergenceFail);
          predict_forward();
        } else {
          prev_error_norm.reset();
          jacobian_updates(h_ * alpha[order], SolverState::FirstConvergenceFail);
          convergence_fail = true;
        }
        continue;
      }
      error_norm = error_control();
      double maxiter = (double)convergence.max_iter;
      double niter = (double)convergence.niter;
      safety = 0.9 * (2.0 * maxiter + 1.0) / (2.0 * maxiter + niter);
      if (error_norm <= 1.0) break;
      double factor = safety * pi_controller_raw(error_norm, prev_error_norm, pr->ode_options.pi_control_integral, pr->ode_options.pi_control_proportional, order + 1);
      prev_error_norm.reset();
      if (factor < minimum_timestep_shrink) factor = minimum_timestep_shrink;
      double new_h = 0.0;
      OdeErr e = update_step_size(factor, &new_h);
      if (e != OdeErr::Ok) return e;
      jacobian_updates(new_h * alpha[order], SolverState::ErrorTestFail);
      predict_forward();
      statistics.number_of_error_test_failures += 1;
      if (statistics.number_of_error_test_failures - old_num_error_test_failures >= maximum_error_test_failures) return OdeErr::TooManyErrorTestFailures;
    }
    // take the accepted step
    update_diff(order_, y_delta, diff);
    for (size_t j = 0; j < sdiff.size(); ++j) update_diff(order_, s_deltas[j], sdiff[j]);  // update_differences_and_integrate_out (:628-643)
    copy_from(y_, y_predict);
    t_ = t_predict;
    dy_ = diff.column(1);
    mul_assign(dy_, 1.0 / h_);
    statistics.number_of_steps += 1;
    jacobian_update.step();
    prev_error_norm = error_norm;
    n_equal_steps += 1;
    if (n_equal_steps > order_) {
      event_counts().order_selections++;
      int order = order_;
      const double inf = std::numeric_limits<double>::infinity();
      double error_m_norm = order > 1 ? predict_error_control(order - 1) : inf;
      double error_p_norm = order < MAX_ORDER ? predict_error_control(order + 1) : inf;
      double pi_i = pr->ode_options.pi_control_integral, pi_p = pr->ode_options.pi_control_proportional;
      double factors[3] = {pi_controller_raw(error_m_norm, prev_error_norm, pi_i, pi_p, order),
                           pi_controller_raw(error_norm, prev_error_norm, pi_i, pi_p, order + 1),
                           pi_controller_raw(error_p_norm, prev_error_norm, pi_i, pi_p, order + 2)};
      // Iterator::max_by returns the LAST maximum on ties
      int max_index = 0;
      for (int k = 1; k < 3; ++k) if (factors[k] >= factors[max_index]) max_index = k;
      int new_order = max_index == 0 ? order - 1 : (max_index == 1 ? order : order + 1);
      order_ = new_order;
      if (max_index != 1) u = compute_r(new_order, 1.0);
      double factor = safety * factors[max_index];
      if (factor > maximum_timestep_growth) factor = maximum_timestep_growth;
      if (factor < minimum_timestep_shrink) factor = minimum_timestep_shrink;
      if (factor >= minimum_timestep_growth || factor <= maximum_timestep_shrink || max_index == 0 || max_index == 2) {
        double new_h = 0.0;
        OdeErr e = update_step_size(factor, &new_h);
        if (e != OdeErr::Ok) return e;
        jacobian_updates(new_h * alpha[new_order], SolverState::StepSuccess);
      }
    }
    if (root_finder) {
      auto interp = [&](double tt, V& yy) { (void)interpolate_inplace(tt, yy); };
      auto ret = root_finder->check_root(interp, *pr->eqn, y_, t_);
      if (root_finder->mismatch) return OdeErr::RootBatchMismatch;
      if (ret) { root_time = ret->first; root_index = ret->second; reason = StopReason::RootFound; return OdeErr::Ok; }
    }
    if (tstop) {
      std::optional<StopReason> r;
      (void)handle_tstop(*tstop, r);  // `.unwrap()` in the reference
      if (r) { reason = *r; return OdeErr::Ok; }
    }
    reason = StopReason::InternalTimestep;
    return OdeErr::Ok;
  }

  OdeErr set_stop_time(double ts) override {  // :1591-1600
    tstop = ts;
    std::optional<StopReason> r;
    OdeErr e = handle_tstop(ts, r);
    if (e != OdeErr::Ok) return e;
    if (r && *r == StopReason::TstopReached) { tstop.reset(); return OdeErr::StopTimeAtCurrentTime; }
    return OdeErr::Ok;
  }

  OdeErr interpolate_inplace(double t, V& y) const override {  // :1081-1108
    if (is_state_modified) { if (t != t_) return OdeErr::InterpolationTimeOutsideCurrentStep; copy_from(y, y_); return OdeErr::Ok; }
    bool is_forward = h_ > 0.0;
    if ((is_forward && t > t_) || (!is_forward && t < t_)) return OdeErr::InterpolationTimeAfterCurrentTime;
    interpolate_from_diff(t, diff, t_, h_, order_, y);
    return OdeErr::Ok;
  }
  OdeErr interpolate_dy_inplace(double t, V& dy) const override {  // :1108-1132
    if (is_state_modified) { if (t != t_) return OdeErr::InterpolationTimeOutsideCurrentStep; copy_from(dy, dy_); return OdeErr::Ok; }
    bool is_forward = h_ > 0.0;
    if ((is_forward && t > t_) || (!is_forward && t < t_)) return OdeErr::InterpolationTimeAfterCurrentTime;
    interpolate_derivative_from_diff(t, diff, t_, h_, order_, dy);
    return OdeErr::Ok;
  }
  // interpolate_sens_inplace (bdf.rs:1162-1215)
  OdeErr interpolate_sens(double t, std::vector<V>& out) const {
    bool is_forward = h_ > 0.0;
    if ((is_forward && t > t_) || (!is_forward && t < t_)) return OdeErr::InterpolationTimeAfterCurrentTime;
    out.assign(sdiff.size(), V(pr->n(), pr->nb()));
    for (size_t j = 0; j < sdiff.size(); ++j) interpolate_from_diff(t, sdiff[j], t_, h_, order_, out[j]);
    return OdeErr::Ok;
  }
  const V& y() const override { return y_; }
  const V& dy() const override { return dy_; }
  double t() const override { return t_; }
  double h() const override { return h_; }
  int order() const override { return order_; }
  const Stats& stats() const override { return statistics; }
  const Problem& problem() const override { return *pr; }
};

}  // namespace orc
