// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of diffsol-nl: Convergence, newton_iteration, NoLineSearch, BacktrackingLineSearch.
// Follows (relative to /root/reference/crates/diffsol-nl/src):
//   convergence.rs:7-140, newton.rs:13-36, line_search.rs:43-72 (NoLineSearch), :84-201 (Backtracking).
#pragma once
#include "oracle_la.hpp"
#include <functional>

namespace orc {

enum class ConvergenceStatus { Converged, Diverged, Continue };

enum class NlErr { Ok = 0, NewtonDiverged, NewtonMaxIterations, LuSolveFailed, LinesearchFailedMinStep, LinesearchFailedMaxIterations, JacobianNotReset };

// convergence.rs:7-140
struct Convergence {
  double rtol;
  const V* atol;
  double tol;
  int max_iter = 10;
  int niter = 0;
  bool has_old_norm = false;
  double old_norm = 0.0;
  double eta;
  Convergence(double rtol_, const V* atol_, double tol_ = 0.2) : rtol(rtol_), atol(atol_), tol(tol_), eta(std::pow(20.0, 1.25)) {}
  void reset_eta() { eta = std::pow(20.0, 1.25); }                    // :36-38
  void reset_eta_timestep_change() { eta = std::pow(100.0, 1.25); }   // :40-42
  void reset() { niter = 0; has_old_norm = false; }                   // :59-62
  double norm(const V& dy, const V& y) const { return std::sqrt(squared_norm(dy, y, *atol, rtol)); }  // :64-66
  ConvergenceStatus check_norm(double norm) {                         // :68-131
    niter += 1;
    if (has_old_norm) {
      if (niter > 2) event_counts().pow_rate++;
      double rate = rpow(norm / old_norm, 1.0 / (double)(niter - 1));
      if (rate > 0.9) return ConvergenceStatus::Diverged;
      event_counts().powi_calls++;
      if (powi(rate, max_iter - niter) / (1.0 - rate) * norm > tol) return ConvergenceStatus::Diverged;
      eta = rate / (1.0 - rate);
    } else {
      double min_eta = 1e4 * std::numeric_limits<double>::epsilon();
      if (eta < min_eta) eta = min_eta;
      event_counts().pow_first_iter++;
      if (eta == std::pow(20.0, 1.25)) event_counts().pow_first_iter_eta_reset++;
      if (eta == std::pow(100.0, 1.25)) event_counts().pow_first_iter_eta_reset_ts++;
      eta = rpow(eta, 0.8);
    }
    if (eta * norm < tol) return ConvergenceStatus::Converged;
    return ConvergenceStatus::Continue;
  }
  ConvergenceStatus check_new_iteration(double norm) {                // :133-139
    ConvergenceStatus s = check_norm(norm);
    if (niter == 1) { has_old_norm = true; old_norm = norm; }
    return s;
  }
};

using FunT = std::function<void(const V&, V&)>;      // F(x) -> y
using LinSolveT = std::function<bool(V&)>;           // in-place solve, false = LuSolveFailed

struct LineSearch {
  virtual ~LineSearch() = default;
  virtual void reset() = 0;
  virtual NlErr take_optimal_step(V& x, V& delta, const V& error_y, const FunT& fun, const LinSolveT& solve, Convergence& conv, ConvergenceStatus& out) = 0;
};

// line_search.rs:43-72
struct NoLineSearch : LineSearch {
  void reset() override {}
  NlErr take_optimal_step(V& x, V& delta, const V& error_y, const FunT& fun, const LinSolveT& solve, Convergence& conv, ConvergenceStatus& out) override {
    fun(x, delta);
    if (!solve(delta)) return NlErr::LuSolveFailed;
    sub_assign(x, delta);
    double norm = conv.norm(delta, error_y);
    out = conv.check_new_iteration(norm);
    return NlErr::Ok;
  }
};

// line_search.rs:84-201 (used only by DAE initialisation, state.rs:978-988)
struct BacktrackingLineSearch : LineSearch {
  double tau = 0.5, c = 1e-4;
  double steptol = std::pow(std::numeric_limits<double>::epsilon(), 2.0 / 3.0);
  int max_iter = 10, n_iters = 0;
  V delta0, x0;
  double norm = 1.0;
  void reset() override { n_iters = 0; }
  NlErr take_optimal_step(V& x, V& delta, const V& error_y, const FunT& fun, const LinSolveT& solve, Convergence& conv, ConvergenceStatus& out) override {
    if (conv.niter == 0) {
      fun(x, delta);
      if (!solve(delta)) return NlErr::LuSolveFailed;
      norm = conv.norm(delta, error_y);
      if (conv.check_norm(norm) == ConvergenceStatus::Converged) {
        sub_assign(x, delta);
        out = ConvergenceStatus::Converged;
        return NlErr::Ok;
      }
    }
    if (x0.size() == 0) { x0 = V(x.n, x.nb); delta0 = V(delta.n, delta.nb); }
    copy_from(x0, x);
    copy_from(delta0, delta);
    const double half = 0.5;
    double nrm = norm;
    double phi0 = nrm * nrm * half;
    double two_phi0 = nrm * nrm;
    double min_alpha = steptol / nrm;
    double alpha = 1.0;
    for (int i = 0; i < max_iter; ++i) {
      axpy(x, -alpha, delta0, 1.0);
      fun(x, delta);
      if (!solve(delta)) return NlErr::LuSolveFailed;
      double new_norm = conv.norm(delta, error_y);
      n_iters = i;
      double phi1 = new_norm * new_norm * half;
      if (phi1 <= phi0 - c * alpha * two_phi0) {
        norm = new_norm;
        out = conv.check_norm(new_norm);
        return NlErr::Ok;
      }
      if (alpha < min_alpha) return NlErr::LinesearchFailedMinStep;
      alpha *= tau;
      copy_from(x, x0);
    }
    return NlErr::LinesearchFailedMaxIterations;
  }
};

// newton.rs:13-36
inline NlErr newton_iteration(V& xn, V& tmp, const V& error_y, const FunT& fun, const LinSolveT& solve, Convergence& conv, LineSearch& ls) {
  conv.reset();
  ls.reset();
  for (int it = 0; it < conv.max_iter; ++it) {
    ConvergenceStatus st = ConvergenceStatus::Continue;
    NlErr e = ls.take_optimal_step(xn, tmp, error_y, fun, solve, conv, st);
    if (e != NlErr::Ok) return e;
    if (st == ConvergenceStatus::Converged) return NlErr::Ok;
    if (st == ConvergenceStatus::Diverged) return NlErr::NewtonDiverged;
  }
  return NlErr::NewtonMaxIterations;
}

}  // namespace orc
