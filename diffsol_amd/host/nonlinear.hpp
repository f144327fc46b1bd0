// Host-side mirror of diffsol-nl (crates/diffsol-nl/src): Convergence, LineSearch {NoLineSearch, BacktrackingLineSearch},
// newton_iteration, NewtonNonlinearSolver — the scalar control logic that north_star keeps on the host.  Every vector operation goes
// through HipVec (one dsh_* call each); Bdf/Sdirk additionally use the fused device entry points for the NoLineSearch step.
//   convergence.rs:7-140   newton.rs:13-36, :88-180   line_search.rs:43-72, :84-201   nonlinear_op.rs:9-84
#pragma once
#include <cmath>
#include <functional>
#include <limits>

#include "../../include/diffsol_detpow.h"
#include "hip_la.hpp"

namespace diffsol_hip {

enum class ConvergenceStatus { Converged, Diverged, Continue };

// diffsol-nl/src/error.rs:22-42
enum class NlError { Ok = 0, NewtonDiverged, NewtonMaxIterations, LuSolveFailed, LinesearchFailedMinStep, LinesearchFailedMaxIterations, JacobianNotReset, WrongStateLength };

inline double powi(double a, int b) {  // f64::powi (compiler-rt __powidf2), used by convergence.rs:85
  const bool recip = b < 0;
  double r = 1.0;
  while (true) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return recip ? 1.0 / r : r;
}

// pow() of the step-size controller, the convergence-rate estimate and the initial-step heuristic: libm (the reference's arithmetic) by default;
// dshs_set_deterministic_pow(1) switches the host-driven integrators to include/diffsol_detpow.h — the pow the device-resident integrators use —
// so that both execution modes produce the same bits.  Constants (20^1.25, eps^(2/3)) are libm's in either mode, as in the device kernels' tables.
inline bool& det_pow_flag() { static bool f = false; return f; }
inline double hpow(double x, double y) { return det_pow_flag() ? dsh_det_pow(x, y) : std::pow(x, y); }

class Convergence {  // convergence.rs:7-140
 public:
  double rtol;
  const HipVec* atol;
  Convergence(double rtol_, const HipVec* atol_, double tol = 0.2) : rtol(rtol_), atol(atol_), tol_(tol), eta_(std::pow(20.0, 1.25)) {}
  int max_iter() const { return max_iter_; }
  void set_max_iter(int v) { max_iter_ = v; }
  int niter() const { return niter_; }
  double eta() const { return eta_; }
  void reset_eta() { eta_ = std::pow(20.0, 1.25); }
  void reset_eta_timestep_change() { eta_ = std::pow(100.0, 1.25); }
  void reset() { niter_ = 0; has_old_norm_ = false; }
  double norm(const HipVec& dy, const HipVec& y) const { return std::sqrt(dy.squared_norm(y, *atol, rtol)); }
  ConvergenceStatus check_norm(double norm) {
    niter_ += 1;
    if (has_old_norm_) {
      double rate = hpow(norm / old_norm_, 1.0 / (double)(niter_ - 1));
      if (rate > 0.9) return ConvergenceStatus::Diverged;
      if (powi(rate, max_iter_ - niter_) / (1.0 - rate) * norm > tol_) return ConvergenceStatus::Diverged;
      eta_ = rate / (1.0 - rate);
    } else {
      double min_eta = 1e4 * std::numeric_limits<double>::epsilon();
      if (eta_ < min_eta) eta_ = min_eta;
      eta_ = hpow(eta_, 0.8);
    }
    if (eta_ * norm < tol_) return ConvergenceStatus::Converged;
    return ConvergenceStatus::Continue;
  }
  ConvergenceStatus check_new_iteration(double norm) {
    ConvergenceStatus s = check_norm(norm);
    if (niter_ == 1) { has_old_norm_ = true; old_norm_ = norm; }
    return s;
  }

 private:
  double tol_;
  int max_iter_ = 10, niter_ = 0;
  bool has_old_norm_ = false;
  double old_norm_ = 0.0, eta_;
};

using FunT = std::function<void(const HipVec&, HipVec&)>;
using LinSolveT = std::function<bool(HipVec&)>;  // false = LuSolveFailed

struct LineSearch {  // line_search.rs:15-41
  virtual ~LineSearch() = default;
  virtual void reset() = 0;
  virtual NlError take_optimal_step(HipVec& x, HipVec& delta, const HipVec& error_y, const FunT& fun, const LinSolveT& solve, Convergence& conv,
                                    ConvergenceStatus& out) = 0;
};

struct NoLineSearch : LineSearch {  // line_search.rs:43-72
  void reset() override {}
  NlError take_optimal_step(HipVec& x, HipVec& delta, const HipVec& error_y, const FunT& fun, const LinSolveT& solve, Convergence& conv,
                            ConvergenceStatus& out) override {
    fun(x, delta);
    if (!solve(delta)) return NlError::LuSolveFailed;
    x.sub_assign(delta);
    double norm = conv.norm(delta, error_y);
    out = conv.check_new_iteration(norm);
    return NlError::Ok;
  }
};

struct BacktrackingLineSearch : LineSearch {  // line_search.rs:84-201
  double tau = 0.5, c = 1e-4;
  double steptol = std::pow(std::numeric_limits<double>::epsilon(), 2.0 / 3.0);
  int max_iter = 10, n_iters = 0;
  HipVec delta0, x0;
  double norm = 1.0;
  void reset() override { n_iters = 0; }
  NlError take_optimal_step(HipVec& x, HipVec& delta, const HipVec& error_y, const FunT& fun, const LinSolveT& solve, Convergence& conv,
                            ConvergenceStatus& out) override {
    if (conv.niter() == 0) {
      fun(x, delta);
      if (!solve(delta)) return NlError::LuSolveFailed;
      norm = conv.norm(delta, error_y);
      if (conv.check_norm(norm) == ConvergenceStatus::Converged) {
        x.sub_assign(delta);
        out = ConvergenceStatus::Converged;
        return NlError::Ok;
      }
    }
    if (x0.len() == 0) { x0 = HipVec::zeros(x.len(), x.context()); delta0 = HipVec::zeros(delta.len(), delta.context()); }
    x0.copy_from(x);
    delta0.copy_from(delta);
    const double half = 0.5;
    double nrm = norm;
    double phi0 = nrm * nrm * half, two_phi0 = nrm * nrm;
    double min_alpha = steptol / nrm;
    double alpha = 1.0;
    for (int i = 0; i < max_iter; ++i) {
      x.axpy(-alpha, delta0, 1.0);
      fun(x, delta);
      if (!solve(delta)) return NlError::LuSolveFailed;
      double new_norm = conv.norm(delta, error_y);
      n_iters = i;
      double phi1 = new_norm * new_norm * half;
      if (phi1 <= phi0 - c * alpha * two_phi0) {
        norm = new_norm;
        out = conv.check_norm(new_norm);
        return NlError::Ok;
      }
      if (alpha < min_alpha) return NlError::LinesearchFailedMinStep;
      alpha *= tau;
      x.copy_from(x0);
    }
    return NlError::LinesearchFailedMaxIterations;
  }
};

// newton.rs:13-36
inline NlError newton_iteration(HipVec& xn, HipVec& tmp, const HipVec& error_y, const FunT& fun, const LinSolveT& solve, Convergence& conv, LineSearch& ls) {
  conv.reset();
  ls.reset();
  for (int it = 0; it < conv.max_iter(); ++it) {
    ConvergenceStatus st = ConvergenceStatus::Continue;
    NlError e = ls.take_optimal_step(xn, tmp, error_y, fun, solve, conv, st);
    if (e != NlError::Ok) return e;
    if (st == ConvergenceStatus::Converged) return NlError::Ok;
    if (st == ConvergenceStatus::Diverged) return NlError::NewtonDiverged;
  }
  return NlError::NewtonMaxIterations;
}

// Time-frozen non-linear operator handed to the Newton solver (nonlinear_op.rs:9-84 + the NonLinearisedRef bridge,
// diffsol/src/nonlinear_solver/mod.rs:26-181)
struct NonLinearOpRef {
  virtual ~NonLinearOpRef() = default;
  virtual int64_t nstates() const = 0;
  virtual const HipContext& context() const = 0;
  virtual void call_inplace(const HipVec& x, double t, HipVec& y) = 0;
  virtual void jacobian_inplace(const HipVec& x, double t, HipMat& y) = 0;
  virtual bool packed_band(int* kl, int* ku) const { (void)kl; (void)ku; return false; }  // the operator's matrices are band containers (LinearOpRef::packed_band)
};

// NewtonNonlinearSolver<M, LS, Lsearch> (newton.rs:88-180) with LS = HipLU
class NewtonNonlinearSolver {
 public:
  void clear_jacobian() { is_jacobian_set_ = false; }
  bool is_jacobian_set() const { return is_jacobian_set_; }
  void set_problem(NonLinearOpRef& op) {
    struct Sp : LinearOpRef {
      NonLinearOpRef& op;
      explicit Sp(NonLinearOpRef& o) : op(o) {}
      int64_t nrows() const override { return op.nstates(); }
      int64_t ncols() const override { return op.nstates(); }
      const HipContext& context() const override { return op.context(); }
      void matrix_inplace(HipMat&) const override {}
      bool packed_band(int* kl, int* ku) const override { return op.packed_band(kl, ku); }
    } sp(op);
    linear_solver_.set_sparsity(sp);
    is_jacobian_set_ = false;
    tmp_ = HipVec::zeros(op.nstates(), op.context());
  }
  void reset_jacobian(NonLinearOpRef& op, const HipVec& x, double t) {
    struct At : LinearOpRef {  // JacobianRef::at (newton.rs:43-86)
      NonLinearOpRef& op; const HipVec& x; double t;
      At(NonLinearOpRef& o, const HipVec& x_, double t_) : op(o), x(x_), t(t_) {}
      int64_t nrows() const override { return op.nstates(); }
      int64_t ncols() const override { return op.nstates(); }
      const HipContext& context() const override { return op.context(); }
      void matrix_inplace(HipMat& y) const override { op.jacobian_inplace(x, t, y); }
    } at(op, x, t);
    linear_solver_.set_linearisation(at);
    is_jacobian_set_ = true;
  }
  // for fused Jacobian refreshes that wrote the factors directly into linear_solver().raw()
  void mark_jacobian_set() { linear_solver_.mark_factored(); is_jacobian_set_ = true; }
  bool solve_linearised_in_place(HipVec& x) const { return linear_solver_.solve_in_place(x); }
  NlError solve_in_place(NonLinearOpRef& op, HipVec& xn, double t, const HipVec& error_y, Convergence& conv, LineSearch& ls) {
    if (!is_jacobian_set_) return NlError::JacobianNotReset;
    if (xn.len() != op.nstates()) return NlError::WrongStateLength;
    FunT fun = [&](const HipVec& x, HipVec& y) { op.call_inplace(x, t, y); };
    LinSolveT solve = [&](HipVec& x) { return linear_solver_.solve_in_place(x); };
    return newton_iteration(xn, tmp_, error_y, fun, solve, conv, ls);
  }
  HipLU& linear_solver() { return linear_solver_; }
  const HipLU& linear_solver() const { return linear_solver_; }

 private:
  HipLU linear_solver_;
  bool is_jacobian_set_ = false;
  HipVec tmp_;
};

}  // namespace diffsol_hip
