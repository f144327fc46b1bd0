// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Speed-oriented build of the oracle's BDF for ONE independent IVP (nbatch = 1) of an ODE model with identity mass and no root function:
// exactly the arithmetic of oracle_ode.hpp (Bdf), oracle_nl.hpp (Convergence, NoLineSearch, newton_iteration) and oracle_la.hpp (DenseLU,
// squared_norm, gemm order) — operation for operation, same order — but on fixed-size stack arrays instead of heap vectors that are allocated
// per operation.  It exists so that bench.py's `cpu_baseline` times the reference ALGORITHM on the host cores rather than the allocation
// pattern of the line-by-line restatement (the fidelity build needs ~2.5 ms per n = 3 solve, the reference publishes
// 3.1e-5 s).  tests/test_oracle_golden.py checks it bit for bit (final state and every counter) against the fidelity build, which is the
// one pinned on the reference's snapshots.
//
// Follows (relative to /root/reference/crates): diffsol/src/ode_solver/bdf.rs:244-368 (_new), :433-463 (_compute_r), :465-506
// (_jacobian_updates), :508-577 (_update_step_size), :646-692 (_update_diff / _predict), :694-731 (handle_tstop), :812-932 (error control),
// :1277-1589 (step); diffsol/src/op/bdf.rs:182-300 (BdfCallable); diffsol/src/ode_solver/state.rs:1209-1277 (set_step_size);
// diffsol-nl/src/{convergence.rs:68-139,newton.rs:13-36,line_search.rs:46-72}; nalgebra 0.35 LU as restated in oracle_la.hpp.
#pragma once
#include "oracle_ode.hpp"

namespace orc {

template <int N>
struct FastBdf {
  static constexpr int MAX_ORDER = 5, NC = MAX_ORDER + 3;
  const Model& mdl;
  const double* p;
  double rtol;
  double atol[N];
  const OdeSolverOptions& o;
  // Bdf::_new tables
  double alpha[6], gamma[6], ec2[6];
  double u[36];  // compute_r(order, 1.0), (order+1)^2 column-major with leading dimension order+1
  int u_dim = 0;
  // state
  double y[N], dy[N], t = 0.0, h = 0.0;
  int order = 1;
  double D[NC][N], Dt[NC][N];
  double psi[N], yp[N], t_predict = 0.0;
  // BdfCallable
  double c = 0.0, J[N * N];
  bool jac_stale = true;
  // LU of M - cJ (column-major), nalgebra pivots
  double A[N * N];
  int P[N];
  // Convergence
  double eta, conv_tol;
  int max_iter, niter = 0;
  bool has_old = false;
  double old_norm = 0.0;
  JacobianUpdate ju;
  int n_equal_steps = 0;
  bool has_prev_err = false;
  double prev_err = 0.0;
  bool has_tstop = false;
  double tstop = 0.0;
  Stats st;
  double min_h, max_growth, min_growth, max_shrink, min_shrink;

  double sqnorm(const double* x, const double* w) const {  // squared_norm, nb = 1 (oracle_la.hpp)
    double acc = 0.0;
    for (int i = 0; i < N; ++i) {
      const double term = x[i] / (std::fabs(w[i]) * rtol + atol[i]);
      acc += term * term;
    }
    const double nrm = acc / (double)N;
    double mx = 0.0;
    if (nrm > mx || nrm != nrm) mx = nrm;
    return mx;
  }
  void compute_u(int ord) {  // compute_r(ord, 1.0)
    u_dim = ord + 1;
    for (int j = 0; j < u_dim; ++j) u[j * u_dim] = 1.0;
    for (int j = 0; j < u_dim; ++j)
      for (int i = 1; i < u_dim; ++i) u[j * u_dim + i] = j == 0 ? 0.0 : u[j * u_dim + i - 1] * ((double)i - 1.0 - 1.0 * (double)j) / (double)i;
  }
  void lu_factor() {  // DenseLU::factor
    for (int i = 0; i < N; ++i) {
      int pv = i;
      double best = std::fabs(A[i * N + i]);
      for (int r = i + 1; r < N; ++r) { const double v = std::fabs(A[i * N + r]); if (v > best) { best = v; pv = r; } }
      P[i] = pv;
      const double diag = A[i * N + pv];
      if (diag == 0.0) { P[i] = i; continue; }
      if (pv != i) for (int cc = 0; cc < N; ++cc) std::swap(A[cc * N + i], A[cc * N + pv]);
      const double inv_diag = 1.0 / diag;
      for (int r = i + 1; r < N; ++r) A[i * N + r] = A[i * N + r] * inv_diag;
      for (int cc = i + 1; cc < N; ++cc) {
        const double pr = A[cc * N + i];
        for (int r = i + 1; r < N; ++r) A[cc * N + r] = (-pr) * A[i * N + r] + A[cc * N + r];
      }
    }
  }
  bool lu_solve(double* v) const {  // DenseLU::solve
    for (int i = 0; i < N; ++i) if (P[i] != i) std::swap(v[i], v[P[i]]);
    for (int i = 0; i + 1 < N; ++i) {
      const double coeff = v[i];
      for (int r = i + 1; r < N; ++r) v[r] = (-coeff) * A[i * N + r] + v[r];
    }
    for (int i = N - 1; i >= 0; --i) {
      const double diag = A[i * N + i];
      if (diag == 0.0) return false;
      const double coeff = v[i] / diag;
      v[i] = coeff;
      for (int r = 0; r < i; ++r) v[r] = (-coeff) * A[i * N + r] + v[r];
    }
    return true;
  }
  void reset_jacobian() {  // NewtonSolver::reset_jacobian over BdfCallable::jacobian_inplace: M - c f'(y), identity mass
    if (jac_stale) {
      double v[N], col[N];
      for (int i = 0; i < N; ++i) v[i] = 0.0;
      for (int j = 0; j < N; ++j) {  // Eqn::jacobian: column j = jac_mul(unit vector j)
        v[j] = 1.0;
        mdl.jac_mul(y, p, t, v, col);
        for (int i = 0; i < N; ++i) J[j * N + i] = col[i];
        v[j] = 0.0;
      }
      jac_stale = false;
    }
    for (int e = 0; e < N * N; ++e) A[e] = J[e] * (-c) + ((e / N == e % N) ? 1.0 : 0.0);  // scale_add_and_assign(y, mass_jac, -c, rhs_jac)
    lu_factor();
  }

  FastBdf(const Model& m, const double* p_, double rtol_, const double* atol_, int natol, double t0, double h0, const OdeSolverOptions& o_)
      : mdl(m), p(p_), rtol(rtol_), o(o_), ju(o_) {
    for (int i = 0; i < N; ++i) atol[i] = atol_[natol == 1 ? 0 : i];
    min_h = o.min_timestep;
    max_growth = o.max_timestep_growth.value_or(2.0); min_growth = o.min_timestep_growth.value_or(2.0);
    max_shrink = o.max_timestep_shrink.value_or(0.9); min_shrink = o.min_timestep_shrink.value_or(0.5);
    conv_tol = o.nonlinear_solver_tolerance;
    max_iter = o.max_nonlinear_solver_iterations;
    eta = std::pow(20.0, 1.25);
    // new_without_initialise + set_step_size (state.rs:1086-1124, :1209-1277), solver order 1
    t = t0;
    mdl.init(p, t, y);
    mdl.rhs(y, p, t, dy);
    {
      const bool is_neg_h = h0 < 0.0;
      const double d0 = std::sqrt(sqnorm(y, y)), d1 = std::sqrt(sqnorm(dy, y));
      const double hh0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
      const double hh = is_neg_h ? -hh0 : hh0;
      double y1[N], f1[N], df[N];
      for (int i = 0; i < N; ++i) y1[i] = dy[i] * hh + y[i];
      mdl.rhs(y1, p, is_neg_h ? t - hh0 : t + hh0, f1);
      for (int i = 0; i < N; ++i) df[i] = f1[i] - dy[i];
      const double d2 = std::sqrt(sqnorm(df, y)) / std::fabs(hh0);
      double max_d = d2;
      if (max_d < d1) max_d = d1;
      double h1;
      if (max_d < 1e-15) { h1 = hh0 * 1e-3; if (h1 < 1e-6) h1 = 1e-6; }
      else h1 = rpow(0.01 / max_d, 1.0 / (1.0 + 1.0));
      h = 100.0 * hh0;
      if (h > h1) h = h1;
      if (is_neg_h) h = -h;
    }
    const double kappa[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
    alpha[0] = 0.0; gamma[0] = 0.0; ec2[0] = 1.0;
    for (int i = 1; i <= MAX_ORDER; ++i) {
      const double i_t = (double)i, one_over_i = 1.0 / i_t, one_over_i_plus_one = 1.0 / (i_t + 1.0);
      gamma[i] = gamma[i - 1] + one_over_i;
      alpha[i] = 1.0 / ((1.0 - kappa[i]) * gamma[i]);
      const double e = kappa[i] * gamma[i] + one_over_i_plus_one;
      ec2[i] = e * e;
    }
    c = h * alpha[order];
    reset_jacobian();
    for (int j = 0; j < NC; ++j) for (int i = 0; i < N; ++i) { D[j][i] = 0.0; Dt[j][i] = 0.0; }
    for (int i = 0; i < N; ++i) { D[0][i] = y[i]; D[1][i] = dy[i] * h; }
    compute_u(order);
    st.number_of_linear_solver_setups = 1;
    st.setups_from_checkpoint = 1;
  }

  void jacobian_updates(double cc, SolverState state) {
    bool did = false;
    if (ju.check_rhs_jacobian_update(cc, state)) {
      jac_stale = true;
      reset_jacobian();
      ju.update_rhs_jacobian(cc);
      ju.update_jacobian(cc);
      eta = std::pow(20.0, 1.25);
      did = true;
    } else if (ju.check_jacobian_update(cc, state)) {
      reset_jacobian();
      ju.update_jacobian(cc);
      eta = std::pow(20.0, 1.25);
      did = true;
    }
    if (did) record_linear_solver_setup(st, state);
  }
  bool update_step_size(double factor, double& new_h) {  // false = StepSizeTooSmall
    new_h = factor * h;
    n_equal_steps = 0;
    const int m = order + 1;
    double r[36], ru[36];
    for (int j = 0; j < m; ++j) r[j * m] = 1.0;
    for (int j = 0; j < m; ++j)
      for (int i = 1; i < m; ++i) r[j * m + i] = j == 0 ? 0.0 : r[j * m + i - 1] * ((double)i - 1.0 - factor * (double)j) / (double)i;
    for (int j = 0; j < m; ++j)  // mat_mul_small(r, u): gemm_cols order
      for (int i = 0; i < m; ++i) {
        double acc = r[0 * m + i] * u[j * m + 0];
        for (int k = 1; k < m; ++k) acc = r[k * m + i] * u[j * m + k] + acc;
        ru[j * m + i] = acc;
      }
    for (int j = 0; j < m; ++j)  // gemm_cols(diff_tmp, diff, order + 1, ru)
      for (int i = 0; i < N; ++i) {
        double acc = D[0][i] * ru[j * m + 0];
        for (int k = 1; k < m; ++k) acc = D[k][i] * ru[j * m + k] + acc;
        Dt[j][i] = acc;
      }
    for (int j = 0; j < NC; ++j) for (int i = 0; i < N; ++i) std::swap(D[j][i], Dt[j][i]);
    c = new_h * alpha[order];
    h = new_h;
    eta = std::pow(100.0, 1.25);
    return std::fabs(h) >= min_h;
  }
  void predict_forward() {
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
      for (int j = 0; j <= order; ++j) s = s + D[j][i];
      double q = gamma[1] * D[1][i];
      for (int j = 2; j <= order; ++j) q = gamma[j] * D[j][i] + 1.0 * q;
      q = q * alpha[order];
      q = q - s;
      yp[i] = s;
      psi[i] = q;
    }
    t_predict = t + h;
  }
  // 0 nothing, 1 TstopReached, 2 StopTimeBeforeCurrentTime
  int handle_tstop() {
    const double eps = std::numeric_limits<double>::epsilon();
    const double troundoff = 100.0 * eps * (std::fabs(t) + std::fabs(h));
    if (std::fabs(t - tstop) <= troundoff) { has_tstop = false; return 1; }
    if ((h > 0.0 && tstop < t - troundoff) || (h < 0.0 && tstop > t + troundoff)) { has_tstop = false; return 2; }
    if ((h > 0.0 && t + h > tstop + troundoff) || (h < 0.0 && t + h < tstop - troundoff)) {
      const double factor = (tstop - t) / h;
      double nh;
      (void)update_step_size(factor, nh);
    }
    return 0;
  }
  OdeErr set_stop_time(double ts) {
    has_tstop = true; tstop = ts;
    const int r = handle_tstop();
    if (r == 2) return OdeErr::StopTimeBeforeCurrentTime;
    if (r == 1) { has_tstop = false; return OdeErr::StopTimeAtCurrentTime; }
    return OdeErr::Ok;
  }

  OdeErr step(StopReason& reason) {
    double safety = 0.0, error_norm = 0.0;
    const long old_fails = st.number_of_error_test_failures;
    bool convergence_fail = false;
    double x[N], ydelta[N];
    predict_forward();
    while (true) {
      const int ord = order;
      for (int i = 0; i < N; ++i) x[i] = yp[i];
      // newton_iteration over NoLineSearch
      niter = 0; has_old = false;
      bool solved = false;
      for (int it = 0; it < max_iter; ++it) {
        double f[N], delta[N];
        mdl.rhs(x, p, t_predict, f);
        for (int i = 0; i < N; ++i) delta[i] = 1.0 * (x[i] + psi[i]) + (-c) * f[i];  // BdfCallable::call_inplace, identity mass
        if (!lu_solve(delta)) break;
        for (int i = 0; i < N; ++i) x[i] = x[i] - delta[i];
        const double norm = std::sqrt(sqnorm(delta, yp));
        niter += 1;
        bool diverged = false;
        if (has_old) {
          const double rate = rpow(norm / old_norm, 1.0 / (double)(niter - 1));
          if (rate > 0.9) diverged = true;
          else if (powi(rate, max_iter - niter) / (1.0 - rate) * norm > conv_tol) diverged = true;
          else eta = rate / (1.0 - rate);
        } else {
          const double min_eta = 1e4 * std::numeric_limits<double>::epsilon();
          if (eta < min_eta) eta = min_eta;
          eta = rpow(eta, 0.8);
        }
        const bool converged = !diverged && eta * norm < conv_tol;
        if (niter == 1) { has_old = true; old_norm = norm; }
        if (diverged) break;
        if (converged) { solved = true; break; }
      }
      st.number_of_nonlinear_solver_iterations += niter;
      if (!solved) {
        st.number_of_nonlinear_solver_fails += 1;
        if (st.number_of_nonlinear_solver_fails > o.max_nonlinear_solver_failures) return OdeErr::TooManyNonlinearSolverFailures;
        has_prev_err = false;
        if (convergence_fail) {
          double new_h;
          if (!update_step_size(0.3, new_h)) return OdeErr::StepSizeTooSmall;
          jacobian_updates(new_h * alpha[ord], SolverState::SecondConvergenceFail);
          predict_forward();
        } else {
          jacobian_updates(h * alpha[ord], SolverState::FirstConvergenceFail);
          convergence_fail = true;
        }
        continue;
      }
      for (int i = 0; i < N; ++i) ydelta[i] = x[i] - yp[i];
      error_norm = std::fmax(0.0, sqnorm(ydelta, y) * ec2[order - 1]);
      const double maxiter = (double)max_iter;
      safety = 0.9 * (2.0 * maxiter + 1.0) / (2.0 * maxiter + (double)niter);
      if (error_norm <= 1.0) break;
      double factor = safety * pi_controller_raw(error_norm, has_prev_err ? std::optional<double>(prev_err) : std::nullopt, o.pi_control_integral, o.pi_control_proportional, ord + 1);
      has_prev_err = false;
      if (factor < min_shrink) factor = min_shrink;
      double new_h;
      if (!update_step_size(factor, new_h)) return OdeErr::StepSizeTooSmall;
      jacobian_updates(new_h * alpha[ord], SolverState::ErrorTestFail);
      predict_forward();
      st.number_of_error_test_failures += 1;
      if (st.number_of_error_test_failures - old_fails >= o.max_error_test_failures) return OdeErr::TooManyErrorTestFailures;
    }
    // _update_diff
    for (int i = 0; i < N; ++i) {
      const double dm = ydelta[i] - D[order + 1][i];
      D[order + 2][i] = dm;
      D[order + 1][i] = ydelta[i];
    }
    for (int j = order; j >= 0; --j) for (int i = 0; i < N; ++i) D[j][i] = D[j][i] + 1.0 * D[j + 1][i];
    for (int i = 0; i < N; ++i) y[i] = yp[i];
    t = t_predict;
    for (int i = 0; i < N; ++i) dy[i] = D[1][i] * (1.0 / h);
    st.number_of_steps += 1;
    ju.step();
    prev_err = error_norm; has_prev_err = true;
    n_equal_steps += 1;
    if (n_equal_steps > order) {
      const int ord = order;
      const double inf = std::numeric_limits<double>::infinity();
      const double error_m_norm = ord > 1 ? sqnorm(D[ord], y) * ec2[ord - 1] : inf;
      const double error_p_norm = ord < MAX_ORDER ? sqnorm(D[ord + 2], y) * ec2[ord + 1] : inf;
      const std::optional<double> pe(prev_err);
      const double factors[3] = {pi_controller_raw(error_m_norm, pe, o.pi_control_integral, o.pi_control_proportional, ord),
                                 pi_controller_raw(error_norm, pe, o.pi_control_integral, o.pi_control_proportional, ord + 1),
                                 pi_controller_raw(error_p_norm, pe, o.pi_control_integral, o.pi_control_proportional, ord + 2)};
      int max_index = 0;
      for (int k = 1; k < 3; ++k) if (factors[k] >= factors[max_index]) max_index = k;
      const int new_order = max_index == 0 ? ord - 1 : (max_index == 1 ? ord : ord + 1);
      order = new_order;
      if (max_index != 1) compute_u(new_order);
      double factor = safety * factors[max_index];
      if (factor > max_growth) factor = max_growth;
      if (factor < min_shrink) factor = min_shrink;
      if (factor >= min_growth || factor <= max_shrink || max_index == 0 || max_index == 2) {
        double new_h;
        if (!update_step_size(factor, new_h)) return OdeErr::StepSizeTooSmall;
        jacobian_updates(new_h * alpha[new_order], SolverState::StepSuccess);
      }
    }
    if (has_tstop) {
      const int r = handle_tstop();
      if (r == 1) { reason = StopReason::TstopReached; return OdeErr::Ok; }
    }
    reason = StopReason::InternalTimestep;
    return OdeErr::Ok;
  }
};

// OdeSolverMethod::solve(final_time) of one member: steps until TstopReached; returns the final state.y and the counters.
template <int N>
inline OdeErr fast_solve(const Model& m, const double* p, double rtol, const double* atol, int natol, double t0, double h0, const OdeSolverOptions& o,
                         double t_final, double* y_out, Stats* stats_out) {
  FastBdf<N> s(m, p, rtol, atol, natol, t0, h0, o);
  OdeErr e = s.set_stop_time(t_final);
  if (e != OdeErr::Ok) return e;
  StopReason r = StopReason::InternalTimestep;
  while (r != StopReason::TstopReached) {
    e = s.step(r);
    if (e != OdeErr::Ok) return e;
  }
  if (y_out) for (int i = 0; i < N; ++i) y_out[i] = s.y[i];
  if (stats_out) *stats_out = s.st;
  return OdeErr::Ok;
}

}  // namespace orc
