// Shared device-side pieces of the device-resident integrators (dsh_adaptive.hip: BDF, dsh_sdirk_resident.hip: TR-BDF2 / ESDIRK34), gfx950.
//
// One lane per ensemble member, solver state in registers.  WAVE selects the control granularity: false = every member decides for itself
// (diffsol's CPU semantics for a sweep of independent IVPs), true = the 64 members of a wavefront advance in lock-step, norms max-reduced over
// the wavefront (the reference's batched semantics, nbatch = 64 per group; every control scalar wavefront-uniform).
//
// Restated here, per lane, from the reference (paths relative to crates/):
//   Convergence                      diffsol-nl/src/convergence.rs:7-140
//   BacktrackingLineSearch           diffsol-nl/src/line_search.rs:84-201      (consistent initialisation only)
//   InitOp / set_consistent          diffsol/src/op/init.rs:14-135, ode_solver/state.rs:84-162
//   set_step_size                    diffsol/src/ode_solver/state.rs:1209-1277
//   RootFinder / root_finding        diffsol/src/nonlinear_solver/root.rs:12-222, diffsol-la/src/vector/nalgebra_serial.rs:484-504, vector/cuda.rs:1153-1177
//   JacobianUpdate                   diffsol/src/ode_solver/jacobian_update.rs:12-79
//   pi controller                    diffsol/src/ode_solver/runge_kutta.rs:1313-1336
// Arithmetic is the oracle's operation for operation (-ffp-contract=off); pow() is ocml's (see dsh_adaptive.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/diffsol_detpow.h"
#include "dsh_device.hpp"
#include "dsh_lu_dev.hpp"
#include "dsh_models.hpp"

namespace dsh {

constexpr double kEps = 2.220446049250313e-16;

enum ResidentStatus : int32_t {  // OdeSolverError ordinals of the host library (host/ode.hpp) where they exist
  kRsOk = 0, kRsStepSizeTooSmall = 1, kRsTooManyErrorTestFailures = 2, kRsTooManyNonlinearSolverFailures = 3, kRsInitialConditionDidNotConverge = 4,
  kRsStopTimeBeforeCurrentTime = 5, kRsStopTimeAtCurrentTime = 6, kRsRootBatchMismatch = 20, kRsMaxStepsExceeded = 99
};

// host-computed constants shared by the resident kernels
struct ResidentConsts {
  double rtol, t0, h0;
  double eta_reset, eta_reset_ts;  // 20^1.25, 100^1.25 (convergence.rs:36-42)
  double ls_steptol;               // eps^(2/3)          (line_search.rs:100)
  dsh_adaptive_options o;
  int n_eval;
  int member_lanes;  // per-member control (group 1) of the lane-per-member kernels: lanes of every wavefront that carry a member (0 = all 64); the other lanes shadow the first
};

// weighted mean square, sequential like Vector::squared_norm (nalgebra_serial.rs:395-408)
template <int N>
__device__ __forceinline__ double wms(const double (&v)[N], const double (&w)[N], const double (&atol)[N], double rtol) {
  double acc = 0.0;
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) {
    const double term = v[i] / (fabs(w[i]) * rtol + atol[i]);
    acc += term * term;
  }
  return acc / (double)N;
}

// Group reduction of a mean-square norm: the member's own value, or the max over the wavefront (Vector::squared_norm's max over the batch,
// vector/cuda.rs:1421-1432, for a batch of 64; a NaN wins like in the oracle).
template <bool WAVE>
__device__ __forceinline__ double group_norm(double v) {
  if constexpr (WAVE) return __longlong_as_double((long long)wave_max_u64(d2u(v)));
  else return v;
}
template <bool WAVE>
__device__ __forceinline__ bool group_all(bool ok) {
  if constexpr (WAVE) return __all(ok);
  else return ok;
}
// value of wavefront lane 0 (batch member 0 of the group) — the reference's batched RootFinder reads g.get_index(i) of batch 0
template <bool WAVE>
__device__ __forceinline__ double group_first(double v) {
  if constexpr (WAVE) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
  } else return v;
}
template <bool WAVE>
__device__ __forceinline__ int group_first_i(int v) {
  if constexpr (WAVE) return __builtin_amdgcn_readfirstlane(v);
  else return v;
}

// pow as the kernels use it: ocml's, or the deterministic one of diffsol_detpow.h (dsh_adaptive_options.deterministic_pow)
// (not inlined: the kernels call it from a dozen sites and both implementations are a few hundred instructions)
__device__ __attribute__((noinline)) double rpow(double x, double y, bool det) { return det ? dsh_det_pow(x, y) : pow(x, y); }

// compiler-rt __powidf2 (what f64::powi lowers to; convergence.rs:85)
__device__ __forceinline__ double powi_rt(double a, int b) {
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

// runge_kutta.rs:1313-1336
__device__ __forceinline__ double pi_controller_raw(double error_norm, bool has_prev, double prev, double pi_i, double pi_p, int eff_order, bool det) {
  const double order_f = (double)eff_order;
  const double ki = pi_i / order_f;
  if (pi_p == 0.0) return rpow(error_norm, -ki, det);
  if (has_prev) {
    const double kp = pi_p / order_f;
    return rpow(error_norm, -(ki + kp), det) * rpow(prev, kp, det);
  }
  return rpow(error_norm, -ki, det);
}

// convergence.rs:7-140
enum class ConvStatus { Converged, Diverged, Continue };
struct ConvState {
  double eta;
  double tol;
  int max_iter;
  bool det = false;
  int niter = 0;
  bool has_old = false;
  double old_norm = 0.0;
  __device__ __forceinline__ void reset() { niter = 0; has_old = false; }
  __device__ __forceinline__ ConvStatus check_norm(double norm) {  // :68-131
    niter += 1;
    if (has_old) {
      // pow(x, 1.0) == x exactly: the common second iteration needs no libm call
      const double rate = niter == 2 ? norm / old_norm : rpow(norm / old_norm, 1.0 / (double)(niter - 1), det);
      if (rate > 0.9) return ConvStatus::Diverged;
      if (powi_rt(rate, max_iter - niter) / (1.0 - rate) * norm > tol) return ConvStatus::Diverged;
      eta = rate / (1.0 - rate);
    } else {
      const double min_eta = 1e4 * kEps;
      if (eta < min_eta) eta = min_eta;
      eta = rpow(eta, 0.8, det);
    }
    if (eta * norm < tol) return ConvStatus::Converged;
    return ConvStatus::Continue;
  }
  __device__ __forceinline__ ConvStatus check_new_iteration(double norm) {  // :133-139
    const ConvStatus s = check_norm(norm);
    if (niter == 1) { has_old = true; old_norm = norm; }
    return s;
  }
};

// JacobianUpdate (jacobian_update.rs:12-79)
enum class JState { StepSuccess, FirstConvergenceFail, SecondConvergenceFail, ErrorTestFail };
struct JacUpdateState {
  int steps_since_jac = 0, steps_since_rhs_jac = 0;
  double h_at_last = 1.0;
  __device__ __forceinline__ void update_jacobian(double h) { steps_since_jac = 0; h_at_last = h; }
  __device__ __forceinline__ void update_rhs_jacobian(double h) { steps_since_rhs_jac = 0; steps_since_jac = 0; h_at_last = h; }
  __device__ __forceinline__ void step() { steps_since_jac += 1; steps_since_rhs_jac += 1; }
  __device__ __forceinline__ bool check_jacobian_update(double h, JState st, const dsh_adaptive_options& o) const {
    if (st == JState::StepSuccess) return steps_since_jac >= o.update_jacobian_after_steps || fabs(h / h_at_last - 1.0) > o.threshold_to_update_jacobian;
    return true;
  }
  __device__ __forceinline__ bool check_rhs_jacobian_update(double h, JState st, const dsh_adaptive_options& o) const {
    switch (st) {
      case JState::StepSuccess: return steps_since_rhs_jac >= o.update_rhs_jacobian_after_steps;
      case JState::FirstConvergenceFail: return fabs(h / h_at_last - 1.0) < o.threshold_to_update_rhs_jacobian;
      case JState::SecondConvergenceFail: return steps_since_rhs_jac > 0;
      case JState::ErrorTestFail: return false;
    }
    return false;
  }
};

// set_step_size (state.rs:1209-1277)
template <class Mdl, bool WAVE>
__device__ __forceinline__ double initial_step_size(double t, double h0_in, const double (&y)[Mdl::N], const double (&f0)[Mdl::N], const double (&p)[Mdl::NP],
                                                    const double (&atol)[Mdl::N], double rtol, int solver_order, bool det) {
  constexpr int N = Mdl::N;
  const bool is_neg_h = h0_in < 0.0;
  const double d0 = sqrt(group_norm<WAVE>(wms<N>(y, y, atol, rtol))), d1 = sqrt(group_norm<WAVE>(wms<N>(f0, y, atol, rtol)));
  const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
  const double hh = is_neg_h ? -h0 : h0;
  double y1[N], f1[N], df[N];
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) y1[i] = f0[i] * hh + y[i];
  Mdl::rhs(is_neg_h ? t - h0 : t + h0, y1, p, f1);
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) df[i] = f1[i] - f0[i];
  const double d2 = sqrt(group_norm<WAVE>(wms<N>(df, y, atol, rtol))) / fabs(h0);
  double max_d = d2;
  if (max_d < d1) max_d = d1;
  double h1;
  if (max_d < 1e-15) { h1 = h0 * 1e-3; if (h1 < 1e-6) h1 = 1e-6; }
  else h1 = rpow(0.01 / max_d, 1.0 / (1.0 + (double)solver_order), det);
  double h = 100.0 * h0;
  if (h > h1) h = h1;
  if (is_neg_h) h = -h;
  return h;
}

// StateRefMut::set_consistent (state.rs:84-162) over InitOp (op/init.rs:14-135), Newton with the backtracking line search (line_search.rs:84-201).
// Returns false for InitialConditionDidNotConverge.
template <class Mdl, bool WAVE>
__device__ __forceinline__ bool set_consistent(double t0, const double (&p)[Mdl::NP], double (&y)[Mdl::N], double (&dy)[Mdl::N], const double (&atol)[Mdl::N],
                                               double rtol, const ResidentConsts& C, bool no_linesearch = false) {  // no_linesearch: apply_reset_with_mass's root solver (state.rs:297-300)
  constexpr int N = Mdl::N;
  if constexpr (!Mdl::HAS_MASS) {
    return true;
  } else {
    const dsh_adaptive_options& o = C.o;
    double Mm[N * N];
    assemble_mass<Mdl>(t0, p, Mm);
    bool is_alg[N];
    bool any_alg = false;
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) { is_alg[i] = Mm[i * N + i] == 0.0; any_alg = any_alg || is_alg[i]; }  // partition_indices_by_zero_diagonal
    if (!any_alg) return true;
    // InitOp::new: jac = (-M_u, df/dv; 0, dg/dv), neg_mass = (-M_u, 0; 0, 0) in the original ordering
    double rj[N * N], jac[N * N], neg_mass[N * N];
    assemble_jacobian<Mdl>(t0, y, p, rj);
DSH_UNROLL_N
    for (int j = 0; j < N; ++j)
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) {
        if (!is_alg[j]) {
          const double v = is_alg[i] ? 0.0 : Mm[j * N + i] * (-1.0);
          jac[j * N + i] = v;
          neg_mass[j * N + i] = v;
        } else {
          jac[j * N + i] = rj[j * N + i];
          neg_mass[j * N + i] = 0.0;
        }
      }
    double y0[N], x[N], yerr[N], delta[N];
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) { y0[i] = y[i]; x[i] = is_alg[i] ? y[i] : dy[i]; yerr[i] = x[i]; delta[i] = 0.0; }
    // InitOp::call_inplace (:103-115): y0[alg] = x[alg]; out = f(y0) ; out = neg_mass x + out  (nalgebra gemv order)
    auto fun = [&](const double (&xx)[N], double (&out)[N]) __attribute__((always_inline)) {
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) if (is_alg[i]) y0[i] = xx[i];
      Mdl::rhs(t0, y0, p, out);
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) {
        double acc = 1.0 * neg_mass[0 * N + i] * xx[0] + 1.0 * out[i];
#pragma unroll
        for (int j = 1; j < N; ++j) acc = 1.0 * neg_mass[j * N + i] * xx[j] + acc;
        out[i] = acc;
      }
    };
    ConvState conv;
    conv.eta = C.eta_reset;
    conv.tol = o.nonlinear_solver_tolerance;
    conv.max_iter = o.ic_max_newton_iterations;
    conv.det = o.deterministic_pow != 0;
    bool ok = false;
    for (int k = 0; k < o.ic_max_linear_solver_setups; ++k) {
      // reset_jacobian: the InitOp Jacobian is constant
      double A[N * N];
      int P[N];
#pragma unroll
      for (int e = 0; e < N * N; ++e) A[e] = jac[e];
      bool sing = false;
      lu_factor_reg<N>(A, P, sing);
      // newton_iteration (newton.rs:13-36)
      conv.reset();
      double ls_norm = 1.0;  // BacktrackingLineSearch::norm persists over the iterations of one solve
      int result = 2;        // 0 ok, 1 fatal (diverged / LU / line search), 2 NewtonMaxIterations
      for (int it = 0; it < conv.max_iter; ++it) {
        ConvStatus st = ConvStatus::Continue;
        bool fatal = false;
        if (!o.ic_use_linesearch || no_linesearch) {  // NoLineSearch::take_optimal_step
          fun(x, delta);
          if (!group_all<WAVE>(lu_solve_reg<N>(A, P, delta))) { fatal = true; }
          else {
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) x[i] = x[i] - delta[i];
            st = conv.check_new_iteration(sqrt(group_norm<WAVE>(wms<N>(delta, yerr, atol, rtol))));
          }
        } else {  // BacktrackingLineSearch::take_optimal_step
          bool returned = false;
          if (conv.niter == 0) {
            fun(x, delta);
            if (!group_all<WAVE>(lu_solve_reg<N>(A, P, delta))) { fatal = true; returned = true; }
            else {
              ls_norm = sqrt(group_norm<WAVE>(wms<N>(delta, yerr, atol, rtol)));
              if (conv.check_norm(ls_norm) == ConvStatus::Converged) {
DSH_UNROLL_N
                for (int i = 0; i < N; ++i) x[i] = x[i] - delta[i];
                st = ConvStatus::Converged;
                returned = true;
              }
            }
          }
          if (!returned) {
            double x0[N], delta0[N];
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) { x0[i] = x[i]; delta0[i] = delta[i]; }
            const double nrm = ls_norm;
            const double phi0 = nrm * nrm * 0.5, two_phi0 = nrm * nrm, min_alpha = C.ls_steptol / nrm;
            double alpha = 1.0;
            bool found = false;
            for (int i = 0; i < o.ic_max_linesearch_iterations; ++i) {
DSH_UNROLL_N
              for (int q = 0; q < N; ++q) x[q] = (-alpha) * delta0[q] + 1.0 * x[q];
              fun(x, delta);
              if (!group_all<WAVE>(lu_solve_reg<N>(A, P, delta))) { fatal = true; break; }
              const double new_norm = sqrt(group_norm<WAVE>(wms<N>(delta, yerr, atol, rtol)));
              const double phi1 = new_norm * new_norm * 0.5;
              if (phi1 <= phi0 - o.ic_armijo_constant * alpha * two_phi0) {
                ls_norm = new_norm;
                st = conv.check_norm(new_norm);
                found = true;
                break;
              }
              if (alpha < min_alpha) { fatal = true; break; }  // LinesearchFailedMinStep
              alpha *= o.ic_step_reduction_factor;
DSH_UNROLL_N
              for (int q = 0; q < N; ++q) x[q] = x0[q];
            }
            if (!found) fatal = true;  // incl. LinesearchFailedMaxIterations
          }
        }
        if (fatal) { result = 1; break; }
        if (st == ConvStatus::Converged) { result = 0; break; }
        if (st == ConvStatus::Diverged) { result = 1; break; }
      }
      if (result == 0) { ok = true; break; }
      if (result != 2) return false;  // anything but NewtonMaxIterations is fatal (state.rs:131-140)
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) yerr[i] = x[i];
    }
    if (!ok) return false;
    // scatter_soln (:76-81) + zero the algebraic derivatives (state.rs:155-158)
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) {
      if (is_alg[i]) { y[i] = x[i]; dy[i] = 0.0; }
      else dy[i] = x[i];
    }
    return true;
  }
}

// Vector::root_finding for one member (nalgebra_serial.rs:484-504): found = any g1 == 0; idx/frac of the largest |g1/(g1-g0)| over sign changes
template <int NR>
__device__ __forceinline__ void root_finding_lane(const double (&g0)[NR], const double (&g1)[NR], bool& found, double& frac, int& idx) {
  found = false; frac = 0.0; idx = -1;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const double a = g0[i], c = g1[i];
    if (c == 0.0) found = true;
    if (a * c < 0.0) { const double f = fabs(c / (c - a)); if (f > frac) { frac = f; idx = i; } }
  }
}

// RootFinder (root.rs:12-222).  g0 holds the root function at the previous step; `interp(t, y)` interpolates inside the last step.
// Returns 0 = no root, 1 = root found (t_root, root_idx set), 2 = batch mismatch (wavefront lock-step only: members disagree on the crossing,
// vector/cuda.rs:1166-1171 panics there).
template <class Mdl, bool WAVE, class Interp>
__device__ __forceinline__ int check_root(double (&g0)[Mdl::NROOTS > 0 ? Mdl::NROOTS : 1], double& rf_t0, const double (&y)[Mdl::N], double t, const double (&p)[Mdl::NP],
                                          Interp&& interp, double& t_root, int& root_idx) {
  constexpr int N = Mdl::N, NR = Mdl::NROOTS > 0 ? Mdl::NROOTS : 1;
  double g1[NR], gmid[NR], ymid[N];
  Mdl::root(t, y, p, g1);
  // batched root_finding: every member must agree with member 0 on (found, idx)
  auto rf = [&](const double (&ga)[NR], const double (&gb)[NR], bool& found, int& idx, bool& mismatch) __attribute__((always_inline)) {
    double frac;
    root_finding_lane<NR>(ga, gb, found, frac, idx);
    if constexpr (WAVE) {
      const int f0 = group_first_i<WAVE>(found ? 1 : 0), i0 = group_first_i<WAVE>(idx);
      if (!__all((found ? 1 : 0) == f0 && idx == i0)) mismatch = true;
      found = f0 != 0; idx = i0;
    }
  };
  auto find_zero_index = [&](const double (&g)[NR]) __attribute__((always_inline)) -> int {  // of member 0
    int mi = 0;
    double mv = fabs(group_first<WAVE>(g[0]));
#pragma unroll
    for (int i = 1; i < NR; ++i) { const double v = fabs(group_first<WAVE>(g[i])); if (v < mv) { mv = v; mi = i; } }
    return mi;
  };
  auto pick = [&](const double (&g)[NR], int i) __attribute__((always_inline)) -> double {  // g[i] of member 0, i dynamic
    double v = g[0];
#pragma unroll
    for (int k = 1; k < NR; ++k) if (k == i) v = g[k];
    return group_first<WAVE>(v);
  };
  bool found, mismatch = false;
  int imax;
  rf(g0, g1, found, imax, mismatch);
  if (mismatch) return 2;
  if (imax < 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) g0[i] = g1[i];
    rf_t0 = t;
    if (found) { t_root = t; root_idx = find_zero_index(g0); return 1; }
    return 0;
  }
  double alpha = 1.0;
  bool sc0 = false, sc1 = true;
  int it = 0;
  double t1 = t, t0l = rf_t0;
  const double tol = 100.0 * kEps * (fabs(t1) + fabs(t1 - t0l));
  while (fabs(t1 - t0l) > tol) {
    const double g1v = pick(g1, imax), g0v = pick(g0, imax);
    double t_mid = t1 - (t1 - t0l) * g1v / (g1v - alpha * g0v);
    if (fabs(t_mid - t0l) < 0.5 * tol) {
      const double fracint = fabs(t1 - t0l) / tol;
      const double fracsub = fracint > 5.0 ? 0.1 : 0.5 / fracint;
      t_mid = t0l + fracsub * (t1 - t0l);
    }
    if (fabs(t1 - t_mid) < 0.5 * tol) {
      const double fracint = fabs(t1 - t0l) / tol;
      const double fracsub = fracint > 5.0 ? 0.1 : 0.5 / fracint;
      t_mid = t1 - fracsub * (t1 - t0l);
    }
    interp(t_mid, ymid);
    Mdl::root(t_mid, ymid, p, gmid);
    bool f2;
    int i2;
    rf(g0, gmid, f2, i2, mismatch);
    if (mismatch) return 2;
    const bool lower = i2 >= 0;
    if (lower) {
      t1 = t_mid; imax = i2;
#pragma unroll
      for (int i = 0; i < NR; ++i) { const double tmp = g1[i]; g1[i] = gmid[i]; gmid[i] = tmp; }
    } else if (f2) {
      Mdl::root(t, y, p, g0);
      rf_t0 = t;  // (the reference leaves t0 stale here; the solve stops at this root, so it is never read again)
      t_root = t_mid; root_idx = imax;
      return 1;
    } else {
      t0l = t_mid;
#pragma unroll
      for (int i = 0; i < NR; ++i) { const double tmp = g0[i]; g0[i] = gmid[i]; gmid[i] = tmp; }
    }
    if ((it & 1) == 0) sc0 = lower; else sc1 = lower;
    if (it >= 2) alpha = (sc0 != sc1) ? 1.0 : (sc0 ? 0.5 * alpha : 2.0 * alpha);
    it += 1;
  }
  Mdl::root(t, y, p, g0);
  t_root = t1; root_idx = imax;
  return 1;
}

}  // namespace dsh
