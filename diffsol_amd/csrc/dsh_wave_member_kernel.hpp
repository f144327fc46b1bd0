// One wavefront per ensemble member: variable-order BDF for run-time-sized models with n <= 64 (launch code and documentation: dsh_wave_member.hip).
// In a header so that the run-time-compiled module of a DiffSL model (dsh_jit.hip, through dsh_jit_wave_member.hpp which defines DSH_JIT_DYNAMIC
// after the generated jit_* functions) instantiates the same kernel with the user's model behind the three hooks below.
#pragma once
#include "dsh_resident.hpp"
#include "dsh_lu_wave.hpp"
#ifndef DSH_JIT_DYNAMIC
#include "dsh_models_dyn.hpp"
#endif

namespace dsh {

// ---- model hooks: component i of f / J v, initial value of component i, stop conditions (at most 2)
#ifdef DSH_JIT_DYNAMIC
template <class XF, class VF, class PF>
__device__ __forceinline__ double wm_component(int, int64_t, double t, int64_t i, XF X, VF V, PF P, bool jac) { return jit_component(t, (long)i, X, V, P, jac); }
template <class PF>
__device__ __forceinline__ double wm_init_value(int, int64_t, int64_t i, double t, PF P) { return jit_init_component(t, (long)i, P); }
template <class XF, class PF>
__device__ __forceinline__ int wm_root_values(int, int64_t, double t, XF X, PF P, double (&g)[2]) {
  if (kJitNRoots > 0) g[0] = jit_root_component(t, 0, X, P);
  if (kJitNRoots > 1) g[1] = jit_root_component(t, 1, X, P);
  return kJitNRoots < 2 ? kJitNRoots : 2;
}
// component i of M(t) x: DiffSL models may carry a (singular) mass matrix — DAEs; the built-in run-time-sized models do not
constexpr bool kWmHasMass = kJitHasMass;
template <class XF, class PF>
__device__ __forceinline__ double wm_mass_component(double t, int64_t i, XF X, PF P) { return jit_mass_component(t, (long)i, X, P); }
// component i of (df/dp) v (init = false) or of (du0/dp) v (init = true): forward sensitivities (k_bdf_wave_member<.., SENS>)
template <class XF, class VF, class PF>
__device__ __forceinline__ double wm_sens_component(double t, int64_t i, XF X, VF V, PF P, bool init) { return jit_sens_component(t, (long)i, X, V, P, init); }
// hybrid models (DiffSL reset_i): component i of the state after an event; the kernels then apply it at every event and go on (solve_dense with a reset operator,
// method.rs:774-797) instead of stopping — models without a mass matrix
constexpr bool kWmResets = kJitHasReset && !kJitHasMass;
template <class XF, class PF>
__device__ __forceinline__ double wm_reset_component(double t, int64_t i, XF X, PF P) { return jit_reset_component(t, (long)i, X, P); }
#else
constexpr bool kWmResets = false;
template <class XF, class PF>
__device__ __forceinline__ double wm_reset_component(double, int64_t i, XF X, PF) { return X(i); }
template <class XF, class VF, class PF>
__device__ __forceinline__ double wm_sens_component(double, int64_t, XF, VF, PF, bool) { return 0.0; }  // the built-in run-time-sized models carry no parameter derivatives here
constexpr bool kWmHasMass = false;
template <class XF, class PF>
__device__ __forceinline__ double wm_mass_component(double, int64_t i, XF X, PF) { return X(i); }
template <class XF, class VF, class PF>
__device__ __forceinline__ double wm_component(int model, int64_t n, double t, int64_t i, XF X, VF V, PF P, bool jac) { return dyn_component(model, n, t, i, X, V, P, jac); }
template <class PF>
__device__ __forceinline__ double wm_init_value(int model, int64_t n, int64_t i, double, PF) { return dyn_init_value(model, n, i); }
template <class XF, class PF>
__device__ __forceinline__ int wm_root_values(int model, int64_t n, double t, XF X, PF P, double (&g)[2]) { return dyn_root_values(model, n, t, X, P, g); }
#endif


constexpr int kMaxOrder = 5, kNC = kMaxOrder + 3;

struct WaveMemberConsts {
  ResidentConsts r;
  double alpha[6], gamma[6], ec2[6];
  double u[kMaxOrder][36];
  int model, n, np, nroots;
  // forward sensitivities (k_bdf_wave_member<.., SENS = true>; bdf.rs:370-432, :934-989), as in AdaptiveConsts: sens_out n_eval x np x n x nb; one sens_atol for every state
  double* sens_out;
  double sens_rtol, sens_atol;
  int sens_error_control, sens_pad;
  // OdeSolverMethod::solve (method.rs:227-258 over :881-961), as in AdaptiveConsts: steps_cap > 0 makes the launch write the state after EVERY accepted step
  // (y_out [steps_cap][n][nb], steps_t_out [steps_cap][nb]; columns beyond steps_cap are counted, not stored) instead of interpolating at save points
  double* steps_t_out;
  int steps_cap, steps_pad;
};
constexpr int kWmMaxSensParams = 16;  // parameters whose sensitivities a lane keeps (np x 10 doubles of per-lane memory)

// sum over lanes 0..n-1 in index order (Vector::squared_norm's sequential accumulation)
template <int NP>
__device__ __forceinline__ double seq_sum(double term, int n) {
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < NP; ++i)
    if (i < n) acc += group_bcast<64>(term, i);
  return acc;
}

// StateRefMut::set_consistent (state.rs:84-162) over InitOp (op/init.rs:14-135), Newton with the backtracking line search (line_search.rs:84-201): the
// per-lane restatement of dsh_resident.hpp (set_consistent<Mdl, WAVE>) with one row per lane, shared by the wavefront-per-member BDF and SDIRK kernels.
// On entry xs need not hold anything; sJ / sM receive the InitOp Jacobian and its neg_mass block (the integrators recompute both afterwards).
// comp_of(): component ln of f at the vector published in xs; mass_e(j) / jac_e(j): entry (ln, j) of M / of f' at the published vector;
// factor_rows(): factor the rows in `a`; solve_rows(v): solve with them, unknown i to lane i; wms(v, w): weighted mean square.  False = did not converge.
template <int NP, class CompF, class MassF, class JacF, class FactorF, class SolveF, class WmsF>
__device__ __forceinline__ bool wm_set_consistent(int n, int ln, bool rowlive, double* xs, double* xs2, double* sJ, double* sM, double (&a)[NP], const ResidentConsts& R,
                                                  CompF comp_of, MassF mass_e, JacF jac_e, FactorF factor_rows, SolveF solve_rows, WmsF wms, double& y, double& f0) {
  const dsh_adaptive_options& o = R.o;
  const bool det = o.deterministic_pow != 0;
  {
    // StateRefMut::set_consistent (state.rs:84-162) over InitOp (op/init.rs:14-135), Newton with the backtracking line search (line_search.rs:84-201):
    // the per-lane restatement of dsh_resident.hpp (set_consistent<Mdl, WAVE>) with one row per lane.  sJ holds the InitOp Jacobian, sM its neg_mass
    // block (both are recomputed for the integrator afterwards: jac_stale starts true).
    __syncthreads();
    xs[ln] = y;
    __syncthreads();
    double mdiag = 1.0;
    for (int j = 0; j < n; ++j) {
      const double mij = mass_e(j);
      const double rij = jac_e(j);
      sM[j * 64 + ln] = mij;
      sJ[j * 64 + ln] = rij;
      if (j == ln) mdiag = mij;
    }
    const bool is_alg = rowlive && mdiag == 0.0;  // partition_indices_by_zero_diagonal
    const unsigned long long alg_mask = __ballot(is_alg);
    if (alg_mask != 0ull) {
      // InitOp::new: jac = (-M_u, df/dv; 0, dg/dv), neg_mass = (-M_u, 0; 0, 0) in the original ordering
      for (int j = 0; j < n; ++j) {
        const bool alg_j = (alg_mask >> j) & 1ull;
        if (!alg_j) {
          const double v = is_alg ? 0.0 : sM[j * 64 + ln] * (-1.0);
          sJ[j * 64 + ln] = v;
          sM[j * 64 + ln] = v;
        } else {
          sM[j * 64 + ln] = 0.0;  // sJ keeps df/dv, dg/dv
        }
      }
      const double y_orig = y;
      double x = is_alg ? y : f0, yerr = x, delta = 0.0;
      // InitOp::call_inplace (:103-115): y0[alg] = x[alg]; out = f(y0); out = neg_mass x + out  (nalgebra gemv order)
      auto init_fun = [&](double x_mine) __attribute__((always_inline)) -> double {
        __syncthreads();
        xs[ln] = is_alg ? x_mine : y_orig;
        xs2[ln] = x_mine;
        __syncthreads();
        const double out = comp_of();
        double acc = 1.0 * sM[0 * 64 + ln] * xs2[0] + 1.0 * out;
        for (int j = 1; j < n; ++j) acc = 1.0 * sM[j * 64 + ln] * xs2[j] + acc;
        return rowlive ? acc : 0.0;
      };
      ConvState conv;
      conv.eta = R.eta_reset;
      conv.tol = o.nonlinear_solver_tolerance;
      conv.max_iter = o.ic_max_newton_iterations;
      conv.det = det;
      bool ok = false, fatal_all = false;
      for (int k = 0; k < o.ic_max_linear_solver_setups && !ok && !fatal_all; ++k) {
#pragma unroll
        for (int j = 0; j < NP; ++j) a[j] = (rowlive && j < n) ? sJ[j * 64 + ln] : 0.0;  // reset_jacobian: the InitOp Jacobian is constant
        factor_rows();
        conv.reset();
        double ls_norm = 1.0;
        int result = 2;  // 0 ok, 1 fatal (diverged / LU / line search), 2 NewtonMaxIterations
        for (int it = 0; it < conv.max_iter; ++it) {
          ConvStatus st = ConvStatus::Continue;
          bool fatal = false;
          if (!o.ic_use_linesearch) {  // NoLineSearch::take_optimal_step
            delta = init_fun(x);
            if (!solve_rows(delta)) fatal = true;
            else { x = x - delta; st = conv.check_new_iteration(sqrt(wms(delta, yerr))); }
          } else {  // BacktrackingLineSearch::take_optimal_step
            bool returned = false;
            if (conv.niter == 0) {
              delta = init_fun(x);
              if (!solve_rows(delta)) { fatal = true; returned = true; }
              else {
                ls_norm = sqrt(wms(delta, yerr));
                if (conv.check_norm(ls_norm) == ConvStatus::Converged) { x = x - delta; st = ConvStatus::Converged; returned = true; }
              }
            }
            if (!returned) {
              const double x0 = x, delta0 = delta;
              const double nrm = ls_norm;
              const double phi0 = nrm * nrm * 0.5, two_phi0 = nrm * nrm, min_alpha = R.ls_steptol / nrm;
              double alpha = 1.0;
              bool found = false;
              for (int i = 0; i < o.ic_max_linesearch_iterations; ++i) {
                x = (-alpha) * delta0 + 1.0 * x;
                delta = init_fun(x);
                if (!solve_rows(delta)) { fatal = true; break; }
                const double new_norm = sqrt(wms(delta, yerr));
                const double phi1 = new_norm * new_norm * 0.5;
                if (phi1 <= phi0 - o.ic_armijo_constant * alpha * two_phi0) { ls_norm = new_norm; st = conv.check_norm(new_norm); found = true; break; }
                if (alpha < min_alpha) { fatal = true; break; }  // LinesearchFailedMinStep
                alpha *= o.ic_step_reduction_factor;
                x = x0;
              }
              if (!found) fatal = true;  // incl. LinesearchFailedMaxIterations
            }
          }
          if (fatal) { result = 1; break; }
          if (st == ConvStatus::Converged) { result = 0; break; }
          if (st == ConvStatus::Diverged) { result = 1; break; }
        }
        if (result == 0) ok = true;
        else if (result != 2) fatal_all = true;  // anything but NewtonMaxIterations is fatal (state.rs:131-140)
        else yerr = x;
      }
      if (!ok) return false;
      if (is_alg) { y = x; f0 = 0.0; }  // scatter_soln (:76-81) + zero the algebraic derivatives (state.rs:155-158)
      else f0 = x;
    }
    }
  return true;
}

// SENS: forward sensitivities of every parameter alongside (run-time-compiled ODE models without a mass matrix and without root functions): the restatement of
// k_bdf_adaptive<.., SENS>'s sensitivity code (dsh_adaptive_kernel.hpp) with a component per lane — J s through the published vectors, the
// sensitivity solves on the state equations' factors in the lanes' registers, sdiff_j one row per lane in per-lane memory.
template <int NP, bool SENS = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_bdf_wave_member(int64_t nb, const double* __restrict__ p_g, const double* __restrict__ atol_g, int atol_broadcast,
                                                       const WaveMemberConsts* __restrict__ Cp, const double* __restrict__ t_eval, double* __restrict__ y_out,
                                                       int32_t* __restrict__ stats_out, int32_t* __restrict__ status_out, double* __restrict__ t_root_out,
                                                       int32_t* __restrict__ root_idx_out, int32_t* __restrict__ ncols_out, unsigned long long* __restrict__ totals) {
  extern __shared__ double lds[];  // xs[64] | ps[64] | sJ[n][64] | with a mass matrix: sM[n][64] | xs2[64]
  double* xs = lds;
  double* ps = lds + 64;
  double* sJ = lds + 128;
  double* sM = sJ + (size_t)Cp->n * 64;    // rows of M (only touched when kWmHasMass)
  double* xs2 = kWmHasMass ? sM + (size_t)Cp->n * 64 : sM;   // a second published vector (InitOp: x next to y0; Newton: y - y0 + psi; SENS: the direction of J v)
  const WaveMemberConsts& C = *Cp;
  const dsh_adaptive_options& o = C.r.o;
  const bool det = o.deterministic_pow != 0;
  const int n = C.n, model = C.model;
  const int64_t b = blockIdx.x;
  const int ln = threadIdx.x;
  const bool rowlive = ln < n;
  const double rtol = C.r.rtol;
  const double atol = rowlive ? (atol_broadcast ? atol_g[ln] : atol_g[(int64_t)ln * nb + b]) : 1.0;
  if (ln < C.np) ps[ln] = p_g[(int64_t)ln * nb + b];
  __syncthreads();
  auto Pf = [&](int64_t k) { return ps[k]; };
  auto Xf = [&](int64_t k) { return xs[k]; };
  auto V0 = [&](int64_t) { return 0.0; };
  // component `ln` of f(x, t); x is published through LDS
  auto rhs_of = [&](double x_mine, double tt) __attribute__((always_inline)) -> double {
    __syncthreads();
    xs[ln] = x_mine;
    __syncthreads();
    return rowlive ? wm_component(model, (int64_t)n, tt, (int64_t)ln, Xf, V0, Pf, false) : 0.0;
  };
  // weighted mean square of a distributed vector: (1/n) sum_i (v_i / (|w_i| rtol + atol_i))^2, summed in index order
  auto wms_wave = [&](double v_mine, double w_mine) __attribute__((always_inline)) -> double {
    const double term = rowlive ? v_mine / (fabs(w_mine) * rtol + atol) : 0.0;
    return seq_sum<NP>(term * term, n) / (double)n;
  };

  // ------------------------------------------------------------ new_and_consistent (identity mass: nothing to make consistent) + set_step_size
  int32_t status = kRsOk;
  double t = C.r.t0, h;
  double y = rowlive ? wm_init_value(model, (int64_t)n, (int64_t)ln, t, Pf) : 0.0;
  double f0 = rhs_of(y, t);
  double a[NP];  // my row of the LU factors (of the InitOp Jacobian during the consistent initialisation, of M - c J afterwards)
  int pos = ln, myinv = ln;
  // factor the rows in `a`; afterwards lane i needs the unknown number i of a solve: remember which lane holds position i
  auto factor_rows = [&]() __attribute__((always_inline)) {
    bool sing = false;
    int mypiv;
    wave_lu_factor_rows<NP, 64>(a, n, true, rowlive, ln, 0, pos, mypiv, sing);
    for (int k = 0; k < n; ++k) {
      const int holder = __ffsll((unsigned long long)__ballot(rowlive && pos == k)) - 1;
      if (ln == k) myinv = holder;
    }
  };
  if constexpr (kWmHasMass) {
    auto comp_of = [&]() __attribute__((always_inline)) { return rowlive ? wm_component(model, (int64_t)n, t, (int64_t)ln, Xf, V0, Pf, false) : 0.0; };
    auto mass_e = [&](int j) __attribute__((always_inline)) { auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; }; return rowlive ? wm_mass_component(t, (int64_t)ln, Ej, Pf) : 0.0; };
    auto jac_e = [&](int j) __attribute__((always_inline)) { auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; }; return rowlive ? wm_component(model, (int64_t)n, t, (int64_t)ln, Xf, Ej, Pf, true) : 0.0; };
    auto solve_rows = [&](double& v) __attribute__((always_inline)) -> bool { const bool ok = wave_lu_solve_rows<NP>(a, n, rowlive, pos, v); v = __shfl(v, myinv, 64); return ok; };
    if (!wm_set_consistent<NP>(n, ln, rowlive, xs, xs2, sJ, sM, a, C.r, comp_of, mass_e, jac_e, factor_rows, solve_rows, wms_wave, y, f0)) status = kRsInitialConditionDidNotConverge;
  }
  {
    const bool is_neg_h = C.r.h0 < 0.0;
    const double d0 = sqrt(wms_wave(y, y)), d1 = sqrt(wms_wave(f0, y));
    const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
    const double hh = is_neg_h ? -h0 : h0;
    const double y1 = f0 * hh + y;
    const double f1 = rhs_of(y1, is_neg_h ? t - h0 : t + h0);
    const double df = f1 - f0;
    const double d2 = sqrt(wms_wave(df, y)) / fabs(h0);
    double max_d = d2;
    if (max_d < d1) max_d = d1;
    double h1;
    if (max_d < 1e-15) { h1 = h0 * 1e-3; if (h1 < 1e-6) h1 = 1e-6; }
    else h1 = rpow(0.01 / max_d, 1.0 / (1.0 + 1.0), det);
    h = 100.0 * h0;
    if (h > h1) h = h1;
    if (is_neg_h) h = -h;
  }

  // ------------------------------------------------------------ Bdf::_new
  int order = 1;
  double D[kNC], Dt[kNC];
#pragma unroll
  for (int j = 0; j < kNC; ++j) { D[j] = 0.0; Dt[j] = 0.0; }
  D[0] = y; D[1] = f0 * h;
  double opc = h * C.alpha[1];
  // ---- forward sensitivities: new_with_sensitivities_and_consistent (state.rs:1032-1083), new_augmented (bdf.rs:384-432): sdiff_j[:, 0] = s_j, sdiff_j[:, 1] = h ds_j
  static_assert(!SENS || !kWmHasMass, "device-resident forward sensitivities: ODE models without a mass matrix");
  constexpr int SP = SENS ? kWmMaxSensParams : 1, SC = SENS ? kNC : 1;
  double S[SP][SC], s_cur[SP], s_delta[SP];
  double s_c = 0.0;  // BdfCallable::c of the sensitivity operator: 0 until the first _update_step_size (op/bdf.rs:61)
  const int nsp = SENS ? C.np : 0;
  auto X2f = [&](int64_t k) { return xs2[k]; };
  // (1/n) sum_i (v_i / (|w_i| sens_rtol + sens_atol))^2, summed in index order
  auto wms_sens = [&](double v_mine, double w_mine) __attribute__((always_inline)) -> double {
    const double term = rowlive ? v_mine / (fabs(w_mine) * C.sens_rtol + C.sens_atol) : 0.0;
    return seq_sum<NP>(term * term, n) / (double)n;
  };
  if constexpr (SENS) {
    for (int j = 0; j < nsp; ++j) {
      auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
      __syncthreads();
      xs[ln] = y;
      __syncthreads();
      const double s0 = rowlive ? wm_sens_component(t, (int64_t)ln, Xf, Ej, Pf, true) : 0.0;
      const double dfdp = rowlive ? wm_sens_component(t, (int64_t)ln, Xf, Ej, Pf, false) : 0.0;  // SensRhs::update_state(y0, t0): column j of df/dp
      xs2[ln] = s0;
      __syncthreads();
      const double jm = rowlive ? wm_component(model, (int64_t)n, t, (int64_t)ln, Xf, X2f, Pf, true) : 0.0;  // SensRhs::call_inplace: J(y0) s_j + (df/dp)_j
      const double ds = jm + dfdp;
#pragma unroll
      for (int k = 0; k < kNC; ++k) S[j][k] = 0.0;
      S[j][0] = s0; S[j][1] = ds * h;
      s_cur[j] = s0; s_delta[j] = 0.0;
    }
  }
  bool jac_stale = true;
  // The factorisation is by far the largest piece of code of this kernel: every request for a new linearisation only records what the reference
  // would have used (the value of c at that moment; state and time do not change before the next Newton solve) and the one inlined copy of
  // reset_jacobian runs at the top of the next solve attempt.
  bool reset_pending = true;
  double c_reset = opc;
  int n_setups = 0, n_steps = 0, n_err_fails = 0, n_newton = 0, n_nl_fails = 0;
  // reset_jacobian: J(x, t) row by row into LDS when stale, A = J * (-c) + I, LU in registers
  auto reset_jacobian = [&](double x_mine, double tt) __attribute__((always_inline)) {
    if (jac_stale) {
      __syncthreads();
      xs[ln] = x_mine;
      __syncthreads();
      for (int j = 0; j < n; ++j) {
        auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
        sJ[j * 64 + ln] = rowlive ? wm_component(model, (int64_t)n, tt, (int64_t)ln, Xf, Ej, Pf, true) : 0.0;
        if constexpr (kWmHasMass) sM[j * 64 + ln] = rowlive ? wm_mass_component(tt, (int64_t)ln, Ej, Pf) : 0.0;  // the mass matrix is evaluated with the Jacobian (op/bdf.rs:138-160)
      }
      jac_stale = false;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      double m = j == ln ? 1.0 : 0.0;
      if constexpr (kWmHasMass) m = (rowlive && j < n) ? sM[j * 64 + ln] : 0.0;
      a[j] = (rowlive && j < n) ? sJ[j * 64 + ln] * (-c_reset) + m : 0.0;
    }
    factor_rows();
  };
  n_setups = 1;
  // RootFinder::init
  double g0[2] = {1.0, 1.0};
  double rf_t0 = t;
  auto root_of = [&](double x_mine, double tt, double (&g)[2]) __attribute__((always_inline)) {
    __syncthreads();
    xs[ln] = x_mine;
    __syncthreads();
    g[0] = 1.0; g[1] = 1.0;  // unused slots: never zero, never a sign change
    double gg[2] = {0.0, 0.0};
    const int nr = wm_root_values(model, (int64_t)n, tt, Xf, Pf, gg);
    if (nr > 0) g[0] = gg[0];
    if (nr > 1) g[1] = gg[1];
  };
  if (C.nroots > 0) root_of(y, t, g0);
  double t_root = 0.0;
  int root_idx = -1;
  int steps_since_jac = 0, steps_since_rhs_jac = 0;
  double h_at_last_jac = 1.0;
  double eta = C.r.eta_reset;
  int n_equal_steps = 0;
  bool has_prev_err = false;
  double prev_err = 0.0;
  double yp = 0.0, psi = 0.0;
  double t_predict = t;

  auto update_step_size = [&](double factor, double& new_h_out) __attribute__((always_inline)) -> bool {
    const double new_h = factor * h;
    n_equal_steps = 0;
    double R[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      R[j][0] = 1.0;
#pragma unroll
      for (int i = 1; i < 6; ++i) R[j][i] = (j == 0) ? 0.0 : R[j][i - 1] * ((double)i - 1.0 - factor * (double)j) / (double)i;
    }
    const double* U = C.u[order - 1];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (j <= order) {
        double ru[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double acc = R[0][k] * U[j * 6 + 0];
#pragma unroll
          for (int m = 1; m < 6; ++m) if (m <= order) acc = R[m][k] * U[j * 6 + m] + acc;
          ru[k] = acc;
        }
        double acc = D[0] * ru[0];
#pragma unroll
        for (int k = 1; k < 6; ++k) if (k <= order) acc = D[k] * ru[k] + acc;
        Dt[j] = acc;
      }
    }
#pragma unroll
    for (int j = 0; j < kNC; ++j) { const double tmp = D[j]; D[j] = Dt[j]; Dt[j] = tmp; }
    if constexpr (SENS) {
      // bdf.rs:546-548: every sdiff goes through the SAME scratch matrix as the states' differences — diff_tmp[:, 0..=order] = sdiff_j[:, 0..=order] (R U),
      // swap(sdiff_j, diff_tmp) — so the columns behind `order` are handed down the chain (states -> s_0 -> s_1 -> ... -> the states at the next change)
      for (int q = 0; q < nsp; ++q) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if (j <= order) {
            double ru[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              double acc = R[0][k] * U[j * 6 + 0];
#pragma unroll
              for (int m = 1; m < 6; ++m) if (m <= order) acc = R[m][k] * U[j * 6 + m] + acc;
              ru[k] = acc;
            }
            double acc = S[q][0] * ru[0];
#pragma unroll
            for (int k = 1; k < 6; ++k) if (k <= order) acc = S[q][k] * ru[k] + acc;
            Dt[j] = acc;
          }
        }
#pragma unroll
        for (int j = 0; j < kNC; ++j) { const double tmp = S[q][j]; S[q][j] = Dt[j]; Dt[j] = tmp; }
      }
      s_c = new_h * C.alpha[order];  // s_op.set_c (bdf.rs:551-553)
    }
    opc = new_h * C.alpha[order];
    h = new_h;
    eta = C.r.eta_reset_ts;
    new_h_out = new_h;
    return fabs(h) < o.min_timestep;
  };
  auto predict_forward = [&]() __attribute__((always_inline)) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) if (j <= order) s = s + D[j];
    double q = C.gamma[1] * D[1];
#pragma unroll
    for (int j = 2; j < 6; ++j) if (j <= order) q = C.gamma[j] * D[j] + 1.0 * q;
    q = q * C.alpha[order];
    q = q - s;
    yp = s;
    psi = q;
    t_predict = t + h;
  };
  auto jacobian_updates = [&](double c, JState st) __attribute__((always_inline)) {
    bool check_rhs = false, check_jac = true;
    const double rel = fabs(c / h_at_last_jac - 1.0);
    switch (st) {
      case JState::StepSuccess:
        check_rhs = steps_since_rhs_jac >= o.update_rhs_jacobian_after_steps;
        check_jac = steps_since_jac >= o.update_jacobian_after_steps || rel > o.threshold_to_update_jacobian;
        break;
      case JState::FirstConvergenceFail: check_rhs = rel < o.threshold_to_update_rhs_jacobian; break;
      case JState::SecondConvergenceFail: check_rhs = steps_since_rhs_jac > 0; break;
      case JState::ErrorTestFail: check_rhs = false; break;
    }
    if (check_rhs) {
      jac_stale = true;
      reset_pending = true; c_reset = opc;
      steps_since_rhs_jac = 0; steps_since_jac = 0; h_at_last_jac = c;
      eta = C.r.eta_reset;
      n_setups++;
    } else if (check_jac) {
      reset_pending = true; c_reset = opc;
      steps_since_jac = 0; h_at_last_jac = c;
      eta = C.r.eta_reset;
      n_setups++;
    }
  };
  bool has_tstop = true;
  const double tstop = t_eval[C.r.n_eval - 1];
  auto handle_tstop = [&]() __attribute__((always_inline)) -> int {
    const double troundoff = 100.0 * kEps * (fabs(t) + fabs(h));
    if (fabs(t - tstop) <= troundoff) { has_tstop = false; return 1; }
    if ((h > 0.0 && tstop < t - troundoff) || (h < 0.0 && tstop > t + troundoff)) { has_tstop = false; return 2; }
    if ((h > 0.0 && t + h > tstop + troundoff) || (h < 0.0 && t + h < tstop - troundoff)) {
      const double factor = (tstop - t) / h;
      double nh;
      (void)update_step_size(factor, nh);
    }
    return 0;
  };
  auto interpolate = [&](double te) __attribute__((always_inline)) -> double {  // interpolate_from_diff, my component
    double time_factor = 1.0;
    double yv = D[0];
#pragma unroll
    for (int j = 0; j < kMaxOrder; ++j) {
      if (j < order) {
        const double jt = (double)j;
        time_factor *= (te - (t - h * jt)) / (h * (1.0 + jt));
        yv = time_factor * D[j + 1] + 1.0 * yv;
      }
    }
    return yv;
  };

  int col = 0;
  const bool steps_mode = !SENS && C.steps_cap > 0;  // every accepted step out (WaveMemberConsts::steps_cap)
  auto steps_write = [&](double tw, double yv_mine) __attribute__((always_inline)) {
    if (col < C.steps_cap) {
      if (ln == 0) C.steps_t_out[(int64_t)col * nb + b] = tw;
      if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = yv_mine;
    }
    col++;
  };
  if (steps_mode) steps_write(t, y);  // write_out before the first step (method.rs:900)
  {
    const int r = handle_tstop();
    if (r == 1) status = kRsStopTimeAtCurrentTime;
    else if (r == 2) status = kRsStopTimeBeforeCurrentTime;
  }
  long guard = 0;
  bool done = status != kRsOk;
  while (!done) {
    if (++guard > o.max_steps) { status = kRsMaxStepsExceeded; break; }
    double safety = 0.0, error_norm = 0.0;
    const int old_err_fails = n_err_fails;
    bool convergence_fail = false;
    double x = 0.0;
    int niter = 0;
    predict_forward();
    while (true) {
      if (reset_pending) { reset_jacobian(y, t); reset_pending = false; }
      x = yp;
      niter = 0;
      bool has_old = false;
      double old_norm = 0.0;
      bool solved = false;
      for (int it = 0; it < o.max_nonlinear_solver_iterations; ++it) {
        const double f = rhs_of(x, t_predict);
        const double tmpv = x + psi;
        double delta;
        if constexpr (kWmHasMass) {  // F(y) = M (y - y0 + psi) - c f(y): M's row times the published vector, then + (-c) f (mass_gemv with beta = -c)
          xs2[ln] = tmpv;
          __syncthreads();
          auto X2f = [&](int64_t k) { return xs2[k]; };
          delta = rowlive ? wm_mass_component(t_predict, (int64_t)ln, X2f, Pf) + (-opc) * f : 0.0;
          __syncthreads();
        } else {
          delta = 1.0 * tmpv + (-opc) * f;  // F(y) = (y - y0 + psi) - c f(y)
        }
        const bool lu_ok = wave_lu_solve_rows<NP>(a, n, rowlive, pos, delta);
        if (!lu_ok) break;
        delta = __shfl(delta, myinv, 64);  // unknown i to lane i
        x = x - delta;
        const double norm = sqrt(wms_wave(delta, yp));
        niter += 1;
        bool diverged = false;
        if (has_old) {
          const double rate = niter == 2 ? norm / old_norm : rpow(norm / old_norm, 1.0 / (double)(niter - 1), det);
          if (rate > 0.9) diverged = true;
          else if (powi_rt(rate, o.max_nonlinear_solver_iterations - niter) / (1.0 - rate) * norm > o.nonlinear_solver_tolerance) diverged = true;
          else eta = rate / (1.0 - rate);
        } else {
          const double min_eta = 1e4 * kEps;
          if (eta < min_eta) eta = min_eta;
          eta = rpow(eta, 0.8, det);
        }
        const bool converged = !diverged && eta * norm < o.nonlinear_solver_tolerance;
        if (niter == 1) { has_old = true; old_norm = norm; }
        if (diverged) break;
        if (converged) { solved = true; break; }
      }
      n_newton += niter;
      if constexpr (SENS) {
        // sensitivity_solve (bdf.rs:934-989): SensRhs linearised about (y_predict, t_new); per parameter the predictor / psi of its difference array and a
        // Newton solve of F(s) = (s - s0 + psi) - c_s (J s + (df/dp)_j) with the factors of the state equations and the SHARED Convergence (eta carries over,
        // state tolerances); a failure is a failure of the step's nonlinear solve
        if (solved) {
          for (int j = 0; j < nsp && solved; ++j) {
            auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
            __syncthreads();
            xs[ln] = yp;
            __syncthreads();
            const double dfdp = rowlive ? wm_sens_component(t_predict, (int64_t)ln, Xf, Ej, Pf, false) : 0.0;
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k <= order) sacc = sacc + S[j][k];
            double q = C.gamma[1] * S[j][1];
#pragma unroll
            for (int k = 2; k < 6; ++k) if (k <= order) q = C.gamma[k] * S[j][k] + 1.0 * q;
            q = q * C.alpha[order];
            q = q - sacc;
            const double sp = sacc, spsi = q;
            double xsv = sacc;
            int sn = 0;
            bool s_has_old = false, s_solved = false;
            double s_old_norm = 0.0;
            for (int it = 0; it < o.max_nonlinear_solver_iterations; ++it) {
              __syncthreads();
              xs2[ln] = xsv;
              __syncthreads();
              const double jm = rowlive ? wm_component(model, (int64_t)n, t_predict, (int64_t)ln, Xf, X2f, Pf, true) : 0.0;
              const double fr = jm + dfdp;
              double delta = 1.0 * (xsv + spsi) + (-s_c) * fr;
              const bool lu_ok = wave_lu_solve_rows<NP>(a, n, rowlive, pos, delta);
              if (!lu_ok) break;
              delta = __shfl(delta, myinv, 64);
              xsv = xsv - delta;
              const double norm = sqrt(wms_wave(delta, sp));
              sn += 1;
              bool diverged = false;
              if (s_has_old) {
                const double rate = sn == 2 ? norm / s_old_norm : rpow(norm / s_old_norm, 1.0 / (double)(sn - 1), det);
                if (rate > 0.9) diverged = true;
                else if (powi_rt(rate, o.max_nonlinear_solver_iterations - sn) / (1.0 - rate) * norm > o.nonlinear_solver_tolerance) diverged = true;
                else eta = rate / (1.0 - rate);
              } else {
                const double min_eta = 1e4 * kEps;
                if (eta < min_eta) eta = min_eta;
                eta = rpow(eta, 0.8, det);
              }
              const bool converged = !diverged && eta * norm < o.nonlinear_solver_tolerance;
              if (sn == 1) { s_has_old = true; s_old_norm = norm; }
              if (diverged) break;
              if (converged) { s_solved = true; break; }
            }
            niter = sn;  // Convergence::niter is the last solve's: the safety factor below reads it
            if (!s_solved) { solved = false; break; }  // `?` before the iteration count is added
            n_newton += sn;
            s_cur[j] = xsv; s_delta[j] = xsv - sp;
          }
        }
      }
      if (!solved) {
        n_nl_fails += 1;
        if (n_nl_fails > o.max_nonlinear_solver_failures) { status = kRsTooManyNonlinearSolverFailures; break; }
        has_prev_err = false;
        if (convergence_fail) {
          double new_h;
          if (update_step_size(0.3, new_h)) { status = kRsStepSizeTooSmall; break; }
          jacobian_updates(new_h * C.alpha[order], JState::SecondConvergenceFail);
          predict_forward();
        } else {
          jacobian_updates(h * C.alpha[order], JState::FirstConvergenceFail);
          convergence_fail = true;
        }
        continue;
      }
      const double ydelta = x - yp;
      error_norm = fmax(0.0, wms_wave(ydelta, y) * C.ec2[order - 1]);
      if constexpr (SENS) {
        if (C.sens_error_control)  // bdf.rs:844-858 — error_const2[order], not [order - 1]
          for (int j = 0; j < nsp; ++j) error_norm = fmax(error_norm, wms_sens(s_delta[j], s_cur[j]) * C.ec2[order]);
      }
      const double maxiter = (double)o.max_nonlinear_solver_iterations;
      safety = 0.9 * (2.0 * maxiter + 1.0) / (2.0 * maxiter + (double)niter);
      if (error_norm <= 1.0) {
        double dk1 = 0.0;
#pragma unroll
        for (int j = 2; j < 7; ++j) if (j == order + 1) dk1 = D[j];
        const double dk2 = ydelta - dk1;
#pragma unroll
        for (int j = 2; j < kNC; ++j) { if (j == order + 2) D[j] = dk2; if (j == order + 1) D[j] = ydelta; }
        double upper = ydelta;
#pragma unroll
        for (int j = 5; j >= 0; --j) if (j <= order) { const double v = D[j] + 1.0 * upper; D[j] = v; upper = v; }
        if constexpr (SENS) {  // update_differences_and_integrate_out (bdf.rs:628-643): _update_diff on every sensitivity difference array
          for (int q = 0; q < nsp; ++q) {
            const double sd = s_delta[q];
            double sk1 = 0.0;
#pragma unroll
            for (int j = 2; j < 7; ++j) if (j == order + 1) sk1 = S[q][j];
            const double sk2 = sd - sk1;
#pragma unroll
            for (int j = 2; j < kNC; ++j) { if (j == order + 2) S[q][j] = sk2; if (j == order + 1) S[q][j] = sd; }
            double up = sd;
#pragma unroll
            for (int j = 5; j >= 0; --j) if (j <= order) { const double v = S[q][j] + 1.0 * up; S[q][j] = v; up = v; }
          }
        }
        y = yp;
        t = t_predict;
        break;
      }
      double factor = safety * pi_controller_raw(error_norm, has_prev_err, prev_err, o.pi_control_integral, o.pi_control_proportional, order + 1, det);
      has_prev_err = false;
      if (factor < o.min_timestep_shrink) factor = o.min_timestep_shrink;
      double new_h;
      if (update_step_size(factor, new_h)) { status = kRsStepSizeTooSmall; break; }
      jacobian_updates(new_h * C.alpha[order], JState::ErrorTestFail);
      predict_forward();
      n_err_fails += 1;
      if (n_err_fails - old_err_fails >= o.max_error_test_failures) { status = kRsTooManyErrorTestFailures; break; }
    }
    if (status != kRsOk) break;
    n_steps += 1;
    steps_since_jac += 1; steps_since_rhs_jac += 1;
    prev_err = error_norm; has_prev_err = true;
    n_equal_steps += 1;
    if (n_equal_steps > order) {
      double vm = 0.0, vp = 0.0;
#pragma unroll
      for (int j = 1; j < kNC; ++j) { if (j == order) vm = D[j]; if (j == order + 2) vp = D[j]; }
      const double inf = __builtin_huge_val();
      double error_m_norm = order > 1 ? wms_wave(vm, y) * C.ec2[order - 1] : inf;
      double error_p_norm = order < kMaxOrder ? wms_wave(vp, y) * C.ec2[order + 1] : inf;
      if constexpr (SENS) {  // predict_error_control with the augmented system (bdf.rs:871-932): the `error_norm.max(err)` chain from zero, then the sensitivities' terms
        if (order > 1) error_m_norm = fmax(0.0, error_m_norm);
        if (order < kMaxOrder) error_p_norm = fmax(0.0, error_p_norm);
        if (C.sens_error_control)
          for (int q = 0; q < nsp; ++q) {
            double cm = 0.0, cp = 0.0;
#pragma unroll
            for (int j = 1; j < kNC; ++j) { if (j == order) cm = S[q][j]; if (j == order + 2) cp = S[q][j]; }
            if (order > 1) error_m_norm = fmax(error_m_norm, wms_sens(cm, s_cur[q]) * C.ec2[order - 1]);
            if (order < kMaxOrder) error_p_norm = fmax(error_p_norm, wms_sens(cp, s_cur[q]) * C.ec2[order + 1]);
          }
      }
      const double pi_i = o.pi_control_integral, pi_p = o.pi_control_proportional;
      const double f0c = pi_controller_raw(error_m_norm, has_prev_err, prev_err, pi_i, pi_p, order, det);
      const double f1c = pi_controller_raw(error_norm, has_prev_err, prev_err, pi_i, pi_p, order + 1, det);
      const double f2c = pi_controller_raw(error_p_norm, has_prev_err, prev_err, pi_i, pi_p, order + 2, det);
      int max_index = 0;
      double fmaxv = f0c;
      if (f1c >= fmaxv) { max_index = 1; fmaxv = f1c; }
      if (f2c >= fmaxv) { max_index = 2; fmaxv = f2c; }
      const int new_order = max_index == 0 ? order - 1 : (max_index == 1 ? order : order + 1);
      order = new_order;
      double factor = safety * fmaxv;
      if (factor > o.max_timestep_growth) factor = o.max_timestep_growth;
      if (factor < o.min_timestep_shrink) factor = o.min_timestep_shrink;
      if (factor >= o.min_timestep_growth || factor <= o.max_timestep_shrink || max_index == 0 || max_index == 2) {
        double new_h;
        if (update_step_size(factor, new_h)) { status = kRsStepSizeTooSmall; break; }
        jacobian_updates(new_h * C.alpha[new_order], JState::StepSuccess);
      }
    }
    int reason = 0;  // 0 internal, 1 tstop, 3 root
    if (C.nroots > 0) {
      // RootFinder::check_root (root.rs:91-222) on wavefront-uniform root values
      double g1[2], gmid[2];
      root_of(y, t, g1);
      bool found;
      double frac;
      int imax;
      root_finding_lane<2>(g0, g1, found, frac, imax);
      if (imax < 0) {
        g0[0] = g1[0]; g0[1] = g1[1];
        rf_t0 = t;
        if (found) { t_root = t; root_idx = fabs(g0[1]) < fabs(g0[0]) && C.nroots > 1 ? 1 : 0; reason = 3; }
      } else {
        double alpha = 1.0;
        bool sc0 = false, sc1 = true;
        int itr = 0;
        double t1 = t, t0l = rf_t0;
        const double tol = 100.0 * kEps * (fabs(t1) + fabs(t1 - t0l));
        bool early = false;
        while (fabs(t1 - t0l) > tol) {
          const double g1v = imax == 0 ? g1[0] : g1[1], g0v = imax == 0 ? g0[0] : g0[1];
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
          root_of(interpolate(t_mid), t_mid, gmid);
          bool f2;
          double fr2;
          int i2;
          root_finding_lane<2>(g0, gmid, f2, fr2, i2);
          const bool lower = i2 >= 0;
          if (lower) {
            t1 = t_mid; imax = i2;
            g1[0] = gmid[0]; g1[1] = gmid[1];
          } else if (f2) {
            root_of(y, t, g0);
            t_root = t_mid; root_idx = imax; early = true;
            break;
          } else {
            t0l = t_mid;
            g0[0] = gmid[0]; g0[1] = gmid[1];
          }
          if ((itr & 1) == 0) sc0 = lower; else sc1 = lower;
          if (itr >= 2) alpha = (sc0 != sc1) ? 1.0 : (sc0 ? 0.5 * alpha : 2.0 * alpha);
          itr += 1;
        }
        if (!early) { root_of(y, t, g0); t_root = t1; root_idx = imax; }
        reason = 3;
      }
    }
    if (reason == 0 && has_tstop) reason = handle_tstop();
    if (reason == 2) reason = 0;
    const double upto = reason == 3 ? t_root : t;
    if (steps_mode) {  // InternalTimestep / TstopReached -> write_out (method.rs:907-921): state.y; a root is written below, at the root
      if (reason != 3) steps_write(t, y);
    } else
    while (col < C.r.n_eval && t_eval[col] <= upto) {
      const double yv = interpolate(t_eval[col]);
      if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = yv;
      if constexpr (SENS) {  // interpolate_sens (bdf.rs:1162-1215): the same polynomial on every sensitivity difference array
        const double te = t_eval[col];
        for (int q = 0; q < nsp; ++q) {
          double time_factor = 1.0;
          double sv = S[q][0];
#pragma unroll
          for (int j = 0; j < kMaxOrder; ++j) {
            if (j < order) {
              const double jt = (double)j;
              time_factor *= (te - (t - h * jt)) / (h * (1.0 + jt));
              sv = time_factor * S[q][j + 1] + 1.0 * sv;
            }
          }
          if (rowlive) C.sens_out[(((int64_t)col * nsp + q) * n + ln) * nb + b] = sv;
        }
      }
      col++;
    }
    if constexpr (kWmResets) {
      if (reason == 3) {
        // A reset operator is configured, as in k_bdf_adaptive / k_bdf_lane_banded: the state goes back to the root (state_mut_back, bdf.rs:1232-1262), y <- reset(y, t),
        // dy <- f(y, t) (bdf.rs:1017-1020 over state.rs:279-306), the stop time is armed again on the OLD differences and order, and the next step restarts from the
        // modified state at first order (bdf.rs:1290-1318).  The save points up to the root were written above.
        const double yb = interpolate(t_root);
        t = t_root;
        __syncthreads();
        xs[ln] = yb;
        __syncthreads();
        y = rowlive ? wm_reset_component(t, (int64_t)ln, Xf, Pf) : 0.0;
        const double dyr = rhs_of(y, t);
        if (steps_mode) steps_write(t, y);  // method.rs:931-932: the reset state at the root time
        if (t < tstop) {
          has_tstop = true;  // set_stop_time (bdf.rs:1591-1600)
          { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          root_of(y, t, g0);  // RootFinder::init
          rf_t0 = t;
          n_equal_steps = 0;
          order = 1;  // initialise_diff_to_first_order: columns 0 and 1 only, the others keep what they hold
          D[0] = y; D[1] = dyr * h;
          opc = h * C.alpha[1];
          jacobian_updates(h * C.alpha[1], JState::StepSuccess);
          has_prev_err = false;
          if (has_tstop) { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          reason = 0;
        } else {
          done = true;  // the event sits on the last save point: TstopReached
          reason = 0;
        }
      }
    }
    if (reason == 3 && steps_mode) {  // method.rs:922-947 without a reset: state_mut_back(t_root), write_out, RootFound
      const double yv = interpolate(t_root);
      steps_write(t_root, yv);
      done = true;
    } else
    if (reason == 3) {
      if (col < C.r.n_eval) {
        const double yv = interpolate(t_root);
        if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = yv;
        col++;
      }
      done = true;
    }
    if (reason == 1) done = true;
  }
  const int ncols = col;
  if (!steps_mode)
  for (; col < C.r.n_eval; ++col) {
    if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = __builtin_nan("");
    if constexpr (SENS)
      for (int q = 0; q < nsp; ++q)
        if (rowlive) C.sens_out[(((int64_t)col * nsp + q) * n + ln) * nb + b] = __builtin_nan("");
  }
  if (ln == 0) {
    if (ncols_out != nullptr) ncols_out[b] = ncols;
    if (t_root_out != nullptr) t_root_out[b] = root_idx >= 0 ? t_root : __builtin_nan("");
    if (root_idx_out != nullptr) root_idx_out[b] = root_idx;
    if (status_out != nullptr) status_out[b] = status;
    if (stats_out != nullptr) {
      stats_out[0 * nb + b] = n_steps;
      stats_out[1 * nb + b] = n_newton;
      stats_out[2 * nb + b] = n_setups;
      stats_out[3 * nb + b] = n_err_fails;
      stats_out[4 * nb + b] = n_nl_fails;
    }
    atomicAdd(&totals[0], (unsigned long long)n_steps);
    atomicAdd(&totals[1], (unsigned long long)n_newton);
    atomicAdd(&totals[2], (unsigned long long)n_setups);
    atomicAdd(&totals[3], (unsigned long long)n_err_fails);
    atomicAdd(&totals[4], (unsigned long long)n_nl_fails);
    if (status != kRsOk) atomicAdd(&totals[5], 1ull);
  }
}

}  // namespace dsh
