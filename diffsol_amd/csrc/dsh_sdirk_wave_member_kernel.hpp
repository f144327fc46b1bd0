// One wavefront per ensemble member: the SDIRK integrators (TR-BDF2, ESDIRK34) for run-time-sized models with n <= 64 (n <= 48 with a mass matrix) — the
// wavefront-distributed form of k_sdirk_resident (dsh_sdirk_kernel.hpp: Sdirk::step sdirk.rs:409-543 over Rk runge_kutta.rs:466-960), built from the
// pieces of k_bdf_wave_member (dsh_wave_member_kernel.hpp): lane i holds component i of the state, of every stage increment and its row of the LU
// factors; the state is published through LDS for the model's component functions; norms are summed in index order.  Launch code: dsh_wave_member.hip.
#pragma once
#include "dsh_team_member_kernel.hpp"
#include "dsh_sdirk_kernel.hpp"

namespace dsh {

struct WaveSdirkConsts {
  SdirkConsts T;
  int model, n, np, nroots;
};

// SENS: forward sensitivities of every parameter alongside (run-time-compiled ODE models without a mass matrix and without root functions, at most kWmMaxSensParams
// parameters): k_sdirk_resident<.., SENS>'s sensitivity code (runge_kutta.rs:196-232, :691-748, :812-822, :1237-1330) with a component per lane.
// TW > 0: ONE WORKGROUP of TW wavefronts per member (64 < n <= 140, identity mass) — the same integrator on k_bdf_team_member's pieces (dsh_team_member_kernel.hpp):
// thread i holds component i, the factors of M - (c h) f' live in LDS (team_lu_factor / team_lu_solve), the cached Jacobian in global scratch (jac_scratch, n^2 doubles
// per member), norms are summed in index order from LDS.  NP is then only the size of an unused register row.
template <int NP, int S, bool SENS = false, int TW = 0>
__global__ __launch_bounds__(TW > 0 ? 64 * TW : 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sdirk_wave_member(int64_t nb, const double* __restrict__ p_g, const double* __restrict__ atol_g, int atol_broadcast,
                                                         const WaveSdirkConsts* __restrict__ Cp, const double* __restrict__ t_eval, double* __restrict__ y_out,
                                                         int32_t* __restrict__ stats_out, int32_t* __restrict__ status_out, double* __restrict__ t_root_out,
                                                         int32_t* __restrict__ root_idx_out, int32_t* __restrict__ ncols_out, unsigned long long* __restrict__ totals,
                                                         double* __restrict__ jac_scratch) {
  constexpr bool TEAM = TW > 0;
  constexpr int TT = TEAM ? 64 * TW : 64;  // threads = rows
  static_assert(!TEAM || !kWmHasMass, "the workgroup-per-member form takes identity-mass models");
  extern __shared__ double lds[];  // xs[64] | ps[64] | sJ[n][64] | with a mass matrix: sM[n][64] | xs2[64];  TEAM: xs[TT] | xs2[TT] | ps[TT] | cand[2 TW] | perm[TT] (int) | A[n][P]
  double* xs = lds;
  double* ps = TEAM ? lds + 2 * TT : lds + 64;
  double* sJ = TEAM ? jac_scratch + (size_t)blockIdx.x * team_scratch_doubles(Cp->n, TEAM ? TW : 2) : lds + 128;  // TEAM: entry (ln, j) at j * n + ln
  double* sM = sJ + (size_t)Cp->n * 64;
  double* xs2 = TEAM ? lds + TT : (kWmHasMass ? sM + (size_t)Cp->n * 64 : sM);  // (SENS: the direction of J v; TEAM: also the LU solve's exchange and the norms' terms)
  double* const cand = lds + 3 * TT;
  int* const perm = reinterpret_cast<int*>(lds + 3 * TT + 2 * (TEAM ? TW : 1));
  double* const A = (TEAM && team_global_factors(TW)) ? sJ + (size_t)Cp->n * Cp->n : lds + 3 * TT + 2 * (TEAM ? TW : 1) + TT / 2;  // n > 140: the factors in global scratch
  constexpr int P = team_pitch_w(TEAM ? TW : 2);
  bool lu_singular = false;
  (void)cand; (void)perm; (void)A; (void)P; (void)lu_singular;
  const SdirkConsts& T = Cp->T;
  const ResidentConsts& C = T.r;
  const dsh_adaptive_options& o = C.o;
  const bool det = o.deterministic_pow != 0;
  const int n = Cp->n, model = Cp->model;
  const int64_t b = blockIdx.x;
  const int ln = threadIdx.x;
  const bool rowlive = ln < n;
  const double rtol = C.rtol;
  const double atol = rowlive ? (atol_broadcast ? atol_g[ln] : atol_g[(int64_t)ln * nb + b]) : 1.0;
  if (ln < Cp->np) ps[ln] = p_g[(int64_t)ln * nb + b];
  __syncthreads();
  auto Pf = [&](int64_t k) { return ps[k]; };
  auto Xf = [&](int64_t k) { return xs[k]; };
  auto V0 = [&](int64_t) { return 0.0; };
  auto rhs_of = [&](double x_mine, double tt) __attribute__((always_inline)) -> double {
    __syncthreads();
    xs[ln] = x_mine;
    __syncthreads();
    return rowlive ? wm_component(model, (int64_t)n, tt, (int64_t)ln, Xf, V0, Pf, false) : 0.0;
  };
  // (1/n) sum of squares in index order; TEAM: every thread the same sequential sum over the terms in LDS
  auto sum_terms = [&](double term) __attribute__((always_inline)) -> double {
    if constexpr (TEAM) {
      __syncthreads();
      xs2[ln] = term * term;
      __syncthreads();
      return team_seq_sum(xs2, n) / (double)n;
    } else {
      return seq_sum<NP>(term * term, n) / (double)n;
    }
  };
  auto wms_wave = [&](double v_mine, double w_mine) __attribute__((always_inline)) -> double {
    const double term = rowlive ? v_mine / (fabs(w_mine) * rtol + atol) : 0.0;
    return sum_terms(term);
  };

  // ------------------------------------------------------------ RkState::new_and_consistent(problem, tableau.order()): identity mass, set_step_size
  int32_t status = kRsOk;
  double t = C.t0, h;
  double y = rowlive ? wm_init_value(model, (int64_t)n, (int64_t)ln, t, Pf) : 0.0;
  double dy = rhs_of(y, t);
  double a[NP];  // my row of the LU factors (InitOp's during the consistent initialisation, of M - (c h) f' afterwards)
  int pos = ln, myinv = ln;
  auto lu_solve = [&](double& v) __attribute__((always_inline)) -> bool {
    if constexpr (TEAM) {
      __syncthreads();  // xs2 may still be read as the direction of a Jacobian product
      return team_lu_solve<TW>(A, P, n, ln, rowlive, perm, xs2, lu_singular, v);  // unknown i comes back to thread i
    } else {
      const bool ok = wave_lu_solve_rows<NP>(a, n, rowlive, pos, v);
      v = __shfl(v, myinv, 64);
      return ok;
    }
  };
  if constexpr (kWmHasMass) {  // DAEs: consistent initial state (shared with k_bdf_wave_member)
    auto factor_init = [&]() __attribute__((always_inline)) {
      bool sing = false;
      int mypiv;
      wave_lu_factor_rows<NP, 64>(a, n, true, rowlive, ln, 0, pos, mypiv, sing);
      for (int k = 0; k < n; ++k) {
        const int holder = __ffsll((unsigned long long)__ballot(rowlive && pos == k)) - 1;
        if (ln == k) myinv = holder;
      }
    };
    auto comp_of = [&]() __attribute__((always_inline)) { return rowlive ? wm_component(model, (int64_t)n, t, (int64_t)ln, Xf, V0, Pf, false) : 0.0; };
    auto mass_e = [&](int j) __attribute__((always_inline)) { auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; }; return rowlive ? wm_mass_component(t, (int64_t)ln, Ej, Pf) : 0.0; };
    auto jac_e = [&](int j) __attribute__((always_inline)) { auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; }; return rowlive ? wm_component(model, (int64_t)n, t, (int64_t)ln, Xf, Ej, Pf, true) : 0.0; };
    if (!wm_set_consistent<NP>(n, ln, rowlive, xs, xs2, sJ, sM, a, C, comp_of, mass_e, jac_e, factor_init, lu_solve, wms_wave, y, dy)) status = kRsInitialConditionDidNotConverge;
  }
  {
    const bool is_neg_h = C.h0 < 0.0;
    const double d0 = sqrt(wms_wave(y, y)), d1 = sqrt(wms_wave(dy, y));
    const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * (d0 / d1);
    const double hh = is_neg_h ? -h0 : h0;
    const double y1 = dy * hh + y;
    const double f1 = rhs_of(y1, is_neg_h ? t - h0 : t + h0);
    const double df = f1 - dy;
    const double d2 = sqrt(wms_wave(df, y)) / fabs(h0);
    double max_d = d2;
    if (max_d < d1) max_d = d1;
    double h1;
    if (max_d < 1e-15) { h1 = h0 * 1e-3; if (h1 < 1e-6) h1 = 1e-6; }
    else h1 = rpow(0.01 / max_d, 1.0 / (1.0 + (double)T.order), det);
    h = 100.0 * h0;
    if (h > h1) h = h1;
    if (is_neg_h) h = -h;
  }

  // ------------------------------------------------------------ Rk::_new (runge_kutta.rs:110-190) + Sdirk::_new (sdirk.rs:172-215)
  double diff[S];
#pragma unroll
  for (int j = 0; j < S; ++j) diff[j] = 0.0;
  double old_y = y, old_dy = dy, old_t = t;
  (void)old_dy;
  // ---- forward sensitivities: RkState::new_with_sensitivities_and_consistent (state.rs:1032-1083), Rk::new_augmented (runge_kutta.rs:196-232)
  static_assert(!SENS || !kWmHasMass, "device-resident forward sensitivities: ODE models without a mass matrix");
  constexpr int SP = SENS ? kWmMaxSensParams : 1, SS = SENS ? S : 1;
  double sv[SP], dsv[SP], old_sv[SP], old_dsv[SP], sdiff[SP][SS];
  const int nsp = SENS ? Cp->np : 0;
  auto X2f = [&](int64_t q) { return xs2[q]; };
  const double s_atol = T.sens_atol[0];
  auto wms_sens = [&](double v_mine, double w_mine) __attribute__((always_inline)) -> double {
    const double term = rowlive ? v_mine / (fabs(w_mine) * T.sens_rtol + s_atol) : 0.0;
    return sum_terms(term);
  };
  if constexpr (SENS) {
    for (int j = 0; j < nsp; ++j) {
      auto Ej = [&](int64_t q) { return q == j ? 1.0 : 0.0; };
      __syncthreads();
      xs[ln] = y;
      __syncthreads();
      const double s0 = rowlive ? wm_sens_component(t, (int64_t)ln, Xf, Ej, Pf, true) : 0.0;
      const double dfdp = rowlive ? wm_sens_component(t, (int64_t)ln, Xf, Ej, Pf, false) : 0.0;  // SensRhs::update_state(y0, t0)
      xs2[ln] = s0;
      __syncthreads();
      const double jm = rowlive ? wm_component(model, (int64_t)n, t, (int64_t)ln, Xf, X2f, Pf, true) : 0.0;  // SensRhs::call_inplace: J(y0) s_j + (df/dp)_j
      const double d0 = jm + dfdp;
      sv[j] = s0; dsv[j] = d0; old_sv[j] = s0; old_dsv[j] = d0;
#pragma unroll
      for (int m = 0; m < S; ++m) sdiff[j][m] = 0.0;
    }
  }
  double g0[2] = {1.0, 1.0};
  double rf_t0 = t;
  auto root_of = [&](double x_mine, double tt, double (&g)[2]) __attribute__((always_inline)) {
    __syncthreads();
    xs[ln] = x_mine;
    __syncthreads();
    g[0] = 1.0; g[1] = 1.0;
    double gg[2] = {0.0, 0.0};
    const int nr = wm_root_values(model, (int64_t)n, tt, Xf, Pf, gg);
    if (nr > 0) g[0] = gg[0];
    if (nr > 1) g[1] = gg[1];
  };
  if (Cp->nroots > 0) root_of(y, t, g0);
  JacUpdateState ju;
  ju.update_jacobian(h);
  ju.update_rhs_jacobian(h);
  ConvState conv;
  conv.eta = C.eta_reset;
  conv.tol = o.nonlinear_solver_tolerance;
  conv.max_iter = o.max_nonlinear_solver_iterations;
  conv.det = det;
  double op_h = h;
  const double op_c = T.gamma;
  double phi = 0.0;  // V::zeros until the first set_phi
  bool is_jacobian_set = false;
  bool has_prev_err = false;
  double prev_err = 0.0;
  int n_setups = 0, n_steps = 0, n_err_fails = 0, n_newton = 0, n_nl_fails = 0;
  // SdirkCallable::jacobian_inplace (op/sdirk.rs:266-296) + LU: I - (c h) f'(phi + c x) — REQUESTED where the reference calls reset_jacobian (recording
  // the linearisation point phi + c y and time of that moment, and the step size), EXECUTED at the top of the next Newton solve: one inlined copy of
  // the factorisation (the largest piece of code of the kernel) instead of three.  A later request supersedes an earlier one exactly as a later
  // factorisation overwrites an earlier one; a pending re-evaluation of f' survives a request that only re-factors.
  bool eval_pending = false, factor_pending = false;
  double lin_point = 0.0, lin_t = t, factor_h = h;
  auto request_reset = [&](bool stale, double x_mine, double tt) __attribute__((always_inline)) {
    if (stale) { eval_pending = true; lin_point = op_c * x_mine + 1.0 * phi; lin_t = tt; }
    factor_pending = true;
    factor_h = op_h;
    is_jacobian_set = true;
  };
  auto execute_reset = [&]() __attribute__((always_inline)) {
    if (eval_pending) {
      __syncthreads();
      xs[ln] = lin_point;
      __syncthreads();
      for (int j = 0; j < n; ++j) {
        auto Ej = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
        if constexpr (TEAM) { if (rowlive) sJ[(size_t)j * n + ln] = wm_component(model, (int64_t)n, lin_t, (int64_t)ln, Xf, Ej, Pf, true); }
        else sJ[j * 64 + ln] = rowlive ? wm_component(model, (int64_t)n, lin_t, (int64_t)ln, Xf, Ej, Pf, true) : 0.0;
        if constexpr (kWmHasMass) sM[j * 64 + ln] = rowlive ? wm_mass_component(lin_t, (int64_t)ln, Ej, Pf) : 0.0;  // the mass matrix is evaluated with the Jacobian (op/sdirk.rs:266-296)
      }
      eval_pending = false;
    }
    const double beta = -(op_c * factor_h);
    if constexpr (TEAM) {  // A = J * beta + I, my row; the factorisation in LDS
      if (rowlive)
        for (int j = 0; j < n; ++j) A[j * P + ln] = sJ[(size_t)j * n + ln] * beta + (j == ln ? 1.0 : 0.0);
      team_lu_factor<TW>(A, P, n, ln, rowlive, cand, perm, lu_singular);
      factor_pending = false;
      return;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      double m = j == ln ? 1.0 : 0.0;
      if constexpr (kWmHasMass) m = (rowlive && j < n) ? sM[j * 64 + ln] : 0.0;
      a[j] = (rowlive && j < n) ? sJ[j * 64 + ln] * beta + m : 0.0;
    }
    bool sing = false;
    int mypiv;
    wave_lu_factor_rows<NP, 64>(a, n, true, rowlive, ln, 0, pos, mypiv, sing);
    for (int k = 0; k < n; ++k) {
      const int holder = __ffsll((unsigned long long)__ballot(rowlive && pos == k)) - 1;
      if (ln == k) myinv = holder;
    }
    factor_pending = false;
  };
  // Sdirk::_jacobian_updates (sdirk.rs:260-303)
  auto jacobian_updates = [&](double hh, JState st) __attribute__((always_inline)) {
    if (ju.check_rhs_jacobian_update(hh, st, o)) {
      request_reset(true, y, t);
      ju.update_rhs_jacobian(hh);
      ju.update_jacobian(hh);
      conv.eta = C.eta_reset;
      n_setups++;
    } else if (ju.check_jacobian_update(hh, st, o)) {
      request_reset(false, y, t);
      ju.update_jacobian(hh);
      conv.eta = C.eta_reset;
      n_setups++;
    }
  };
  if constexpr (SENS) {
    // Sdirk::new_augmented ends with jacobian_updates(h, Checkpoint) (sdirk.rs:251): with sensitivities the first linearisation is made at construction, at t0 and —
    // phi still being zero — about gamma y0 (requested here, carried out at the top of the first solve like every other request)
    if (status == kRsOk) {
      request_reset(true, y, t);
      ju.update_rhs_jacobian(h);
      ju.update_jacobian(h);
      conv.eta = C.eta_reset;
      n_setups++;
    }
  }
  bool has_tstop = true;
  const double tstop = t_eval[C.n_eval - 1];
  auto handle_tstop = [&]() __attribute__((always_inline)) -> int {  // runge_kutta.rs:752-781: 0 nothing, 1 reached, 2 StopTimeBeforeCurrentTime
    const double troundoff = 100.0 * kEps * (fabs(t) + fabs(h));
    if (fabs(t - tstop) <= troundoff) return 1;
    if ((h > 0.0 && tstop < t - troundoff) || (h < 0.0 && tstop > t + troundoff)) return 2;
    if ((h > 0.0 && t + h > tstop + troundoff) || (h < 0.0 && t + h < tstop - troundoff)) {
      const double factor = (tstop - t) / h;
      h *= factor;
    }
    return 0;
  };
  auto interpolate = [&](double tt) __attribute__((always_inline)) -> double {  // interpolate_inplace (runge_kutta.rs:1080-1127) inside [old_t, t], my component
    const double dt = t - old_t;
    const double theta = dt == 0.0 ? 1.0 : (tt - old_t) / dt;
    if (T.has_beta) {
      double thetav[kMaxPoly];
      thetav[0] = theta;
#pragma unroll
      for (int q = 1; q < kMaxPoly; ++q) thetav[q] = theta * thetav[q - 1];
      double bf[S];
#pragma unroll
      for (int i = 0; i < S; ++i) {
        double acc = 1.0 * T.beta[0 * S + i] * thetav[0];
#pragma unroll
        for (int q = 1; q < kMaxPoly; ++q) if (q < T.poly_order) acc = 1.0 * T.beta[q * S + i] * thetav[q] + acc;
        bf[i] = acc;
      }
      double acc = 1.0 * diff[0] * bf[0] + 1.0 * old_y;
#pragma unroll
      for (int j = 1; j < S; ++j) acc = 1.0 * diff[j] * bf[j] + acc;
      return acc;
    }
    double r = y - old_y;  // interpolate_hermite (runge_kutta.rs:1016-1035)
    r = (1.0 * (theta - 1.0)) * diff[0] + (1.0 - 2.0 * theta) * r;
    r = (1.0 * theta) * diff[S - 1] + 1.0 * r;
    r = (1.0 - theta) * old_y + (theta * (theta - 1.0)) * r;
    r = theta * y + 1.0 * r;
    return r;
  };

  int col = 0;
  double t_root = 0.0;
  int root_idx = -1;
  const bool steps_mode = !SENS && T.steps_cap > 0;  // every accepted step out (SdirkConsts::steps_cap), as in k_sdirk_adaptive
  auto steps_write = [&](double tw, double yv_mine) __attribute__((always_inline)) {
    if (col < T.steps_cap) {
      if (ln == 0) T.steps_t_out[(int64_t)col * nb + b] = tw;
      if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = yv_mine;
    }
    col++;
  };
  if (steps_mode && status == kRsOk) steps_write(t, y);  // write_out before the first step (method.rs:900)
  {
    const int r = handle_tstop();
    if (r == 1 && status == kRsOk) status = kRsStopTimeAtCurrentTime;
    else if (r == 2 && status == kRsOk) status = kRsStopTimeBeforeCurrentTime;
  }
  long guard = 0;
  bool done = status != kRsOk;
  while (!done) {
    if (++guard > o.max_steps) { status = kRsMaxStepsExceeded; break; }
    // ================================================================ Sdirk::step (sdirk.rs:409-543)
    double hh = h;
    if (fabs(hh) < o.min_timestep) { status = kRsStepSizeTooSmall; break; }
    op_h = hh;
    int nattempts = 0;
    bool updated_jacobian = false;
    const bool skip_first = T.a[0] == 0.0;
    double fac = 1.0, error_norm = 0.0;
    double k = 0.0;  // the stage increment being solved for, my component
    while (true) {
      if (skip_first) {
        diff[0] = hh * dy;  // start_step_attempt (runge_kutta.rs:505-516)
        if constexpr (SENS)
          for (int j = 0; j < nsp; ++j) sdiff[j][0] = hh * dsv[j];  // "sensitivities too" (:518-523)
      }
      // The one place where a requested linearisation is carried out.  Requests are made after a step, after a failed attempt (both come back here
      // before the next Newton solve) and by the Checkpoint of the very first solve — whose stage (the first one that runs) and phi are known here.
      if (!is_jacobian_set) {
        const int i0 = skip_first ? 1 : 0;
        phi = i0 == 0 ? y * 1.0 : 1.0 * diff[0] * T.a[0 * S + 1] + 1.0 * y;
        request_reset(true, y, t + T.c[i0] * hh);
        n_setups++;
      }
      if (factor_pending) execute_reset();
      bool failed = false;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        if (failed || (skip_first && i == 0)) continue;
        // ---- do_stage_sdirk (runge_kutta.rs:631-689)
        const double ts = t + T.c[i] * hh;
        if (i == 0) phi = y * 1.0;  // set_phi: phi = y0 + diff[:, 0..i] a_row_i   (nalgebra gemv order)
        else {
          double acc = 1.0 * diff[0] * T.a[0 * S + i] + 1.0 * y;
#pragma unroll
          for (int j = 1; j < i; ++j) acc = 1.0 * diff[j] * T.a[j * S + i] + acc;
          phi = acc;
        }
        if (i == 0) k = hh * dy;  // predict_stage_sdirk (:610-629)
        else if (i == 1) k = diff[0];
        else {
          const double cc = (T.c[i] - T.c[i - 2]) / (T.c[i - 1] - T.c[i - 2]);
          k = (-cc) * diff[i - 2] + (1.0 + cc) * diff[i - 1];
        }
        // Newton (newton.rs:13-36 over NoLineSearch); error_y = state.y
        conv.reset();
        bool solved = false;
        for (int it = 0; it < conv.max_iter; ++it) {
          const double tmp = op_c * k + 1.0 * phi;
          const double f = rhs_of(tmp, ts);
          const double beta = -op_h;
          double delta;
          if constexpr (kWmHasMass) {  // F(k) = M k - h f(phi + c k): the model's mass product of the published increment, plus beta f (mass_gemv)
            xs2[ln] = k;
            __syncthreads();
            auto X2f = [&](int64_t q) { return xs2[q]; };
            delta = rowlive ? wm_mass_component(ts, (int64_t)ln, X2f, Pf) + beta * f : 0.0;
            __syncthreads();
          } else {
            delta = 1.0 * k + beta * f;
          }
          if (!lu_solve(delta)) break;  // LuSolveFailed
          k = k - delta;
          const ConvStatus st = conv.check_new_iteration(sqrt(wms_wave(delta, y)));
          if (st == ConvStatus::Converged) { solved = true; break; }
          if (st == ConvStatus::Diverged) break;
        }
        n_newton += conv.niter;
        if (solved) {
          old_y = op_c * k + 1.0 * phi;  // get_f_eval
          diff[i] = k;
          if constexpr (SENS) {
            // the sensitivity half of do_stage_sdirk (:691-748): SensRhs linearised about this stage's state (old_state.y = f_eval, ts); per parameter phi and the stage
            // predictor from ITS difference array, a Newton solve of F(k) = k - h (J (phi + c k) + (df/dp)_j) with the factors of the state equations and the shared
            // Convergence (the norm against s_j with the states' tolerances); the iteration count is added before the failure test
            for (int j = 0; j < nsp && solved; ++j) {
              auto Ej = [&](int64_t q) { return q == j ? 1.0 : 0.0; };
              __syncthreads();
              xs[ln] = old_y;
              __syncthreads();
              const double dfdp = rowlive ? wm_sens_component(ts, (int64_t)ln, Xf, Ej, Pf, false) : 0.0;
              double sphi;
              if (i == 0) sphi = sv[j] * 1.0;
              else {
                double acc = 1.0 * sdiff[j][0] * T.a[0 * S + i] + 1.0 * sv[j];
#pragma unroll
                for (int m = 1; m < i; ++m) acc = 1.0 * sdiff[j][m] * T.a[m * S + i] + acc;
                sphi = acc;
              }
              double ks;
              if (i == 0) ks = hh * dsv[j];
              else if (i == 1) ks = sdiff[j][0];
              else {
                const double cc = (T.c[i] - T.c[i - 2]) / (T.c[i - 1] - T.c[i - 2]);
                ks = (-cc) * sdiff[j][i - 2] + (1.0 + cc) * sdiff[j][i - 1];
              }
              conv.reset();
              bool s_solved = false;
              for (int it = 0; it < conv.max_iter; ++it) {
                const double tmp2 = op_c * ks + 1.0 * sphi;
                __syncthreads();
                xs2[ln] = tmp2;
                __syncthreads();
                const double jm = rowlive ? wm_component(model, (int64_t)n, ts, (int64_t)ln, Xf, X2f, Pf, true) : 0.0;
                const double fr = jm + dfdp;
                double sdelta = 1.0 * ks + (-op_h) * fr;
                if (!lu_solve(sdelta)) break;
                ks = ks - sdelta;
                const ConvStatus st = conv.check_new_iteration(sqrt(wms_wave(sdelta, sv[j])));
                if (st == ConvStatus::Converged) { s_solved = true; break; }
                if (st == ConvStatus::Diverged) break;
              }
              n_newton += conv.niter;
              if (!s_solved) { solved = false; break; }
              old_sv[j] = op_c * ks + 1.0 * sphi; old_dsv[j] = ks; sdiff[j][i] = ks;
            }
          }
        }
        if (!solved) {
          if (!updated_jacobian) {
            updated_jacobian = true;
            jacobian_updates(hh, JState::FirstConvergenceFail);
          } else {
            hh *= 0.3;
            conv.eta = C.eta_reset_ts;
            op_h = hh;
            jacobian_updates(hh, JState::SecondConvergenceFail);
          }
          has_prev_err = false;
          n_nl_fails += 1;  // solve_fail (runge_kutta.rs:868-892)
          if (n_nl_fails > o.max_nonlinear_solver_failures) status = kRsTooManyNonlinearSolverFailures;
          else if (fabs(hh) < o.min_timestep) status = kRsStepSizeTooSmall;
          failed = true;
        }
      }
      if (status != kRsOk) break;
      if (failed) continue;
      // ---- error estimate (runge_kutta.rs:783-800, sdirk.rs:474-495): diff d, one LU solve (identity mass)
      double err = 1.0 * diff[0] * T.d[0];
#pragma unroll
      for (int j = 1; j < S; ++j) err = 1.0 * diff[j] * T.d[j] + err;
      if constexpr (kWmHasMass) {  // through the mass matrix as it was last evaluated (current_mass().gemv, nalgebra order)
        __syncthreads();
        xs2[ln] = err;
        __syncthreads();
        double acc = 1.0 * sM[0 * 64 + ln] * xs2[0];
        for (int j = 1; j < n; ++j) acc = 1.0 * sM[j * 64 + ln] * xs2[j] + acc;
        err = rowlive ? acc : 0.0;
      }
      if (!lu_solve(err)) { status = kRsTooManyNonlinearSolverFailures; break; }
      error_norm = fmax(0.0, wms_wave(err, y));
      if constexpr (SENS) {
        if (T.sens_error_control)  // runge_kutta.rs:812-822 — no linear solve on the sensitivity error estimates
          for (int j = 0; j < nsp; ++j) {
            double se = 1.0 * sdiff[j][0] * T.d[0];
#pragma unroll
            for (int m = 1; m < S; ++m) se = 1.0 * sdiff[j][m] * T.d[m] + se;
            error_norm = fmax(error_norm, wms_sens(se, sv[j]));
          }
      }
      const double maxiter = (double)conv.max_iter, niter = (double)conv.niter;
      const double safety_factor = (2.0 * maxiter + 1.0) / (2.0 * maxiter + niter);
      {  // Rk::factor (runge_kutta.rs:466-495)
        const double safety = 0.9 * safety_factor;
        double f = safety * pi_controller_raw(error_norm, has_prev_err, prev_err, o.pi_control_integral, o.pi_control_proportional, T.order + 1, det);
        if (f > o.max_timestep_shrink && f < o.min_timestep_growth) f = 1.0;
        if (f < o.min_timestep_shrink) f = o.min_timestep_shrink;
        if (f > o.max_timestep_growth) f = o.max_timestep_growth;
        fac = f;
      }
      if (error_norm < 1.0) break;
      hh *= fac;
      conv.eta = C.eta_reset_ts;
      op_h = hh;
      jacobian_updates(hh, JState::ErrorTestFail);
      nattempts += 1;
      has_prev_err = false;
      n_err_fails += 1;  // error_test_fail (runge_kutta.rs:841-866)
      if (nattempts >= o.max_error_test_failures) { status = kRsTooManyErrorTestFailures; break; }
      if (fabs(hh) < o.min_timestep) { status = kRsStepSizeTooSmall; break; }
    }
    if (status != kRsOk) break;
    const double new_h = hh * fac;
    if (fac != 1.0) conv.eta = C.eta_reset_ts;
    op_h = new_h;
    jacobian_updates(new_h, JState::StepSuccess);
    ju.step();
    prev_err = error_norm; has_prev_err = true;
    {  // step_accepted (runge_kutta.rs:894-960): old_state <- (f_eval of the last stage, k/h, t+h, new_h); swap
      const double inv_h = 1.0 / hh;
      const double ny = old_y, ndy = k * inv_h;
      old_y = y; old_dy = dy;
      y = ny; dy = ndy;
      if constexpr (SENS)  // old_ds_j *= 1/h; swap(old_s, s); swap(old_ds, ds)
        for (int j = 0; j < nsp; ++j) {
          const double ns_ = old_sv[j], nds = old_dsv[j] * inv_h;
          old_sv[j] = sv[j]; old_dsv[j] = dsv[j];
          sv[j] = ns_; dsv[j] = nds;
        }
      const double nt = t + hh;
      old_t = t;
      t = nt;
      h = new_h;
    }
    n_steps += 1;
    int reason = 0;  // 0 internal, 1 tstop, 3 root
    if (Cp->nroots > 0) {
      // RootFinder::check_root (root.rs:91-222) on wavefront-uniform root values, as in k_bdf_wave_member
      double g1[2], gmid[2];
      root_of(y, t, g1);
      bool found;
      double frac;
      int imax;
      root_finding_lane<2>(g0, g1, found, frac, imax);
      if (imax < 0) {
        g0[0] = g1[0]; g0[1] = g1[1];
        rf_t0 = t;
        if (found) { t_root = t; root_idx = fabs(g0[1]) < fabs(g0[0]) && Cp->nroots > 1 ? 1 : 0; reason = 3; }
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
    if (reason == 0 && has_tstop) {
      const int r = handle_tstop();
      if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; }
      if (r == 1) { has_tstop = false; reason = 1; }
    }
    // ================================================================ solve_dense (method.rs:467-520)
    const double upto = reason == 3 ? t_root : t;
    if (steps_mode) {  // InternalTimestep / TstopReached -> write_out (method.rs:907-921): state.y; a root is written below, at the root
      if (reason != 3) steps_write(t, y);
    } else
    while (col < C.n_eval && t_eval[col] <= upto) {
      const double yv = interpolate(t_eval[col]);
      if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = yv;
      if constexpr (SENS) {  // interpolate_sens_inplace (runge_kutta.rs:1237-1330): the state's interpolant on (old_s, s, sdiff_j)
        const double tt = t_eval[col];
        const double dt = t - old_t;
        const double theta = dt == 0.0 ? 1.0 : (tt - old_t) / dt;
        double bf[S];
        if (T.has_beta) {
          double thetav[kMaxPoly];
          thetav[0] = theta;
#pragma unroll
          for (int q = 1; q < kMaxPoly; ++q) thetav[q] = theta * thetav[q - 1];
#pragma unroll
          for (int i = 0; i < S; ++i) {
            double acc = 1.0 * T.beta[0 * S + i] * thetav[0];
#pragma unroll
            for (int q = 1; q < kMaxPoly; ++q) if (q < T.poly_order) acc = 1.0 * T.beta[q * S + i] * thetav[q] + acc;
            bf[i] = acc;
          }
        }
        for (int j = 0; j < nsp; ++j) {
          double ret;
          if (T.has_beta) {
            double acc = 1.0 * sdiff[j][0] * bf[0] + 1.0 * old_sv[j];
#pragma unroll
            for (int m = 1; m < S; ++m) acc = 1.0 * sdiff[j][m] * bf[m] + acc;
            ret = acc;
          } else {
            double r = sv[j] - old_sv[j];
            r = (1.0 * (theta - 1.0)) * sdiff[j][0] + (1.0 - 2.0 * theta) * r;
            r = (1.0 * theta) * sdiff[j][S - 1] + 1.0 * r;
            r = (1.0 - theta) * old_sv[j] + (theta * (theta - 1.0)) * r;
            r = theta * sv[j] + 1.0 * r;
            ret = r;
          }
          if (rowlive) T.sens_out[(((int64_t)col * nsp + j) * n + ln) * nb + b] = ret;
        }
      }
      col++;
    }
    if constexpr (kWmResets) {
      if (reason == 3) {
        // A reset operator is configured, as in k_sdirk_resident: state_mut_back(t_root) (runge_kutta.rs:396-434), apply_reset (sdirk.rs:368-374 over state.rs:279-306:
        // y <- reset(y, t), dy <- f(y, t)), the stop time armed again, then Rk::start_step's branch for a mutated state (:444-464: root finder re-initialised, stop
        // time checked once more) — a one-step method restarts from (t, y, dy, h) as they are
        const double yb = interpolate(t_root);
        t = t_root;
        __syncthreads();
        xs[ln] = yb;
        __syncthreads();
        y = rowlive ? wm_reset_component(t, (int64_t)ln, Xf, Pf) : 0.0;
        dy = rhs_of(y, t);
        if (steps_mode) steps_write(t, y);  // method.rs:931-932: the reset state at the root time
        if (t < tstop) {
          has_tstop = true;
          { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          root_of(y, t, g0);
          rf_t0 = t;
          { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
        } else done = true;
        reason = 0;
      }
    }
    if (reason == 3 && steps_mode) {  // method.rs:922-947 without a reset: state_mut_back(t_root), write_out, RootFound
      steps_write(t_root, interpolate(t_root));
      done = true;
    } else
    if (reason == 3) {
      if (col < C.n_eval) {
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
  for (; col < C.n_eval; ++col) {
    if (rowlive) y_out[((int64_t)col * n + ln) * nb + b] = __builtin_nan("");
    if constexpr (SENS)
      for (int j = 0; j < nsp; ++j)
        if (rowlive) T.sens_out[(((int64_t)col * nsp + j) * n + ln) * nb + b] = __builtin_nan("");
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
