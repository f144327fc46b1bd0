// Device-resident SDIRK integrator (launch code and documentation: dsh_sdirk_resident.hip).  In a header so that run-time-compiled model modules
// (dsh_jit.hip) instantiate the same kernel for user models.
#pragma once
#include "dsh_resident.hpp"
#if !defined(__HIPCC_RTC__)
#include <cmath>
#endif

namespace dsh {

constexpr int kMaxStages = 4, kMaxPoly = 2;

struct SdirkConsts {
  ResidentConsts r;
  int s, order, has_beta, poly_order;
  double a[kMaxStages * kMaxStages];  // column-major s x s
  double b[kMaxStages], c[kMaxStages], d[kMaxStages];
  double beta[kMaxStages * kMaxPoly];  // column-major s x poly_order
  double gamma;                        // a(1,1)
  // forward sensitivities (k_sdirk_resident<.., SENS = true>; problem.tr_bdf2_sens() / esdirk34_sens()): as in AdaptiveConsts (dsh_adaptive_kernel.hpp)
  double* sens_out;                    // n_eval x NP x N x nb
  double sens_rtol, sens_atol[4];
  int sens_error_control, sens_pad;
  // OdeSolverMethod::solve (method.rs:227-258 over :881-961), as in AdaptiveConsts: steps_cap > 0 makes the launch write the state after EVERY accepted step
  // (y_out [steps_cap][N][nb], steps_t_out [steps_cap][nb]; columns beyond steps_cap are counted, not stored) instead of interpolating at save points
  double* steps_t_out;
  int steps_cap, steps_pad;
};

// the run-time-compiled banded form (state in per-lane memory) is built for a fixed occupancy like the BDF kernel (dsh_jit.hip defines the macro)
#ifdef DSH_ADAPTIVE_WAVES_PER_EU
#define DSH_SDIRK_OCCUPANCY __attribute__((amdgpu_waves_per_eu(DSH_ADAPTIVE_WAVES_PER_EU, DSH_ADAPTIVE_WAVES_PER_EU)))
#else
#define DSH_SDIRK_OCCUPANCY
#endif
// FAST (dsh_adaptive_options::deterministic_pow == 2; instantiated only in dsh_sdirk_fast.hip, which is compiled with -ffp-contract=fast -freciprocal-math): a tag that
// gives the fast-arithmetic build of the same source its own symbols — multiply-adds fused, divisions by reciprocal + refinement.  Same algorithm; on BASELINE config 5
// at full size the same step / Newton / refactorisation decisions for every member, states and event times within 1e-9 (tests/test_gpu_configs.py); 18 - 20 % shorter
// dependent chain (profiles/r06_c5_fast.md).  Not bit-comparable with the oracle: the parity tier runs the exact instantiation.
template <class Mdl, bool BA, bool WAVE, int S, bool SENS = false, bool FAST = false>
__global__ __launch_bounds__(64) DSH_SDIRK_OCCUPANCY void k_sdirk_resident(int64_t nb, const double* __restrict__ p_g, const double* __restrict__ atol_g, const SdirkConsts* __restrict__ Cp,
                                                       const double* __restrict__ t_eval, double* __restrict__ y_out, int32_t* __restrict__ stats_out,
                                                       int32_t* __restrict__ status_out, double* __restrict__ t_root_out, int32_t* __restrict__ root_idx_out,
                                                       int32_t* __restrict__ ncols_out, unsigned long long* __restrict__ totals) {
  constexpr int N = Mdl::N, NP = Mdl::NP, NR = Mdl::NROOTS > 0 ? Mdl::NROOTS : 1;
  const SdirkConsts& T = *Cp;
  const ResidentConsts& C = T.r;
  const dsh_adaptive_options& o = C.o;
  // per-member control may run with fewer than 64 members per wavefront (C.member_lanes: more wavefronts per SIMD, a wavefront pays for the union of fewer paths)
  const int ML = (!WAVE && C.member_lanes > 0) ? C.member_lanes : 64;
  const int64_t bglobal = (int64_t)blockIdx.x * ML + threadIdx.x;
  const bool active = (int)threadIdx.x < ML && bglobal < nb;  // the other lanes shadow the wavefront's first member (no stores): invisible in the group reductions
  const int64_t b = active ? bglobal : (int64_t)blockIdx.x * ML;
  const int ln = threadIdx.x;
  const double rtol = C.rtol;
  double p[NP], atol[N];
  load_vec<NP>(p_g, nb, b, p);
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) atol[i] = BA ? atol_g[i] : atol_g[(int64_t)i * nb + b];
  // banded models (Mdl::BAND_K, n up to 64): Jacobian band and banded LU factors in per-lane memory, see dsh_adaptive_kernel.hpp
  constexpr int BK = model_band_k<Mdl>::value;
  constexpr bool BANDED = BK > 0;
  static_assert(!BANDED || !Mdl::HAS_MASS, "banded device-resident models need an identity mass matrix");
  // hybrid models: events handled in the launch (the banded lane-per-member form included); round 5: with a mass matrix too (register-resident form) — hybrid DAEs
  constexpr bool kResets = model_has_reset<Mdl>::value && (!Mdl::HAS_MASS || !BANDED) && Mdl::NROOTS > 0;
  constexpr int LN = BANDED ? 1 : N;
  __shared__ double sJ[LN * LN][64];
  double Jb[BANDED ? (2 * BK + 1) * N : 1], Lf[BANDED ? BK * N : 1], Uf[BANDED ? (2 * BK + 1) * N : 1];

  // ------------------------------------------------------------ RkState::new_and_consistent(problem, tableau.order())
  int32_t status = kRsOk;
  double t = C.t0, h = 0.0;
  double y[N], dy[N];
  Mdl::init(t, p, y);
  Mdl::rhs(t, y, p, dy);
  if (!group_all<WAVE>(set_consistent<Mdl, WAVE>(t, p, y, dy, atol, rtol, C))) status = kRsInitialConditionDidNotConverge;
  const bool det = o.deterministic_pow != 0;
  h = initial_step_size<Mdl, WAVE>(t, C.h0, y, dy, p, atol, rtol, T.order, det);

  // ------------------------------------------------------------ Rk::_new (runge_kutta.rs:110-190) + Sdirk::_new (sdirk.rs:172-215)
  double diff[S][N];
#pragma unroll
  for (int j = 0; j < S; ++j)
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) diff[j][i] = 0.0;
  double old_y[N], old_dy[N], old_t = t;  // old_state (its h is never read)
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) { old_y[i] = y[i]; old_dy[i] = dy[i]; }
  double g0[NR] = {0.0};
  double rf_t0 = t;
  if constexpr (Mdl::NROOTS > 0) Mdl::root(t, y, p, g0);
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
  double phi[N];
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) phi[i] = 0.0;  // V::zeros until the first set_phi
  double A[BANDED ? 1 : N * N];
  int P[N];
  bool is_jacobian_set = false, jac_stale = true;
  bool has_prev_err = false;
  double prev_err = 0.0;
  int n_setups = 0, n_steps = 0, n_err_fails = 0, n_newton = 0, n_nl_fails = 0;
  // ---- forward sensitivities (runge_kutta.rs:196-232 new_augmented, :691-748 the sensitivity half of do_stage_sdirk, :812-822 error norm, :1237-1330
  // interpolate_sens; RkState::new_with_sensitivities_and_consistent state.rs:1032-1083).  Per-lane arrays in scratch memory, indexed by the parameter at run time.
  constexpr int NPS = SENS ? NP : 1, NS = SENS ? N : 1, SS = SENS ? S : 1;
  // banded lane-per-member models (model_band_k > 0: the state in per-lane memory) take the same code — the sensitivity arrays are per-lane memory too, the
  // sensitivity solves run on the banded factors (round 4; runge_kutta.rs:691-748 for the PDE / battery models)
  static_assert(!SENS || (!Mdl::HAS_MASS && Mdl::NROOTS == 0), "device-resident forward sensitivities: ODE models without root functions");
  double sv[NPS][NS], dsv[NPS][NS], old_sv[NPS][NS], old_dsv[NPS][NS], sdiff[NPS][SS][NS];
  double s_atol[NS];
  if constexpr (SENS) {
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) s_atol[i] = T.sens_atol[(T.sens_pad || i >= 4) ? 0 : i];  // sens_pad: one tolerance for every state (models with more than 4)
    for (int j = 0; j < NP; ++j) {
      double ev[NP], s0[N], jm[N], dfdp[N];
#pragma unroll
      for (int q = 0; q < NP; ++q) ev[q] = q == j ? 1.0 : 0.0;
      Mdl::init_sens_mul(t, p, ev, s0);
      Mdl::sens_mul(t, y, p, ev, dfdp);  // SensRhs::update_state(y0, t0)
      Mdl::jac_mul(t, y, p, s0, jm);     // SensRhs::call_inplace: J(y0) s_j + (df/dp)_j
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) {
        const double d0 = jm[i] + dfdp[i];
        sv[j][i] = s0[i]; dsv[j][i] = d0; old_sv[j][i] = s0[i]; old_dsv[j][i] = d0;
      }
      for (int m = 0; m < S; ++m)
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) sdiff[j][m][i] = 0.0;
    }
  }

  // SdirkCallable::jacobian_inplace (op/sdirk.rs:266-296) + LU: M - (c h) f'(phi + c x)
  auto reset_jacobian = [&](const double (&xx)[N], double tt) __attribute__((always_inline)) {
    if constexpr (BANDED) {
      if (jac_stale) {
        double tmp[N];
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) tmp[i] = op_c * xx[i] + 1.0 * phi[i];
        Mdl::jac_band(tt, tmp, p, Jb);
        jac_stale = false;
      }
      bool sing = false;
      band_factor_lane<N, BK>(Jb, op_c * op_h, Lf, Uf, P, sing);  // J * (-(c h)) + I
      is_jacobian_set = true;
    } else {
    double J[N * N], Mm[N * N];
    if (jac_stale) {
      double tmp[N];
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) tmp[i] = op_c * xx[i] + 1.0 * phi[i];
      assemble_jacobian<Mdl>(tt, tmp, p, J);
#pragma unroll
      for (int e = 0; e < N * N; ++e) sJ[e][ln] = J[e];
      jac_stale = false;
    } else {
#pragma unroll
      for (int e = 0; e < N * N; ++e) J[e] = sJ[e][ln];
    }
    if constexpr (Mdl::HAS_MASS) assemble_mass<Mdl>(tt, p, Mm);
    else {
#pragma unroll
      for (int e = 0; e < N * N; ++e) Mm[e] = (e / N == e % N) ? 1.0 : 0.0;
    }
    const double beta = -(op_c * op_h);
#pragma unroll
    for (int e = 0; e < N * N; ++e) A[e] = J[e] * beta + Mm[e];
    bool sing = false;
    lu_factor_reg<N>(A, P, sing);
    is_jacobian_set = true;
    }
  };
  // Sdirk::_jacobian_updates (sdirk.rs:260-303)
  auto jacobian_updates = [&](double hh, JState st) __attribute__((always_inline)) {
    if (ju.check_rhs_jacobian_update(hh, st, o)) {
      jac_stale = true;
      reset_jacobian(y, t);
      ju.update_rhs_jacobian(hh);
      ju.update_jacobian(hh);
      conv.eta = C.eta_reset;
      n_setups++;
    } else if (ju.check_jacobian_update(hh, st, o)) {
      reset_jacobian(y, t);
      ju.update_jacobian(hh);
      conv.eta = C.eta_reset;
      n_setups++;
    }
  };
  if constexpr (SENS) {
    // Sdirk::new_augmented ends with jacobian_updates(h, Checkpoint) (sdirk.rs:251): with sensitivities the first linearisation is made at construction, at t0 and
    // — phi still being zero — about gamma y0; Checkpoint always refreshes the right-hand side's Jacobian (jacobian_update.rs)
    if (status == kRsOk) {
      jac_stale = true;
      reset_jacobian(y, t);
      ju.update_rhs_jacobian(h);
      ju.update_jacobian(h);
      conv.eta = C.eta_reset;
      n_setups++;
    }
  }
  // handle_tstop (runge_kutta.rs:752-781): 0 nothing, 1 reached, 2 StopTimeBeforeCurrentTime
  bool has_tstop = true;
  const double tstop = t_eval[C.n_eval - 1];
  auto handle_tstop = [&]() __attribute__((always_inline)) -> int {
    const double troundoff = 100.0 * kEps * (fabs(t) + fabs(h));
    if (fabs(t - tstop) <= troundoff) return 1;
    if ((h > 0.0 && tstop < t - troundoff) || (h < 0.0 && tstop > t + troundoff)) return 2;
    if ((h > 0.0 && t + h > tstop + troundoff) || (h < 0.0 && t + h < tstop - troundoff)) {
      const double factor = (tstop - t) / h;
      h *= factor;
    }
    return 0;
  };
  // interpolate_inplace (runge_kutta.rs:1080-1127) inside the last step [old_t, t]
  auto interpolate = [&](double tt, double (&ret)[N]) __attribute__((always_inline)) {
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
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) {
        double acc = 1.0 * diff[0][i] * bf[0] + 1.0 * old_y[i];
#pragma unroll
        for (int j = 1; j < S; ++j) acc = 1.0 * diff[j][i] * bf[j] + acc;
        ret[i] = acc;
      }
    } else {  // interpolate_hermite (runge_kutta.rs:1016-1035)
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) {
        double r = y[i] - old_y[i];
        r = (1.0 * (theta - 1.0)) * diff[0][i] + (1.0 - 2.0 * theta) * r;
        r = (1.0 * theta) * diff[S - 1][i] + 1.0 * r;
        r = (1.0 - theta) * old_y[i] + (theta * (theta - 1.0)) * r;
        r = theta * y[i] + 1.0 * r;
        ret[i] = r;
      }
    }
  };

  int col = 0;
  double t_root = 0.0;
  int root_idx = -1;
  const bool steps_mode = !SENS && T.steps_cap > 0;  // every accepted step out (SdirkConsts::steps_cap)
  auto steps_write = [&](double tw, const double (&yw)[N]) __attribute__((always_inline)) {
    if (col < T.steps_cap && active) {
      T.steps_t_out[(int64_t)col * nb + b] = tw;
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) y_out[((int64_t)col * N + i) * nb + b] = yw[i];
    }
    col++;
  };
  if (steps_mode && status == kRsOk) steps_write(t, y);  // write_out before the first step (method.rs:900)
  {  // set_stop_time (runge_kutta.rs:436-447); t_eval[0] >= t0 is checked on the host
    const int r = handle_tstop();
    if (r == 1 && status == kRsOk) status = kRsStopTimeAtCurrentTime;
    else if (r == 2 && status == kRsOk) status = kRsStopTimeBeforeCurrentTime;
  }

  long guard = 0;
  bool done = status != kRsOk || (!WAVE && !active);
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
    double k[N];  // old_state.dy: the stage increment being solved for
    while (true) {
      if (skip_first) {
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) diff[0][i] = hh * dy[i];  // start_step_attempt (runge_kutta.rs:505-516)
        if constexpr (SENS) {  // "sensitivities too" (:518-523)
          for (int j = 0; j < NP; ++j)
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) sdiff[j][0][i] = hh * dsv[j][i];
        }
      }
      bool failed = false;
#pragma unroll
      for (int i = 0; i < S; ++i) {
        if (failed || (skip_first && i == 0)) continue;
        // ---- do_stage_sdirk (runge_kutta.rs:631-689)
        const double ts = t + T.c[i] * hh;
        // set_phi: phi = y0 + diff[:, 0..i] a_row_i   (nalgebra gemv order)
DSH_UNROLL_N
        for (int r = 0; r < N; ++r) {
          if (i == 0) phi[r] = y[r] * 1.0;
          else {
            double acc = 1.0 * diff[0][r] * T.a[0 * S + i] + 1.0 * y[r];
#pragma unroll
            for (int j = 1; j < i; ++j) acc = 1.0 * diff[j][r] * T.a[j * S + i] + acc;
            phi[r] = acc;
          }
        }
        // predict_stage_sdirk (:610-629)
        if (i == 0) {
DSH_UNROLL_N
          for (int r = 0; r < N; ++r) k[r] = hh * dy[r];
        } else if (i == 1) {
DSH_UNROLL_N
          for (int r = 0; r < N; ++r) k[r] = diff[0][r];
        } else {
          const double cc = (T.c[i] - T.c[i - 2]) / (T.c[i - 1] - T.c[i - 2]);
DSH_UNROLL_N
          for (int r = 0; r < N; ++r) k[r] = (-cc) * diff[i - 2][r] + (1.0 + cc) * diff[i - 1][r];
        }
        if (!is_jacobian_set) { reset_jacobian(y, ts); n_setups++; }  // Checkpoint
        // Newton (newton.rs:13-36 over NoLineSearch); error_y = state.y
        conv.reset();
        bool solved = false;
        for (int it = 0; it < conv.max_iter; ++it) {
          double tmp[N], f[N], delta[N];
DSH_UNROLL_N
          for (int r = 0; r < N; ++r) tmp[r] = op_c * k[r] + 1.0 * phi[r];
          Mdl::rhs(ts, tmp, p, f);
          const double beta = -op_h;
          if constexpr (Mdl::HAS_MASS) {
DSH_UNROLL_N
            for (int r = 0; r < N; ++r) delta[r] = f[r];
            Mdl::mass_gemv(ts, k, p, beta, delta);
          } else {
DSH_UNROLL_N
            for (int r = 0; r < N; ++r) delta[r] = 1.0 * k[r] + beta * f[r];
          }
          bool solved_ok;
          if constexpr (BANDED) solved_ok = band_solve_lane<N, BK>(Lf, Uf, P, delta);
          else solved_ok = lu_solve_reg<N>(A, P, delta);
          if (!group_all<WAVE>(solved_ok)) break;  // LuSolveFailed
DSH_UNROLL_N
          for (int r = 0; r < N; ++r) k[r] = k[r] - delta[r];
          const ConvStatus st = conv.check_new_iteration(sqrt(group_norm<WAVE>(wms<N>(delta, y, atol, rtol))));
          if (st == ConvStatus::Converged) { solved = true; break; }
          if (st == ConvStatus::Diverged) break;
        }
        n_newton += conv.niter;
        if (solved) {
DSH_UNROLL_N
          for (int r = 0; r < N; ++r) { old_y[r] = op_c * k[r] + 1.0 * phi[r]; diff[i][r] = k[r]; }  // get_f_eval; diff.column_mut(i)
        }
        if constexpr (SENS) {
          if (solved) {
            // the sensitivity half of do_stage_sdirk (:691-748): SensRhs linearised about this stage's state (old_state.y = f_eval, ts); per parameter phi and
            // the stage predictor from ITS difference array, a Newton solve of F(k) = k - h (J (phi + c k) + (df/dp)_j) with the factors of the state
            // equations and the shared Convergence (the norm against s_j with the states' tolerances); the iteration count is added before the failure test
            for (int j = 0; j < NP && solved; ++j) {
              double ev[NP], dfdp[N], sphi[N], ks[N];
#pragma unroll
              for (int q = 0; q < NP; ++q) ev[q] = q == j ? 1.0 : 0.0;
              Mdl::sens_mul(ts, old_y, p, ev, dfdp);
DSH_UNROLL_N
              for (int r = 0; r < N; ++r) {
                if (i == 0) sphi[r] = sv[j][r] * 1.0;
                else {
                  double acc = 1.0 * sdiff[j][0][r] * T.a[0 * S + i] + 1.0 * sv[j][r];
#pragma unroll
                  for (int m = 1; m < i; ++m) acc = 1.0 * sdiff[j][m][r] * T.a[m * S + i] + acc;
                  sphi[r] = acc;
                }
              }
              if (i == 0) {
DSH_UNROLL_N
                for (int r = 0; r < N; ++r) ks[r] = hh * dsv[j][r];
              } else if (i == 1) {
DSH_UNROLL_N
                for (int r = 0; r < N; ++r) ks[r] = sdiff[j][0][r];
              } else {
                const double cc = (T.c[i] - T.c[i - 2]) / (T.c[i - 1] - T.c[i - 2]);
DSH_UNROLL_N
                for (int r = 0; r < N; ++r) ks[r] = (-cc) * sdiff[j][i - 2][r] + (1.0 + cc) * sdiff[j][i - 1][r];
              }
              conv.reset();
              bool s_solved = false;
              for (int it = 0; it < conv.max_iter; ++it) {
                double tmp[N], jm[N], delta[N];
DSH_UNROLL_N
                for (int r = 0; r < N; ++r) tmp[r] = op_c * ks[r] + 1.0 * sphi[r];
                Mdl::jac_mul(ts, old_y, p, tmp, jm);
                const double beta = -op_h;
DSH_UNROLL_N
                for (int r = 0; r < N; ++r) {
                  const double fr = jm[r] + dfdp[r];
                  delta[r] = 1.0 * ks[r] + beta * fr;
                }
                bool s_ok;
                if constexpr (BANDED) s_ok = band_solve_lane<N, BK>(Lf, Uf, P, delta);
                else s_ok = lu_solve_reg<N>(A, P, delta);
                if (!group_all<WAVE>(s_ok)) break;
DSH_UNROLL_N
                for (int r = 0; r < N; ++r) ks[r] = ks[r] - delta[r];
                const ConvStatus st = conv.check_new_iteration(sqrt(group_norm<WAVE>(wms<N>(delta, sv[j], atol, rtol))));
                if (st == ConvStatus::Converged) { s_solved = true; break; }
                if (st == ConvStatus::Diverged) break;
              }
              n_newton += conv.niter;
              if (!s_solved) { solved = false; break; }
DSH_UNROLL_N
              for (int r = 0; r < N; ++r) { old_sv[j][r] = op_c * ks[r] + 1.0 * sphi[r]; old_dsv[j][r] = ks[r]; sdiff[j][i][r] = ks[r]; }
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
      // ---- error estimate (runge_kutta.rs:783-800, sdirk.rs:474-495): diff d, through the mass matrix and one LU solve
      double err[N];
DSH_UNROLL_N
      for (int r = 0; r < N; ++r) {
        double acc = 1.0 * diff[0][r] * T.d[0];
#pragma unroll
        for (int j = 1; j < S; ++j) acc = 1.0 * diff[j][r] * T.d[j] + acc;
        err[r] = acc;
      }
      if constexpr (Mdl::HAS_MASS) {
        double Mm[N * N], e2[N];
        assemble_mass<Mdl>(t, p, Mm);
DSH_UNROLL_N
        for (int r = 0; r < N; ++r) {
          double acc = 1.0 * Mm[0 * N + r] * err[0];
#pragma unroll
          for (int j = 1; j < N; ++j) acc = 1.0 * Mm[j * N + r] * err[j] + acc;
          e2[r] = acc;
        }
DSH_UNROLL_N
        for (int r = 0; r < N; ++r) err[r] = e2[r];
      }
      bool err_ok;
      if constexpr (BANDED) err_ok = band_solve_lane<N, BK>(Lf, Uf, P, err);
      else err_ok = lu_solve_reg<N>(A, P, err);
      if (!group_all<WAVE>(err_ok)) { status = kRsTooManyNonlinearSolverFailures; break; }
      error_norm = fmax(0.0, group_norm<WAVE>(wms<N>(err, y, atol, rtol)));
      if constexpr (SENS) {
        if (T.sens_error_control)  // runge_kutta.rs:812-822 — no linear solve on the sensitivity error estimates
          for (int j = 0; j < NP; ++j) {
            double se[N];
DSH_UNROLL_N
            for (int r = 0; r < N; ++r) {
              double acc = 1.0 * sdiff[j][0][r] * T.d[0];
#pragma unroll
              for (int m = 1; m < S; ++m) acc = 1.0 * sdiff[j][m][r] * T.d[m] + acc;
              se[r] = acc;
            }
            error_norm = fmax(error_norm, group_norm<WAVE>(wms<N>(se, sv[j], s_atol, T.sens_rtol)));
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
    // ---- step_accepted (runge_kutta.rs:894-960): old_state <- (f_eval of the last stage, k/h, t+h, new_h); swap
    {
      const double inv_h = 1.0 / hh;
DSH_UNROLL_N
      for (int r = 0; r < N; ++r) {
        const double ny = old_y[r], ndy = k[r] * inv_h;
        old_y[r] = y[r]; old_dy[r] = dy[r];
        y[r] = ny; dy[r] = ndy;
      }
      if constexpr (SENS) {  // old_ds_j *= 1/h; swap(old_s, s); swap(old_ds, ds)
        for (int j = 0; j < NP; ++j)
DSH_UNROLL_N
          for (int r = 0; r < N; ++r) {
            const double ns = old_sv[j][r], nds = old_dsv[j][r] * inv_h;
            old_sv[j][r] = sv[j][r]; old_dsv[j][r] = dsv[j][r];
            sv[j][r] = ns; dsv[j][r] = nds;
          }
      }
      const double nt = t + hh;
      old_t = t;
      t = nt;
      h = new_h;
    }
    n_steps += 1;
    int reason = 0;  // 0 internal, 1 tstop, 3 root
    if constexpr (Mdl::NROOTS > 0) {
      const int rr = check_root<Mdl, WAVE>(g0, rf_t0, y, t, p, interpolate, t_root, root_idx);
      if (rr == 2) { status = kRsRootBatchMismatch; break; }
      if (rr == 1) reason = 3;
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
      double yv[N];
      interpolate(t_eval[col], yv);
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) if (active) y_out[((int64_t)col * N + i) * nb + b] = yv[i];
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
        for (int j = 0; j < NP; ++j) {
          double ret[N];
          if (T.has_beta) {
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) {
              double acc = 1.0 * sdiff[j][0][i] * bf[0] + 1.0 * old_sv[j][i];
#pragma unroll
              for (int m = 1; m < S; ++m) acc = 1.0 * sdiff[j][m][i] * bf[m] + acc;
              ret[i] = acc;
            }
          } else {
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) {
              double r = sv[j][i] - old_sv[j][i];
              r = (1.0 * (theta - 1.0)) * sdiff[j][0][i] + (1.0 - 2.0 * theta) * r;
              r = (1.0 * theta) * sdiff[j][S - 1][i] + 1.0 * r;
              r = (1.0 - theta) * old_sv[j][i] + (theta * (theta - 1.0)) * r;
              r = theta * sv[j][i] + 1.0 * r;
              ret[i] = r;
            }
          }
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) if (active) T.sens_out[(((int64_t)col * NP + j) * N + i) * nb + b] = ret[i];
        }
      }
      col++;
    }
    if constexpr (kResets) {
      if (reason == 3) {
        // A reset operator is configured (solve_dense, method.rs:774-797): state_mut_back(t_root) (runge_kutta.rs:396-434), apply_reset (sdirk.rs:368-374 over
        // state.rs:279-306: y <- reset(y, t), dy <- f(y, t)), the stop time armed again, then Rk::start_step's branch for a mutated state (:444-464: root finder
        // re-initialised, stop time checked once more) — a one-step method restarts from (t, y, dy, h) as they are
        double yb[N], yr[N];
        interpolate(t_root, yb);
        if constexpr (Mdl::HAS_MASS) {
          // state_mut_back stores the derivative of the step's dense output at the root in state.dy (interpolate_dy_inplace, runge_kutta.rs:1129-1181): the starting
          // guess of the differential unknowns of set_consistent below
          const double dt = t - old_t;
          if (dt != 0.0) {
            const double theta = (t_root - old_t) / dt;
            double dyb[N];
            if (T.has_beta) {  // interpolate_beta_function_deriv (:985-1002): d_thetav = [1, 2 theta, 3 theta^2, ...]
              double dth[kMaxPoly];
              dth[0] = 1.0;
              double theta_pow = theta;
#pragma unroll
              for (int q = 1; q < kMaxPoly; ++q) { dth[q] = ((double)q + 1.0) * theta_pow; theta_pow *= theta; }
              double dbf[S];
#pragma unroll
              for (int i = 0; i < S; ++i) {
                double acc = 1.0 * T.beta[0 * S + i] * dth[0];
#pragma unroll
                for (int q = 1; q < kMaxPoly; ++q) if (q < T.poly_order) acc = 1.0 * T.beta[q * S + i] * dth[q] + acc;
                dbf[i] = acc;
              }
              const double al = 1.0 / dt;
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) {
                double acc = al * diff[0][i] * dbf[0];
#pragma unroll
                for (int j = 1; j < S; ++j) acc = al * diff[j][i] * dbf[j] + acc;
                dyb[i] = acc;
              }
            } else {  // interpolate_hermite_deriv (:1037-1078)
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) {
                double q = y[i] - old_y[i];
                q = (1.0 * (theta - 1.0)) * diff[0][i] + (1.0 - 2.0 * theta) * q;
                q = (1.0 * theta) * diff[S - 1][i] + 1.0 * q;
                double d = y[i] - old_y[i];
                d = ((2.0 * theta - 1.0) / dt) * q + (1.0 / dt) * d;
                double q2 = old_y[i] - y[i];
                q2 = 1.0 * diff[0][i] + 2.0 * q2;
                q2 = 1.0 * diff[S - 1][i] + 1.0 * q2;
                d = (theta * (theta - 1.0) / dt) * q2 + 1.0 * d;
                dyb[i] = d;
              }
            }
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) dy[i] = dyb[i];
          }  // (dt == 0: state.dy as it is)
        }
        t = t_root;
        Mdl::reset(t, yb, p, yr);
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) y[i] = yr[i];
        if constexpr (Mdl::HAS_MASS) {
          // apply_reset_with_mass (state.rs:279-306): (y, dy) consistent again by Newton on InitOp WITHOUT line search
          if (!group_all<WAVE>(set_consistent<Mdl, WAVE>(t, p, y, dy, atol, rtol, C, true))) { status = kRsInitialConditionDidNotConverge; break; }
        } else
        Mdl::rhs(t, y, p, dy);
        if (steps_mode) steps_write(t, y);  // method.rs:931-932: the reset state at the root time
        if (t < tstop) {
          has_tstop = true;
          { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          Mdl::root(t, y, p, g0);
          rf_t0 = t;
          { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
        } else done = true;
        reason = 0;
      }
    }
    if (reason == 3 && steps_mode) {  // method.rs:922-947 without a reset: state_mut_back(t_root), write_out, RootFound
      double yv[N];
      interpolate(t_root, yv);
      steps_write(t_root, yv);
      done = true;
    } else
    if (reason == 3) {  // state_mut_back(root_time); the column after the drained ones holds the state at the root
      if (col < C.n_eval) {
        double yv[N];
        interpolate(t_root, yv);
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) if (active) y_out[((int64_t)col * N + i) * nb + b] = yv[i];
        col++;
      }
      done = true;
    }
    if (reason == 1) done = true;
  }
  if (active) {
    if (ncols_out != nullptr) ncols_out[b] = col;
    if (!steps_mode)
    for (; col < C.n_eval; ++col) {  // columns that were never reached (root stop or error exit): NaN
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) y_out[((int64_t)col * N + i) * nb + b] = __builtin_nan("");
      if constexpr (SENS)
        for (int j = 0; j < NP; ++j)
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) T.sens_out[(((int64_t)col * NP + j) * N + i) * nb + b] = __builtin_nan("");
    }
    if (status_out != nullptr) status_out[b] = status;
    if (t_root_out != nullptr) t_root_out[b] = root_idx >= 0 ? t_root : __builtin_nan("");
    if (root_idx_out != nullptr) root_idx_out[b] = root_idx;
    if (stats_out != nullptr) {
      stats_out[0 * nb + b] = n_steps;
      stats_out[1 * nb + b] = n_newton;
      stats_out[2 * nb + b] = n_setups;
      stats_out[3 * nb + b] = n_err_fails;
      stats_out[4 * nb + b] = n_nl_fails;
    }
  }
  const unsigned long long mine[6] = {active ? (unsigned long long)n_steps : 0ull, active ? (unsigned long long)n_newton : 0ull,
                                      active ? (unsigned long long)n_setups : 0ull, active ? (unsigned long long)n_err_fails : 0ull,
                                      active ? (unsigned long long)n_nl_fails : 0ull, (active && status != kRsOk) ? 1ull : 0ull};
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const unsigned long long sum = wave_sum_u64(mine[q]);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&totals[q], sum);
  }
}

#if !defined(__HIPCC_RTC__)
// Tableau::tr_bdf2 / esdirk34 (tableau.rs:41-160), column-major a (host side)
inline void fill_tableau(int method, SdirkConsts& T) {
  T.steps_t_out = nullptr; T.steps_cap = 0; T.steps_pad = 0;  // save points unless the caller asks for every step
  for (double& v : T.a) v = 0.0;
  for (double& v : T.beta) v = 0.0;
  if (method == 1) {
    T.s = 3; T.order = 2;
    const double gamma = 2.0 - std::sqrt(2.0), d = gamma / 2.0, w = std::sqrt(2.0) / 4.0;
    const double a[9] = {0.0, d, w, 0.0, d, w, 0.0, 0.0, d};
    for (int i = 0; i < 9; ++i) T.a[i] = a[i];
    T.b[0] = w; T.b[1] = w; T.b[2] = d;
    const double b_hat[3] = {(1.0 - w) / 3.0, (3.0 * w + 1.0) / 3.0, d / 3.0};
    for (int i = 0; i < 3; ++i) T.d[i] = T.b[i] - b_hat[i];
    T.has_beta = 1; T.poly_order = 2;
    const double beta[6] = {2.0 * w, 2.0 * w, gamma - 1.0, -w, -w, 2.0 * w};
    for (int i = 0; i < 6; ++i) T.beta[i] = beta[i];
    T.c[0] = 0.0; T.c[1] = gamma; T.c[2] = 1.0;
    T.gamma = T.a[1 * 3 + 1];
  } else {
    T.s = 4; T.order = 3;
    const double gamma = 0.435866521508459;
    const double a[16] = {0.0, gamma, 0.1407377747247062, 0.102399400619911, 0.0, gamma, -0.1083655513813208, -0.3768784522555561,
                          0.0, 0.0, gamma, 0.8386125301271861, 0.0, 0.0, 0.0, gamma};
    for (int i = 0; i < 16; ++i) T.a[i] = a[i];
    for (int j = 0; j < 4; ++j) T.b[j] = T.a[j * 4 + 3];
    const double c[4] = {0.0, 0.871733043016918, 0.4682387448518444, 1.0};
    const double d[4] = {-0.05462549724041394, -0.49420889362599496, 0.22193449973506466, 0.32689989113134427};
    for (int i = 0; i < 4; ++i) { T.c[i] = c[i]; T.d[i] = d[i]; }
    T.has_beta = 0; T.poly_order = 0;
    T.gamma = T.a[1 * 4 + 1];
  }
}

#endif  // !__HIPCC_RTC__

}  // namespace dsh
