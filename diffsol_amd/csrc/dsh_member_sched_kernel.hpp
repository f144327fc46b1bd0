// Phase-scheduled variant of the per-member device-resident BDF ("divergent-step repacking", BASELINE config 5 names it).
//
// k_bdf_adaptive<.., WAVE = false> runs every member's own control flow in nested loops: a wavefront pays for the UNION of its 64 members' paths — at
// any moment some lanes are in a Newton iteration, some in the error test / order selection, some rescaling the difference array by R U and
// refactoring — 2.6x the instructions of the lock-step kernel (profiles/pmc_resident.json).  Here the same per-member algorithm is a state machine
// with three scheduled phases,
//     NEWTON   one Newton iteration (residual, LU solve, norm, Convergence)
//     ERRTEST  error test; on acceptance the difference update, statistics, order selection, root / stop-time / output handling and the next prediction
//     RESCALE  step-size change (R U rescaling) and / or Jacobian refresh + LU, with its continuation (new prediction, or the post-step handling)
// and every pass of the wavefront executes ONE phase: the one most of its live lanes are waiting for (ballots; ties go to the cheaper phase).  Lanes in
// other phases wait for a pass.  Expensive, rare blocks thus run once for many lanes instead of once per lane that happens to need them.  Every member
// still performs exactly its own sequence of operations, so results, counters and event times are bit-identical to k_bdf_adaptive's and to the oracle's
// (tests/test_gpu_adaptive.py runs both).  All state stays in registers / LDS exactly as in k_bdf_adaptive; no data moves between lanes.
// A simulation of the schedule on the C2 phase statistics (scripts/phase_schedule_sim.py) puts the most-populated-phase rule at ~1.05 M wave
// instructions per wavefront (~1700 passes serving 33 of 64 lanes on average; no other ballot rule does better) against ~0.62 M for the slowest
// single lane (the in-wave bound) and 0.64 M for the lock-step kernel.
// MEASURED (MI355X, C2, 100 000 members): 10.75 ms against 9.23 ms for the nested-loop kernel — the per-pass bookkeeping (ballots, the state that has
// to survive between passes: 304 B of scratch spill at 256 VGPRs) costs more than the schedule saves.  The kernel is therefore OPT-IN
// (DSH_MEMBER_SCHED=1); it stays in the tree, bit-exact and tested, as the measured answer to "bin members by phase inside a wavefront".
#pragma once
#include "dsh_adaptive_kernel.hpp"

namespace dsh {

template <class Mdl, bool BA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DSH_ADAPTIVE_WAVES_PER_EU, DSH_ADAPTIVE_WAVES_PER_EU))) void k_bdf_member_sched(int64_t nb, const double* __restrict__ p_g, const double* __restrict__ atol_g, const AdaptiveConsts* __restrict__ Cp,
                                                    const double* __restrict__ t_eval, double* __restrict__ y_out, int32_t* __restrict__ stats_out,
                                                    int32_t* __restrict__ status_out, double* __restrict__ t_root_out, int32_t* __restrict__ root_idx_out,
                                                    int32_t* __restrict__ ncols_out, unsigned long long* __restrict__ totals) {
  constexpr bool WAVE = false;
  static_assert(model_band_k<Mdl>::value == 0, "the phase-scheduled kernel is the register-resident (n <= 4) form");
  constexpr int N = Mdl::N, NP = Mdl::NP;
  constexpr int NR = Mdl::NROOTS > 0 ? Mdl::NROOTS : 1;
  const AdaptiveConsts& C = *Cp;
  const int64_t bglobal = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = bglobal < nb;  // lanes past the ensemble shadow a live member (no stores) so that the whole wavefront reaches every reduction
  const int64_t b = active ? bglobal : (int64_t)blockIdx.x * blockDim.x;  // shadow the wavefront's first member: invisible in the group max
  const dsh_adaptive_options& o = C.r.o;
  const bool det = o.deterministic_pow != 0;
  const double rtol = C.r.rtol;
  double p[NP], atol[N];
  load_vec<NP>(p_g, nb, b, p);
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) atol[i] = BA ? atol_g[i] : atol_g[(int64_t)i * nb + b];

  // ------------------------------------------------------------ OdeSolverState::new_and_consistent (state.rs:969-997, :1086-1124)
  double t = C.r.t0, h;
  double y[N], f0[N];
  Mdl::init(t, p, y);
  Mdl::rhs(t, y, p, f0);
  int32_t status = kRsOk;
  if (!group_all<WAVE>(set_consistent<Mdl, WAVE>(t, p, y, f0, atol, rtol, C.r))) status = kRsInitialConditionDidNotConverge;
  h = initial_step_size<Mdl, WAVE>(t, C.r.h0, y, f0, p, atol, rtol, 1, det);

  // ------------------------------------------------------------ Bdf::_new (bdf.rs:244-368) + BdfState::initialise_diff_to_first_order
  int order = 1;
  // D lives in registers; its swap partner (bdf.rs `diff_tmp`, touched only when the step size changes) and the cached Jacobian (touched only
  // when refactoring) live in LDS, one column of 64 lanes per value: that keeps the kernel at two wavefronts per SIMD.
  // Banded models (Mdl::BAND_K, n up to 64): nothing of that fits registers / LDS — D, its swap partner, the band of the Jacobian and the banded LU
  // factors are per-lane arrays (scratch memory: the hardware interleaves it by lane, so every access is a coalesced 512-byte transaction).
  constexpr int BK = model_band_k<Mdl>::value;
  constexpr bool BANDED = BK > 0;
  static_assert(!BANDED || !Mdl::HAS_MASS, "banded device-resident models need an identity mass matrix");
  constexpr int LN = BANDED ? 1 : N;
  __shared__ double sDt[kNC * LN][64];
  __shared__ double sJ[LN * LN][64];
  const int ln = threadIdx.x;
  // Bdf::_new tables in LDS: every lookup is indexed by the current order and sits in the serial chain of the step (h alpha_order, the error
  // constants, the R U rescaling) — an LDS read instead of a global load there.
  __shared__ double sAlpha[6], sGamma[6], sEc2[6], sU[kMaxOrder * 36];
  if (ln < 6) { sAlpha[ln] = C.alpha[ln]; sGamma[ln] = C.gamma[ln]; sEc2[ln] = C.ec2[ln]; }
  for (int k = ln; k < kMaxOrder * 36; k += 64) sU[k] = C.u[k / 36][k % 36];
  __syncthreads();
  double Dt_p[BANDED ? kNC : 1][BANDED ? N : 1];
  double Jb[BANDED ? (2 * BK + 1) * N : 1], Lf[BANDED ? BK * N : 1], Uf[BANDED ? (2 * BK + 1) * N : 1];
  auto dt_get = [&](int j, int i) __attribute__((always_inline)) -> double { if constexpr (BANDED) return Dt_p[j][i]; else return sDt[j * N + i][ln]; };
  auto dt_set = [&](int j, int i, double v) __attribute__((always_inline)) { if constexpr (BANDED) Dt_p[j][i] = v; else sDt[j * N + i][ln] = v; };
  double D[kNC][N];
#pragma unroll
  for (int j = 0; j < kNC; ++j)
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) { D[j][i] = 0.0; dt_set(j, i, 0.0); }
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) { D[0][i] = y[i]; D[1][i] = f0[i] * h; }
  double opc = h * sAlpha[1];  // BdfCallable::c
  double A[BANDED ? 1 : N * N];
  int P[N];
  bool jac_stale = true;
  // statistics (ode_solver/mod.rs:28-69)
  int n_setups = 0, n_steps = 0, n_err_fails = 0, n_newton = 0, n_nl_fails = 0;
  // NonLinearSolver::reset_jacobian: M - c f'(x)  (op/bdf.rs:273-300) + LU
  auto reset_jacobian = [&](const double (&xx)[N], double tt) __attribute__((always_inline)) {
    if constexpr (BANDED) {
      if (jac_stale) { Mdl::jac_band(tt, xx, p, Jb); jac_stale = false; }
      bool sing = false;
      band_factor_lane<N, BK>(Jb, opc, Lf, Uf, P, sing);
    } else {
    double J[N * N];
    if (jac_stale) {
      assemble_jacobian<Mdl>(tt, xx, p, J);
#pragma unroll
      for (int e = 0; e < N * N; ++e) sJ[e][ln] = J[e];
      jac_stale = false;
    } else {
#pragma unroll
      for (int e = 0; e < N * N; ++e) J[e] = sJ[e][ln];
    }
    double Mm[N * N];
    if constexpr (Mdl::HAS_MASS) assemble_mass<Mdl>(tt, p, Mm);
    else {
#pragma unroll
      for (int e = 0; e < N * N; ++e) Mm[e] = (e / N == e % N) ? 1.0 : 0.0;  // Matrix::from_diagonal(ones), op/bdf.rs:138-141
    }
#pragma unroll
    for (int e = 0; e < N * N; ++e) A[e] = J[e] * (-opc) + Mm[e];
    bool sing = false;
    lu_factor_reg<N>(A, P, sing);
    }
  };
  reset_jacobian(y, t);
  n_setups = 1;
  // RootFinder::init (root.rs:44-49)
  double g0[NR] = {0.0};
  double rf_t0 = t;
  if constexpr (Mdl::NROOTS > 0) Mdl::root(t, y, p, g0);
  double t_root = 0.0;
  int root_idx = -1;
  // JacobianUpdate (jacobian_update.rs:12-36)
  int steps_since_jac = 0, steps_since_rhs_jac = 0;
  double h_at_last_jac = 1.0;
  // Convergence (convergence.rs:7-57)
  double eta = C.r.eta_reset;
  int n_equal_steps = 0;
  bool has_prev_err = false;
  double prev_err = 0.0;
  double yp[N], psi[N];
  double t_predict = t;

  // _update_step_size (bdf.rs:508-566) with _update_diff_for_step_size (:568-577): diff_tmp[:, 0..=order] = diff[:, 0..=order] * (R U); swap
  auto update_step_size = [&](double factor, double& new_h_out) __attribute__((always_inline)) -> bool {
    const double new_h = factor * h;
    n_equal_steps = 0;
    double R[6][6];  // R[j][i] = element (row i, col j) of compute_r(order, factor)   (bdf.rs:433-463)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      R[j][0] = 1.0;
#pragma unroll
      for (int i = 1; i < 6; ++i) R[j][i] = (j == 0) ? 0.0 : R[j][i - 1] * ((double)i - 1.0 - factor * (double)j) / (double)i;
    }
    const double* U = sU + (order - 1) * 36;  // element (row k, col j) at U[j*6 + k]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (j <= order) {
        // column j of RU: ru[k] = sum_m R(k,m) U(m,j), gemm order: first term, then acc = a*b + acc
        double ru[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          double acc = R[0][k] * U[j * 6 + 0];
#pragma unroll
          for (int m = 1; m < 6; ++m) if (m <= order) acc = R[m][k] * U[j * 6 + m] + acc;
          ru[k] = acc;
        }
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) {
          double acc = D[0][i] * ru[0];
#pragma unroll
          for (int k = 1; k < 6; ++k) if (k <= order) acc = D[k][i] * ru[k] + acc;
          dt_set(j, i, acc);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kNC; ++j)
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) { const double tmp = D[j][i]; D[j][i] = dt_get(j, i); dt_set(j, i, tmp); }
    opc = new_h * sAlpha[order];
    h = new_h;
    eta = C.r.eta_reset_ts;  // reset_eta_timestep_change
    new_h_out = new_h;
    return fabs(h) < o.min_timestep;  // true = StepSizeTooSmall
  };

  // _predict_forward (bdf.rs:674-692): y_predict = sum_{j<=order} D_j ; psi_neg_y0 = alpha_order * sum_{1<=j<=order} gamma_j D_j - y_predict
  auto predict_forward = [&]() __attribute__((always_inline)) {
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) if (j <= order) s = s + D[j][i];
      double q = sGamma[1] * D[1][i];
#pragma unroll
      for (int j = 2; j < 6; ++j) if (j <= order) q = sGamma[j] * D[j][i] + 1.0 * q;
      q = q * sAlpha[order];
      q = q - s;
      yp[i] = s;
      psi[i] = q;
    }
    t_predict = t + h;
  };

  // _jacobian_updates (bdf.rs:465-506) over JacobianUpdate::check_* (jacobian_update.rs:38-79)
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
      reset_jacobian(y, t);
      steps_since_rhs_jac = 0; steps_since_jac = 0; h_at_last_jac = c;  // update_rhs_jacobian, then update_jacobian
      eta = C.r.eta_reset;
      n_setups++;
    } else if (check_jac) {
      reset_jacobian(y, t);
      steps_since_jac = 0; h_at_last_jac = c;
      eta = C.r.eta_reset;
      n_setups++;
    }
  };

  // handle_tstop (bdf.rs:694-731): 0 = nothing, 1 = TstopReached, 2 = StopTimeBeforeCurrentTime
  bool has_tstop = true;
  const double tstop = t_eval[C.r.n_eval - 1];
  auto handle_tstop = [&]() __attribute__((always_inline)) -> int {
    const double eps = 2.220446049250313e-16;
    const double troundoff = 100.0 * eps * (fabs(t) + fabs(h));
    if (fabs(t - tstop) <= troundoff) { has_tstop = false; return 1; }
    if ((h > 0.0 && tstop < t - troundoff) || (h < 0.0 && tstop > t + troundoff)) { has_tstop = false; return 2; }
    if ((h > 0.0 && t + h > tstop + troundoff) || (h < 0.0 && t + h < tstop - troundoff)) {
      const double factor = (tstop - t) / h;
      double nh;
      (void)update_step_size(factor, nh);  // "step size too small" is ignored here like in the reference
    }
    return 0;
  };

  int col = 0;
  double te_next = t_eval[0];  // t_eval[col], kept in a register: it is compared after every step
  // solve_dense (method.rs:467-520): t_eval[0] >= t0 is checked on the host; set_stop_time(t_eval.last())
  {
    const int r = handle_tstop();
    if (r == 1) status = kRsStopTimeAtCurrentTime;
    else if (r == 2) status = kRsStopTimeBeforeCurrentTime;
  }


  enum : int { PH_NEWTON = 0, PH_ERRTEST = 1, PH_RESCALE = 2, PH_DONE = 3 };
  enum : int { RK_NLFAIL1 = 0, RK_NLFAIL2 = 1, RK_ERRFAIL = 2, RK_STEPOK = 3 };
  // per-step state of the member (locals of Bdf::step in the nested-loop kernel)
  double x[N];
  int niter = 0;
  bool has_old = false;
  double old_norm = 0.0;
  bool convergence_fail = false;
  int old_err_fails = 0;
  double safety = 0.0, error_norm = 0.0;
  double rs_factor = 1.0;
  int rs_kind = RK_NLFAIL1;
  long guard = 0;
  int phase = (status != kRsOk || !active) ? PH_DONE : PH_NEWTON;

  // interpolate_from_diff (bdf.rs:767-782)
  auto interpolate = [&](double te, double (&yv)[N]) __attribute__((always_inline)) {
    double time_factor = 1.0;
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) yv[i] = D[0][i];
#pragma unroll
    for (int j = 0; j < kMaxOrder; ++j) {
      if (j < order) {
        const double jt = (double)j;
        time_factor *= (te - (t - h * jt)) / (h * (1.0 + jt));
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) yv[i] = time_factor * D[j + 1][i] + 1.0 * yv[i];
      }
    }
  };

  auto restart_newton = [&]() __attribute__((always_inline)) {
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) x[i] = yp[i];
    niter = 0;
    has_old = false;
  };
  // top of Bdf::step (bdf.rs:1277-1322): guard, per-step locals, first prediction
  auto begin_step = [&]() __attribute__((always_inline)) {
    if (++guard > o.max_steps) { status = kRsMaxStepsExceeded; phase = PH_DONE; return; }
    safety = 0.0; error_norm = 0.0;
    old_err_fails = n_err_fails;
    convergence_fail = false;
    predict_forward();
    restart_newton();
    phase = PH_NEWTON;
  };
  // everything of the step after the (optional) step-size change of the order selection: root check, stop time, solve_dense output, next step
  auto post_accept = [&]() __attribute__((always_inline)) {
    int reason = 0;  // 0 internal, 1 tstop, 3 root
    if constexpr (Mdl::NROOTS > 0) {
      const int rr = check_root<Mdl, WAVE>(g0, rf_t0, y, t, p, interpolate, t_root, root_idx);
      if (rr == 1) reason = 3;
    }
    if (reason == 0 && has_tstop) reason = handle_tstop();
    if (reason == 2) reason = 0;
    const double upto = reason == 3 ? t_root : t;
    while (col < C.r.n_eval && te_next <= upto) {
      double yv[N];
      interpolate(te_next, yv);
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) y_out[((int64_t)col * N + i) * nb + b] = yv[i];
      col++;
      if (col < C.r.n_eval) te_next = t_eval[col];
    }
    if (reason == 3) {
      if (col < C.r.n_eval) {
        double yv[N];
        interpolate(t_root, yv);
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) y_out[((int64_t)col * N + i) * nb + b] = yv[i];
        col++;
      }
      phase = PH_DONE;
    } else if (reason == 1) {
      phase = PH_DONE;
    } else {
      begin_step();
    }
  };

  if (phase != PH_DONE) begin_step();
  while (true) {
    const unsigned long long mN = __ballot(phase == PH_NEWTON), mE = __ballot(phase == PH_ERRTEST), mR = __ballot(phase == PH_RESCALE);
    if ((mN | mE | mR) == 0ull) break;
    const int cN = __popcll(mN), cE = __popcll(mE), cR = __popcll(mR);
    const int pick = (cN >= cE && cN >= cR) ? PH_NEWTON : (cE >= cR ? PH_ERRTEST : PH_RESCALE);  // wavefront-uniform
    if (pick == PH_NEWTON) {
      if (phase == PH_NEWTON) {
        // ---- one iteration of NewtonNonlinearSolver::solve_in_place over NoLineSearch (newton.rs:13-36, line_search.rs:46-72)
        double f[N], delta[N], tmpv[N];
        Mdl::rhs(t_predict, x, p, f);
        if constexpr (Mdl::HAS_MASS) {
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) tmpv[i] = x[i] + psi[i];
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) delta[i] = f[i];
          Mdl::mass_gemv(t_predict, tmpv, p, -opc, delta);
        } else {
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) delta[i] = 1.0 * (x[i] + psi[i]) + (-opc) * f[i];
        }
        const bool lu_ok = lu_solve_reg<N>(A, P, delta);
        bool failed = !lu_ok, converged = false;
        if (lu_ok) {
          double acc = 0.0;
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) {
            const double d = delta[i];
            x[i] = x[i] - d;
            const double term = d / (fabs(yp[i]) * rtol + atol[i]);
            acc += term * term;
          }
          const double norm = sqrt(acc / (double)N);
          niter += 1;
          bool diverged = false;
          if (has_old) {
            const double rate = niter == 2 ? norm / old_norm : rpow(norm / old_norm, 1.0 / (double)(niter - 1), det);
            if (rate > 0.9) diverged = true;
            else if (powi_rt(rate, o.max_nonlinear_solver_iterations - niter) / (1.0 - rate) * norm > o.nonlinear_solver_tolerance) diverged = true;
            else eta = rate / (1.0 - rate);
          } else {
            const double min_eta = 1e4 * 2.220446049250313e-16;
            if (eta < min_eta) eta = min_eta;
            eta = rpow(eta, 0.8, det);
          }
          converged = !diverged && eta * norm < o.nonlinear_solver_tolerance;
          if (niter == 1) { has_old = true; old_norm = norm; }
          failed = diverged || (!converged && niter >= o.max_nonlinear_solver_iterations);
        }
        if (converged) {
          n_newton += niter;
          phase = PH_ERRTEST;
        } else if (failed) {
          n_newton += niter;
          n_nl_fails += 1;
          if (n_nl_fails > o.max_nonlinear_solver_failures) { status = kRsTooManyNonlinearSolverFailures; phase = PH_DONE; }
          else {
            has_prev_err = false;
            rs_kind = convergence_fail ? RK_NLFAIL2 : RK_NLFAIL1;
            rs_factor = 0.3;
            phase = PH_RESCALE;
          }
        }
      }
    } else if (pick == PH_ERRTEST) {
      if (phase == PH_ERRTEST) {
        double ydelta[N];
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) ydelta[i] = x[i] - yp[i];
        // error_control (bdf.rs:812-843): norm against the CURRENT state y
        error_norm = fmax(0.0, wms<N>(ydelta, y, atol, rtol) * sEc2[order - 1]);
        const double maxiter = (double)o.max_nonlinear_solver_iterations;
        safety = 0.9 * (2.0 * maxiter + 1.0) / (2.0 * maxiter + (double)niter);
        if (error_norm <= 1.0) {
          // ---- accepted: _update_diff (bdf.rs:646-664), state update
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) {
            double dk1 = 0.0;
#pragma unroll
            for (int j = 2; j < 7; ++j) if (j == order + 1) dk1 = D[j][i];
            const double dk2 = ydelta[i] - dk1;
#pragma unroll
            for (int j = 2; j < kNC; ++j) { if (j == order + 2) D[j][i] = dk2; if (j == order + 1) D[j][i] = ydelta[i]; }
            double upper = ydelta[i];
#pragma unroll
            for (int j = 5; j >= 0; --j) if (j <= order) { const double v = D[j][i] + 1.0 * upper; D[j][i] = v; upper = v; }
            y[i] = yp[i];
          }
          t = t_predict;
          n_steps += 1;
          steps_since_jac += 1; steps_since_rhs_jac += 1;
          prev_err = error_norm; has_prev_err = true;
          n_equal_steps += 1;
          bool rescale = false;
          if (n_equal_steps > order) {
            // order selection (bdf.rs:1494-1560)
            double col_m[N], col_p[N];
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) {
              double vm = 0.0, vp = 0.0;
#pragma unroll
              for (int j = 1; j < kNC; ++j) { if (j == order) vm = D[j][i]; if (j == order + 2) vp = D[j][i]; }
              col_m[i] = vm; col_p[i] = vp;
            }
            const double inf = __builtin_huge_val();
            const double error_m_norm = order > 1 ? wms<N>(col_m, y, atol, rtol) * sEc2[order - 1] : inf;
            const double error_p_norm = order < kMaxOrder ? wms<N>(col_p, y, atol, rtol) * sEc2[order + 1] : inf;
            const double pi_i = o.pi_control_integral, pi_p = o.pi_control_proportional;
            const double f0c = pi_controller_raw(error_m_norm, has_prev_err, prev_err, pi_i, pi_p, order, det);
            const double f1c = pi_controller_raw(error_norm, has_prev_err, prev_err, pi_i, pi_p, order + 1, det);
            const double f2c = pi_controller_raw(error_p_norm, has_prev_err, prev_err, pi_i, pi_p, order + 2, det);
            int max_index = 0;
            double fmaxv = f0c;
            if (f1c >= fmaxv) { max_index = 1; fmaxv = f1c; }
            if (f2c >= fmaxv) { max_index = 2; fmaxv = f2c; }
            order = max_index == 0 ? order - 1 : (max_index == 1 ? order : order + 1);
            double factor = safety * fmaxv;
            if (factor > o.max_timestep_growth) factor = o.max_timestep_growth;
            if (factor < o.min_timestep_shrink) factor = o.min_timestep_shrink;
            if (factor >= o.min_timestep_growth || factor <= o.max_timestep_shrink || max_index == 0 || max_index == 2) {
              rescale = true;
              rs_factor = factor;
              rs_kind = RK_STEPOK;
              phase = PH_RESCALE;
            }
          }
          if (!rescale) post_accept();
        } else {
          double factor = safety * pi_controller_raw(error_norm, has_prev_err, prev_err, o.pi_control_integral, o.pi_control_proportional, order + 1, det);
          has_prev_err = false;
          if (factor < o.min_timestep_shrink) factor = o.min_timestep_shrink;
          rs_factor = factor;
          rs_kind = RK_ERRFAIL;
          phase = PH_RESCALE;
        }
      }
    } else {
      if (phase == PH_RESCALE) {
        // ---- _update_step_size (all kinds but the first convergence failure) and _jacobian_updates: ONE site each for the whole wavefront
        double new_h = h;
        bool too_small = false;
        if (rs_kind != RK_NLFAIL1) too_small = update_step_size(rs_factor, new_h);
        if (too_small) { status = kRsStepSizeTooSmall; phase = PH_DONE; }
        else {
          const JState st = rs_kind == RK_NLFAIL1 ? JState::FirstConvergenceFail : (rs_kind == RK_NLFAIL2 ? JState::SecondConvergenceFail : (rs_kind == RK_ERRFAIL ? JState::ErrorTestFail : JState::StepSuccess));
          jacobian_updates(new_h * sAlpha[order], st);
          if (rs_kind == RK_STEPOK) post_accept();
          else {
            if (rs_kind == RK_NLFAIL1) convergence_fail = true;
            else predict_forward();
            bool fail = false;
            if (rs_kind == RK_ERRFAIL) {
              n_err_fails += 1;
              if (n_err_fails - old_err_fails >= o.max_error_test_failures) { status = kRsTooManyErrorTestFailures; phase = PH_DONE; fail = true; }
            }
            if (!fail) { restart_newton(); phase = PH_NEWTON; }
          }
        }
      }
    }
  }
  if (active) {
    if (ncols_out != nullptr) ncols_out[b] = col;
    if (t_root_out != nullptr) t_root_out[b] = root_idx >= 0 ? t_root : __builtin_nan("");
    if (root_idx_out != nullptr) root_idx_out[b] = root_idx;
    // columns that were never reached (root stop or error exit): NaN
    for (; col < C.r.n_eval; ++col)
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) y_out[((int64_t)col * N + i) * nb + b] = __builtin_nan("");
    if (status_out != nullptr) status_out[b] = status;
    if (stats_out != nullptr) {
      stats_out[0 * nb + b] = n_steps;
      stats_out[1 * nb + b] = n_newton;
      stats_out[2 * nb + b] = n_setups;
      stats_out[3 * nb + b] = n_err_fails;
      stats_out[4 * nb + b] = n_nl_fails;
    }
  }
  // ensemble totals: wavefront sums, one atomic per wavefront and counter
  const unsigned long long mine[6] = {active ? (unsigned long long)n_steps : 0ull, active ? (unsigned long long)n_newton : 0ull,
                                      active ? (unsigned long long)n_setups : 0ull, active ? (unsigned long long)n_err_fails : 0ull,
                                      active ? (unsigned long long)n_nl_fails : 0ull, (active && status != kRsOk) ? 1ull : 0ull};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const unsigned long long sum = wave_sum_u64(mine[k]);
    if ((threadIdx.x & 63) == 0 && sum) atomicAdd(&totals[k], sum);
  }
}

}  // namespace dsh
