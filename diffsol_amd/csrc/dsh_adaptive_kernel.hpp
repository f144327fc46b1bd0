// Device-resident variable-order BDF integrator (launch code and documentation: dsh_adaptive.hip).  In a header so that run-time-compiled model
// modules (dsh_jit.hip) instantiate the same kernel for user models.
#pragma once
#include "dsh_resident.hpp"

namespace dsh {

constexpr int kMaxOrder = 5;
constexpr int kNC = kMaxOrder + 3;  // columns of the difference array

struct AdaptiveConsts {
  ResidentConsts r;
  double alpha[6], gamma[6], ec2[6];  // Bdf::_new tables (bdf.rs:286-306), computed on the host
  double u[kMaxOrder][36];            // compute_r(order, 1.0), 6x6 column-major (unused entries 0)
  double eta_reset_p08, eta_reset_ts_p08;  // dsh_det_pow(eta_reset, 0.8), dsh_det_pow(eta_reset_ts, 0.8): the first Newton iteration after a reset (appended: older code objects ignore them)
  // Segmented per-member runs (k_bdf_adaptive<.., SEG = true>, dsh_adaptive.hip): the launch integrates every member until it has produced seg_col_end
  // save points, writes the member's whole integrator state to seg_dbl / seg_int ([slot][member]) and a sort key to seg_key; the next launch resumes from
  // it with the members dealt to lanes in a new order (seg_lane_member).  Nothing of the arithmetic depends on where a launch ends.
  double* seg_dbl;
  int* seg_int;
  unsigned long long* seg_key;
  const int* seg_lane_member;
  int seg_fresh, seg_last, seg_col_end, seg_step_budget;  // seg_step_budget > 0: the launch also ends for a member after that many trips of its step loop
  unsigned int* seg_remaining;                             // number of members that are not finished when the launch ends (one atomic per wavefront)
  // Forward sensitivities (k_bdf_adaptive<.., SENS = true>; problem.bdf_sens(), bdf.rs:370-432, :934-989): s_j = dy/dp_j of every parameter integrated
  // alongside.  sens_out: n_eval x NP x N x nb (device, batch-fastest); sens_error_control != 0: the sensitivities take part in the error test and in the
  // order selection with sens_rtol / sens_atol (the same for every parameter and member; builder.rs build_atols), else turn_off_sensitivities_error_control.
  double* sens_out;
  double sens_rtol, sens_atol[4];
  int sens_error_control, sens_pad;
  // OdeSolverMethod::solve (method.rs:227-258 over :881-961): steps_cap > 0 makes the kernel write the state after EVERY accepted step instead of interpolating at
  // save points — column 0 = (t0, y0), then one column per internal step, the last one at the stop time t_eval[0] (or at a member's event) — into
  // y_out [steps_cap][N][nb] and steps_t_out [steps_cap][nb]; ncols_out = the columns the member produced (columns beyond steps_cap are counted, not stored).
  double* steps_t_out;
  int steps_cap, steps_pad;
};
constexpr int kSegDbl = 128, kSegInt = 32;  // slots per member (upper bounds for n <= 4)

// `if (c) body` that STAYS a branch.  In wavefront lock-step groups the BDF order is wavefront-uniform (a scalar register), and the loops over the
// difference columns test it per column: the compiler turns such tiny guarded blocks into selects (v_cndmask on every 32-bit half: 56 % of the
// kernel's VALU instructions were not FP64, profiles/r02) although a scalar branch would skip them for nothing.  An empty volatile asm inside the block
// keeps it a block.  The arithmetic and its order per component are untouched.  Per-member control (order differs by lane) keeps the select form.
template <bool WAVE, class F>
__device__ __forceinline__ void guarded(bool c, F&& f) {
  if (c) {
    if constexpr (WAVE) asm volatile("" ::);
    f();
  }
}

// wavefronts per SIMD the kernel is compiled for: 2 for the register-resident models (256 VGPRs hold the whole BDF state); the run-time-compiled banded
// form, whose state lives in per-lane memory anyway, overrides it (dsh_jit.hip) to trade registers for latency hiding
#ifndef DSH_ADAPTIVE_WAVES_PER_EU
#define DSH_ADAPTIVE_WAVES_PER_EU 2
#endif
// occupancy experiment (profiles/r04_headline_occupancy.md): D's swap partner in per-lane memory instead of LDS (12 of the 19 KB a wavefront of the Robertson
// kernel holds), so that LDS allows more than two wavefronts per SIMD when DSH_ADAPTIVE_WAVES_PER_EU asks for them
#ifndef DSH_ADAPTIVE_DT_PRIVATE
#define DSH_ADAPTIVE_DT_PRIVATE 0
#endif
// FAST (opt-in, dsh_adaptive_options::deterministic_pow == 2; instantiated only in dsh_adaptive_fast.hip, which is compiled with -ffp-contract=fast and
// reciprocal-math division): ocml's pow, fused multiply-adds, the Newton norm's weights as reciprocals computed once per solve.  NOT bit-comparable with the
// oracle: held to 1e-6 relative on the states at tight tolerances (tests/test_gpu_adaptive.py).
template <class Mdl, bool BA, bool WAVE, bool SEG = false, bool SENS = false, bool FAST = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DSH_ADAPTIVE_WAVES_PER_EU, DSH_ADAPTIVE_WAVES_PER_EU))) void k_bdf_adaptive(int64_t nb, const double* __restrict__ p_g, const double* __restrict__ atol_g, const AdaptiveConsts* __restrict__ Cp,
                                                    const double* __restrict__ t_eval, double* __restrict__ y_out, int32_t* __restrict__ stats_out,
                                                    int32_t* __restrict__ status_out, double* __restrict__ t_root_out, int32_t* __restrict__ root_idx_out,
                                                    int32_t* __restrict__ ncols_out, unsigned long long* __restrict__ totals) {
  // @phase set-up (init, consistent state, first factorisation)
  constexpr int N = Mdl::N, NP = Mdl::NP;
  constexpr int NR = Mdl::NROOTS > 0 ? Mdl::NROOTS : 1;
  const AdaptiveConsts& C = *Cp;
  const int ML = (!WAVE && !SEG && C.r.member_lanes > 0) ? C.r.member_lanes : 64;  // members per wavefront under per-member control (see ResidentConsts)
  const int64_t bglobal = (int64_t)blockIdx.x * ML + threadIdx.x;
  const bool active = (int)threadIdx.x < ML && bglobal < nb;  // the other lanes shadow a live member (no stores) so that the whole wavefront reaches every reduction
  int64_t b_ = active ? bglobal : (int64_t)blockIdx.x * ML;  // shadow the wavefront's first member: invisible in the group max
  if constexpr (SEG) { if (C.seg_lane_member != nullptr) b_ = C.seg_lane_member[b_]; }  // segmented runs: the member this lane works for in this launch
  const int64_t b = b_;
  const bool fresh = !SEG || C.seg_fresh != 0;  // not a resumed launch
  const dsh_adaptive_options& o = C.r.o;
  const bool det = !FAST && o.deterministic_pow != 0;
  const double rtol = C.r.rtol;
  double p[NP], atol[N];
  load_vec<NP>(p_g, nb, b, p);
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) atol[i] = BA ? atol_g[i] : atol_g[(int64_t)i * nb + b];

  // ------------------------------------------------------------ OdeSolverState::new_and_consistent (state.rs:969-997, :1086-1124)
  double t = C.r.t0, h = 0.0;
  double y[N], f0[N];
  int32_t status = kRsOk;
  if (fresh) {
    Mdl::init(t, p, y);
    Mdl::rhs(t, y, p, f0);
    if (!group_all<WAVE>(set_consistent<Mdl, WAVE>(t, p, y, f0, atol, rtol, C.r))) status = kRsInitialConditionDidNotConverge;
    h = initial_step_size<Mdl, WAVE>(t, C.r.h0, y, f0, p, atol, rtol, 1, det);
  } else {
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) { y[i] = 0.0; f0[i] = 0.0; }
  }

  // ------------------------------------------------------------ Bdf::_new (bdf.rs:244-368) + BdfState::initialise_diff_to_first_order
  int order = 1;
  // D lives in registers; its swap partner (bdf.rs `diff_tmp`, touched only when the step size changes) and the cached Jacobian (touched only
  // when refactoring) live in LDS, one column of 64 lanes per value: that keeps the kernel at two wavefronts per SIMD.
  // Banded models (Mdl::BAND_K, n up to 64): nothing of that fits registers / LDS — D, its swap partner, the band of the Jacobian and the banded LU
  // factors are per-lane arrays (scratch memory: the hardware interleaves it by lane, so every access is a coalesced 512-byte transaction).
  constexpr int BK = model_band_k<Mdl>::value;
  constexpr bool BANDED = BK > 0;
  static_assert(!BANDED || !Mdl::HAS_MASS, "banded device-resident models need an identity mass matrix");
  // hybrid models (a reset operator: OdeEquations::reset, DiffSL reset_i) apply the reset at every event and go on, as the reference's solve_dense does
  // (round 5: also with a mass matrix — hybrid DAEs: the state is made consistent again after the reset, apply_reset_with_mass, state.rs:279-306)
  constexpr bool kResets = model_has_reset<Mdl>::value && !BANDED && Mdl::NROOTS > 0;
  constexpr bool DTP = BANDED || DSH_ADAPTIVE_DT_PRIVATE != 0;  // the swap partner of D in per-lane memory
  constexpr int LN = BANDED ? 1 : N;
  __shared__ double sDt[DTP ? 1 : kNC * LN][64];
  __shared__ double sJ[LN * LN][64];
  const int ln = threadIdx.x;
  // Bdf::_new tables in LDS: every lookup is indexed by the current order and sits in the serial chain of the step (h alpha_order, the error
  // constants, the R U rescaling) — an LDS read instead of a global load there.
  __shared__ double sAlpha[6], sGamma[6], sEc2[6], sU[kMaxOrder * 36];
  __shared__ double sRw[48], sRUw[48];  // R and R U of a step-size change in wavefront lock-step groups (a workgroup is one wavefront)
  if (ln < 6) { sAlpha[ln] = C.alpha[ln]; sGamma[ln] = C.gamma[ln]; sEc2[ln] = C.ec2[ln]; }
  for (int k = ln; k < kMaxOrder * 36; k += 64) sU[k] = C.u[k / 36][k % 36];
  __syncthreads();
  double Dt_p[DTP ? kNC : 1][DTP ? N : 1];
  double Jb[BANDED ? (2 * BK + 1) * N : 1], Lf[BANDED ? BK * N : 1], Uf[BANDED ? (2 * BK + 1) * N : 1];
  auto dt_get = [&](int j, int i) __attribute__((always_inline)) -> double { if constexpr (DTP) return Dt_p[j][i]; else return sDt[j * N + i][ln]; };
  auto dt_set = [&](int j, int i, double v) __attribute__((always_inline)) { if constexpr (DTP) Dt_p[j][i] = v; else sDt[j * N + i][ln] = v; };
  double D[kNC][N];
#pragma unroll
  for (int j = 0; j < kNC; ++j)
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) { D[j][i] = 0.0; dt_set(j, i, 0.0); }
DSH_UNROLL_N
  for (int i = 0; i < N; ++i) { D[0][i] = y[i]; D[1][i] = f0[i] * h; }
  double opc = h * sAlpha[1];  // BdfCallable::c
  // ---- forward sensitivities: new_with_sensitivities_and_consistent (state.rs:1032-1083: s_j = SensInit(t0) e_j, ds_j = SensRhs(s_j) about (y0, t0)),
  // new_augmented (bdf.rs:384-432): sdiff_j[:, 0] = s_j, sdiff_j[:, 1] = h ds_j.  Per-lane arrays in scratch memory (indexed by the parameter at run time).
  constexpr int NPS = SENS ? NP : 1, NCS = SENS ? kNC : 1, NS = SENS ? N : 1;
  // banded lane-per-member models (model_band_k > 0: the state in per-lane memory) take the same code: the sensitivity arrays are per-lane memory too, the
  // sensitivity solves run on the banded factors (bdf.rs:934-989 for the battery / PDE models)
  static_assert(!SENS || (!Mdl::HAS_MASS && Mdl::NROOTS == 0 && !SEG),
                "device-resident forward sensitivities: identity-mass ODE models without root functions");
  double S[NPS][NCS][NS];    // sdiff
  double s_cur[NPS][NS];     // state.s: the latest solution of the sensitivity equations (the scale of the sensitivity error norms)
  double s_delta[NPS][NS];   // s - s_predict of the last sensitivity_solve
  double s_c = 0.0;          // BdfCallable::c of the sensitivity operator: 0 until the first _update_step_size (new_no_jacobian, bdf.rs:403; op/bdf.rs:61)
  double s_atol[NS];
  if constexpr (SENS) {
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) s_atol[i] = C.sens_atol[C.sens_pad != 0 ? 0 : (i < 4 ? i : 0)];  // sens_pad = 1: one value for every state (the only form for n > 4)
    for (int j = 0; j < NP; ++j) {
      double ev[NP], s0[N], jm[N], dfdp[N];
#pragma unroll
      for (int k = 0; k < NP; ++k) ev[k] = k == j ? 1.0 : 0.0;
      Mdl::init_sens_mul(t, p, ev, s0);
      Mdl::sens_mul(t, y, p, ev, dfdp);   // SensRhs::update_state(y0, t0): column j of df/dp
      Mdl::jac_mul(t, y, p, s0, jm);      // SensRhs::call_inplace: J(y0) s_j + (df/dp)_j
      for (int k = 0; k < kNC; ++k)
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) S[j][k][i] = 0.0;
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) {
        const double ds = jm[i] + dfdp[i];
        S[j][0][i] = s0[i];
        S[j][1][i] = ds * h;
        s_cur[j][i] = s0[i];
        s_delta[j][i] = 0.0;
      }
    }
  }
  double A[BANDED ? 1 : N * N];
  double Adinv[(FAST && !BANDED) ? N : 1];  // fast variant: reciprocals of U's diagonal, once per factorisation
  double wyinv[FAST ? N : 1];               // fast variant: 1 / (|y_i| rtol + atol_i) for the current state (error test, order selection)
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
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < N; ++i) Adinv[i] = 1.0 / A[i * N + i];
    }
    }
  };
  if (fresh) {
    reset_jacobian(y, t);
    n_setups = 1;
  }
  // RootFinder::init (root.rs:44-49)
  double g0[NR] = {0.0};
  double rf_t0 = t;
  if constexpr (Mdl::NROOTS > 0) { if (fresh) Mdl::root(t, y, p, g0); }
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

  // fast variant: the weighted mean square against the current state with the cached reciprocal weights
  auto wms_y = [&](const double (&v)[N]) __attribute__((always_inline)) -> double {
    if constexpr (FAST) {
      double acc = 0.0;
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) { const double term = v[i] * wyinv[i]; acc += term * term; }
      return acc / (double)N;
    } else return wms<N>(v, y, atol, rtol);
  };
  auto refresh_wyinv = [&]() __attribute__((always_inline)) {
    if constexpr (FAST) {
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) wyinv[i] = 1.0 / (fabs(y[i]) * rtol + atol[i]);
    }
  };
  refresh_wyinv();

  // @phase update_step_size (R, R U, D <- D R U)
  // _update_step_size (bdf.rs:508-566) with _update_diff_for_step_size (:568-577): diff_tmp[:, 0..=order] = diff[:, 0..=order] * (R U); swap
  auto update_step_size = [&](double factor, double& new_h_out) __attribute__((always_inline)) -> bool {
    const double new_h = factor * h;
    n_equal_steps = 0;
    if constexpr (WAVE) {
      // R and R U are wavefront-uniform 6 x 6 matrices: every lane used to compute all of both (25 recurrences with 10 true divisions, 216 multiply-add
      // pairs).  Here lane m computes column m of R (its recurrence down the rows), lane k + 8 j the entry (k, j) of R U, both through LDS — each
      // entry by the operations, in the order, of the sequential code, so the bits are the same; then every lane applies R U to its own differences.
      const int ou = __builtin_amdgcn_readfirstlane(order);
      {
        const int m = ln & 7;  // lanes 0..5: column m
        double r = 1.0;
        sRw[m * 6 + 0] = r;
#pragma unroll
        for (int i = 1; i < 6; ++i) { r = (m == 0) ? 0.0 : r * ((double)i - 1.0 - factor * (double)m) / (double)i; if (m < 6) sRw[m * 6 + i] = r; }
      }
      {
        const int k = ln & 7, j = ln >> 3;
        const bool mine = (k < 6) & (j < 6);
        const double* U = sU + (ou - 1) * 36 + (mine ? j : 0) * 6;
        const int kk = mine ? k : 0;
        double acc = sRw[0 * 6 + kk] * U[0];
#pragma unroll
        for (int m = 1; m < 6; ++m)
          guarded<WAVE>(m <= ou, [&]() __attribute__((always_inline)) { acc = sRw[m * 6 + kk] * U[m] + acc; });
        if (mine) sRUw[j * 6 + k] = acc;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j)
        guarded<WAVE>(j <= ou, [&]() __attribute__((always_inline)) {
          double acc[N];
          const double ru0 = sRUw[j * 6 + 0];
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) acc[i] = D[0][i] * ru0;
#pragma unroll
          for (int k = 1; k < 6; ++k)
            guarded<WAVE>(k <= ou, [&]() __attribute__((always_inline)) {
              const double ruk = sRUw[j * 6 + k];
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) acc[i] = D[k][i] * ruk + acc[i];
            });
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) dt_set(j, i, acc[i]);
        });
    } else {
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
          // U = compute_r(order, 1) is stored zero-padded to 6 x 6: the terms m > order add R * (+0) = +-0 to a sum that starts at +0 or 1 and is therefore
          // never -0 — the sum is unchanged bit for bit, and the per-lane test `m <= order` (two selects per term: a third of this function's instructions
          // when every lane has its own order) is not needed
          double acc = R[0][k] * U[j * 6 + 0];
#pragma unroll
          for (int m = 1; m < 6; ++m) acc = R[m][k] * U[j * 6 + m] + acc;
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
    }
#pragma unroll
    for (int j = 0; j < kNC; ++j)
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) { const double tmp = D[j][i]; D[j][i] = dt_get(j, i); dt_set(j, i, tmp); }
    if constexpr (SENS) {
      // bdf.rs:546-548: every sdiff goes through the SAME scratch matrix as the states' differences — diff_tmp[:, 0..=order] = sdiff_j[:, 0..=order] (R U),
      // swap(sdiff_j, diff_tmp) — so the columns behind `order` are handed down the chain (states -> s_0 -> s_1 -> ... -> the states at the next change)
      double RU[6][6];  // RU[j][k] = element (row k, col j); the operations of the code above
      {
        double R[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          R[j][0] = 1.0;
#pragma unroll
          for (int i = 1; i < 6; ++i) R[j][i] = (j == 0) ? 0.0 : R[j][i - 1] * ((double)i - 1.0 - factor * (double)j) / (double)i;
        }
        const double* U = sU + (order - 1) * 36;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            double acc = R[0][k] * U[j * 6 + 0];
#pragma unroll
            for (int m = 1; m < 6; ++m) if (m <= order) acc = R[m][k] * U[j * 6 + m] + acc;
            RU[j][k] = acc;
          }
      }
      for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if (j <= order) {
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) {
              double acc = S[q][0][i] * RU[j][0];
#pragma unroll
              for (int k = 1; k < 6; ++k) if (k <= order) acc = S[q][k][i] * RU[j][k] + acc;
              dt_set(j, i, acc);
            }
          }
#pragma unroll
        for (int j = 0; j < kNC; ++j)
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) { const double tmp = S[q][j][i]; S[q][j][i] = dt_get(j, i); dt_set(j, i, tmp); }
      }
      s_c = new_h * sAlpha[order];  // s_op.set_c (bdf.rs:551-553)
    }
    opc = new_h * sAlpha[order];
    h = new_h;
    eta = C.r.eta_reset_ts;  // reset_eta_timestep_change
    new_h_out = new_h;
    return fabs(h) < o.min_timestep;  // true = StepSizeTooSmall
  };

  // @phase predict_forward
  // _predict_forward (bdf.rs:674-692): y_predict = sum_{j<=order} D_j ; psi_neg_y0 = alpha_order * sum_{1<=j<=order} gamma_j D_j - y_predict
  auto predict_forward = [&]() __attribute__((always_inline)) {
    if constexpr (WAVE) {
      const int ou = __builtin_amdgcn_readfirstlane(order);
      double s[N], q[N];
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) { s[i] = 0.0; q[i] = sGamma[1] * D[1][i]; }
#pragma unroll
      for (int j = 0; j < 6; ++j)
        guarded<WAVE>(j <= ou, [&]() __attribute__((always_inline)) {
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) s[i] = s[i] + D[j][i];
          if (j >= 2) {
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) q[i] = sGamma[j] * D[j][i] + 1.0 * q[i];
          }
        });
      const double al = sAlpha[ou];
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) {
        double qq = q[i] * al;
        qq = qq - s[i];
        yp[i] = s[i];
        psi[i] = qq;
      }
    } else {
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
    }
    t_predict = t + h;
  };

  // @phase jacobian_updates (re-evaluation, M - cJ, LU factor)
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

  // @phase handle_tstop / loop head
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
  const bool steps_mode = !SEG && !SENS && C.steps_cap > 0;  // every accepted step out (see AdaptiveConsts::steps_cap)
  auto steps_write = [&](double tw, const double (&yw)[N]) __attribute__((always_inline)) {
    if (col < C.steps_cap && active) {
      C.steps_t_out[(int64_t)col * nb + b] = tw;
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) y_out[((int64_t)col * N + i) * nb + b] = yw[i];
    }
    col++;
  };
  if (steps_mode && fresh) steps_write(t, y);  // write_out before the first step (method.rs:900)
  // solve_dense (method.rs:467-520): t_eval[0] >= t0 is checked on the host; set_stop_time(t_eval.last())
  if (fresh) {
    const int r = handle_tstop();
    if (r == 1) status = kRsStopTimeAtCurrentTime;
    else if (r == 2) status = kRsStopTimeBeforeCurrentTime;
  }

  long guard = 0;
  bool done = status != kRsOk || (!WAVE && !active);  // wavefront lock-step: shadow lanes run along (their reductions must not be masked off)
  // ---- segmented runs: the whole per-member state, enumerated ONCE for both directions
  auto seg_xfer = [&](auto&& fd, auto&& fi) __attribute__((always_inline)) {
    int kd = 0, ki = 0;
    auto d_ = [&](double& v) __attribute__((always_inline)) { fd(kd++, v); };
    auto i_ = [&](int& v) __attribute__((always_inline)) { fi(ki++, v); };
    d_(t); d_(h); d_(opc); d_(h_at_last_jac); d_(eta); d_(prev_err); d_(te_next); d_(rf_t0); d_(t_root); d_(t_predict);
#pragma unroll
    for (int j = 0; j < kNC; ++j)
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) { d_(D[j][i]); double v = dt_get(j, i); d_(v); dt_set(j, i, v); }
    if constexpr (!BANDED) {
#pragma unroll
      for (int e = 0; e < N * N; ++e) { double v = sJ[e][ln]; d_(v); sJ[e][ln] = v; d_(A[e]); }
    }
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) { d_(y[i]); i_(P[i]); }
#pragma unroll
    for (int r = 0; r < NR; ++r) d_(g0[r]);
    int js = jac_stale ? 1 : 0, hp = has_prev_err ? 1 : 0, ht = has_tstop ? 1 : 0, dn = done ? 1 : 0, glo = (int)(guard & 0x7fffffff), ghi = (int)(guard >> 31), st = (int)status;
    i_(order); i_(js); i_(n_setups); i_(n_steps); i_(n_err_fails); i_(n_newton); i_(n_nl_fails); i_(steps_since_jac); i_(steps_since_rhs_jac); i_(n_equal_steps);
    i_(hp); i_(col); i_(root_idx); i_(st); i_(ht); i_(dn); i_(glo); i_(ghi);
    jac_stale = js != 0; has_prev_err = hp != 0; has_tstop = ht != 0; done = dn != 0; guard = ((long)ghi << 31) | (long)glo; status = (int32_t)st;
  };
  if constexpr (SEG) {
    static_assert(!WAVE && !BANDED, "segmented runs: per-member control of the register-resident models");
    static_assert(10 + 2 * kNC * N + 2 * N * N + N + NR <= kSegDbl && N + 18 <= kSegInt, "segment state does not fit its slots");
    if (!fresh && active) {
      seg_xfer([&](int k, double& v) __attribute__((always_inline)) { v = C.seg_dbl[(int64_t)k * nb + b]; },
               [&](int k, int& v) __attribute__((always_inline)) { v = C.seg_int[(int64_t)k * nb + b]; });
    }
  }
  int seg_trips = 0;
  while (!done) {
    if (++guard > o.max_steps) { status = kRsMaxStepsExceeded; break; }
    // ================================================================ Bdf::step (bdf.rs:1277-1589)
    double safety = 0.0, error_norm = 0.0;
    const int old_err_fails = n_err_fails;
    bool convergence_fail = false;
    double x[N];
    int niter = 0;
    // @phase Newton iteration (residual, update, norm, convergence test)
    predict_forward();
    while (true) {
      // ---- NewtonNonlinearSolver::solve_in_place over NoLineSearch (newton.rs:13-36, line_search.rs:46-72)
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) x[i] = yp[i];
      double winv[FAST ? N : 1];
      if constexpr (FAST) {
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) winv[i] = 1.0 / (fabs(yp[i]) * rtol + atol[i]);
      }
      niter = 0;
      bool has_old = false;
      double old_norm = 0.0;
      bool solved = false;
      for (int it = 0; it < o.max_nonlinear_solver_iterations; ++it) {
        double f[N], delta[N], tmpv[N];
        Mdl::rhs(t_predict, x, p, f);
        if constexpr (Mdl::HAS_MASS) {
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) tmpv[i] = x[i] + psi[i];
        }
        // F(y) = M (y - y0 + psi) - c f(y)   (op/bdf.rs:240-256)
        if constexpr (Mdl::HAS_MASS) {
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) delta[i] = f[i];
          Mdl::mass_gemv(t_predict, tmpv, p, -opc, delta);
        } else {
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) delta[i] = 1.0 * (x[i] + psi[i]) + (-opc) * f[i];  // one pass: the sum is not stored (it matters when the vectors live in memory)
        }
        bool solved_ok;
        if constexpr (BANDED) solved_ok = band_solve_lane<N, BK>(Lf, Uf, P, delta);
        else if constexpr (FAST) solved_ok = lu_solve_reg_inv<N>(A, Adinv, P, delta);
        else solved_ok = lu_solve_reg<N>(A, P, delta);
        const bool lu_ok = group_all<WAVE>(solved_ok);
        if (!lu_ok) break;  // LuSolveFailed
        double delta_ms;  // Convergence::norm of the update (wms) fused with the update itself: one pass over the vectors
        {
          double acc = 0.0;
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) {
            const double d = delta[i];
            x[i] = x[i] - d;
            double term;
            if constexpr (FAST) term = d * winv[i]; else term = d / (fabs(yp[i]) * rtol + atol[i]);
            acc += term * term;
          }
          delta_ms = acc / (double)N;
        }
        const double norm = sqrt(group_norm<WAVE>(delta_ms));
        // Convergence::check_new_iteration (convergence.rs:68-139)
        niter += 1;
        bool diverged = false;
        if (has_old) {
          // pow(x, 1.0) == x exactly: the common second iteration needs no libm call
          const double rate = niter == 2 ? norm / old_norm : rpow(norm / old_norm, 1.0 / (double)(niter - 1), det);
          if (rate > 0.9) diverged = true;
          else if (powi_rt(rate, o.max_nonlinear_solver_iterations - niter) / (1.0 - rate) * norm > o.nonlinear_solver_tolerance) diverged = true;
          else eta = rate / (1.0 - rate);
        } else {
          const double min_eta = 1e4 * 2.220446049250313e-16;
          if (eta < min_eta) eta = min_eta;
          // after a reset eta is one of two constants (convergence.rs:36-42): their 0.8th powers come from the host (the same deterministic pow)
          if ((det || FAST) && eta == C.r.eta_reset) eta = C.eta_reset_p08;
          else if ((det || FAST) && eta == C.r.eta_reset_ts) eta = C.eta_reset_ts_p08;
          else eta = rpow(eta, 0.8, det);
        }
        const bool converged = !diverged && eta * norm < o.nonlinear_solver_tolerance;
        if (niter == 1) { has_old = true; old_norm = norm; }
        if (diverged) break;
        if (converged) { solved = true; break; }
      }
      // @phase Newton failure handling
      n_newton += niter;
      if constexpr (SENS) {
        // sensitivity_solve (bdf.rs:934-989): SensRhs linearised about (y_predict, t_new); per parameter the predictor / psi of its difference array and a
        // Newton solve of F(s) = (s - s0 + psi) - c_s (J s + (df/dp)_j) with the factors of the state equations and the SHARED Convergence (eta carries over,
        // state tolerances); a failure is a failure of the step's nonlinear solve (SensitivitySolveFailed, :1355-1360)
        if (solved) {
          for (int j = 0; j < NP && solved; ++j) {
            double ev[NP], dfdp[N], sp[N], spsi[N], xs[N];
#pragma unroll
            for (int k = 0; k < NP; ++k) ev[k] = k == j ? 1.0 : 0.0;
            Mdl::sens_mul(t_predict, yp, p, ev, dfdp);
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) {
              double sacc = 0.0;
#pragma unroll
              for (int k = 0; k < 6; ++k) if (k <= order) sacc = sacc + S[j][k][i];
              double q = sGamma[1] * S[j][1][i];
#pragma unroll
              for (int k = 2; k < 6; ++k) if (k <= order) q = sGamma[k] * S[j][k][i] + 1.0 * q;
              q = q * sAlpha[order];
              q = q - sacc;
              sp[i] = sacc; spsi[i] = q; xs[i] = sacc;
            }
            int sn = 0;
            bool s_has_old = false, s_solved = false;
            double s_old_norm = 0.0;
            for (int it = 0; it < o.max_nonlinear_solver_iterations; ++it) {
              double jm[N], delta[N];
              Mdl::jac_mul(t_predict, yp, p, xs, jm);
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) {
                const double fr = jm[i] + dfdp[i];
                delta[i] = 1.0 * (xs[i] + spsi[i]) + (-s_c) * fr;
              }
              bool s_lu_ok;
              if constexpr (BANDED) s_lu_ok = band_solve_lane<N, BK>(Lf, Uf, P, delta);
              else s_lu_ok = lu_solve_reg<N>(A, P, delta);
              const bool lu_ok = group_all<WAVE>(s_lu_ok);
              if (!lu_ok) break;
              double acc = 0.0;
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) {
                const double d = delta[i];
                xs[i] = xs[i] - d;
                const double term = d / (fabs(sp[i]) * rtol + atol[i]);
                acc += term * term;
              }
              const double norm = sqrt(group_norm<WAVE>(acc / (double)N));
              sn += 1;
              bool diverged = false;
              if (s_has_old) {
                const double rate = sn == 2 ? norm / s_old_norm : rpow(norm / s_old_norm, 1.0 / (double)(sn - 1), det);
                if (rate > 0.9) diverged = true;
                else if (powi_rt(rate, o.max_nonlinear_solver_iterations - sn) / (1.0 - rate) * norm > o.nonlinear_solver_tolerance) diverged = true;
                else eta = rate / (1.0 - rate);
              } else {
                const double min_eta = 1e4 * 2.220446049250313e-16;
                if (eta < min_eta) eta = min_eta;
                if (det && eta == C.r.eta_reset) eta = C.eta_reset_p08;
                else if (det && eta == C.r.eta_reset_ts) eta = C.eta_reset_ts_p08;
                else eta = rpow(eta, 0.8, det);
              }
              const bool converged = !diverged && eta * norm < o.nonlinear_solver_tolerance;
              if (sn == 1) { s_has_old = true; s_old_norm = norm; }
              if (diverged) break;
              if (converged) { s_solved = true; break; }
            }
            niter = sn;  // Convergence::niter is the last solve's: the safety factor below reads it
            if (!s_solved) { solved = false; break; }  // `?` before the iteration count is added
            n_newton += sn;
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) { s_cur[j][i] = xs[i]; s_delta[j][i] = xs[i] - sp[i]; }
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
          jacobian_updates(new_h * sAlpha[order], JState::SecondConvergenceFail);
          predict_forward();
        } else {
          jacobian_updates(h * sAlpha[order], JState::FirstConvergenceFail);
          convergence_fail = true;
        }
        continue;
      }
      // @phase error norm
      double ydelta[N];
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) ydelta[i] = x[i] - yp[i];
      // error_control (bdf.rs:812-843): norm against the CURRENT state y
      error_norm = fmax(0.0, group_norm<WAVE>(wms_y(ydelta)) * sEc2[order - 1]);
      if constexpr (SENS) {
        if (C.sens_error_control)  // bdf.rs:844-858 — error_const2[order], not [order - 1]
          for (int j = 0; j < NP; ++j) error_norm = fmax(error_norm, group_norm<WAVE>(wms<N>(s_delta[j], s_cur[j], s_atol, C.sens_rtol)) * sEc2[order]);
      }
      const double maxiter = (double)o.max_nonlinear_solver_iterations;
      safety = 0.9 * (2.0 * maxiter + 1.0) / (2.0 * maxiter + (double)niter);
      if (error_norm <= 1.0) {
        // @phase accept: update of the differences
        // ---- accepted: _update_diff (bdf.rs:646-664), state update
        if constexpr (WAVE) {
          const int ou = __builtin_amdgcn_readfirstlane(order);
          double dk1[N], upper[N];
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) dk1[i] = 0.0;
#pragma unroll
          for (int j = 2; j < 7; ++j)
            guarded<WAVE>(j == ou + 1, [&]() __attribute__((always_inline)) {
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) dk1[i] = D[j][i];
            });
#pragma unroll
          for (int j = 2; j < kNC; ++j) {
            guarded<WAVE>(j == ou + 2, [&]() __attribute__((always_inline)) {
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) D[j][i] = ydelta[i] - dk1[i];
            });
            guarded<WAVE>(j == ou + 1, [&]() __attribute__((always_inline)) {
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) D[j][i] = ydelta[i];
            });
          }
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) upper[i] = ydelta[i];
#pragma unroll
          for (int j = 5; j >= 0; --j)
            guarded<WAVE>(j <= ou, [&]() __attribute__((always_inline)) {
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) { const double v = D[j][i] + 1.0 * upper[i]; D[j][i] = v; upper[i] = v; }
            });
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) y[i] = yp[i];
        } else {
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
        }
        if constexpr (SENS) {  // update_differences_and_integrate_out (bdf.rs:628-643): _update_diff on every sensitivity difference array
          for (int q = 0; q < NP; ++q)
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) {
              const double sd = s_delta[q][i];
              double dk1 = 0.0;
#pragma unroll
              for (int j = 2; j < 7; ++j) if (j == order + 1) dk1 = S[q][j][i];
              const double dk2 = sd - dk1;
#pragma unroll
              for (int j = 2; j < kNC; ++j) { if (j == order + 2) S[q][j][i] = dk2; if (j == order + 1) S[q][j][i] = sd; }
              double upper = sd;
#pragma unroll
              for (int j = 5; j >= 0; --j) if (j <= order) { const double v = S[q][j][i] + 1.0 * upper; S[q][j][i] = v; upper = v; }
            }
        }
        refresh_wyinv();
        t = t_predict;
        break;
      }
      // @phase error-test failure
      double factor = safety * pi_controller_raw(error_norm, has_prev_err, prev_err, o.pi_control_integral, o.pi_control_proportional, order + 1, det);
      has_prev_err = false;
      if (factor < o.min_timestep_shrink) factor = o.min_timestep_shrink;
      double new_h;
      if (update_step_size(factor, new_h)) { status = kRsStepSizeTooSmall; break; }
      jacobian_updates(new_h * sAlpha[order], JState::ErrorTestFail);
      predict_forward();
      n_err_fails += 1;
      if (n_err_fails - old_err_fails >= o.max_error_test_failures) { status = kRsTooManyErrorTestFailures; break; }
    }
    // @phase order selection
    if (status != kRsOk) break;
    n_steps += 1;
    steps_since_jac += 1; steps_since_rhs_jac += 1;  // JacobianUpdate::step
    prev_err = error_norm; has_prev_err = true;
    n_equal_steps += 1;
    if (n_equal_steps > order) {
      // order selection (bdf.rs:1494-1560): predict_error_control(order-1) / (order+1) on the updated differences
      double col_m[N], col_p[N];
      if constexpr (WAVE) {
        const int ou = __builtin_amdgcn_readfirstlane(order);
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) { col_m[i] = 0.0; col_p[i] = 0.0; }
#pragma unroll
        for (int j = 1; j < kNC; ++j) {
          guarded<WAVE>(j == ou, [&]() __attribute__((always_inline)) {
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) col_m[i] = D[j][i];
          });
          guarded<WAVE>(j == ou + 2, [&]() __attribute__((always_inline)) {
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) col_p[i] = D[j][i];
          });
        }
      } else {
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) {
        double vm = 0.0, vp = 0.0;
#pragma unroll
        for (int j = 1; j < kNC; ++j) { if (j == order) vm = D[j][i]; if (j == order + 2) vp = D[j][i]; }
        col_m[i] = vm; col_p[i] = vp;
      }
      }
      const double inf = __builtin_huge_val();
      double error_m_norm = order > 1 ? group_norm<WAVE>(wms_y(col_m)) * sEc2[order - 1] : inf;
      double error_p_norm = order < kMaxOrder ? group_norm<WAVE>(wms_y(col_p)) * sEc2[order + 1] : inf;
      if constexpr (SENS) {  // predict_error_control with the augmented system (bdf.rs:871-932): the `error_norm.max(err)` chain from zero, then the sensitivities' terms
        if (order > 1) error_m_norm = fmax(0.0, error_m_norm);
        if (order < kMaxOrder) error_p_norm = fmax(0.0, error_p_norm);
        if (C.sens_error_control)
          for (int q = 0; q < NP; ++q) {
            double cm[N], cp[N];
DSH_UNROLL_N
            for (int i = 0; i < N; ++i) {
              double vm = 0.0, vp = 0.0;
#pragma unroll
              for (int j = 1; j < kNC; ++j) { if (j == order) vm = S[q][j][i]; if (j == order + 2) vp = S[q][j][i]; }
              cm[i] = vm; cp[i] = vp;
            }
            if (order > 1) error_m_norm = fmax(error_m_norm, group_norm<WAVE>(wms<N>(cm, s_cur[q], s_atol, C.sens_rtol)) * sEc2[order - 1]);
            if (order < kMaxOrder) error_p_norm = fmax(error_p_norm, group_norm<WAVE>(wms<N>(cp, s_cur[q], s_atol, C.sens_rtol)) * sEc2[order + 1]);
          }
      }
      const double pi_i = o.pi_control_integral, pi_p = o.pi_control_proportional;
      double f0c, f1c, f2c;
      if constexpr (WAVE) {
        // the three controller values (runge_kutta.rs:1313-1336) are wavefront-uniform, and each is one pow or a product of two: lanes 0..5 take one
        // (base, exponent) pair each and ONE call of pow serves all of them — the same function on the same arguments, so the same bits, for a sixth
        // (or a third) of the instructions; v_readlane brings the results back
        const bool two = (pi_p != 0.0) & has_prev_err;
        const int l6 = ln & 7, which = l6 >> 1;
        const double eo = (double)(order + which);
        const double ki = pi_i / eo, kp = pi_p / eo;
        const double errs = which == 0 ? error_m_norm : (which == 1 ? error_norm : error_p_norm);
        const double base = (l6 & 1) ? prev_err : errs;
        const double expo = (l6 & 1) ? kp : (two ? -(ki + kp) : -ki);
        const double r = rpow(l6 < 6 ? base : 1.0, expo, det);
        auto rl = [&](int lane) __attribute__((always_inline)) -> double {
          return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(r), lane), __builtin_amdgcn_readlane(__double2loint(r), lane));
        };
        f0c = two ? rl(0) * rl(1) : rl(0);
        f1c = two ? rl(2) * rl(3) : rl(2);
        f2c = two ? rl(4) * rl(5) : rl(4);
      } else {
        f0c = pi_controller_raw(error_m_norm, has_prev_err, prev_err, pi_i, pi_p, order, det);
        f1c = pi_controller_raw(error_norm, has_prev_err, prev_err, pi_i, pi_p, order + 1, det);
        f2c = pi_controller_raw(error_p_norm, has_prev_err, prev_err, pi_i, pi_p, order + 2, det);
      }
      int max_index = 0;  // Iterator::max_by keeps the LAST maximum
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
        jacobian_updates(new_h * sAlpha[new_order], JState::StepSuccess);
      }
    }
    // @phase interpolation / output / stop tests
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
    int reason = 0;  // 0 internal, 1 tstop, 3 root
    if constexpr (Mdl::NROOTS > 0) {
      const int rr = check_root<Mdl, WAVE>(g0, rf_t0, y, t, p, interpolate, t_root, root_idx);
      if (rr == 2) { status = kRsRootBatchMismatch; break; }
      if (rr == 1) reason = 3;
    }
    if (reason == 0 && has_tstop) reason = handle_tstop();
    if (reason == 2) reason = 0;  // the reference unwraps / ignores this inside step()
    // ================================================================ solve_dense (method.rs:467-520): interpolated output
    const double upto = reason == 3 ? t_root : t;
    if (steps_mode) {  // OdeSolverMethod::solve: InternalTimestep / TstopReached -> write_out (method.rs:907-921); a root is written below, at the root
      if (reason != 3) steps_write(t, y);  // state.y (the converged Newton iterate), not D[0] (the same value summed in another order)
    } else
    while (col < C.r.n_eval && te_next <= upto) {
      double yv[N];
      interpolate(te_next, yv);
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) if (active) y_out[((int64_t)col * N + i) * nb + b] = yv[i];
      if constexpr (SENS) {  // interpolate_sens (bdf.rs:1162-1215): the same polynomial on every sensitivity difference array
        for (int q = 0; q < NP; ++q) {
          double sv[N];
          double time_factor = 1.0;
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) sv[i] = S[q][0][i];
#pragma unroll
          for (int j = 0; j < kMaxOrder; ++j) {
            if (j < order) {
              const double jt = (double)j;
              time_factor *= (te_next - (t - h * jt)) / (h * (1.0 + jt));
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) sv[i] = time_factor * S[q][j + 1][i] + 1.0 * sv[i];
            }
          }
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) if (active) C.sens_out[(((int64_t)col * NP + q) * N + i) * nb + b] = sv[i];
        }
      }
      col++;
      if (col < C.r.n_eval) te_next = t_eval[col];
    }
    if constexpr (kResets) {
      if (reason == 3) {
        // A reset operator is configured (solve_dense, method.rs:774-797): move the state back to the root (state_mut_back, bdf.rs:1232-1262), apply the reset
        // (apply_reset, bdf.rs:1017-1020 over state.rs:279-306: y <- reset(y, t), dy <- f(y, t)), arm the stop time again and go on — the next step restarts
        // from the modified state at first order (bdf.rs:1290-1318).  The save points up to the root were written from the step's polynomial above.
        double yb[N], yr[N], dyr[N];
        interpolate(t_root, yb);
        if constexpr (Mdl::HAS_MASS) {
          // state_mut_back stores the derivative of the step's polynomial at the root in state.dy (interpolate_derivative_from_diff, bdf.rs:784-811): the starting
          // guess of the differential unknowns of set_consistent below
          double pi = 1.0, d_pi = 0.0;
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) dyr[i] = 0.0;
#pragma unroll
          for (int j = 0; j < kMaxOrder; ++j) {
            if (j < order) {
              const double i_t = (double)j;
              const double denom = h * (1.0 + i_t);
              const double w = (t_root - (t - h * i_t)) / denom;
              const double dw = 1.0 / denom;
              const double new_d_pi = d_pi * w + pi * dw;
              pi *= w;
              d_pi = new_d_pi;
DSH_UNROLL_N
              for (int i = 0; i < N; ++i) dyr[i] = d_pi * D[j + 1][i] + 1.0 * dyr[i];
            }
          }
        }
        t = t_root;
        Mdl::reset(t, yb, p, yr);
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) y[i] = yr[i];
        if constexpr (Mdl::HAS_MASS) {
          // apply_reset_with_mass (state.rs:279-306): (y, dy) consistent with the algebraic equations again — Newton on InitOp WITHOUT line search, whatever ic_options say
          if (!group_all<WAVE>(set_consistent<Mdl, WAVE>(t, p, y, dyr, atol, rtol, C.r, true))) { status = kRsInitialConditionDidNotConverge; break; }
        } else
        Mdl::rhs(t, y, p, dyr);
        if (steps_mode) steps_write(t, y);  // method.rs:931-932: the reset state at the root time
        if (t < tstop) {
          has_tstop = true;  // set_stop_time (bdf.rs:1591-1600): on the OLD differences and order, like the reference (the step size may change here)
          { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          // ---- the `is_state_modified` branch of the next Bdf::step
          Mdl::root(t, y, p, g0);  // RootFinder::init
          rf_t0 = t;
          n_equal_steps = 0;
          order = 1;               // initialise_diff_to_first_order: columns 0 and 1 only, the others keep what they hold
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) { D[0][i] = y[i]; D[1][i] = dyr[i] * h; }
          opc = h * sAlpha[1];
          jacobian_updates(h * sAlpha[1], JState::StepSuccess);
          has_prev_err = false;
          if (has_tstop) { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          refresh_wyinv();
          reason = 0;
        } else {
          done = true;  // the event sits on the last save point: TstopReached
          reason = 0;
        }
      }
    }
    if (reason == 3 && steps_mode) {  // method.rs:922-947 without a reset: state_mut_back(t_root), write_out, RootFound
      double yv[N];
      interpolate(t_root, yv);
      steps_write(t_root, yv);
      done = true;
    } else
    if (reason == 3) {  // state_mut_back(root_time): the column after the drained ones holds the state at the root
      if (col < C.r.n_eval) {
        double yv[N];
        interpolate(t_root, yv);
DSH_UNROLL_N
        for (int i = 0; i < N; ++i) if (active) y_out[((int64_t)col * N + i) * nb + b] = yv[i];
        col++;
      }
      done = true;
    }
    if (reason == 1) done = true;
    if constexpr (SEG) {  // this launch's share of the save points (or of the steps) is out: the next launch goes on from here
      if (!C.seg_last && (col >= C.seg_col_end || (C.seg_step_budget > 0 && ++seg_trips >= C.seg_step_budget))) break;
    }
  }
  if constexpr (SEG) {
    if (status != kRsOk) done = true;
    if (active) {
      seg_xfer([&](int k, double& v) __attribute__((always_inline)) { C.seg_dbl[(int64_t)k * nb + b] = v; },
               [&](int k, int& v) __attribute__((always_inline)) { C.seg_int[(int64_t)k * nb + b] = v; });
      // members that go on together should look alike: same order, same distance to the next order selection, neighbouring step sizes
      const int ne = n_equal_steps < 15 ? n_equal_steps : 15;
      C.seg_key[b] = done ? ~0ull : (((unsigned long long)order << 60) | ((unsigned long long)ne << 56) | ((unsigned long long)__double_as_longlong(fabs(h)) >> 8));
    }
    if (!C.seg_last) {
      if (C.seg_remaining != nullptr) {
        const int left = __popcll(__ballot(active && !done));
        if ((threadIdx.x & 63) == 0 && left) atomicAdd(C.seg_remaining, (unsigned int)left);
      }
      return;
    }
  }
  // @phase epilogue
  if (active) {
    if (ncols_out != nullptr) ncols_out[b] = col;
    if (t_root_out != nullptr) t_root_out[b] = root_idx >= 0 ? t_root : __builtin_nan("");
    if (root_idx_out != nullptr) root_idx_out[b] = root_idx;
    // columns that were never reached (root stop or error exit): NaN
    if (!steps_mode)
    for (; col < C.r.n_eval; ++col) {
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) y_out[((int64_t)col * N + i) * nb + b] = __builtin_nan("");
      if constexpr (SENS)
        for (int q = 0; q < NP; ++q)
DSH_UNROLL_N
          for (int i = 0; i < N; ++i) C.sens_out[(((int64_t)col * NP + q) * N + i) * nb + b] = __builtin_nan("");
    }
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
