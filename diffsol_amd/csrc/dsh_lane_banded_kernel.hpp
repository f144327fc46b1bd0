// Device-resident variable-order BDF for BANDED models with 8 < n <= 64 states, one lane per ensemble member, the whole solver state in per-lane
// memory (launch code: dsh_adaptive.hip; instantiated by hiprtc for the model at hand, dsh_jit.hip, DSH_JIT_FORM_STATIC_BANDED).
//
// Same algorithm, control flow and arithmetic as k_bdf_adaptive (dsh_adaptive_kernel.hpp; restated from bdf.rs:244-1589, see there) — bit-identical
// results — but written for a state that does NOT fit registers: with n = 42 (BASELINE config 4, the single-particle battery model) the difference
// array alone is 8 x 42 doubles per member, so every vector operation is a stream of 512-byte wavefront transactions to HBM (the working set of all
// resident wavefronts is gigabytes: nothing is cached) and the kernel is bound by the BYTES it moves.  k_bdf_adaptive's banded branch, written for
// register arrays (value-selects over all candidate rows instead of indexing, one loop per vector operation), moved 36 KB per member-step
// (profiles/pmc_lane_banded.json).  Here every pass is fused down to the traffic the algorithm needs (q = order; counts in n-vectors of 8 n bytes):
//   predict                         q+1 loads, 2 stores   — and skipped altogether when the previous step's difference update could provide it
//   Newton iteration                f(x): 1 + 1;  residual fused into the forward sweep of the banded LU: 3 + K + 1/2 loads, 1 store;  backward sweep
//                                   fused with the update of x: 2 + (2K+1) loads, 2 stores;  norm: 2 loads.  The first iteration reads y_predict
//                                   in place of x (no copy).
//   error test                      3 loads
//   accepted step                   ONE pass: difference update D_{q+2}, D_{q+1}, D_q..D_0 (q+4 loads, q+4 stores), state y, the two order-selection
//                                   norms when they are due, and the NEXT step's prediction y_predict / psi from the values in registers (2 stores)
//   step-size change                D <- D R U in one pass with R U in registers (q+1 loads, q+1 stores); the reference's swap of `diff` and
//                                   `diff_tmp` (bdf.rs:568-577) is a flip of a buffer index — which also reproduces what the swap does to the
//                                   columns beyond the order (they come back from two swaps ago)
//   absolute tolerances             read from the operand (uniform address) instead of a per-lane copy
// ~16 KB per member-step at order 4 instead of 36 KB.  Loops over the state components stay rolled (unrolled by 4 for memory-level parallelism):
// the code is ~10x smaller than the fully unrolled form and independent of n.
#pragma once
#include "dsh_adaptive_kernel.hpp"

namespace dsh {

#ifndef DSH_LANE_BANDED_WAVES_PER_EU
#define DSH_LANE_BANDED_WAVES_PER_EU 3
#endif
#ifndef DSH_LANE_BANDED_PAD
#define DSH_LANE_BANDED_PAD 0  // extra doubles at the end of the difference arrays: changes the size of the per-lane scratch frame (and with it the address pattern of the wavefronts)
#endif
#ifndef DSH_LANE_BANDED_UNROLL
#define DSH_LANE_BANDED_UNROLL 4
#endif
#define DSH_LB_PRAGMA_(x) _Pragma(#x)
#define DSH_LB_PRAGMA(x) DSH_LB_PRAGMA_(x)
#define DSH_LB_STREAM DSH_LB_PRAGMA(unroll DSH_LANE_BANDED_UNROLL)

// Chunked streaming: the operands of CH consecutive components are loaded into registers first (CH x operands loads in flight per wavefront — the
// kernel is bound by memory latency x bytes in flight, and the compiler keeps loads next to their uses otherwise), then the arithmetic runs in
// index order.  The chunk is a divisor of n where one is near the wanted size (no tail code); otherwise the tail is clamped loads + predicates.
constexpr int lb_chunk(int n, int want) {
  for (int c = want; c * 2 > want && c > 1; --c) if (n % c == 0) return c;
  return want;
}
#ifndef DSH_LANE_BANDED_CHUNK_SCALE
#define DSH_LANE_BANDED_CHUNK_SCALE 1
#endif


// StateRefMut::set_consistent (state.rs:84-162) over InitOp (op/init.rs:14-135) for a banded model with a DIAGONAL mass matrix, every array in per-lane
// memory: the restatement of dsh_resident.hpp's set_consistent (Newton on (du for the differential, v for the algebraic unknowns), backtracking line
// search, line_search.rs:84-201) with the banded LU in place of the dense one — the same eliminations in the same order on a banded matrix, so the same
// bits as the oracle's dense solve.  md: the diagonal of M (0 = algebraic row).  y / dy in-out; Jb, Lf, Uf, P: the kernel's band / factor storage, free at
// this point.  Returns false for InitialConditionDidNotConverge.
template <class Mdl, bool WAVE, bool BA>
__device__ __forceinline__ bool set_consistent_banded(double t0, const double (&p)[Mdl::NP], double* y, double* dy, const double* md, const double* Mb, const double* atol_g, int64_t nb,
                                                      int64_t b, double rtol, const ResidentConsts& C, double* Jb, double* Lf, double* Uf, int* P) {
  constexpr int N = Mdl::N, K = model_band_k<Mdl>::value;
  constexpr bool MBAND = model_mass_band_k<Mdl>::value > 0;  // M banded, its band in Mb (layout of Jb); else diag(md)
  const dsh_adaptive_options& o = C.o;
  auto AT = [&](int i) __attribute__((always_inline)) -> double { return BA ? atol_g[i] : atol_g[(int64_t)i * nb + b]; };
  bool any_alg = false;
  for (int i = 0; i < N; ++i) any_alg = any_alg || md[i] == 0.0;  // partition_indices_by_zero_diagonal
  if (!any_alg) return true;
  // InitOp::new: jac = (-M_u, df/dv; 0, dg/dv): column j of f_y for an algebraic unknown j, -M's column for a differential one
  Mdl::jac_band(t0, *reinterpret_cast<const double (*)[N]>(y), p, *reinterpret_cast<double (*)[(2 * K + 1) * N]>(Jb));
  alignas(16) double y0[N], x[N], yerr[N], delta[N], x0[N], delta0[N];
  for (int i = 0; i < N; ++i) { y0[i] = y[i]; x[i] = md[i] == 0.0 ? y[i] : dy[i]; yerr[i] = x[i]; delta[i] = 0.0; }
  // InitOp::call_inplace (:103-115): y0[alg] = x[alg]; out = f(y0); out = neg_mass x + out — with a diagonal neg_mass the gemv leaves (-m_i) x_i + out_i; with a banded
  // one (neg_mass = -M restricted to differential rows AND columns, init.rs:46-63) the nonzero terms of the row in ascending column order (the gemv's order; its
  // zero terms change nothing)
  auto fun = [&](const double* xx, double* out) __attribute__((always_inline)) {
    for (int i = 0; i < N; ++i) if (md[i] == 0.0) y0[i] = xx[i];
    Mdl::rhs(t0, *reinterpret_cast<const double (*)[N]>(y0), p, *reinterpret_cast<double (*)[N]>(out));
    if constexpr (MBAND) {
      for (int i = 0; i < N; ++i)
        if (md[i] != 0.0) {
          double acc = out[i];
          for (int col = (i - K > 0 ? i - K : 0); col <= (i + K < N - 1 ? i + K : N - 1); ++col)
            if (md[col] != 0.0) acc = (Mb[(col - i + K) * N + i] * (-1.0)) * xx[col] + acc;
          out[i] = acc;
        }
    } else {
      for (int i = 0; i < N; ++i) if (md[i] != 0.0) out[i] = (md[i] * (-1.0)) * xx[i] + out[i];
    }
  };
  auto norm_of = [&](const double* v) __attribute__((always_inline)) -> double {  // Convergence::norm against yerr
    double acc = 0.0;
    for (int i = 0; i < N; ++i) { const double term = v[i] / (fabs(yerr[i]) * rtol + AT(i)); acc += term * term; }
    return sqrt(group_norm<WAVE>(acc / (double)N));
  };
  ConvState conv;
  conv.eta = C.eta_reset;
  conv.tol = o.nonlinear_solver_tolerance;
  conv.max_iter = o.ic_max_newton_iterations;
  conv.det = o.deterministic_pow != 0;
  bool ok = false;
  for (int k = 0; k < o.ic_max_linear_solver_setups; ++k) {
    // reset_jacobian: the InitOp Jacobian is constant
    bool sing = false;
    band_factor_lane_fn<N, K>([&](int i, int col) -> double {
      if (md[col] == 0.0) return Jb[(col - i + K) * N + i];
      if constexpr (MBAND) return md[i] == 0.0 ? 0.0 : Mb[(col - i + K) * N + i] * (-1.0);  // -M_u: differential rows and columns only
      else return i == col ? md[col] * (-1.0) : 0.0;
    }, Lf, Uf, P, sing);
    conv.reset();
    double ls_norm = 1.0;
    int result = 2;  // 0 ok, 1 fatal (diverged / LU / line search), 2 NewtonMaxIterations
    for (int it = 0; it < conv.max_iter; ++it) {
      ConvStatus st = ConvStatus::Continue;
      bool fatal = false;
      if (!o.ic_use_linesearch) {  // NoLineSearch::take_optimal_step
        fun(x, delta);
        if (!group_all<WAVE>(band_solve_lane<N, K>(Lf, Uf, P, delta))) fatal = true;
        else {
          for (int i = 0; i < N; ++i) x[i] = x[i] - delta[i];
          st = conv.check_new_iteration(norm_of(delta));
        }
      } else {  // BacktrackingLineSearch::take_optimal_step
        bool returned = false;
        if (conv.niter == 0) {
          fun(x, delta);
          if (!group_all<WAVE>(band_solve_lane<N, K>(Lf, Uf, P, delta))) { fatal = true; returned = true; }
          else {
            ls_norm = norm_of(delta);
            if (conv.check_norm(ls_norm) == ConvStatus::Converged) {
              for (int i = 0; i < N; ++i) x[i] = x[i] - delta[i];
              st = ConvStatus::Converged;
              returned = true;
            }
          }
        }
        if (!returned) {
          for (int i = 0; i < N; ++i) { x0[i] = x[i]; delta0[i] = delta[i]; }
          const double nrm = ls_norm;
          const double phi0 = nrm * nrm * 0.5, two_phi0 = nrm * nrm, min_alpha = C.ls_steptol / nrm;
          double alpha = 1.0;
          bool found = false;
          for (int i = 0; i < o.ic_max_linesearch_iterations; ++i) {
            for (int q = 0; q < N; ++q) x[q] = (-alpha) * delta0[q] + 1.0 * x[q];
            fun(x, delta);
            if (!group_all<WAVE>(band_solve_lane<N, K>(Lf, Uf, P, delta))) { fatal = true; break; }
            const double new_norm = norm_of(delta);
            const double phi1 = new_norm * new_norm * 0.5;
            if (phi1 <= phi0 - o.ic_armijo_constant * alpha * two_phi0) {
              ls_norm = new_norm;
              st = conv.check_norm(new_norm);
              found = true;
              break;
            }
            if (alpha < min_alpha) { fatal = true; break; }  // LinesearchFailedMinStep
            alpha *= o.ic_step_reduction_factor;
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
    for (int i = 0; i < N; ++i) yerr[i] = x[i];
  }
  if (!ok) return false;
  // scatter_soln (:76-81) + zero the algebraic derivatives (state.rs:155-158)
  for (int i = 0; i < N; ++i) {
    if (md[i] == 0.0) { y[i] = x[i]; dy[i] = 0.0; }
    else dy[i] = x[i];
  }
  return true;
}

template <class Mdl, bool BA, bool WAVE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DSH_LANE_BANDED_WAVES_PER_EU, DSH_LANE_BANDED_WAVES_PER_EU))) void k_bdf_lane_banded(
    int64_t nb, const double* __restrict__ p_g, const double* __restrict__ atol_g, const AdaptiveConsts* __restrict__ Cp, const double* __restrict__ t_eval,
    double* __restrict__ y_out, int32_t* __restrict__ stats_out, int32_t* __restrict__ status_out, double* __restrict__ t_root_out,
    int32_t* __restrict__ root_idx_out, int32_t* __restrict__ ncols_out, unsigned long long* __restrict__ totals) {
  constexpr int N = Mdl::N, NP = Mdl::NP;
  constexpr int NR = Mdl::NROOTS > 0 ? Mdl::NROOTS : 1;
  constexpr int K = model_band_k<Mdl>::value, RW = K + 1, CW = 2 * K + 1;
  static_assert(K > 0, "k_bdf_lane_banded: banded models");  // a mass matrix is diagonal (the fast paths) or banded within K (model_mass_band_k)
  const AdaptiveConsts& C = *Cp;
  const int64_t bglobal = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = bglobal < nb;  // lanes past the ensemble shadow a live member (no stores) so that the whole wavefront reaches every reduction
  const int64_t b = active ? bglobal : (int64_t)blockIdx.x * blockDim.x;
  const dsh_adaptive_options& o = C.r.o;
  const bool det = o.deterministic_pow != 0;
  const double rtol = C.r.rtol;
  double p[NP];
  load_vec<NP>(p_g, nb, b, p);
  auto AT = [&](int i) __attribute__((always_inline)) -> double { return BA ? atol_g[i] : atol_g[(int64_t)i * nb + b]; };

  const int ln = threadIdx.x;
  __shared__ double sAlpha[6], sGamma[6], sEc2[6], sU[kMaxOrder * 36];
  if (ln < 6) { sAlpha[ln] = C.alpha[ln]; sGamma[ln] = C.gamma[ln]; sEc2[ln] = C.ec2[ln]; }
  for (int k = ln; k < kMaxOrder * 36; k += 64) sU[k] = C.u[k / 36][k % 36];
  __syncthreads();

  // ---- per-lane memory (scratch: interleaved by lane in hardware, every access one coalesced 512-byte transaction per wavefront)
  alignas(16) double Dm[2 * kNC * N + DSH_LANE_BANDED_PAD];                       // diff and diff_tmp: row j of the current one at Dm[(cur * kNC + j) * N]
  alignas(16) double Jb[CW * N], Lf[K * N], Uf[CW * N];     // band of f_y, banded LU factors of I - c f_y
  int P[N];
  alignas(16) double y[N], xy[2 * N], psi[N], w[N];         // state, (y_predict | Newton iterate), psi_neg_y0, work vector (f, then the solve in place)
  constexpr bool MASS = Mdl::HAS_MASS;
  // M x == diag(md) x bit for bit (model_mass_rows_scaled): the residual entry is formed where the sweep consumes it, (md_i == 0 ? 0 : md_i (x_i + psi_i)) + (-c) f_i —
  // the arithmetic of the generated mass_gemv row (literal 0 / x_i / coefficient * x_i, then + beta y_i) without the two extra passes over per-lane memory
  constexpr bool MSCALED = MASS && model_mass_rows_scaled<Mdl>::value;
  alignas(16) double md[MASS ? N : 1];  // models with a mass matrix: its diagonal (0 = algebraic row)
  constexpr bool MBAND = MASS && model_mass_band_k<Mdl>::value > 0;
  static_assert(!MBAND || (model_mass_band_k<Mdl>::value <= K && !MSCALED), "the mass matrix's band must fit the factored band");
  alignas(16) double Mb[MBAND ? CW * N : 1];  // banded mass matrix: entry (i, col) at Mb[(col - i + K) * N + i], like Jb
  double* const yp = xy;
  double* const X = xy + N;
  int cur = 0;
  auto Drow = [&](int j) __attribute__((always_inline)) -> double* { return Dm + (cur * kNC + j) * N; };

  // ------------------------------------------------------------ OdeSolverState::new_and_consistent, set_step_size (state.rs:969-997, :1209-1277)
  double t = C.r.t0, h;
  int32_t status = kRsOk;
  {
    double atol_arr[N];
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) atol_arr[i] = AT(i);
    Mdl::init(t, p, y);
    Mdl::rhs(t, y, p, w);
    if constexpr (MASS) {
      alignas(16) double tmpv[N];
      if constexpr (MBAND) {
        // the band of M from CW products with 0/1 vectors (columns c, c + CW, ... at once: within the band of a row at most one of them is met)
        alignas(16) double col_of[N];
        for (int e = 0; e < CW * N; ++e) Mb[e] = 0.0;
        for (int c = 0; c < CW; ++c) {
          for (int i = 0; i < N; ++i) { tmpv[i] = i % CW == c ? 1.0 : 0.0; col_of[i] = 0.0; }
          Mdl::mass_gemv(t, *reinterpret_cast<const double (*)[N]>(tmpv), p, 0.0, *reinterpret_cast<double (*)[N]>(col_of));
          for (int i = 0; i < N; ++i)
            for (int col = (i - K > 0 ? i - K : 0); col <= (i + K < N - 1 ? i + K : N - 1); ++col)
              if (col % CW == c) Mb[(col - i + K) * N + i] = col_of[i];
        }
        for (int i = 0; i < N; ++i) md[i] = Mb[K * N + i];
      } else {
        // the diagonal of M: M applied to a vector of ones (a diagonal matrix's row sums are its diagonal)
        for (int i = 0; i < N; ++i) { tmpv[i] = 1.0; md[i] = 0.0; }
        Mdl::mass_gemv(t, *reinterpret_cast<const double (*)[N]>(tmpv), p, 0.0, *reinterpret_cast<double (*)[N]>(md));
      }
      if (!group_all<WAVE>(set_consistent_banded<Mdl, WAVE, BA>(t, p, y, w, md, Mb, atol_g, nb, b, rtol, C.r, Jb, Lf, Uf, P))) status = kRsInitialConditionDidNotConverge;
    }
    h = initial_step_size<Mdl, WAVE>(t, C.r.h0, y, w, p, atol_arr, rtol, 1, det);
  }

  // ------------------------------------------------------------ Bdf::_new (bdf.rs:244-368) + BdfState::initialise_diff_to_first_order
  int order = 1;
DSH_LB_STREAM
  for (int e = 0; e < 2 * kNC * N; ++e) Dm[e] = 0.0;
DSH_LB_STREAM
  for (int i = 0; i < N; ++i) { Dm[i] = y[i]; Dm[N + i] = w[i] * h; }
  double opc = h * sAlpha[1];  // BdfCallable::c
  bool jac_stale = true;
  bool predicted = false;  // y_predict / psi already hold the prediction of the coming step (made by the accept pass of the last one)
  int n_setups = 0, n_steps = 0, n_err_fails = 0, n_newton = 0, n_nl_fails = 0;
  // NonLinearSolver::reset_jacobian: M - c f'(x) (op/bdf.rs:273-300) + banded LU
  auto reset_jacobian = [&](double tt) __attribute__((always_inline)) {
    if (jac_stale) { Mdl::jac_band(tt, y, p, Jb); jac_stale = false; }
    bool sing = false;
    if constexpr (MASS)  // J (-c) + M (op/bdf.rs:273-300), M diagonal
      band_factor_lane_fn<N, K>([&](int i, int col) -> double { return Jb[(col - i + K) * N + i] * (-opc) + (MBAND ? Mb[(col - i + K) * N + i] : (i == col ? md[i] : 0.0)); }, Lf, Uf, P, sing);
    else
    band_factor_lane<N, K>(Jb, opc, Lf, Uf, P, sing);
  };
  reset_jacobian(t);
  n_setups = 1;
  double g0[NR] = {0.0};
  double rf_t0 = t;
  if constexpr (Mdl::NROOTS > 0) Mdl::root(t, y, p, g0);
  double t_root = 0.0;
  int root_idx = -1;
  int steps_since_jac = 0, steps_since_rhs_jac = 0;
  double h_at_last_jac = 1.0;
  double eta = C.r.eta_reset;
  int n_equal_steps = 0;
  bool has_prev_err = false;
  double prev_err = 0.0;
  double t_predict = t;

  // _update_step_size (bdf.rs:508-566) with _update_diff_for_step_size (:568-577): diff_tmp[:, 0..=order] = diff[:, 0..=order] * (R U); swap
  auto update_step_size = [&](double factor, double& new_h_out) __attribute__((always_inline)) -> bool {
    const double new_h = factor * h;
    n_equal_steps = 0;
    predicted = false;
    // ru[j][k] = (R U)(k, j) = sum_m R(k, m) U(m, j), gemm order (first term, then acc = a b + acc); R(k, m) = compute_r(order, factor) (bdf.rs:433-463)
    // row by row of R from its recurrence in k: only one row of R is live at a time
    const double* U = sU + (order - 1) * 36;  // element (row m, col j) at U[j*6 + m]
    double ru[6][6], rrow[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
      for (int m = 0; m < 6; ++m) rrow[m] = (k == 0) ? 1.0 : ((m == 0) ? 0.0 : rrow[m] * ((double)k - 1.0 - factor * (double)m) / (double)k);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double acc = rrow[0] * U[j * 6 + 0];
#pragma unroll
        for (int m = 1; m < 6; ++m) if (m <= order) acc = rrow[m] * U[j * 6 + m] + acc;
        ru[j][k] = acc;
      }
    }
    const double* Dc = Dm + (cur * kNC) * N;
    double* Dn = Dm + ((cur ^ 1) * kNC) * N;
    constexpr int CH = lb_chunk(N, 2 * DSH_LANE_BANDED_CHUNK_SCALE);
    constexpr bool EXACT = N % CH == 0;
#pragma unroll 1
    for (int i0 = 0; i0 < N; i0 += CH) {
      double d[CH][6];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = EXACT ? i0 + c : (i0 + c < N ? i0 + c : N - 1);
#pragma unroll
        for (int k = 0; k < 6; ++k) d[c][k] = (k <= order) ? Dc[k * N + i] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (EXACT || i0 + c < N) {
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            if (j <= order) {
              double acc = d[c][0] * ru[j][0];
#pragma unroll
              for (int k = 1; k < 6; ++k) if (k <= order) acc = d[c][k] * ru[j][k] + acc;
              Dn[j * N + i0 + c] = acc;
            }
          }
        }
      }
    }
    cur ^= 1;  // std::mem::swap(diff, diff_tmp)
    opc = new_h * sAlpha[order];
    h = new_h;
    eta = C.r.eta_reset_ts;  // reset_eta_timestep_change
    new_h_out = new_h;
    return fabs(h) < o.min_timestep;  // true = StepSizeTooSmall
  };

  // _predict_forward (bdf.rs:674-692): y_predict = sum_{j<=order} D_j ; psi_neg_y0 = alpha_order * sum_{1<=j<=order} gamma_j D_j - y_predict
  auto predict_forward = [&]() __attribute__((always_inline)) {
    const double* Dc = Dm + (cur * kNC) * N;
    constexpr int CH = lb_chunk(N, 3 * DSH_LANE_BANDED_CHUNK_SCALE);
    constexpr bool EXACT = N % CH == 0;
    const double al = sAlpha[order];
#pragma unroll 1
    for (int i0 = 0; i0 < N; i0 += CH) {
      double d[CH][6];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = EXACT ? i0 + c : (i0 + c < N ? i0 + c : N - 1);
#pragma unroll
        for (int j = 0; j < 6; ++j) d[c][j] = (j <= order) ? Dc[j * N + i] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (EXACT || i0 + c < N) {
          double s = 0.0;
#pragma unroll
          for (int j = 0; j < 6; ++j) if (j <= order) s = s + d[c][j];
          double q = sGamma[1] * d[c][1];
#pragma unroll
          for (int j = 2; j < 6; ++j) if (j <= order) q = sGamma[j] * d[c][j] + 1.0 * q;
          q = q * al;
          q = q - s;
          yp[i0 + c] = s;
          psi[i0 + c] = q;
        }
      }
    }
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
      reset_jacobian(t);
      steps_since_rhs_jac = 0; steps_since_jac = 0; h_at_last_jac = c;
      eta = C.r.eta_reset;
      n_setups++;
    } else if (check_jac) {
      reset_jacobian(t);
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

  // interpolate_from_diff (bdf.rs:767-782), one pass: the time factors first, then every component through `put(i, value)`
  auto interpolate_to = [&](double te, auto&& put) __attribute__((always_inline)) {
    double tf[kMaxOrder];
    double time_factor = 1.0;
#pragma unroll
    for (int j = 0; j < kMaxOrder; ++j) {
      const double jt = (double)j;
      if (j < order) time_factor *= (te - (t - h * jt)) / (h * (1.0 + jt));
      tf[j] = time_factor;
    }
    const double* Dc = Dm + (cur * kNC) * N;
    constexpr int CH = lb_chunk(N, 3 * DSH_LANE_BANDED_CHUNK_SCALE);
    constexpr bool EXACT = N % CH == 0;
#pragma unroll 1
    for (int i0 = 0; i0 < N; i0 += CH) {
      double d[CH][6];
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = EXACT ? i0 + c : (i0 + c < N ? i0 + c : N - 1);
#pragma unroll
        for (int j = 0; j < 6; ++j) d[c][j] = (j <= order) ? Dc[j * N + i] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (EXACT || i0 + c < N) {
          double acc = d[c][0];
#pragma unroll
          for (int j = 0; j < kMaxOrder; ++j) if (j < order) acc = tf[j] * d[c][j + 1] + 1.0 * acc;
          put(i0 + c, acc);
        }
      }
    }
  };
  auto interpolate = [&](double te, double (&yv)[N]) __attribute__((always_inline)) { interpolate_to(te, [&](int i, double v) __attribute__((always_inline)) { yv[i] = v; }); };

  int col = 0;
  double te_next = t_eval[0];
  // OdeSolverMethod::solve (method.rs:227-258 over :881-961): C.steps_cap > 0 makes the launch write the state after EVERY accepted step (y_out [steps_cap][N][nb],
  // C.steps_t_out [steps_cap][nb]; columns beyond steps_cap are counted, not stored) instead of interpolating at save points — as in k_bdf_adaptive
  const bool steps_mode = C.steps_cap > 0;
  auto steps_write = [&](double tw, const double* yw) __attribute__((always_inline)) {
    if (col < C.steps_cap && active) {
      C.steps_t_out[(int64_t)col * nb + b] = tw;
      const int64_t c0 = (int64_t)col * N;
DSH_LB_STREAM
      for (int i = 0; i < N; ++i) y_out[(c0 + i) * nb + b] = yw[i];
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
  bool done = status != kRsOk || (!WAVE && !active);
  while (!done) {
    if (++guard > o.max_steps) { status = kRsMaxStepsExceeded; break; }
    // ================================================================ Bdf::step (bdf.rs:1277-1589)
    double safety = 0.0, error_norm = 0.0;
    const int old_err_fails = n_err_fails;
    bool convergence_fail = false;
    int niter = 0;
    if (!predicted) predict_forward();
    predicted = false;
    t_predict = t + h;
    double acc_m = 0.0, acc_p = 0.0;
    while (true) {
      // ---- NewtonNonlinearSolver::solve_in_place over NoLineSearch (newton.rs:13-36, line_search.rs:46-72)
      niter = 0;
      bool has_old = false;
      double old_norm = 0.0;
      bool solved = false;
      for (int it = 0; it < o.max_nonlinear_solver_iterations; ++it) {
        const double* xin = it == 0 ? yp : X;  // the iterate: y_predict itself in the first iteration
        Mdl::rhs(t_predict, *reinterpret_cast<const double (*)[N]>(xin), p, *reinterpret_cast<double (*)[N]>(w));
        if constexpr (MASS && !MSCALED) {  // F(y) = M (y - y0 + psi) - c f(y) as one more pass: w <- M (x + psi) + (-c) w   (op/bdf.rs:240-256)
          alignas(16) double tmpv[N];
DSH_LB_STREAM
          for (int i = 0; i < N; ++i) tmpv[i] = xin[i] + psi[i];
          Mdl::mass_gemv(t_predict, *reinterpret_cast<const double (*)[N]>(tmpv), p, -opc, *reinterpret_cast<double (*)[N]>(w));
        }
        // F(y) = M (y - y0 + psi) - c f(y) (op/bdf.rs:240-256), element by element as it enters the window of the forward sweep:
        // interchanges interleaved with the unit-lower-triangular solve; win[0..K] = entries j..j+K
        {
          constexpr int CH = lb_chunk(N, 6 * DSH_LANE_BANDED_CHUNK_SCALE);
          constexpr bool EXACT = N % CH == 0;
          double win[RW];
#pragma unroll
          for (int r = 0; r < RW; ++r) {
            if constexpr (MSCALED) win[r] = r < N ? (md[r] == 0.0 ? 0.0 : md[r] * (xin[r] + psi[r])) + (-opc) * w[r] : 0.0;
            else win[r] = r < N ? (MASS ? w[r] : 1.0 * (xin[r] + psi[r]) + (-opc) * w[r]) : 0.0;
          }
#pragma unroll 1
          for (int j0 = 0; j0 < N; j0 += CH) {
            // operands of the CH steps j0.. : pivot offset and multipliers of step j, and x, psi, f of the entry that enters the window after it
            int pv[CH];
            double lf[CH][K], xn[CH], pn[CH], fn[CH], mn[MSCALED ? CH : 1];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const int j = EXACT ? j0 + c : (j0 + c < N ? j0 + c : N - 1);
              const int jn = j + 1 + K < N ? j + 1 + K : N - 1;
              pv[c] = P[j] - j;
#pragma unroll
              for (int r = 0; r < K; ++r) lf[c][r] = Lf[r * N + j];
              xn[c] = xin[jn]; pn[c] = psi[jn]; fn[c] = w[jn];
              if constexpr (MSCALED) mn[c] = md[jn];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              if (EXACT || j0 + c < N) {
                const int j = j0 + c;
                const double top = win[0];
                double xv = top;
#pragma unroll
                for (int r = 1; r < RW; ++r) {
                  const bool sel = (r == pv[c]);
                  const double curv = win[r];
                  xv = sel ? curv : xv;
                  win[r] = sel ? top : curv;
                }
                w[j] = xv;
#pragma unroll
                for (int r = 1; r < RW; ++r) win[r] = (-xv) * lf[c][r - 1] + win[r];
#pragma unroll
                for (int r = 0; r + 1 < RW; ++r) win[r] = win[r + 1];
                if constexpr (MSCALED) win[RW - 1] = j + 1 + K < N ? (mn[c] == 0.0 ? 0.0 : mn[c] * (xn[c] + pn[c])) + (-opc) * fn[c] : 0.0;
                else win[RW - 1] = j + 1 + K < N ? (MASS ? fn[c] : 1.0 * (xn[c] + pn[c]) + (-opc) * fn[c]) : 0.0;
              }
            }
          }
        }
        // backward sweep with U (bandwidth 2K), fused with x <- x - delta
        bool solved_ok = true;
        {
          constexpr int CH = lb_chunk(N, 6 * DSH_LANE_BANDED_CHUNK_SCALE);
          constexpr bool EXACT = N % CH == 0;
          double u[CW];
#pragma unroll
          for (int q = 0; q < CW; ++q) { const int r = N - 1 - (CW - 1) + q; u[q] = r >= 0 ? w[r] : 0.0; }
#pragma unroll 1
          for (int it0 = N - 1; it0 >= 0; it0 -= CH) {
            double uf[CH][CW], wn[CH], xo[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              const int i = EXACT ? it0 - c : (it0 - c >= 0 ? it0 - c : 0);
#pragma unroll
              for (int d = 0; d < CW; ++d) uf[c][d] = Uf[d * N + (i - d >= 0 ? i - d : 0)];
              wn[c] = w[i - CW >= 0 ? i - CW : 0];
              xo[c] = xin[i];
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              if (EXACT || it0 - c >= 0) {
                const int i = it0 - c;
                const double diag = uf[c][0];
                if (diag == 0.0) solved_ok = false;
                const double xv = u[CW - 1] / diag;
                w[i] = xv;
                X[i] = xo[c] - xv;
#pragma unroll
                for (int d = 1; d < CW; ++d) u[CW - 1 - d] = (i - d >= 0) ? (-xv) * uf[c][d] + u[CW - 1 - d] : u[CW - 1 - d];
#pragma unroll
                for (int q = CW - 1; q > 0; --q) u[q] = u[q - 1];
                u[0] = i - CW >= 0 ? wn[c] : 0.0;
              }
            }
          }
        }
        const bool lu_ok = group_all<WAVE>(solved_ok);
        if (!lu_ok) break;  // LuSolveFailed
        double delta_ms;  // Convergence::norm of the update
        {
          double acc = 0.0;
          constexpr int CH = lb_chunk(N, 8 * DSH_LANE_BANDED_CHUNK_SCALE);
          constexpr bool EXACT = N % CH == 0;
#pragma unroll 1
          for (int i0 = 0; i0 < N; i0 += CH) {
            double dv[CH], yv[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) { const int i = EXACT ? i0 + c : (i0 + c < N ? i0 + c : N - 1); dv[c] = w[i]; yv[c] = yp[i]; }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
              if (EXACT || i0 + c < N) {
                const double term = dv[c] / (fabs(yv[c]) * rtol + AT(i0 + c));
                acc += term * term;
              }
            }
          }
          delta_ms = acc / (double)N;
        }
        const double norm = sqrt(group_norm<WAVE>(delta_ms));
        // Convergence::check_new_iteration (convergence.rs:68-139)
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
        const bool converged = !diverged && eta * norm < o.nonlinear_solver_tolerance;
        if (niter == 1) { has_old = true; old_norm = norm; }
        if (diverged) break;
        if (converged) { solved = true; break; }
      }
      n_newton += niter;
      if (!solved) {
        n_nl_fails += 1;
        if (n_nl_fails > o.max_nonlinear_solver_failures) { status = kRsTooManyNonlinearSolverFailures; break; }
        has_prev_err = false;
        if (convergence_fail) {
          double new_h;
          if (update_step_size(0.3, new_h)) { status = kRsStepSizeTooSmall; break; }
          jacobian_updates(new_h * sAlpha[order], JState::SecondConvergenceFail);
          predict_forward();
          t_predict = t + h;
        } else {
          jacobian_updates(h * sAlpha[order], JState::FirstConvergenceFail);
          convergence_fail = true;
        }
        continue;
      }
      // error_control (bdf.rs:812-843): norm of y_delta = x - y_predict against the CURRENT state y
      {
        double acc = 0.0;
        constexpr int CH = lb_chunk(N, 8 * DSH_LANE_BANDED_CHUNK_SCALE);
        constexpr bool EXACT = N % CH == 0;
#pragma unroll 1
        for (int i0 = 0; i0 < N; i0 += CH) {
          double xv[CH], pv[CH], yv[CH];
#pragma unroll
          for (int c = 0; c < CH; ++c) { const int i = EXACT ? i0 + c : (i0 + c < N ? i0 + c : N - 1); xv[c] = X[i]; pv[c] = yp[i]; yv[c] = y[i]; }
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            if (EXACT || i0 + c < N) {
              const double ydelta = xv[c] - pv[c];
              const double term = ydelta / (fabs(yv[c]) * rtol + AT(i0 + c));
              acc += term * term;
            }
          }
        }
        error_norm = fmax(0.0, group_norm<WAVE>(acc / (double)N) * sEc2[order - 1]);
      }
      const double maxiter = (double)o.max_nonlinear_solver_iterations;
      safety = 0.9 * (2.0 * maxiter + 1.0) / (2.0 * maxiter + (double)niter);
      if (error_norm <= 1.0) {
        // ---- accepted: _update_diff (bdf.rs:646-664), state update; in the same pass the two order-selection norms when they are due
        // (predict_error_control(order -+ 1) on the updated differences against the new state, bdf.rs:1494-1560) and the prediction of the next step
        double* Dc = Dm + (cur * kNC) * N;
        const double al = sAlpha[order];
        const bool order_selection_due = n_equal_steps + 1 > order;  // what the order selection below will see
        constexpr int CH = lb_chunk(N, 3 * DSH_LANE_BANDED_CHUNK_SCALE);
        constexpr bool EXACT = N % CH == 0;
#pragma unroll 1
        for (int i0 = 0; i0 < N; i0 += CH) {
          double xl[CH], pl[CH], d1[CH], dl[CH][6];
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            const int i = EXACT ? i0 + c : (i0 + c < N ? i0 + c : N - 1);
            xl[c] = X[i]; pl[c] = yp[i]; d1[c] = Dc[(order + 1) * N + i];
#pragma unroll
            for (int j = 0; j < 6; ++j) dl[c][j] = (j <= order) ? Dc[j * N + i] : 0.0;
          }
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            if (EXACT || i0 + c < N) {
              const int i = i0 + c;
              const double ypv = pl[c];
              const double ydelta = xl[c] - ypv;
              const double dk1 = d1[c];
              const double dk2 = ydelta - dk1;
              Dc[(order + 2) * N + i] = dk2;
              Dc[(order + 1) * N + i] = ydelta;
              double v[6];
              double upper = ydelta;
#pragma unroll
              for (int j = 5; j >= 0; --j) {
                v[j] = 0.0;
                if (j <= order) { const double nv = dl[c][j] + 1.0 * upper; Dc[j * N + i] = nv; upper = nv; v[j] = nv; }
              }
              y[i] = ypv;
              if (order_selection_due) {
                const double wgt = fabs(ypv) * rtol + AT(i);
                double vm = v[1];
#pragma unroll
                for (int j = 2; j < 6; ++j) vm = (j == order) ? v[j] : vm;
                const double tm = vm / wgt, tp = dk2 / wgt;
                acc_m += tm * tm;
                acc_p += tp * tp;
              }
              // _predict_forward of the next step at this order and step size; discarded if either changes before it
              double s = 0.0;
#pragma unroll
              for (int j = 0; j < 6; ++j) if (j <= order) s = s + v[j];
              double q = sGamma[1] * v[1];
#pragma unroll
              for (int j = 2; j < 6; ++j) if (j <= order) q = sGamma[j] * v[j] + 1.0 * q;
              q = q * al;
              q = q - s;
              yp[i] = s;
              psi[i] = q;
            }
          }
        }
        predicted = true;
        t = t_predict;
        break;
      }
      double factor = safety * pi_controller_raw(error_norm, has_prev_err, prev_err, o.pi_control_integral, o.pi_control_proportional, order + 1, det);
      has_prev_err = false;
      if (factor < o.min_timestep_shrink) factor = o.min_timestep_shrink;
      double new_h;
      if (update_step_size(factor, new_h)) { status = kRsStepSizeTooSmall; break; }
      jacobian_updates(new_h * sAlpha[order], JState::ErrorTestFail);
      predict_forward();
      t_predict = t + h;
      n_err_fails += 1;
      if (n_err_fails - old_err_fails >= o.max_error_test_failures) { status = kRsTooManyErrorTestFailures; break; }
    }
    if (status != kRsOk) break;
    n_steps += 1;
    steps_since_jac += 1; steps_since_rhs_jac += 1;  // JacobianUpdate::step
    prev_err = error_norm; has_prev_err = true;
    n_equal_steps += 1;
    if (n_equal_steps > order) {
      // order selection (bdf.rs:1494-1560); the two sums were accumulated by the accept pass
      const double inf = __builtin_huge_val();
      const double error_m_norm = order > 1 ? group_norm<WAVE>(acc_m / (double)N) * sEc2[order - 1] : inf;
      const double error_p_norm = order < kMaxOrder ? group_norm<WAVE>(acc_p / (double)N) * sEc2[order + 1] : inf;
      const double pi_i = o.pi_control_integral, pi_p = o.pi_control_proportional;
      const double f0c = pi_controller_raw(error_m_norm, has_prev_err, prev_err, pi_i, pi_p, order, det);
      const double f1c = pi_controller_raw(error_norm, has_prev_err, prev_err, pi_i, pi_p, order + 1, det);
      const double f2c = pi_controller_raw(error_p_norm, has_prev_err, prev_err, pi_i, pi_p, order + 2, det);
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
    int reason = 0;  // 0 internal, 1 tstop, 3 root
    if constexpr (Mdl::NROOTS > 0) {
      const int rr = check_root<Mdl, WAVE>(g0, rf_t0, *reinterpret_cast<const double (*)[N]>(y), t, p, interpolate, t_root, root_idx);
      if (rr == 2) { status = kRsRootBatchMismatch; break; }
      if (rr == 1) reason = 3;
    }
    if (reason == 0 && has_tstop) reason = handle_tstop();
    if (reason == 2) reason = 0;  // the reference unwraps / ignores this inside step()
    // ================================================================ solve_dense (method.rs:467-520): interpolated output
    const double upto = reason == 3 ? t_root : t;
    if (steps_mode) {  // InternalTimestep / TstopReached -> write_out (method.rs:907-921): state.y; a root is written below, at the root
      if (reason != 3) steps_write(t, y);
    } else
    while (col < C.r.n_eval && te_next <= upto) {
      const int64_t c0 = (int64_t)col * N;
      interpolate_to(te_next, [&](int i, double v) __attribute__((always_inline)) { if (active) y_out[(c0 + i) * nb + b] = v; });
      col++;
      if (col < C.r.n_eval) te_next = t_eval[col];
    }
    if constexpr (model_has_reset<Mdl>::value && !MASS && Mdl::NROOTS > 0) {
      if (reason == 3) {
        // A reset operator is configured (hybrid model; solve_dense, method.rs:774-797), as in k_bdf_adaptive: the state goes back to the root (state_mut_back,
        // bdf.rs:1232-1262: the step's polynomial at t_root), y <- reset(y, t), dy <- f(y, t) (apply_reset, bdf.rs:1017-1020 over state.rs:279-306), the stop time is
        // armed again — on the OLD differences and order, like the reference, so the step size may change here — and the next step restarts from the modified
        // state at first order (the `is_state_modified` branch of Bdf::step, bdf.rs:1290-1318).  The save points up to the root were written above.
        interpolate_to(t_root, [&](int i, double v) __attribute__((always_inline)) { w[i] = v; });
        t = t_root;
        Mdl::reset(t, *reinterpret_cast<const double (*)[N]>(w), p, *reinterpret_cast<double (*)[N]>(y));
        Mdl::rhs(t, *reinterpret_cast<const double (*)[N]>(y), p, *reinterpret_cast<double (*)[N]>(w));
        if (steps_mode) steps_write(t, y);  // method.rs:931-932: the reset state at the root time
        if (t < tstop) {
          has_tstop = true;  // set_stop_time (bdf.rs:1591-1600)
          { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          Mdl::root(t, *reinterpret_cast<const double (*)[N]>(y), p, g0);  // RootFinder::init
          rf_t0 = t;
          n_equal_steps = 0;
          order = 1;  // initialise_diff_to_first_order: columns 0 and 1 only, the others keep what they hold
          {
            double* const D0 = Drow(0);
            double* const D1 = Drow(1);
DSH_LB_STREAM
            for (int i = 0; i < N; ++i) { D0[i] = y[i]; D1[i] = w[i] * h; }
          }
          opc = h * sAlpha[1];
          jacobian_updates(h * sAlpha[1], JState::StepSuccess);
          has_prev_err = false;
          predicted = false;
          if (has_tstop) { const int r = handle_tstop(); if (r == 1) { status = kRsStopTimeAtCurrentTime; break; } if (r == 2) { status = kRsStopTimeBeforeCurrentTime; break; } }
          reason = 0;
        } else {
          done = true;  // the event sits on the last save point: TstopReached
          reason = 0;
        }
      }
    }
    if (reason == 3 && steps_mode) {  // method.rs:922-947 without a reset: state_mut_back(t_root), write_out, RootFound
      interpolate_to(t_root, [&](int i, double v) __attribute__((always_inline)) { w[i] = v; });
      steps_write(t_root, w);
      done = true;
    } else
    if (reason == 3) {  // state_mut_back(root_time): the column after the drained ones holds the state at the root
      if (col < C.r.n_eval) {
        const int64_t c0 = (int64_t)col * N;
        interpolate_to(t_root, [&](int i, double v) __attribute__((always_inline)) { if (active) y_out[(c0 + i) * nb + b] = v; });
        col++;
      }
      done = true;
    }
    if (reason == 1) done = true;
  }
  if (active) {
    if (ncols_out != nullptr) ncols_out[b] = col;
    if (t_root_out != nullptr) t_root_out[b] = root_idx >= 0 ? t_root : __builtin_nan("");
    if (root_idx_out != nullptr) root_idx_out[b] = root_idx;
    if (!steps_mode)
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
