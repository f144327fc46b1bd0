// Fused kernels of the host-driven BDF / SDIRK path (launch code: dsh_fused.hip).  Kernel templates live in this header so that the run-time-compiled
// model modules (dsh_jit.hip) instantiate exactly the same code for a user model as the library does for the built-in ones.
#pragma once
#include "dsh_device.hpp"
#include "dsh_lu_dev.hpp"
#include "dsh_models.hpp"

namespace dsh {

// ---- weighted mean-square of v against (|w| rtol + atol), sequential like Vector::squared_norm (nalgebra_serial.rs:395-408)
template <int N>
__device__ __forceinline__ double wms(const double (&v)[N], const double (&w)[N], const double (&atol)[N], double rtol) {
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double term = v[i] / (fabs(w[i]) * rtol + atol[i]);
    acc += term * term;
  }
  return acc / (double)N;
}

template <int N, bool BA>
__device__ __forceinline__ void load_atol(const double* __restrict__ atol, int64_t nb, int64_t b, double (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = BA ? atol[i] : atol[(int64_t)i * nb + b];
}

// NIT consecutive Newton iterations of the BDF residual in one launch (IS_SDIRK selects the SDIRK stage residual instead).
//   BDF  : delta = M(y + psi_neg_y0) - c f(y, t)             (op/bdf.rs:240-256)
//   SDIRK: delta = M k - h f(phi + c k, t)                   (op/sdirk.rs:229-244)
// The convergence decision stays on the host (it needs the max over ALL systems), so the launch is speculative by construction: it
// writes every intermediate iterate (y_out + i*n*nb) and one result record group per iteration; the host runs the reference's
// Convergence state machine over the NIT norms in order and takes the first iterate that converged.  Iterations 2..NIT reuse the
// factors / parameters / psi already in registers: they cost flops and 8n bytes of stores each, not another HBM pass nor another
// launch + host round trip.
template <class Mdl, bool IS_SDIRK, bool BA, bool WITH_ERR, int NIT>
__global__ void k_newton_iter(int64_t nb, double t, double c, double h, const double* y_in, double* y_out, const double* __restrict__ aux /*psi_neg_y0 | phi*/,
                              const double* __restrict__ p, const double* __restrict__ factors, const int32_t* __restrict__ piv,
                              const double* error_y, const double* __restrict__ y_old, const double* __restrict__ atol, double rtol,
                              unsigned long long* rec, unsigned int seq, unsigned long long* clk) {
  constexpr int N = Mdl::N, NP = Mdl::NP;
  if (clk != nullptr && threadIdx.x == 0) clk[2 * blockIdx.x] = wall_clock64();  // timing mode only: 100 MHz device clock at block start
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = b < nb;
  const int64_t bb = active ? b : 0;  // inactive lanes shadow system 0 (loads only) so that every lane reaches the block reductions
  double x[N], a[N], pp[NP], A[N * N], ey[N], at[N], yo[N];
  int P[N];
  // The iterate is read from y_in (first iteration of a step: the predictor — y_delta.copy_from(y_predict), bdf.rs:1326, without a copy
  // launch).  All loads are issued before the first use: one HBM round trip per launch.
  load_vec<N>(y_in, nb, bb, x);
  load_vec<N>(error_y, nb, bb, ey);
  load_vec<N>(aux, nb, bb, a);
  load_vec<NP>(p, nb, bb, pp);
  load_mat<N>(factors, nb, bb, A);
  load_piv<N>(piv, nb, bb, P);
  load_atol<N, BA>(atol, nb, bb, at);
  if constexpr (WITH_ERR) load_vec<N>(y_old, nb, bb, yo);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    double f[N], tmp[N], delta[N];
    if constexpr (!IS_SDIRK) {
      Mdl::rhs(t, x, pp, f);
#pragma unroll
      for (int i = 0; i < N; ++i) tmp[i] = x[i] + a[i];
      if constexpr (Mdl::HAS_MASS) {
#pragma unroll
        for (int i = 0; i < N; ++i) delta[i] = f[i];
        Mdl::mass_gemv(t, tmp, pp, -c, delta);
      } else {
#pragma unroll
        for (int i = 0; i < N; ++i) delta[i] = 1.0 * tmp[i] + (-c) * f[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) tmp[i] = c * x[i] + 1.0 * a[i];
      Mdl::rhs(t, tmp, pp, f);
      double beta = -h;
      if constexpr (Mdl::HAS_MASS) {
#pragma unroll
        for (int i = 0; i < N; ++i) delta[i] = f[i];
        Mdl::mass_gemv(t, x, pp, beta, delta);
      } else {
#pragma unroll
        for (int i = 0; i < N; ++i) delta[i] = 1.0 * x[i] + beta * f[i];
      }
    }
    const bool ok = lu_solve_reg<N>(A, P, delta);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] - delta[i];
    unsigned long long nrm_bits = 0ull, err_bits = 0ull, bad = 0ull;
    if (active) {
      store_vec<N>(y_out + (int64_t)it * N * nb, nb, b, x);
      nrm_bits = d2u(wms<N>(delta, ey, at, rtol));
      if constexpr (WITH_ERR) {
        double d[N];
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = x[i] - ey[i];
        err_bits = d2u(wms<N>(d, yo, at, rtol));
      }
      bad = ok ? 0ull : 1ull;
    }
    block_publish(nrm_bits, err_bits, bad, rec + (size_t)it * gridDim.x * kRecWords, seq);
  }
  if (clk != nullptr && threadIdx.x == 0) clk[2 * blockIdx.x + 1] = wall_clock64();
}

// Jacobian refresh + assembly of M - cJ + LU factorisation, one lane per system, A never leaves registers.
template <class Mdl>
__global__ void k_jac_factor(int64_t nb, double t, double c, const double* __restrict__ x, const double* __restrict__ p, int recompute,
                             double* __restrict__ rhs_jac, double* __restrict__ mass_jac, double* __restrict__ factors, int32_t* __restrict__ piv,
                             unsigned long long* singular_count, unsigned int epoch) {
  constexpr int N = Mdl::N, NP = Mdl::NP;
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long sing = 0ull;
  if (b < nb) {
    double J[N * N], Mm[N * N], A[N * N];
    int P[N];
    if (recompute) {
      double xr[N], pp[NP];
      load_vec<N>(x, nb, b, xr);
      load_vec<NP>(p, nb, b, pp);
      assemble_jacobian<Mdl>(t, xr, pp, J);
      store_mat<N>(rhs_jac, nb, b, J);
      if constexpr (Mdl::HAS_MASS) {
        assemble_mass<Mdl>(t, pp, Mm);
        store_mat<N>(mass_jac, nb, b, Mm);
      }
    } else {
      load_mat<N>(rhs_jac, nb, b, J);
      if constexpr (Mdl::HAS_MASS) load_mat<N>(mass_jac, nb, b, Mm);
    }
    if constexpr (!Mdl::HAS_MASS) {
#pragma unroll
      for (int e = 0; e < N * N; ++e) Mm[e] = (e / N == e % N) ? 1.0 : 0.0;  // Matrix::from_diagonal(ones), op/bdf.rs:138-141
    }
#pragma unroll
    for (int e = 0; e < N * N; ++e) A[e] = J[e] * (-c) + Mm[e];  // scale_add_and_assign(mass, -c, rhs_jac)
    bool s = false;
    lu_factor_reg<N>(A, P, s);
    store_mat<N>(factors, nb, b, A);
    store_piv<N>(piv, nb, b, P);
    sing = s ? 1ull : 0ull;
  }
  sing = wave_sum_u64(sing);
  if ((threadIdx.x & 63) == 0 && sing) publish_singular(singular_count, sing, epoch);
}

struct BdfCoeffs {
  double ru[36];    // (order+1)^2 column-major
  double gamma[6];
  double alpha;
  int order;
  int rescale;
};

// flat over n*nb elements: optional D <- D*RU (into diff_tmp), predictor and psi
__global__ void k_bdf_prepare(int64_t total, const double* __restrict__ diff, double* __restrict__ diff_tmp, BdfCoeffs cf, double* __restrict__ y_predict,
                              double* __restrict__ psi_neg_y0) {
  const int order = cf.order, ncol = cf.order + 1;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    double d[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) d[j] = j < ncol ? diff[(int64_t)j * total + idx] : 0.0;
    if (cf.rescale) {
      double nd[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        if (j < ncol) {
          double acc = d[0] * cf.ru[j * ncol + 0];
#pragma unroll
          for (int k = 1; k < 6; ++k) if (k < ncol) acc = d[k] * cf.ru[j * ncol + k] + acc;
          nd[j] = acc;
          diff_tmp[(int64_t)j * total + idx] = acc;
        } else nd[j] = 0.0;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) d[j] = nd[j];
    }
    if (y_predict == nullptr) continue;  // rescale only
    double yp = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) if (j < ncol) yp = yp + d[j];
    double psi = cf.gamma[1] * d[1];
#pragma unroll
    for (int j = 2; j < 6; ++j) if (j <= order) psi = cf.gamma[j] * d[j] + 1.0 * psi;
    psi = psi * cf.alpha;
    psi = psi - yp;
    y_predict[idx] = yp;
    psi_neg_y0[idx] = psi;
  }
}

// one lane per system: difference-array update, state update and the two order-selection norms.  NS > 0: compile-time number of
// states, the loop over the states is fully unrolled so that all loads of the launch are in flight together (one HBM round trip).
template <bool BA, int NS>
__global__ void k_bdf_accept(int64_t n_rt, int64_t nb, int order, double inv_h, double* __restrict__ diff, double* y_predict,
                             const double* __restrict__ y_new, double* __restrict__ y, double* __restrict__ dy, const double* __restrict__ atol, double rtol,
                             BdfCoeffs cf, double* __restrict__ psi_next, unsigned long long* rec, unsigned int seq) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long m_bits = 0ull, p_bits = 0ull;
  if (b < nb) {
    const int64_t n = NS > 0 ? NS : n_rt;
    const int64_t cs = n * nb;  // column stride
    double acc_m = 0.0, acc_p = 0.0;
#pragma unroll
    for (int64_t i = 0; i < n; ++i) {
      const int64_t e = i * nb + b;
      double yp = y_predict[e];
      double d = y_new[e] - yp;
      // D[:,k+2] = d - D[:,k+1]; D[:,k+1] = d; D[:,j] += D[:,j+1] for j = k..0   (all register indices compile-time: no scratch)
      double col[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) col[j] = j <= order + 1 ? diff[(int64_t)j * cs + e] : 0.0;
      double dk1 = 0.0;
#pragma unroll
      for (int j = 2; j < 7; ++j) if (j == order + 1) dk1 = col[j];
      double dk2 = d - dk1;
      diff[(int64_t)(order + 2) * cs + e] = dk2;
      diff[(int64_t)(order + 1) * cs + e] = d;
      double upper = d;  // value of column j+1 while walking down
      double new_k = 0.0, new_1 = 0.0;
      double nd[6];  // updated columns 0..order
#pragma unroll
      for (int j = 5; j >= 0; --j) {
        nd[j] = 0.0;
        if (j <= order) {
          double v = col[j] + 1.0 * upper;
          diff[(int64_t)j * cs + e] = v;
          nd[j] = v;
          if (j == order) new_k = v;
          if (j == 1) new_1 = v;
          upper = v;
        }
      }
      y[e] = yp;
      dy[e] = new_1 * inv_h;
      if (psi_next != nullptr) {
        // speculative prediction for the NEXT step at unchanged order and step size (same arithmetic as k_bdf_prepare):
        // y_predict = sum_{j<=k} D_j ; psi_neg_y0 = alpha*(sum_{1<=j<=k} gamma_j D_j) - y_predict
        double ypn = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) if (j <= order) ypn = ypn + nd[j];
        double psi = cf.gamma[1] * nd[1];
#pragma unroll
        for (int j = 2; j < 6; ++j) if (j <= order) psi = cf.gamma[j] * nd[j] + 1.0 * psi;
        psi = psi * cf.alpha;
        psi = psi - ypn;
        y_predict[e] = ypn;
        psi_next[e] = psi;
      }
      double ai = BA ? atol[i] : atol[e];
      double w = fabs(yp) * rtol + ai;
      double tm = new_k / w;
      double tp = dk2 / w;
      acc_m += tm * tm;
      acc_p += tp * tp;
    }
    m_bits = d2u(acc_m / (double)n);
    p_bits = d2u(acc_p / (double)n);
  }
  block_publish(m_bits, p_bits, 0ull, rec, seq);
}

// Accepted-step bookkeeping of step k AND the first NIT Newton iterations of step k+1 in ONE launch (the common case: the controller keeps the
// order, the step size and the LU factors): k_bdf_accept followed by k_newton_iter<..., WITH_ERR = true, NIT>, the new state, prediction and psi
// handed over in registers instead of through HBM.  Record group 0 = the accept launch's order-selection norms, groups 1..NIT = the Newton
// iterations'.  Every value is the same function of the same inputs as in the two separate launches.
template <class Mdl, bool BA, int NIT>
__global__ void k_accept_newton(int64_t nb, int order, double inv_h, double* __restrict__ diff, double* __restrict__ y_predict, const double* y_new /* may alias y_out */,
                                double* __restrict__ y, double* __restrict__ dy, const double* __restrict__ atol, double rtol, BdfCoeffs cf,
                                double* __restrict__ psi_next, double t_next, double c, double* y_out, const double* __restrict__ p,
                                const double* __restrict__ factors, const int32_t* __restrict__ piv, unsigned long long* rec, unsigned int seq) {
  constexpr int N = Mdl::N, NP = Mdl::NP;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = b < nb;
  const int64_t bb = active ? b : 0;
  const int64_t cs = (int64_t)N * nb;  // column stride of the difference array
  double x[N], a[N], ey[N], yo[N], at[N], pp[NP], A[N * N];
  int P[N];
  load_vec<NP>(p, nb, bb, pp);
  load_mat<N>(factors, nb, bb, A);
  load_piv<N>(piv, nb, bb, P);
  load_atol<N, BA>(atol, nb, bb, at);
  unsigned long long m_bits = 0ull, p_bits = 0ull;
  {
    double acc_m = 0.0, acc_p = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int64_t e = (int64_t)i * nb + bb;
      const double yp = y_predict[e];
      const double d = y_new[e] - yp;
      double col[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) col[j] = j <= order + 1 ? diff[(int64_t)j * cs + e] : 0.0;
      double dk1 = 0.0;
#pragma unroll
      for (int j = 2; j < 7; ++j) if (j == order + 1) dk1 = col[j];
      const double dk2 = d - dk1;
      if (active) { diff[(int64_t)(order + 2) * cs + e] = dk2; diff[(int64_t)(order + 1) * cs + e] = d; }
      double upper = d, new_k = 0.0, new_1 = 0.0;
      double nd[6];
#pragma unroll
      for (int j = 5; j >= 0; --j) {
        nd[j] = 0.0;
        if (j <= order) {
          const double v = col[j] + 1.0 * upper;
          if (active) diff[(int64_t)j * cs + e] = v;
          nd[j] = v;
          if (j == order) new_k = v;
          if (j == 1) new_1 = v;
          upper = v;
        }
      }
      double ypn = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) if (j <= order) ypn = ypn + nd[j];
      double psi = cf.gamma[1] * nd[1];
#pragma unroll
      for (int j = 2; j < 6; ++j) if (j <= order) psi = cf.gamma[j] * nd[j] + 1.0 * psi;
      psi = psi * cf.alpha;
      psi = psi - ypn;
      if (active) { y[e] = yp; dy[e] = new_1 * inv_h; y_predict[e] = ypn; psi_next[e] = psi; }
      const double w = fabs(yp) * rtol + at[i];
      const double tm = new_k / w, tp = dk2 / w;
      acc_m += tm * tm;
      acc_p += tp * tp;
      yo[i] = yp;   // the new state
      ey[i] = ypn;  // the new prediction = error_y = first iterate
      x[i] = ypn;
      a[i] = psi;
    }
    if (active) { m_bits = d2u(acc_m / (double)N); p_bits = d2u(acc_p / (double)N); }
  }
  block_publish(m_bits, p_bits, 0ull, rec, seq);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    double f[N], tmp[N], delta[N];
    Mdl::rhs(t_next, x, pp, f);
#pragma unroll
    for (int i = 0; i < N; ++i) tmp[i] = x[i] + a[i];
    if constexpr (Mdl::HAS_MASS) {
#pragma unroll
      for (int i = 0; i < N; ++i) delta[i] = f[i];
      Mdl::mass_gemv(t_next, tmp, pp, -c, delta);
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) delta[i] = 1.0 * tmp[i] + (-c) * f[i];
    }
    const bool ok = lu_solve_reg<N>(A, P, delta);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] - delta[i];
    unsigned long long nrm_bits = 0ull, err_bits = 0ull, bad = 0ull;
    if (active) {
      store_vec<N>(y_out + (int64_t)it * N * nb, nb, b, x);
      nrm_bits = d2u(wms<N>(delta, ey, at, rtol));
      double dd[N];
#pragma unroll
      for (int i = 0; i < N; ++i) dd[i] = x[i] - ey[i];
      err_bits = d2u(wms<N>(dd, yo, at, rtol));
      bad = ok ? 0ull : 1ull;
    }
    block_publish(nrm_bits, err_bits, bad, rec + (size_t)(1 + it) * gridDim.x * kRecWords, seq);
  }
}

}  // namespace dsh
