// Fused hot-path kernels of libdiffsol_hip.so (gfx950).
//
// The reference issues, per Newton iteration on its CUDA backend: the user RHS closure (one kernel + one D2H per batch member), copy,
// add_assign, axpy, nbatch x cusolverDnDgetrs (host loop), sub_assign, squared-norm kernel + alloc + blocking D2H + host reduction
// (diffsol-nl/src/line_search.rs:48-69, diffsol/src/op/bdf.rs:240-256, diffsol-la/src/linear_solver/cuda/lu.rs:127-145).
// Here one launch does all of it for the whole ensemble: one lane per system, state / parameters / LU factors streamed once from HBM with
// fully coalesced 8-byte-per-lane accesses (batch-fastest layout), everything else in registers, per-wave shuffle reduction of the
// weighted norms and one conditional atomicMax per workgroup.  Per n=3 system and iteration: 228 algorithmic bytes, ~60 flop => HBM-bound.
// Arithmetic order is exactly that of the unfused ops (and of the CPU oracle); built with -ffp-contract=off.
#include <algorithm>
#include <vector>

#include "dsh_internal.hpp"
#include "dsh_lu_dev.hpp"
#include "dsh_models.hpp"

using namespace dsh;

namespace {

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

}  // namespace

extern "C" {

int dsh_model_has_fused(int model, int64_t size) {
  return dispatch_static_model(model, size, [](auto) {}) ? 1 : 0;
}

static int newton_launch(dsh_ctx* ctx, bool is_sdirk, int model, int64_t size, int64_t nb, double t, double c, double h, int nit, const double* y_in,
                         double* y_out, const double* aux, const double* p, const dsh_lu* lu, const double* error_y, const double* y_old,
                         const double* atol, int64_t anb, double rtol, int64_t* ticket) {
  DSH_CHECK_NB(anb, nb);
  DSH_REQUIRE(ticket != nullptr && lu != nullptr, "null argument");
  DSH_REQUIRE(nit >= 1 && nit <= 4, "nit must be in 1..4");
  if (!lu->factored) { set_error("newton iteration: LU not initialised"); return DSH_E_NOT_SETUP; }
  unsigned long long* rec; unsigned int seq;
  const dim3 g = grid_for(nb, ctx->block), blk(ctx->block);
  int rc = begin_records(ctx, (int64_t)g.x * nit, &rec, &seq);
  if (rc != DSH_OK) return rc;
  const bool ba = anb == 1 && nb != 1;
  const bool with_err = y_old != nullptr;
  unsigned long long* clk = nullptr;
  if (ctx->timing) {
    rc = ensure_i32_scratch(ctx, 4 * (int64_t)g.x);  // 2 x u64 per workgroup
    if (rc != DSH_OK) return rc;
    clk = reinterpret_cast<unsigned long long*>(ctx->i32_scratch);
    DSH_HIP_CHECK(hipEventRecord(ctx->ev_start, ctx->stream));
  }
  bool ok = dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
#define DSH_NEWTON_LAUNCH(SD, BA, WE, NIT)                                                                                                      \
  hipLaunchKernelGGL((k_newton_iter<Mdl, SD, BA, WE, NIT>), g, blk, 0, ctx->stream, nb, t, c, h, y_in, y_out, aux, p, (const double*)lu->factors, \
                     (const int32_t*)lu->pivots, error_y, y_old, atol, rtol, rec, seq, clk)
#define DSH_NEWTON_NIT(SD, BA, WE)                                                                           \
  switch (nit) {                                                                                             \
    case 1: DSH_NEWTON_LAUNCH(SD, BA, WE, 1); break;                                                          \
    case 2: DSH_NEWTON_LAUNCH(SD, BA, WE, 2); break;                                                          \
    case 3: DSH_NEWTON_LAUNCH(SD, BA, WE, 3); break;                                                          \
    default: DSH_NEWTON_LAUNCH(SD, BA, WE, 4);                                                                \
  }
    if (is_sdirk) { if (ba) { DSH_NEWTON_NIT(true, true, false) } else { DSH_NEWTON_NIT(true, false, false) } }
    else if (with_err) { if (ba) { DSH_NEWTON_NIT(false, true, true) } else { DSH_NEWTON_NIT(false, false, true) } }
    else { if (ba) { DSH_NEWTON_NIT(false, true, false) } else { DSH_NEWTON_NIT(false, false, false) } }
#undef DSH_NEWTON_NIT
#undef DSH_NEWTON_LAUNCH
  });
  if (!ok) { set_error("newton iteration: model has no fused (register-resident) specialisation"); return DSH_E_UNSUPPORTED; }
  DSH_HIP_CHECK(hipGetLastError());
  if (ctx->timing) {  // event timing needs completed events: timed launches are synchronous
    DSH_HIP_CHECK(hipEventRecord(ctx->ev_stop, ctx->stream));
    DSH_HIP_CHECK(hipEventSynchronize(ctx->ev_stop));
    float ms = 0.f;
    DSH_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    ctx->timed_ms += (double)ms;
    ctx->timed_launches += 1;
    // device-clock span of the launch: max(block end) - min(block start), 100 MHz ticks
    std::vector<unsigned long long> h((size_t)2 * g.x);
    DSH_HIP_CHECK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    unsigned long long t_min = ~0ull, t_max = 0ull;
    for (unsigned i = 0; i < g.x; ++i) { t_min = std::min(t_min, h[2 * i]); t_max = std::max(t_max, h[2 * i + 1]); }
    ctx->timed_clock_ms += (double)(t_max - t_min) * 1e-5;  // 10 ns per tick
  }
  *ticket = ((int64_t)seq << 32) | ((int64_t)nit << 24) | (int64_t)g.x;
  return DSH_OK;
}

// ticket = seq << 32 | first group << 28 | groups << 24 | workgroups; out receives 3 doubles per record group
int dsh_reduction_wait(dsh_ctx* ctx, int64_t ticket, double* out) {
  DSH_REQUIRE(out != nullptr, "out is null");
  const unsigned int seq = (unsigned int)((uint64_t)ticket >> 32);
  const int groups = (int)((ticket >> 24) & 0xf), first = (int)((ticket >> 28) & 0xf);
  const int64_t nblocks = ticket & 0xffffffll;
  if (ctx->seq - seq >= (unsigned int)kRecRegions) { set_error("dsh_reduction_wait: ticket is too old, its result records have been reused"); return DSH_E_STALE; }
  for (int gi = 0; gi < (groups > 0 ? groups : 1); ++gi) {
    int rc = fetch_records(ctx, nblocks, seq, (int64_t)(first + gi) * nblocks);
    if (rc != DSH_OK) return rc;
    out[3 * gi + 0] = bits_to_double(ctx->res_m0);
    out[3 * gi + 1] = bits_to_double(ctx->res_m1);
    out[3 * gi + 2] = (double)ctx->res_cnt;
  }
  return DSH_OK;
}

int dsh_bdf_newton_iter_async(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double c, int nit, const double* y_in, double* y_out,
                              const double* psi_neg_y0, const double* p, const dsh_lu* lu, const double* error_y, const double* y_old, const double* atol,
                              int64_t anb, double rtol, int64_t* ticket) {
  return newton_launch(ctx, false, model, size, nb, t, c, 0.0, nit, y_in, y_out, psi_neg_y0, p, lu, error_y, y_old, atol, anb, rtol, ticket);
}
int dsh_bdf_newton_iter(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double c, const double* y_in, double* y_out, const double* psi_neg_y0,
                        const double* p, const dsh_lu* lu, const double* error_y, const double* y_old, const double* atol, int64_t anb, double rtol,
                        double* out) {
  int64_t ticket = 0;
  int rc = newton_launch(ctx, false, model, size, nb, t, c, 0.0, 1, y_in, y_out, psi_neg_y0, p, lu, error_y, y_old, atol, anb, rtol, &ticket);
  if (rc != DSH_OK) return rc;
  return dsh_reduction_wait(ctx, ticket, out);
}
int dsh_sdirk_newton_iter_async(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double h, double c, int nit, const double* k_in, double* k_out,
                                const double* phi, const double* p, const dsh_lu* lu, const double* error_y, const double* atol, int64_t anb, double rtol,
                                int64_t* ticket) {
  return newton_launch(ctx, true, model, size, nb, t, c, h, nit, k_in, k_out, phi, p, lu, error_y, nullptr, atol, anb, rtol, ticket);
}
int dsh_sdirk_newton_iter(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double h, double c, const double* k_in, double* k_out,
                          const double* phi, const double* p, const dsh_lu* lu, const double* error_y, const double* atol, int64_t anb, double rtol,
                          double* out) {
  int64_t ticket = 0;
  int rc = newton_launch(ctx, true, model, size, nb, t, c, h, 1, k_in, k_out, phi, p, lu, error_y, nullptr, atol, anb, rtol, &ticket);
  if (rc != DSH_OK) return rc;
  return dsh_reduction_wait(ctx, ticket, out);
}

int dsh_jac_factor(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double c, const double* x, const double* p, int recompute, double* rhs_jac,
                   double* mass_jac, dsh_lu* lu) {
  DSH_REQUIRE(lu != nullptr && rhs_jac != nullptr, "null argument");
  lu->singular_epoch += 1;  // the kernel adds (epoch << 32 | 1) per singular system: no reset launch needed between factorisations
  bool ok = dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    hipLaunchKernelGGL((k_jac_factor<Mdl>), grid_for(nb, ctx->block), dim3(ctx->block), 0, ctx->stream, nb, t, c, x, p, recompute, rhs_jac, mass_jac,
                       lu->factors, lu->pivots, lu->singular, lu->singular_epoch);
  });
  if (!ok) { set_error("dsh_jac_factor: model has no fused (register-resident) specialisation"); return DSH_E_UNSUPPORTED; }
  DSH_HIP_CHECK(hipGetLastError());
  lu->factored = true;
  return DSH_OK;
}

int dsh_bdf_prepare_step(dsh_ctx* ctx, int64_t n, int64_t nb, int order, const double* diff, double* diff_tmp, const double* ru_host,
                         const double* gamma_host, double alpha, double* y_predict, double* psi_neg_y0) {
  DSH_REQUIRE(order >= 1 && order <= 5, "order must be in 1..5");
  BdfCoeffs cf;
  for (int k = 0; k < 36; ++k) cf.ru[k] = 0.0;
  for (int k = 0; k < 6; ++k) cf.gamma[k] = k <= order ? gamma_host[k] : 0.0;
  cf.alpha = alpha; cf.order = order; cf.rescale = ru_host != nullptr;
  if (ru_host) for (int k = 0; k < (order + 1) * (order + 1); ++k) cf.ru[k] = ru_host[k];
  int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_bdf_prepare, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, total, diff, diff_tmp, cf, y_predict, psi_neg_y0);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

static int accept_launch(dsh_ctx* ctx, int64_t n, int64_t nb, int order, double h, double* diff, double* y_predict, const double* y_new, double* y, double* dy,
                         const double* atol, int64_t anb, double rtol, const double* gamma_host, double alpha, double* psi_neg_y0_next, int64_t* ticket) {
  DSH_REQUIRE(order >= 1 && order <= 5, "order must be in 1..5");
  DSH_CHECK_NB(anb, nb);
  unsigned long long* rec; unsigned int seq;
  dim3 g = grid_for(nb, ctx->block), blk(ctx->block);
  int rc = begin_records(ctx, g.x, &rec, &seq);
  if (rc != DSH_OK) return rc;
  const double inv_h = 1.0 / h;
  BdfCoeffs cf;
  for (int k = 0; k < 36; ++k) cf.ru[k] = 0.0;
  for (int k = 0; k < 6; ++k) cf.gamma[k] = (psi_neg_y0_next && k <= order) ? gamma_host[k] : 0.0;
  cf.alpha = alpha; cf.order = order; cf.rescale = 0;
  const bool ba = anb == 1 && nb != 1;
#define DSH_ACCEPT_LAUNCH(BA, NS) \
  hipLaunchKernelGGL((k_bdf_accept<BA, NS>), g, blk, 0, ctx->stream, n, nb, order, inv_h, diff, y_predict, y_new, y, dy, atol, rtol, cf, psi_neg_y0_next, rec, seq)
#define DSH_ACCEPT_CASE(NS) case NS: if (ba) DSH_ACCEPT_LAUNCH(true, NS); else DSH_ACCEPT_LAUNCH(false, NS); break;
  switch (n) {
    DSH_ACCEPT_CASE(1) DSH_ACCEPT_CASE(2) DSH_ACCEPT_CASE(3) DSH_ACCEPT_CASE(4)
    default: if (ba) DSH_ACCEPT_LAUNCH(true, 0); else DSH_ACCEPT_LAUNCH(false, 0);
  }
#undef DSH_ACCEPT_CASE
#undef DSH_ACCEPT_LAUNCH
  DSH_HIP_CHECK(hipGetLastError());
  *ticket = ((int64_t)seq << 32) | ((int64_t)1 << 24) | (int64_t)g.x;
  return DSH_OK;
}

int dsh_bdf_accept_step_async(dsh_ctx* ctx, int64_t n, int64_t nb, int order, double h, double* diff, double* y_predict, const double* y_new, double* y,
                              double* dy, const double* atol, int64_t anb, double rtol, const double* gamma_host, double alpha, double* psi_neg_y0_next,
                              int64_t* ticket) {
  DSH_REQUIRE(ticket != nullptr, "ticket is null");
  if (n * nb == 0) { *ticket = 0; return DSH_OK; }
  return accept_launch(ctx, n, nb, order, h, diff, y_predict, y_new, y, dy, atol, anb, rtol, gamma_host, alpha, psi_neg_y0_next, ticket);
}

int dsh_bdf_accept_newton_async(dsh_ctx* ctx, int model, int64_t size, int64_t nb, int order, double h, double* diff, double* y_predict, const double* y_new,
                                double* y, double* dy, const double* atol, int64_t anb, double rtol, const double* gamma_host, double alpha,
                                double* psi_neg_y0_next, double t_next, double c, int nit, double* y_out, const double* p, const dsh_lu* lu,
                                int64_t* accept_ticket, int64_t* newton_ticket) {
  DSH_REQUIRE(accept_ticket != nullptr && newton_ticket != nullptr && lu != nullptr && psi_neg_y0_next != nullptr, "null argument");
  DSH_REQUIRE(order >= 1 && order <= 5, "order must be in 1..5");
  DSH_REQUIRE(nit >= 1 && nit <= 4, "nit must be in 1..4");
  DSH_CHECK_NB(anb, nb);
  if (!lu->factored) { set_error("accept+newton: LU not initialised"); return DSH_E_NOT_SETUP; }
  if (nb == 0) { *accept_ticket = 0; *newton_ticket = 0; return DSH_OK; }
  unsigned long long* rec; unsigned int seq;
  const dim3 g = grid_for(nb, ctx->block), blk(ctx->block);
  int rc = begin_records(ctx, (int64_t)g.x * (nit + 1), &rec, &seq);
  if (rc != DSH_OK) return rc;
  BdfCoeffs cf;
  for (int k = 0; k < 36; ++k) cf.ru[k] = 0.0;
  for (int k = 0; k < 6; ++k) cf.gamma[k] = k <= order ? gamma_host[k] : 0.0;
  cf.alpha = alpha; cf.order = order; cf.rescale = 0;
  const double inv_h = 1.0 / h;
  const bool ba = anb == 1 && nb != 1;
  bool ok = dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
#define DSH_AN_LAUNCH(BA, NIT)                                                                                                                          \
  hipLaunchKernelGGL((k_accept_newton<Mdl, BA, NIT>), g, blk, 0, ctx->stream, nb, order, inv_h, diff, y_predict, y_new, y, dy, atol, rtol, cf,        \
                     psi_neg_y0_next, t_next, c, y_out, p, (const double*)lu->factors, (const int32_t*)lu->pivots, rec, seq)
#define DSH_AN_NIT(BA)                                                                                                                                  \
  switch (nit) { case 1: DSH_AN_LAUNCH(BA, 1); break; case 2: DSH_AN_LAUNCH(BA, 2); break; case 3: DSH_AN_LAUNCH(BA, 3); break; default: DSH_AN_LAUNCH(BA, 4); }
    if (ba) { DSH_AN_NIT(true) } else { DSH_AN_NIT(false) }
#undef DSH_AN_NIT
#undef DSH_AN_LAUNCH
  });
  if (!ok) { set_error("accept+newton: model has no fused (register-resident) specialisation"); return DSH_E_UNSUPPORTED; }
  DSH_HIP_CHECK(hipGetLastError());
  *accept_ticket = ((int64_t)seq << 32) | ((int64_t)0 << 28) | ((int64_t)1 << 24) | (int64_t)g.x;
  *newton_ticket = ((int64_t)seq << 32) | ((int64_t)1 << 28) | ((int64_t)nit << 24) | (int64_t)g.x;
  return DSH_OK;
}

int dsh_bdf_accept_step(dsh_ctx* ctx, int64_t n, int64_t nb, int order, double h, double* diff, double* y_predict, const double* y_new, double* y,
                        double* dy, const double* atol, int64_t anb, double rtol, const double* gamma_host, double alpha, double* psi_neg_y0_next,
                        int want_norms, double* out) {
  if (n * nb == 0) return DSH_OK;
  int64_t ticket = 0;
  int rc = accept_launch(ctx, n, nb, order, h, diff, y_predict, y_new, y, dy, atol, anb, rtol, gamma_host, alpha, psi_neg_y0_next, &ticket);
  if (rc != DSH_OK) return rc;
  if (want_norms) {
    DSH_REQUIRE(out != nullptr, "out is null");
    double r[3];
    rc = dsh_reduction_wait(ctx, ticket, r);
    if (rc != DSH_OK) return rc;
    out[0] = r[0];
    out[1] = r[1];
  }
  return DSH_OK;
}

}  // extern "C"
