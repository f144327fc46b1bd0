// Model registry entry points of libdiffsol_hip.so (gfx950): dsh_model_{info,rhs,jac_mul,jacobian,mass_gemv,mass_matrix,init,root}.
// This is the 1:1 (unfused) form of the OdeEquations boundary: each call is one launch over the whole ensemble, one lane per system
// for the register-resident ("static") models, one thread per (state, system) for the run-time-sized models.  The reference's batched
// closures instead loop over the batch on the host with one D2H copy + one kernel per batch member
// (diffsol/src/ode_equations/test_models/exponential_decay.rs:14-21).
#include "dsh_internal.hpp"
#include "dsh_lu_dev.hpp"
#include "dsh_models.hpp"

using namespace dsh;

namespace {

enum class Op { Rhs, JacMul, Jacobian, MassGemv, MassMatrix, Init, Root };

template <class Mdl, Op OP>
__global__ void k_static_model(int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, const double* __restrict__ v,
                               double beta, double* __restrict__ y) {
  constexpr int N = Mdl::N, NP = Mdl::NP;
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  double pp[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) pp[k] = p[(int64_t)k * nb + b];
  if constexpr (OP == Op::Rhs) {
    double xr[N], yr[N];
    load_vec<N>(x, nb, b, xr);
    Mdl::rhs(t, xr, pp, yr);
    store_vec<N>(y, nb, b, yr);
  } else if constexpr (OP == Op::JacMul) {
    double xr[N], vr[N], yr[N];
    load_vec<N>(x, nb, b, xr);
    load_vec<N>(v, nb, b, vr);
    Mdl::jac_mul(t, xr, pp, vr, yr);
    store_vec<N>(y, nb, b, yr);
  } else if constexpr (OP == Op::Jacobian) {
    double xr[N], J[N * N];
    load_vec<N>(x, nb, b, xr);
    assemble_jacobian<Mdl>(t, xr, pp, J);
    store_mat<N>(y, nb, b, J);
  } else if constexpr (OP == Op::MassGemv) {
    double xr[N], yr[N];
    load_vec<N>(x, nb, b, xr);
    load_vec<N>(y, nb, b, yr);
    Mdl::mass_gemv(t, xr, pp, beta, yr);
    store_vec<N>(y, nb, b, yr);
  } else if constexpr (OP == Op::MassMatrix) {
    double Mm[N * N];
    assemble_mass<Mdl>(t, pp, Mm);
    store_mat<N>(y, nb, b, Mm);
  } else if constexpr (OP == Op::Init) {
    double yr[N];
    Mdl::init(t, pp, yr);
    store_vec<N>(y, nb, b, yr);
  } else if constexpr (OP == Op::Root) {
    if constexpr (Mdl::NROOTS > 0) {
      double xr[N], g[1];
      load_vec<N>(x, nb, b, xr);
      Mdl::root(t, xr, pp, g);
      y[b] = g[0];
    }
  }
}

template <Op OP>
int launch_static(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, const double* v, double beta,
                  double* y, bool* handled) {
  *handled = dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    hipLaunchKernelGGL((k_static_model<Mdl, OP>), grid_for(nb, ctx->block), dim3(ctx->block), 0, ctx->stream, nb, t, x, p, v, beta, y);
  });
  if (*handled) DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

// ------------------------------------------------------------------ run-time-sized models: one thread per (state i, system b)
constexpr int kBlock = 256;
inline dim3 ew_grid(int64_t total) {
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

// value of component i of f(x) (or of J(x) v when `v` is given) for the dynamic models; X(i)/V(i) read system b
template <class XF, class VF, class PF>
__device__ __forceinline__ double dyn_component(int model, int64_t n, double t, int64_t i, XF X, VF V, PF P, bool jac) {
  switch (model) {
    case DSH_MODEL_DYDT_Y2:  // test_models/dydt_y2.rs:9-19
      return jac ? V(i) * X(i) * 2.0 : X(i) * X(i);
    case DSH_MODEL_GAUSSIAN_DECAY:  // test_models/gaussian_decay.rs:12-23
      return (jac ? V(i) : X(i)) * P(i) * (-t);
    case DSH_MODEL_HEAT1D: {  // test_models/heat1d.rs:16-52 : D*(A u)/h^2, A = tridiag(1,-2,1), h = 1/(n+1)
      double h = 1.0 / (double)(n + 1);
      auto U = [&](int64_t k) { return jac ? V(k) : X(k); };
      double left = i > 0 ? U(i - 1) : 0.0, right = i + 1 < n ? U(i + 1) : 0.0;
      double heat = left + (-2.0) * U(i) + right;
      return P(0) * heat / (h * h);
    }
    case DSH_MODEL_SPM: {  // book/src/primer/src/spm.ds F_i: (I/3600, |I|/3600, A_neg x_neg + flux_neg e_last, A_pos x_pos + flux_pos e_last)
      const int64_t m = (n - 2) / 2;
      if (i < 2) return jac ? 0.0 : (i == 0 ? 0.0002777777777777778 * P(0) : 0.0002777777777777778 * fabs(P(0)));
      const bool pos = i >= 2 + m;
      const int64_t base = pos ? 2 + m : 2, k = i - base;
      auto U = [&](int64_t q) { return jac ? V(base + q) : X(base + q); };
      // spherical finite-volume Laplacian on m uniform shells times D/R^2 (constant6_ij / constant7_ij of spm.ds)
      const double s = pos ? 1.0e-3 : 0.39e-3, dr = 1.0 / (double)m, i0 = (double)k, i1 = (double)(k + 1);
      const double vol = i1 * i1 * i1 - i0 * i0 * i0;
      const double lower = 3.0 * i0 * i0 / vol / (dr * dr) * s, upper = k + 1 < m ? 3.0 * i1 * i1 / vol / (dr * dr) * s : 0.0;
      double acc = (-(lower + upper)) * U(k);
      if (k > 0) acc += lower * U(k - 1);
      if (k + 1 < m) acc += upper * U(k + 1);
      if (!jac && k == m - 1) acc += (pos ? 4.106800547504748e-12 * 243644455.17866704 : 3.2835305549534856e-12 * -520607810.21082705) * P(0);
      return acc;
    }
    case DSH_MODEL_ROBERTSON_ODE: {  // test_models/robertson_ode.rs:71-90
      int64_t g = (i / 3) * 3, r = i % 3;
      if (!jac) {
        if (r == 0) return -P(0) * X(g) + P(1) * X(g + 1) * X(g + 2);
        if (r == 1) return P(0) * X(g) - P(1) * X(g + 1) * X(g + 2) - P(2) * X(g + 1) * X(g + 1);
        return P(2) * X(g + 1) * X(g + 1);
      }
      if (r == 0) return -P(0) * V(g) + P(1) * V(g + 1) * X(g + 2) + P(1) * X(g + 1) * V(g + 2);
      if (r == 1) return P(0) * V(g) - P(1) * V(g + 1) * X(g + 2) - P(1) * X(g + 1) * V(g + 2) - 2.0 * P(2) * X(g + 1) * V(g + 1);
      return 2.0 * P(2) * X(g + 1) * V(g + 1);
    }
  }
  return 0.0;
}

__global__ void k_dyn_rhs(int model, int64_t n, int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, const double* __restrict__ v,
                          double* __restrict__ y) {
  int64_t total = n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = idx / nb, b = idx % nb;
    auto X = [&](int64_t k) { return x[k * nb + b]; };
    auto V = [&](int64_t k) { return v[k * nb + b]; };
    auto P = [&](int64_t k) { return p[k * nb + b]; };
    y[idx] = dyn_component(model, n, t, i, X, V, P, v != nullptr);
  }
}
// dense Jacobian entry (i,j) = component i of J e_j (same arithmetic as jac_mul with a unit vector)
__global__ void k_dyn_jacobian(int model, int64_t n, int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, double* __restrict__ jac) {
  int64_t total = n * n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t e = idx / nb, b = idx % nb;
    int64_t i = e % n, j = e / n;
    auto X = [&](int64_t k) { return x[k * nb + b]; };
    auto V = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
    auto P = [&](int64_t k) { return p[k * nb + b]; };
    jac[idx] = dyn_component(model, n, t, i, X, V, P, true);
  }
}
__global__ void k_dyn_init(int model, int64_t n, int64_t nb, double* __restrict__ y) {
  int64_t total = n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = idx / nb;
    double val = 0.0;
    switch (model) {
      case DSH_MODEL_DYDT_Y2: val = -200.0; break;
      case DSH_MODEL_GAUSSIAN_DECAY: val = 1.0; break;
      case DSH_MODEL_HEAT1D: { double h = 1.0 / (double)(n + 1); double xx = (double)(i + 1) * h; val = xx < 0.5 ? 2.0 * xx : 2.0 * (1.0 - xx); break; }
      case DSH_MODEL_ROBERTSON_ODE: val = (i % 3 == 0) ? 1.0 : 0.0; break;
      case DSH_MODEL_SPM: val = i < 2 ? 0.0 : (i < 2 + (n - 2) / 2 ? 0.8000000000000016 : 0.6000000000000001); break;  // spm.ds u_i
    }
    y[idx] = val;
  }
}

// ---- terminal voltage of the single-particle model (spm.ds varying2..5, out_i) and its two stop conditions
__device__ __forceinline__ double spm_clamp(double v, double lo, double hi) { return v < hi ? (v > lo ? v : lo) : hi; }
__device__ double spm_ocp_pos(double s) {
  return 2.16216 + 0.07645 * tanh(30.834 - 57.858397200000006 * s) + 2.1581 * tanh(52.294 - 53.412228 * s) - 0.14169 * tanh(11.0923 - 21.0852666 * s) +
         0.2051 * tanh(1.4684 - 5.829105600000001 * s) + 0.2531 * tanh(4.291641337386018 - 8.069908814589667 * s) - 0.02167 * tanh(-87.5 + 177.0 * s) +
         1e-06 * (1.0 / s + 1.0 / (-1.0 + s));
}
__device__ double spm_ocp_neg(double s) {
  return 0.194 + 1.5 * exp(-120.0 * s) + 0.0351 * tanh(-3.44578313253012 + 12.048192771084336 * s) - 0.0045 * tanh(-7.1344537815126055 + 8.403361344537815 * s) -
         0.035 * tanh(-18.466 + 20.0 * s) - 0.0147 * tanh(-14.705882352941176 + 29.41176470588235 * s) - 0.102 * tanh(-1.3661971830985917 + 7.042253521126761 * s) -
         0.022 * tanh(-54.8780487804878 + 60.975609756097555 * s) - 0.011 * tanh(-5.486725663716814 + 44.24778761061947 * s) +
         0.0155 * tanh(-3.6206896551724133 + 34.48275862068965 * s) + 1e-06 * (1.0 / s + 1.0 / (-1.0 + s));
}
__device__ double spm_voltage(double neg_in, double neg_out, double pos_in, double pos_out, double current) {
  const double sp = -0.4999999999999983 * pos_in + 1.4999999999999982 * pos_out, sn = -0.4999999999999983 * neg_in + 1.4999999999999984 * neg_out;
  const double cp = spm_clamp(-25608.96286546366 * pos_in + 76826.88859639116 * pos_out, 0.000512179257309275, 51217.92521874824);
  const double cn = spm_clamp(-12491.630996921805 * neg_in + 37474.892990765504 * neg_out, 0.000249832619938437, 24983.261744011077);
  const double stp = spm_clamp(sp, 1e-10, 0.9999999999), stn = spm_clamp(sn, 1e-10, 0.9999999999);
  const double eta_p = 0.05138515824298745 * asinh((-2.3508116177110145 * current) / (2.0 * ((1.8973665961010275e-05 * sqrt(cp)) * sqrt(51217.9257309275 - cp))));
  const double eta_n = 0.05138515824298745 * asinh((1.9590096814258458 * current) / (2.0 * ((0.0006324555320336759 * sqrt(cn)) * sqrt(24983.2619938437 - cn))));
  return (eta_p + spm_ocp_pos(stp)) - (eta_n + spm_ocp_neg(stn));
}
// roots of the run-time-sized models, one thread per system: g is nroots x nb batch-fastest
__global__ void k_dyn_root(int model, int64_t n, int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, double* __restrict__ g) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  if (model == DSH_MODEL_SPM) {
    const int64_t m = (n - 2) / 2;
    const double v = spm_voltage(x[(2 + m - 2) * nb + b], x[(2 + m - 1) * nb + b], x[(2 + 2 * m - 2) * nb + b], x[(2 + 2 * m - 1) * nb + b], p[b]);
    g[b] = -3.105 + v;
    g[nb + b] = 4.1 - v;
  }
}

bool is_dynamic_model(int model, int64_t size) {
  return model == DSH_MODEL_DYDT_Y2 || model == DSH_MODEL_GAUSSIAN_DECAY || model == DSH_MODEL_HEAT1D || model == DSH_MODEL_SPM ||
         (model == DSH_MODEL_ROBERTSON_ODE && size > 1);
}

}  // namespace

extern "C" {

int dsh_model_info(int model, int64_t size, int64_t* nstates, int64_t* nparams, int* has_mass, int64_t* nroots) {
  int64_t n = 0, np = 0, nr = 0;
  int hm = 0;
  bool ok = dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    n = Mdl::N; np = Mdl::NP; hm = Mdl::HAS_MASS ? 1 : 0; nr = Mdl::NROOTS;
  });
  if (!ok) {
    switch (model) {
      case DSH_MODEL_DYDT_Y2: n = size; np = 0; break;
      case DSH_MODEL_GAUSSIAN_DECAY: n = size; np = size; break;
      case DSH_MODEL_HEAT1D: n = size; np = 1; break;
      case DSH_MODEL_ROBERTSON_ODE: n = 3 * size; np = 3; break;
      case DSH_MODEL_SPM: n = 2 + 2 * (size <= 0 ? 20 : size); np = 1; nr = 2; break;
      default: set_error("dsh_model_info: unknown model id"); return DSH_E_INVALID;
    }
    DSH_REQUIRE(n > 0, "model size must be positive");
  }
  if (nstates) *nstates = n;
  if (nparams) *nparams = np;
  if (has_mass) *has_mass = hm;
  if (nroots) *nroots = nr;
  return DSH_OK;
}

int dsh_model_rhs(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, double* y) {
  bool handled = false;
  int rc = launch_static<Op::Rhs>(ctx, model, size, nb, t, x, p, nullptr, 0.0, y, &handled);
  if (rc != DSH_OK || handled) return rc;
  DSH_REQUIRE(is_dynamic_model(model, size), "unknown model id");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_rhs, ew_grid(n * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, t, x, p, (const double*)nullptr, y);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_model_jac_mul(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, const double* v, double* y) {
  bool handled = false;
  int rc = launch_static<Op::JacMul>(ctx, model, size, nb, t, x, p, v, 0.0, y, &handled);
  if (rc != DSH_OK || handled) return rc;
  DSH_REQUIRE(is_dynamic_model(model, size), "unknown model id");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_rhs, ew_grid(n * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, t, x, p, v, y);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_model_jacobian(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, double* jac) {
  bool handled = false;
  int rc = launch_static<Op::Jacobian>(ctx, model, size, nb, t, x, p, nullptr, 0.0, jac, &handled);
  if (rc != DSH_OK || handled) return rc;
  DSH_REQUIRE(is_dynamic_model(model, size), "unknown model id");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_jacobian, ew_grid(n * n * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, t, x, p, jac);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_model_mass_gemv(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, double beta, double* y) {
  bool handled = false;
  int rc = launch_static<Op::MassGemv>(ctx, model, size, nb, t, x, p, nullptr, beta, y, &handled);
  if (rc != DSH_OK || handled) return rc;
  // run-time-sized models have identity mass: y = x + beta*y
  DSH_REQUIRE(is_dynamic_model(model, size), "unknown model id");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  return dsh_vec_axpy(ctx, n, nb, 1.0, x, nb, beta, y);
}
int dsh_model_mass_matrix(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* p, double* mass) {
  bool handled = false;
  int rc = launch_static<Op::MassMatrix>(ctx, model, size, nb, t, nullptr, p, nullptr, 0.0, mass, &handled);
  if (rc != DSH_OK || handled) return rc;
  set_error("dsh_model_mass_matrix: model has no mass matrix");
  return DSH_E_UNSUPPORTED;
}
int dsh_model_init(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* p, double* y) {
  bool handled = false;
  int rc = launch_static<Op::Init>(ctx, model, size, nb, t, nullptr, p, nullptr, 0.0, y, &handled);
  if (rc != DSH_OK || handled) return rc;
  DSH_REQUIRE(is_dynamic_model(model, size), "unknown model id");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_init, ew_grid(n * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, y);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_model_root(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, double* g) {
  int64_t nroots = 0;
  int rc = dsh_model_info(model, size, nullptr, nullptr, nullptr, &nroots);
  if (rc != DSH_OK) return rc;
  DSH_REQUIRE(nroots > 0, "model has no root function");
  bool handled = false;
  rc = launch_static<Op::Root>(ctx, model, size, nb, t, x, p, nullptr, 0.0, g, &handled);
  if (rc != DSH_OK || handled) return rc;
  DSH_REQUIRE(is_dynamic_model(model, size), "unknown model id");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_root, grid_for(nb, ctx->block), dim3(ctx->block), 0, ctx->stream, model, n, nb, t, x, p, g);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

}  // extern "C"
