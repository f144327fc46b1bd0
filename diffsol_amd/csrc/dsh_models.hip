// Model registry entry points of libdiffsol_hip.so (gfx950): dsh_model_{info,rhs,jac_mul,jacobian,mass_gemv,mass_matrix,init,root}.
// This is the 1:1 (unfused) form of the OdeEquations boundary: each call is one launch over the whole ensemble, one lane per system
// for the register-resident ("static") models, one thread per (state, system) for the run-time-sized models.  The reference's batched
// closures instead loop over the batch on the host with one D2H copy + one kernel per batch member
// (diffsol/src/ode_equations/test_models/exponential_decay.rs:14-21).
#include "dsh_internal.hpp"
#include "dsh_lu_dev.hpp"
#include "dsh_models.hpp"
#include "dsh_models_dyn.hpp"

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
    y[idx] = dyn_init_value(model, n, i);
  }
}

// roots of the run-time-sized models, one thread per system: g is nroots x nb batch-fastest
__global__ void k_dyn_root(int model, int64_t n, int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, double* __restrict__ g) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  double gg[2] = {0.0, 0.0};
  auto X = [&](int64_t k) { return x[k * nb + b]; };
  auto P = [&](int64_t k) { return p[k * nb + b]; };
  const int nr = dyn_root_values(model, n, t, X, P, gg);
  for (int r = 0; r < nr; ++r) g[(int64_t)r * nb + b] = gg[r];
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
