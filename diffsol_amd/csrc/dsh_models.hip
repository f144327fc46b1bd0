// Model registry entry points of libdiffsol_hip.so (gfx950): dsh_model_{info,rhs,jac_mul,jacobian,mass_gemv,mass_matrix,init,root}.
// This is the 1:1 (unfused) form of the OdeEquations boundary: each call is one launch over the whole ensemble, one lane per system
// for the register-resident ("static") models, one thread per (state, system) for the run-time-sized models.  The reference's batched
// closures instead loop over the batch on the host with one D2H copy + one kernel per batch member
// (diffsol/src/ode_equations/test_models/exponential_decay.rs:14-21).
#include "dsh_internal.hpp"
#include "dsh_lu_dev.hpp"
#include "dsh_models.hpp"
#include "dsh_models_dyn.hpp"
#include "dsh_model_kernels.hpp"
#include "dsh_jit.hpp"

using namespace dsh;

namespace {

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
// SdirkCallable::call_inplace (op/sdirk.rs:229-244) of an identity-mass model as ONE pass: tmp = phi + c k (copy, then axpy: c * k + 1.0 * phi), f = rhs(tmp, t),
// out = k - h f (axpy: 1.0 * k + (-h) * f) — the arithmetic of the three vector kernels and k_dyn_rhs, entry by entry; the stencil's neighbours of tmp are
// formed again from phi and k instead of being read back
__global__ void k_dyn_sdirk_residual(int model, int64_t n, int64_t nb, double t, double c, double h, const double* __restrict__ phi, const double* __restrict__ kk,
                                     const double* __restrict__ p, double* __restrict__ out) {
  int64_t total = n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = idx / nb, b = idx % nb;
    auto X = [&](int64_t k) { return c * kk[k * nb + b] + 1.0 * phi[k * nb + b]; };
    auto V = [&](int64_t) { return 0.0; };
    auto P = [&](int64_t k) { return p[k * nb + b]; };
    const double f = dyn_component(model, n, t, i, X, V, P, false);
    out[idx] = 1.0 * kk[idx] + (-h) * f;
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
// the same entries on the declared band only (row i, columns i-kl .. i+ku); everything else of the container is left as it is (zero, by the caller's promise)
// PACKED: into a band container (entry (i, j) at ((j - i + kl) * n + i) * nb + b — the loop index itself; the corners outside the matrix are written as zeros)
template <bool PACKED>
__global__ void k_dyn_jacobian_band(int model, int64_t n, int64_t nb, int kl, int ku, double t, const double* __restrict__ x, const double* __restrict__ p,
                                    double* __restrict__ jac) {
  const int64_t w = kl + ku + 1, total = n * w * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / nb, b = idx % nb;
    const int64_t i = e % n, j = i + e / n - kl;
    if (j < 0 || j >= n) { if (PACKED) jac[idx] = 0.0; continue; }
    auto X = [&](int64_t k) { return x[k * nb + b]; };
    auto V = [&](int64_t k) { return k == j ? 1.0 : 0.0; };
    auto P = [&](int64_t k) { return p[k * nb + b]; };
    jac[PACKED ? idx : (j * n + i) * nb + b] = dyn_component(model, n, t, i, X, V, P, true);
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

// y = M x + beta y for the run-time-sized registry models with a (diagonal, 0 / 1) mass matrix: the reference's closures entry by entry (x + beta y, or beta y on an algebraic row)
__global__ void k_dyn_mass_gemv(int model, int64_t n, int64_t nb, const double* __restrict__ x, double beta, double* __restrict__ y) {
  int64_t total = n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / nb;
    y[idx] = dyn_mass_diag(model, n, i) != 0.0 ? x[idx] + beta * y[idx] : beta * y[idx];
  }
}
// the dense n x n mass matrix (what LinearOp::matrix assembles from gemv with unit vectors: 1 * e_j + 0 * y)
__global__ void k_dyn_mass_matrix(int model, int64_t n, int64_t nb, double* __restrict__ mass) {
  int64_t total = n * n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / nb, i = e % n, j = e / n;
    mass[idx] = i == j ? dyn_mass_diag(model, n, i) : 0.0;
  }
}
bool dyn_model_has_mass(int model) { return model == DSH_MODEL_HEAT2D || model == DSH_MODEL_FOODWEB; }

bool is_dynamic_model(int model, int64_t size) {
  return model == DSH_MODEL_DYDT_Y2 || model == DSH_MODEL_GAUSSIAN_DECAY || model == DSH_MODEL_HEAT1D || model == DSH_MODEL_SPM || model == DSH_MODEL_HEAT2D || model == DSH_MODEL_FOODWEB ||
         (model == DSH_MODEL_ROBERTSON_ODE && size > 1);
}

// ------------------------------------------------------------------ run-time-compiled models (dsh_jit.hip)
const char* static_op_name(Op op) {
  switch (op) {
    case Op::Rhs: return "dsh::k_static_model<dsh::JitModel, dsh::Op::Rhs>";
    case Op::JacMul: return "dsh::k_static_model<dsh::JitModel, dsh::Op::JacMul>";
    case Op::Jacobian: return "dsh::k_static_model<dsh::JitModel, dsh::Op::Jacobian>";
    case Op::MassGemv: return "dsh::k_static_model<dsh::JitModel, dsh::Op::MassGemv>";
    case Op::MassMatrix: return "dsh::k_static_model<dsh::JitModel, dsh::Op::MassMatrix>";
    case Op::Init: return "dsh::k_static_model<dsh::JitModel, dsh::Op::Init>";
    case Op::Root: return "dsh::k_static_model<dsh::JitModel, dsh::Op::Root>";
    case Op::RhsSens: return "dsh::k_static_model<dsh::JitModel, dsh::Op::RhsSens>";
    case Op::InitSens: return "dsh::k_static_model<dsh::JitModel, dsh::Op::InitSens>";
    case Op::Reset: return "dsh::k_static_model<dsh::JitModel, dsh::Op::Reset>";
    default: return "dsh::k_static_model<dsh::JitModel, dsh::Op::Out>";
  }
}
int jit_model_op(dsh_ctx* ctx, int model, Op op, int64_t nb, double t, const double* x, const double* p, const double* v, double beta, double* y) {
  const JitInfo* ji = jit_info(model);
  if (!ji) return DSH_E_INVALID;
  if (op == Op::Root) DSH_REQUIRE(ji->nroots > 0, "model has no root function");
  if (op == Op::Out) DSH_REQUIRE(ji->nout > 0, "model has no out_i");
  if (ji->form == DSH_JIT_FORM_STATIC)
    return jit_launch(ctx, model, "dsh_model_kernels.hpp", "ops", jit_static_op_names(), static_op_name(op), grid_for(nb, ctx->block), dim3(ctx->block), 0, nb, t, x, p,
                      v, beta, y);
  static const std::vector<std::string> none;
  const char* hdr = "dsh_jit_dyn_kernels.hpp";
  const int64_t n = ji->n;
  switch (op) {
    case Op::Rhs:
    case Op::JacMul: return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_rhs", ew_grid(n * nb), dim3(kBlock), 0, nb, t, x, p, op == Op::JacMul ? v : (const double*)nullptr, y);
    case Op::Jacobian:
      if (ji->jac_nnz > 0 && !(std::getenv("DSH_JAC_SPARSE") && std::getenv("DSH_JAC_SPARSE")[0] == '0')) {
        // large sparse model: zero the matrix, then the structural nonzeros only (the same evaluation per entry; the n^2 - nnz others are the +0 the dense form computes as
        // 0 * x for FINITE states — with a non-finite state the dense form's 0 * inf is NaN where this one writes 0: the two agree on every state an integrator accepts)
        DSH_HIP_CHECK(hipMemsetAsync(y, 0, sizeof(double) * (size_t)(n * n * nb), ctx->stream));
        return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_jacobian_sparse", ew_grid(ji->jac_nnz * nb), dim3(kBlock), 0, nb, t, x, p, y);
      }
      return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_jacobian", ew_grid(n * n * nb), dim3(kBlock), 0, nb, t, x, p, y);
    case Op::MassGemv: return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_mass_gemv", ew_grid(n * nb), dim3(kBlock), 0, nb, t, x, p, beta, y);
    case Op::MassMatrix: return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_mass_matrix", ew_grid(n * n * nb), dim3(kBlock), 0, nb, t, p, y);
    case Op::Init: return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_init", ew_grid(n * nb), dim3(kBlock), 0, nb, t, p, y);
    case Op::Root: return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_root_out", ew_grid(ji->nroots * nb), dim3(kBlock), 0, nb, t, x, p, (int)0, y);
    case Op::Reset: return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_reset", ew_grid(n * nb), dim3(kBlock), 0, nb, t, x, p, y);
    case Op::RhsSens: return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_sens", ew_grid(n * ji->np * nb), dim3(kBlock), 0, nb, t, x, p, (int)0, y);
    case Op::InitSens: return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_sens", ew_grid(n * ji->np * nb), dim3(kBlock), 0, nb, t, x, p, (int)1, y);
    default: return jit_launch(ctx, model, hdr, "ops", none, "k_jit_dyn_root_out", ew_grid(ji->nout * nb), dim3(kBlock), 0, nb, t, x, p, (int)1, y);
  }
}

}  // namespace

extern "C" {

int dsh_model_info(int model, int64_t size, int64_t* nstates, int64_t* nparams, int* has_mass, int64_t* nroots) {
  int64_t n = 0, np = 0, nr = 0;
  int hm = 0;
  if (is_jit_model(model)) {
    const JitInfo* ji = jit_info(model);
    if (!ji) return DSH_E_INVALID;
    if (nstates) *nstates = ji->n;
    if (nparams) *nparams = ji->np;
    if (has_mass) *has_mass = ji->has_mass;
    if (nroots) *nroots = ji->nroots;
    return DSH_OK;
  }
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
      case DSH_MODEL_HEAT2D: { const int64_t m = size <= 0 ? 10 : size; DSH_REQUIRE(m >= 3, "heat2d: a grid of at least 3 x 3"); n = m * m; np = 1; hm = 1; break; }
      case DSH_MODEL_FOODWEB: { const int64_t m = size <= 0 ? 10 : size; DSH_REQUIRE(m >= 2, "foodweb: a grid of at least 2 x 2"); n = 2 * m * m; np = 2; hm = 1; break; }
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
  DSH_ENTER(ctx);
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::Rhs, nb, t, x, p, nullptr, 0.0, y);
  bool handled = false;
  int rc = launch_static<Op::Rhs>(ctx, model, size, nb, t, x, p, nullptr, 0.0, y, &handled);
  if (rc != DSH_OK || handled) return rc;
  DSH_REQUIRE(is_dynamic_model(model, size), "unknown model id");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_rhs, ew_grid(n * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, t, x, p, (const double*)nullptr, y);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
}  // extern "C"
namespace dsh {
bool model_has_staged_newton(int model, int64_t size) {
  if (is_jit_model(model) || !is_dynamic_model(model, size)) return false;
  int64_t n; int hm = 0;
  return dsh_model_info(model, size, &n, nullptr, &hm, nullptr) == DSH_OK && hm == 0;
}
// out = k - h f(phi + c k, t) for a run-time-sized registry model without a mass matrix, one launch; false when the model is of another kind
bool model_dyn_sdirk_residual(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double c, double h, const double* phi, const double* k, const double* p,
                              double* out) {
  if (!model_has_staged_newton(model, size)) return false;
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_sdirk_residual, ew_grid(n * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, t, c, h, phi, k, p, out);
  return true;
}
}  // namespace dsh
extern "C" {
int dsh_model_jac_mul(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, const double* v, double* y) {
  DSH_ENTER(ctx);
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::JacMul, nb, t, x, p, v, 0.0, y);
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
  DSH_ENTER(ctx);
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::Jacobian, nb, t, x, p, nullptr, 0.0, jac);
  bool handled = false;
  int rc = launch_static<Op::Jacobian>(ctx, model, size, nb, t, x, p, nullptr, 0.0, jac, &handled);
  if (rc != DSH_OK || handled) return rc;
  DSH_REQUIRE(is_dynamic_model(model, size), "unknown model id");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_jacobian, ew_grid(n * n * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, t, x, p, jac);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
// The Jacobian of a run-time-sized registry model with a declared band (dsh_model_band), written on the band only: for a dense container whose entries
// outside the band are already zero (a freshly zeroed matrix that only this function has written).  n (kl + ku + 1) entries instead of n^2 per member —
// config 3: 50 MB instead of 8.6 GB per evaluation.  Same arithmetic per entry as dsh_model_jacobian.  Other models: DSH_E_UNSUPPORTED.
int dsh_model_has_band_jacobian(int model, int64_t size) {
  int jl = -1, ju = -1, ml = -1, mu = -1;
  return !is_jit_model(model) && is_dynamic_model(model, size) && dsh_model_band(model, size, &jl, &ju, &ml, &mu) == DSH_OK && jl >= 0 && ju >= 0 ? 1 : 0;
}
int dsh_model_jacobian_band(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, int kl, int ku, double* jac) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(!is_jit_model(model) && is_dynamic_model(model, size), "dsh_model_jacobian_band: run-time-sized registry models only");
  int jl = -1, ju = -1, ml = -1, mu = -1;
  DSH_REQUIRE(dsh_model_band(model, size, &jl, &ju, &ml, &mu) == DSH_OK && jl >= 0 && ju >= 0 && kl >= jl && ku >= ju, "dsh_model_jacobian_band: the band must cover the declared one");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_jacobian_band<false>, ew_grid(n * (kl + ku + 1) * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, kl, ku, t, x, p, jac);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
// the same entries into a band container of bandwidths (kl, ku): (kl + ku + 1) n doubles per member, every one of them written
int dsh_model_jacobian_band_packed(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, int kl, int ku, double* band) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(!is_jit_model(model) && is_dynamic_model(model, size), "dsh_model_jacobian_band_packed: run-time-sized registry models only");
  int jl = -1, ju = -1, ml = -1, mu = -1;
  DSH_REQUIRE(dsh_model_band(model, size, &jl, &ju, &ml, &mu) == DSH_OK && jl >= 0 && ju >= 0 && kl >= jl && ku >= ju, "dsh_model_jacobian_band_packed: the band must cover the declared one");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_dyn_jacobian_band<true>, ew_grid(n * (kl + ku + 1) * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, kl, ku, t, x, p, band);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
// Reset operator of hybrid models (OdeEquations::reset, ode_equations/mod.rs; DiffSL reset_i): y = reset(x, t), the state after an event.  Models
// compiled from DiffSL text with a reset_i tensor have one; the built-in registry models do not.
int dsh_model_has_reset(int model, int64_t size) {
  (void)size;
  if (!is_jit_model(model)) return 0;
  const JitInfo* ji = jit_info(model);
  return ji && ji->has_reset && ji->form != DSH_JIT_FORM_STATIC_BANDED ? 1 : 0;
}
int dsh_model_reset(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, double* y) {
  DSH_ENTER(ctx);
  if (!dsh_model_has_reset(model, size)) { set_error("dsh_model_reset: model has no reset operator"); return DSH_E_UNSUPPORTED; }
  return jit_model_op(ctx, model, Op::Reset, nb, t, x, p, nullptr, 0.0, y);
}
// forward sensitivities: df/dp and dy0/dp as n x np batched matrices, one launch each
int dsh_model_has_sens(int model, int64_t size) {
  if (is_jit_model(model)) { const JitInfo* ji = jit_info(model); return ji && ji->has_sens && ji->form != DSH_JIT_FORM_STATIC_BANDED ? 1 : 0; }
  bool ok = false;
  dispatch_static_model(model, size, [&](auto mdl) { ok = model_has_sens<decltype(mdl)>::value; });
  return ok ? 1 : 0;
}
int dsh_model_rhs_sens(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, double* sens) {
  DSH_ENTER(ctx);
  if (!dsh_model_has_sens(model, size)) { set_error("dsh_model_rhs_sens: model has no parameter sensitivities"); return DSH_E_UNSUPPORTED; }
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::RhsSens, nb, t, x, p, nullptr, 0.0, sens);
  bool handled = false;
  return launch_static<Op::RhsSens>(ctx, model, size, nb, t, x, p, nullptr, 0.0, sens, &handled);
}
int dsh_model_init_sens(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* p, double* sens0) {
  DSH_ENTER(ctx);
  if (!dsh_model_has_sens(model, size)) { set_error("dsh_model_init_sens: model has no parameter sensitivities"); return DSH_E_UNSUPPORTED; }
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::InitSens, nb, t, p, p, nullptr, 0.0, sens0);  // x is not read by du0/dp
  bool handled = false;
  return launch_static<Op::InitSens>(ctx, model, size, nb, t, nullptr, p, nullptr, 0.0, sens0, &handled);
}
int dsh_model_mass_gemv(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, double beta, double* y) {
  DSH_ENTER(ctx);
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::MassGemv, nb, t, x, p, nullptr, beta, y);
  bool handled = false;
  int rc = launch_static<Op::MassGemv>(ctx, model, size, nb, t, x, p, nullptr, beta, y, &handled);
  if (rc != DSH_OK || handled) return rc;
  DSH_REQUIRE(is_dynamic_model(model, size), "unknown model id");
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  if (dyn_model_has_mass(model)) {  // heat2d / foodweb: diagonal 0 / 1 mass
    hipLaunchKernelGGL(k_dyn_mass_gemv, ew_grid(n * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, x, beta, y);
    DSH_HIP_CHECK(hipGetLastError());
    return DSH_OK;
  }
  // the other run-time-sized models have identity mass: y = x + beta*y
  return dsh_vec_axpy(ctx, n, nb, 1.0, x, nb, beta, y);
}
int dsh_model_mass_matrix(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* p, double* mass) {
  DSH_ENTER(ctx);
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::MassMatrix, nb, t, nullptr, p, nullptr, 0.0, mass);
  bool handled = false;
  int rc = launch_static<Op::MassMatrix>(ctx, model, size, nb, t, nullptr, p, nullptr, 0.0, mass, &handled);
  if (rc != DSH_OK || handled) return rc;
  if (is_dynamic_model(model, size) && dyn_model_has_mass(model)) {
    int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(k_dyn_mass_matrix, ew_grid(n * n * nb), dim3(kBlock), 0, ctx->stream, model, n, nb, mass);
    DSH_HIP_CHECK(hipGetLastError());
    return DSH_OK;
  }
  set_error("dsh_model_mass_matrix: model has no mass matrix");
  return DSH_E_UNSUPPORTED;
}
int dsh_model_init(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* p, double* y) {
  DSH_ENTER(ctx);
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::Init, nb, t, nullptr, p, nullptr, 0.0, y);
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
  DSH_ENTER(ctx);
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::Root, nb, t, x, p, nullptr, 0.0, g);
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

// Structural bandwidth of the Jacobian f_y and of the mass matrix of a model: entries (i, j) with i - j > kl or j - i > ku are never non-zero.
// -1 = not declared (treat as dense).  Built-in run-time-sized models: heat1d and the single-particle model are tridiagonal, dydt_y2 / gaussian_decay
// diagonal, robertson_ode block diagonal 3 x 3; their mass matrix is the identity.
int dsh_model_band(int model, int64_t size, int* jac_kl, int* jac_ku, int* mass_kl, int* mass_ku) {
  int jl = -1, ju = -1, ml = -1, mu = -1;
  if (is_jit_model(model)) {
    const JitInfo* ji = jit_info(model);
    if (!ji) return DSH_E_INVALID;
    jl = ji->jac_kl; ju = ji->jac_ku; ml = ji->mass_kl; mu = ji->mass_ku;
  } else if (is_dynamic_model(model, size)) {
    ml = mu = 0;
    switch (model) {
      case DSH_MODEL_HEAT1D: case DSH_MODEL_SPM: jl = ju = 1; break;
      case DSH_MODEL_DYDT_Y2: case DSH_MODEL_GAUSSIAN_DECAY: jl = ju = 0; break;
      case DSH_MODEL_ROBERTSON_ODE: jl = ju = 2; break;
      case DSH_MODEL_HEAT2D: jl = ju = (int)(size <= 0 ? 10 : size); break;       // the 5-point stencil's +-m neighbours
      case DSH_MODEL_FOODWEB: jl = ju = (int)(2 * (size <= 0 ? 10 : size)); break;  // two species per grid point
      default: break;
    }
  }
  if (jac_kl) *jac_kl = jl;
  if (jac_ku) *jac_ku = ju;
  if (mass_kl) *mass_kl = ml;
  if (mass_ku) *mass_ku = mu;
  return DSH_OK;
}

// out_i of a DiffSL model (calc_out): out is nout x nb, batch-fastest.  The registry models have no out_i (their output is the state).
int dsh_model_out(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, const double* x, const double* p, double* out) {
  DSH_ENTER(ctx);
  (void)size;
  if (is_jit_model(model)) return jit_model_op(ctx, model, Op::Out, nb, t, x, p, nullptr, 0.0, out);
  set_error("dsh_model_out: the built-in models have no out_i");
  return DSH_E_UNSUPPORTED;
}

}  // extern "C"
