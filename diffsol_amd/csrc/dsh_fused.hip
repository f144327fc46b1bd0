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

#include "dsh_fused_kernels.hpp"
#include "dsh_jit.hpp"

using namespace dsh;


namespace {
const char* tf(bool b) { return b ? "true" : "false"; }
// name expressions of the fused kernels of a run-time-compiled model (one hiprtc module per flag combination, its four NIT variants together)
std::vector<std::string> jit_newton_group(bool sd, bool ba, bool we) {
  std::vector<std::string> g;
  for (int nit = 1; nit <= 4; ++nit)
    g.push_back(std::string("dsh::k_newton_iter<dsh::JitModel, ") + tf(sd) + ", " + tf(ba) + ", " + tf(we) + ", " + std::to_string(nit) + ">");
  return g;
}
std::vector<std::string> jit_accept_group(bool ba) {
  std::vector<std::string> g;
  for (int nit = 1; nit <= 4; ++nit) g.push_back(std::string("dsh::k_accept_newton<dsh::JitModel, ") + tf(ba) + ", " + std::to_string(nit) + ">");
  return g;
}
bool jit_static(int model) {
  if (!is_jit_model(model)) return false;
  const JitInfo* ji = jit_info(model);
  return ji && ji->form == DSH_JIT_FORM_STATIC;
}
}  // namespace

extern "C" {

int dsh_model_has_fused(int model, int64_t size) {
  if (is_jit_model(model)) { const JitInfo* ji = jit_info(model); return ji && ji->form == DSH_JIT_FORM_STATIC ? 1 : 0; }
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
  if (timing_on(ctx)) {
    rc = ensure_i32_scratch(ctx, 4 * (int64_t)g.x);  // 2 x u64 per workgroup
    if (rc != DSH_OK) return rc;
    clk = reinterpret_cast<unsigned long long*>(ctx->i32_scratch);
    DSH_HIP_CHECK(hipEventRecord(ctx->ev_start, ctx->stream));
  }
  bool ok = false;
  if (jit_static(model)) {
    const bool sd = is_sdirk, we = !is_sdirk && with_err;
    const std::vector<std::string> grp = jit_newton_group(sd, ba, we);
    rc = jit_launch(ctx, model, "dsh_fused_kernels.hpp", std::string("newton") + tf(sd) + tf(ba) + tf(we), grp, grp[std::min(std::max(nit, 1), 4) - 1], g, blk, 0, nb, t, c, h,
                    y_in, y_out, aux, p, (const double*)lu->factors, (const int32_t*)lu->pivots, error_y, y_old, atol, rtol, rec, seq, clk);
    if (rc != DSH_OK) return rc;
    ok = true;
  } else
  ok = dispatch_static_model(model, size, [&](auto mdl) {
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
  if (timing_on(ctx)) {  // event timing needs completed events: timed launches are synchronous
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
  DSH_ENTER(ctx);
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
  DSH_ENTER(ctx);
  return newton_launch(ctx, false, model, size, nb, t, c, 0.0, nit, y_in, y_out, psi_neg_y0, p, lu, error_y, y_old, atol, anb, rtol, ticket);
}
int dsh_bdf_newton_iter(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double c, const double* y_in, double* y_out, const double* psi_neg_y0,
                        const double* p, const dsh_lu* lu, const double* error_y, const double* y_old, const double* atol, int64_t anb, double rtol,
                        double* out) {
  DSH_ENTER(ctx);
  int64_t ticket = 0;
  int rc = newton_launch(ctx, false, model, size, nb, t, c, 0.0, 1, y_in, y_out, psi_neg_y0, p, lu, error_y, y_old, atol, anb, rtol, &ticket);
  if (rc != DSH_OK) return rc;
  return dsh_reduction_wait(ctx, ticket, out);
}
int dsh_sdirk_newton_iter_async(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double h, double c, int nit, const double* k_in, double* k_out,
                                const double* phi, const double* p, const dsh_lu* lu, const double* error_y, const double* atol, int64_t anb, double rtol,
                                int64_t* ticket) {
  DSH_ENTER(ctx);
  return newton_launch(ctx, true, model, size, nb, t, c, h, nit, k_in, k_out, phi, p, lu, error_y, nullptr, atol, anb, rtol, ticket);
}
// Run-time-sized registry models without a mass matrix (heat1d, spm, ...: no register-resident specialisation): the same iteration as three launches
// instead of seven — residual (tmp, rhs and the axpy in one pass), the LU solve, Newton update + norm in one pass — and ONE wait for both result records
// (zero-pivot count of the solve, norm).  Entry by entry the arithmetic of SdirkCallable::call_inplace + NoLineSearch::take_optimal_step through the vector
// kernels: bit-identical iterates.
int dsh_model_has_staged_newton(int model, int64_t size) { return model_has_staged_newton(model, size) ? 1 : 0; }
static int sdirk_newton_staged(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double h, double c, const double* k_in, double* k_out, const double* phi,
                               const double* p, const dsh_lu* lu, const double* error_y, const double* atol, int64_t anb, double rtol, double* out) {
  DSH_REQUIRE(lu != nullptr && out != nullptr, "null argument");
  if (!lu->factored) { set_error("newton iteration: LU not initialised"); return DSH_E_NOT_SETUP; }
  int64_t n; dsh_model_info(model, size, &n, nullptr, nullptr, nullptr);
  int rc = ensure_f64_scratch(ctx, n * nb);
  if (rc != DSH_OK) return rc;
  double* delta = ctx->f64_scratch;
  if (!model_dyn_sdirk_residual(ctx, model, size, nb, t, c, h, phi, k_in, p, delta)) { set_error("newton iteration: model has no staged form"); return DSH_E_UNSUPPORTED; }
  unsigned int gs = 0, ss = 0, gn = 0, sn = 0;
  {  // banded factors, small ensemble: the Newton update and the norm ride in the solve's launch (k_lu_band_solve_team<.., EPI>); one record carries both results
    bool fused = false;
    rc = lu_solve_norm_launch(lu, delta, k_in, k_out, error_y, nb, atol, anb, rtol, &gs, &ss, &fused);
    if (rc != DSH_OK) return rc;
    if (fused) {
      rc = fetch_records(ctx, gs, ss);
      if (rc != DSH_OK) return rc;
      out[0] = bits_to_double(ctx->res_m0);
      out[1] = 0.0;
      out[2] = (double)ctx->res_cnt;
      return DSH_OK;
    }
  }
  rc = lu_solve_launch(lu, delta, &gs, &ss);
  if (rc != DSH_OK) return rc;
  rc = vec_sub_squared_norm_launch(ctx, n, nb, delta, k_in, k_out, error_y, nb, atol, anb, rtol, &gn, &sn);
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipGetLastError());
  rc = fetch_records(ctx, gn, sn);  // the later launch first: when its records are there the solve's are too
  if (rc != DSH_OK) return rc;
  out[0] = bits_to_double(ctx->res_m0);
  out[1] = 0.0;
  rc = fetch_records(ctx, gs, ss);
  if (rc != DSH_OK) return rc;
  out[2] = (double)ctx->res_cnt;
  return DSH_OK;
}

// x <- A^-1 x with the factors of `lu`, then max_b mean_i (x_i / (|y_i| rtol + atol_i))^2: the error estimate of the SDIRK step (sdirk.rs:474-495 filtered
// through the Newton matrix, runge_kutta.rs:783-800) as two launches and ONE wait for both results.  DSH_E_SINGULAR like dsh_lu_solve.
int dsh_lu_solve_squared_norm(const dsh_lu* lu, double* x, const double* y, int64_t ynb, const double* atol, int64_t anb, double rtol, double* out_norm) {
  DSH_ENTER(lu ? lu->ctx : nullptr);
  DSH_REQUIRE(lu != nullptr && x != nullptr && out_norm != nullptr, "null argument");
  dsh_ctx* ctx = lu->ctx;
  if (!lu->factored) { set_error("dsh_lu_solve_squared_norm: LU not initialised"); return DSH_E_NOT_SETUP; }
  const int64_t n = lu->n, nb = lu->nbatch;
  if (n == 0) { *out_norm = 0.0; return DSH_OK; }
  unsigned int gs = 0, ss = 0;
  {
    bool fused = false;
    int rcf = lu_solve_norm_launch(lu, x, nullptr, nullptr, y, ynb, atol, anb, rtol, &gs, &ss, &fused);
    if (rcf != DSH_OK) return rcf;
    if (fused) {
      rcf = fetch_records(ctx, gs, ss);
      if (rcf != DSH_OK) return rcf;
      *out_norm = bits_to_double(ctx->res_m0);
      if (ctx->res_cnt != 0ull) {
        set_error("dsh_lu_solve: zero pivot in " + std::to_string((long long)ctx->res_cnt) + " system(s) (LuSolveFailed)");
        return DSH_E_SINGULAR;
      }
      return DSH_OK;
    }
  }
  int rc = lu_solve_launch(lu, x, &gs, &ss);
  if (rc != DSH_OK) return rc;
  rc = dsh_vec_squared_norm(ctx, n, nb, x, y, ynb, atol, anb, rtol, out_norm, nullptr);  // waits for its own records: the solve's are there by then
  if (rc != DSH_OK) return rc;
  rc = fetch_records(ctx, gs, ss);
  if (rc != DSH_OK) return rc;
  if (ctx->res_cnt != 0ull) {
    set_error("dsh_lu_solve: zero pivot in " + std::to_string((long long)ctx->res_cnt) + " system(s) (LuSolveFailed)");
    return DSH_E_SINGULAR;
  }
  return DSH_OK;
}

int dsh_sdirk_newton_iter(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double h, double c, const double* k_in, double* k_out,
                          const double* phi, const double* p, const dsh_lu* lu, const double* error_y, const double* atol, int64_t anb, double rtol,
                          double* out) {
  DSH_ENTER(ctx);
  if (dsh_model_has_staged_newton(model, size)) return sdirk_newton_staged(ctx, model, size, nb, t, h, c, k_in, k_out, phi, p, lu, error_y, atol, anb, rtol, out);
  int64_t ticket = 0;
  int rc = newton_launch(ctx, true, model, size, nb, t, c, h, 1, k_in, k_out, phi, p, lu, error_y, nullptr, atol, anb, rtol, &ticket);
  if (rc != DSH_OK) return rc;
  return dsh_reduction_wait(ctx, ticket, out);
}

int dsh_jac_factor(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double c, const double* x, const double* p, int recompute, double* rhs_jac,
                   double* mass_jac, dsh_lu* lu) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(lu != nullptr && rhs_jac != nullptr, "null argument");
  { const int rc = lu_ensure_storage(lu); if (rc != DSH_OK) return rc; }
  lu->singular_epoch += 1;  // the kernel adds (epoch << 32 | 1) per singular system: no reset launch needed between factorisations
  bool ok = false;
  if (jit_static(model)) {
    const std::vector<std::string> grp = {"dsh::k_jac_factor<dsh::JitModel>"};
    int rc = jit_launch(ctx, model, "dsh_fused_kernels.hpp", "jac_factor", grp, grp[0], grid_for(nb, ctx->block), dim3(ctx->block), 0, nb, t, c, x, p, recompute, rhs_jac,
                        mass_jac, (double*)lu->factors, (int32_t*)lu->pivots, (unsigned long long*)lu->singular, (unsigned int)lu->singular_epoch);
    if (rc != DSH_OK) return rc;
    ok = true;
  } else
  ok = dispatch_static_model(model, size, [&](auto mdl) {
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
  DSH_ENTER(ctx);
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
  DSH_ENTER(ctx);
  DSH_REQUIRE(ticket != nullptr, "ticket is null");
  if (n * nb == 0) { *ticket = 0; return DSH_OK; }
  return accept_launch(ctx, n, nb, order, h, diff, y_predict, y_new, y, dy, atol, anb, rtol, gamma_host, alpha, psi_neg_y0_next, ticket);
}

int dsh_bdf_accept_newton_async(dsh_ctx* ctx, int model, int64_t size, int64_t nb, int order, double h, double* diff, double* y_predict, const double* y_new,
                                double* y, double* dy, const double* atol, int64_t anb, double rtol, const double* gamma_host, double alpha,
                                double* psi_neg_y0_next, double t_next, double c, int nit, double* y_out, const double* p, const dsh_lu* lu,
                                int64_t* accept_ticket, int64_t* newton_ticket) {
  DSH_ENTER(ctx);
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
  bool ok = false;
  if (jit_static(model)) {
    const std::vector<std::string> grp = jit_accept_group(ba);
    rc = jit_launch(ctx, model, "dsh_fused_kernels.hpp", std::string("accept") + tf(ba), grp, grp[std::min(std::max(nit, 1), 4) - 1], g, blk, 0, nb, order, inv_h, diff, y_predict,
                    y_new, y, dy, atol, rtol, cf, psi_neg_y0_next, t_next, c, y_out, p, (const double*)lu->factors, (const int32_t*)lu->pivots, rec, seq);
    if (rc != DSH_OK) return rc;
    ok = true;
  } else
  ok = dispatch_static_model(model, size, [&](auto mdl) {
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

}  // extern "C"
namespace {
// ---- SDIRK stage bookkeeping in single passes (round 5).  Between two Newton solves the host-driven Sdirk::step runs a handful of elementwise / tiny-gemv operations over
// the n x nbatch vectors (runge_kutta.rs:505-516, 610-689, 783-800; op/sdirk.rs:174-203): each was one launch and one pass over memory.  The kernels below do the
// operations that follow each other with no host decision in between in ONE pass — per element the same expressions in the same order as the separate kernels
// (k_axpby_to: alpha * x + beta * y0;  k_gemv: alpha * A(0) * X(0) + beta * yin, then alpha * A(j) * X(j) + acc;  FAxpy0: alpha * x): the same bits.
struct sdirk_coef { double v[8]; };
constexpr int kStageBlock = 256;
inline dim3 stage_grid(int64_t total) { int64_t b = (total + kStageBlock - 1) / kStageBlock; if (b > 8192) b = 8192; if (b < 1) b = 1; return dim3((unsigned)b); }

// start of a step attempt with an explicit first stage: diff[:,0] = h dy (start_step_attempt), phi = y0 + diff[:,0] a10 (set_phi of stage 1), k = diff[:,0] (its predictor)
__global__ void k_sdirk_begin_attempt(int64_t total, double h, double a10, const double* __restrict__ dy, const double* __restrict__ y0, double* __restrict__ diff0,
                                      double* __restrict__ phi, double* __restrict__ k) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const double alpha = 1.0, beta = 1.0;
    const double d0 = h * dy[idx];
    diff0[idx] = d0;
    phi[idx] = alpha * d0 * a10 + beta * y0[idx];
    k[idx] = d0;
  }
}
// stage i has converged (k = its increment), stage i + 1 follows: y_stage = c k + phi (get_f_eval), diff[:,i] = k, phi = y0 + diff[:,0..=i] a (set_phi of stage i + 1),
// k = pa diff[:,i-1] + pb diff[:,i] (its predictor, runge_kutta.rs:610-629; stage i + 1 >= 2)
__global__ void k_sdirk_next_stage(int64_t total, int i, double c, double* __restrict__ k, double* __restrict__ phi, const double* __restrict__ y0, double* __restrict__ y_stage,
                                   double* __restrict__ diff, sdirk_coef a, double pa, double pb) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const double alpha = 1.0, beta = 1.0;
    const double xv = k[idx];
    y_stage[idx] = c * xv + beta * phi[idx];
    diff[(int64_t)i * total + idx] = xv;
    double prev = 0.0;  // diff[:, i-1]
    double acc = 0.0;
    for (int j = 0; j <= i; ++j) {
      const double aj = j < i ? diff[(int64_t)j * total + idx] : xv;
      if (j == i - 1) prev = aj;
      acc = j == 0 ? alpha * aj * a.v[0] + beta * y0[idx] : alpha * aj * a.v[j] + acc;
    }
    phi[idx] = acc;
    k[idx] = pa * prev + pb * xv;
  }
}
// the last stage has converged: y_stage = c k + phi, diff[:,s-1] = k, err = diff d (the error estimate before its filter solve, runge_kutta.rs:783-800)
__global__ void k_sdirk_finish_error(int64_t total, int s, double c, const double* __restrict__ k, const double* __restrict__ phi, double* __restrict__ y_stage,
                                     double* __restrict__ diff, sdirk_coef d, double* __restrict__ err) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const double alpha = 1.0, beta = 1.0;
    const double xv = k[idx];
    y_stage[idx] = c * xv + beta * phi[idx];
    diff[(int64_t)(s - 1) * total + idx] = xv;
    double acc = 0.0;
    for (int j = 0; j < s; ++j) {
      const double aj = j < s - 1 ? diff[(int64_t)j * total + idx] : xv;
      acc = j == 0 ? alpha * aj * d.v[0] : alpha * aj * d.v[j] + acc;
    }
    err[idx] = acc;
  }
}
}  // namespace
extern "C" {

int dsh_sdirk_begin_attempt(dsh_ctx* ctx, int64_t n, int64_t nb, double h, double a10, const double* dy, const double* y0, double* diff0, double* phi, double* k) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(dy && y0 && diff0 && phi && k, "null argument");
  const int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  hipLaunchKernelGGL(k_sdirk_begin_attempt, stage_grid(total), dim3(kStageBlock), 0, ctx->stream, total, h, a10, dy, y0, diff0, phi, k);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_sdirk_next_stage(dsh_ctx* ctx, int64_t n, int64_t nb, int stage, double c, double* k, double* phi, const double* y0, double* y_stage, double* diff,
                         const double* a_next_host, double pred_a, double pred_b) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(k && phi && y0 && y_stage && diff && a_next_host, "null argument");
  DSH_REQUIRE(stage >= 1 && stage < 7, "dsh_sdirk_next_stage: the finished stage must be 1 ... 6");
  const int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  sdirk_coef a;
  for (int j = 0; j < 8; ++j) a.v[j] = j <= stage ? a_next_host[j] : 0.0;
  hipLaunchKernelGGL(k_sdirk_next_stage, stage_grid(total), dim3(kStageBlock), 0, ctx->stream, total, stage, c, k, phi, y0, y_stage, diff, a, pred_a, pred_b);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_sdirk_finish_error(dsh_ctx* ctx, int64_t n, int64_t nb, int nstages, double c, const double* k, const double* phi, double* y_stage, double* diff,
                           const double* d_host, double* err) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(k && phi && y_stage && diff && d_host && err, "null argument");
  DSH_REQUIRE(nstages >= 1 && nstages <= 8, "dsh_sdirk_finish_error: 1 ... 8 stages");
  const int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  sdirk_coef d;
  for (int j = 0; j < 8; ++j) d.v[j] = j < nstages ? d_host[j] : 0.0;
  hipLaunchKernelGGL(k_sdirk_finish_error, stage_grid(total), dim3(kStageBlock), 0, ctx->stream, total, nstages, c, k, phi, y_stage, diff, d, err);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

int dsh_bdf_accept_step(dsh_ctx* ctx, int64_t n, int64_t nb, int order, double h, double* diff, double* y_predict, const double* y_new, double* y,
                        double* dy, const double* atol, int64_t anb, double rtol, const double* gamma_host, double alpha, double* psi_neg_y0_next,
                        int want_norms, double* out) {
  DSH_ENTER(ctx);
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
