// Device-resident (E)SDIRK integration — TR-BDF2 and ESDIRK34 — for ensembles of small systems (gfx950): SURVEY §8(f) row 1 for BASELINE config 5
// ("electrical-circuits DAE with event/root-finding, ESDIRK34, divergent-step repacking").
//
// One launch integrates the whole ensemble, one lane per member, everything of Sdirk::step (crates/diffsol/src/ode_solver/sdirk.rs:409-543) and the
// Rk core (runge_kutta.rs:466-960) per lane: stage predictor and Newton solves with the shared M - γhJ (linearised at φ + γ·state.y like the
// reference), embedded error estimate filtered through one LU solve, PI controller, JacobianUpdate policy, consistent DAE initialisation
// (state.rs:84-162), root finding on the dense output (root.rs), tstop handling and solve_dense (method.rs:467-520).  With per-member control
// (group = 1) every member stops at ITS OWN event time — what the lock-step backend cannot do (vector/cuda.rs:1166-1171 panics when the batch
// disagrees on the crossing); with wavefront lock-step (group = 64) the reference's batched semantics hold per 64-member group.
//
// State per lane (n = 4, s = 4): stage derivatives diff (n x s), state/old state (y, dy), φ, LU factors in registers; cached Jacobian in LDS.
// Arithmetic is the oracle's operation for operation (oracle/oracle_sdirk.hpp); pow()/sin() are ocml's.
#include <cmath>
#include <cstring>
#include <vector>

#include "dsh_internal.hpp"
#include "dsh_resident.hpp"

#include "dsh_sdirk_kernel.hpp"
#include "dsh_jit.hpp"

using namespace dsh;

namespace dsh {
bool sdirk_fast_launch(int method, int model, int64_t size, bool ba, bool wave, dim3 grid, hipStream_t stream, int64_t nb, const double* p, const double* atol,
                       const SdirkConsts* consts, const double* t_eval, double* y_out, int32_t* stats, int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols,
                       unsigned long long* totals);  // dsh_sdirk_fast.hip
}

extern "C" {

int dsh_model_has_resident(int method, int model, int64_t size) {
  if (method == 0) return dsh_model_has_adaptive(model, size);
  if (method != 1 && method != 2) return 0;
  if (is_jit_model(model)) {
    const JitInfo* ji = jit_info(model);
    if (!ji) return 0;
    // banded lane-per-member form with a mass matrix: BDF only so far (k_bdf_lane_banded); TR-BDF2 / ESDIRK34 of such a model run wavefront per member
    return (ji->form == DSH_JIT_FORM_STATIC && ji->n <= 4) || (ji->form == DSH_JIT_FORM_STATIC_BANDED && ji->n <= 512 && !ji->has_mass) ? 1 : 0;
  }
  bool ok = false;
  dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    ok = Mdl::N <= 4;
  });
  return ok ? 1 : 0;
}

}  // extern "C"
namespace {
struct SdirkSensSpec { double* out; double rtol; const double* atol_host; int64_t natol; };
struct SdirkStepsSpec { double* t_out; int64_t cap; };  // OdeSolverMethod::solve: every accepted step out (SdirkConsts::steps_cap)
template <class Mdl> constexpr bool sdirk_sens_ok() {
  if constexpr (model_has_sens<Mdl>::value) return Mdl::N <= 4 && !Mdl::HAS_MASS && Mdl::NROOTS == 0 && model_band_k<Mdl>::value == 0;
  else return false;
}
int sdirk_solve_resident_impl(dsh_ctx* ctx, int method, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                              double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                              int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host, const SdirkSensSpec* sens,
                              const SdirkStepsSpec* steps = nullptr);
}  // namespace
extern "C" {
int dsh_sdirk_solve_resident(dsh_ctx* ctx, int method, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                             double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                             int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host) {
  DSH_ENTER(ctx);
  return sdirk_solve_resident_impl(ctx, method, model, size, nb, p, atol, atol_nb, rtol, t0, h0, opts, t_eval_host, n_eval, y_out, stats, status, t_root, root_idx, ncols,
                                   totals_host, nullptr);
}
// OdeSolverMethod::solve (method.rs:227-258 over :881-961) inside the launch of the device-resident TR-BDF2 / ESDIRK34: the state after every accepted step of every member
// (arguments as dsh_bdf_solve_adaptive_steps; the models of dsh_model_has_resident(method, ..))
int dsh_sdirk_solve_resident_steps(dsh_ctx* ctx, int method, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                   double t0, double h0, const dsh_adaptive_options* opts, double t_final, int64_t max_cols, double* y_out, double* t_out, int32_t* stats,
                                   int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(max_cols >= 2 && max_cols <= 0x7fffffff && y_out != nullptr && t_out != nullptr && ncols != nullptr, "dsh_sdirk_solve_resident_steps: max_cols >= 2, y_out, t_out and ncols are needed");
  const SdirkStepsSpec st{t_out, max_cols};
  return sdirk_solve_resident_impl(ctx, method, model, size, nb, p, atol, atol_nb, rtol, t0, h0, opts, &t_final, 1, y_out, stats, status, t_root, root_idx, ncols, totals_host, nullptr, &st);
}
// TR-BDF2 / ESDIRK34 with forward sensitivities in the same launch (dsh_bdf_solve_adaptive_sens is the BDF): the same models (dsh_model_has_adaptive_sens)
int dsh_sdirk_solve_resident_sens(dsh_ctx* ctx, int method, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                  double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double sens_rtol,
                                  const double* sens_atol_host, int64_t nsens_atol, double* y_out, double* sens_out, int32_t* stats, int32_t* status, int64_t* totals_host) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(sens_out != nullptr, "sens_out is null");
  DSH_REQUIRE(nsens_atol == 0 || sens_atol_host != nullptr, "sens_atol is null");
  if (!dsh_model_has_adaptive_sens(model, size)) {
    set_error("dsh_sdirk_solve_resident_sens: the model has no device-resident integrator with forward sensitivities (identity-mass ODE model with parameter derivatives and no root functions: register-resident n <= 4, or the banded lane-per-member form)");
    return DSH_E_UNSUPPORTED;
  }
  const SdirkSensSpec sp{sens_out, sens_rtol, sens_atol_host, nsens_atol};
  return sdirk_solve_resident_impl(ctx, method, model, size, nb, p, atol, atol_nb, rtol, t0, h0, opts, t_eval_host, n_eval, y_out, stats, status, nullptr, nullptr, nullptr,
                                   totals_host, &sp);
}
}  // extern "C"
namespace {
int sdirk_solve_resident_impl(dsh_ctx* ctx, int method, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                              double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                              int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host, const SdirkSensSpec* sens,
                              const SdirkStepsSpec* steps) {
  DSH_REQUIRE(ctx != nullptr, "ctx is null");
  DSH_REQUIRE(method == 1 || method == 2, "method must be 1 (TR-BDF2) or 2 (ESDIRK34)");
  DSH_REQUIRE(n_eval >= 1 && t_eval_host != nullptr, "t_eval must hold at least one time");
  DSH_REQUIRE(atol_nb == 1 || atol_nb == nb, "atol must be broadcast (nbatch 1) or per member");
  for (int64_t q = 0; q + 1 < n_eval; ++q) DSH_REQUIRE(t_eval_host[q] <= t_eval_host[q + 1], "t_eval must be increasing (InvalidTEval)");
  DSH_REQUIRE(t_eval_host[0] >= t0, "t_eval[0] before t0 (InvalidTEval)");
  if (!dsh_model_has_resident(method, model, size)) { set_error("dsh_sdirk_solve_resident: model has no device-resident kernel (needs a static model, n <= 4)"); return DSH_E_UNSUPPORTED; }
  if (nb == 0) return DSH_OK;
  SdirkConsts T;
  std::memset((void*)&T, 0, sizeof T);
  T.r.rtol = rtol; T.r.t0 = t0; T.r.h0 = h0; T.r.n_eval = (int)n_eval; T.r.member_lanes = 0;
  if (opts) T.r.o = *opts; else dsh_adaptive_default_options(&T.r.o);
  if (T.r.o.max_steps <= 0) T.r.o.max_steps = 10000000;
  DSH_REQUIRE(T.r.o.group == 1 || T.r.o.group == 64, "group must be 1 (per member) or 64 (wavefront lock-step)");
  {
    // per-member control: DSH_MEMBER_LANES = 32 | 16 | 8 puts that many members on a wavefront (measured in profiles/r05_member_lanes.md); 0 / unset: 64
    static const int member_lanes_env = [] { const char* e = std::getenv("DSH_MEMBER_LANES"); const int v = e && *e ? std::atoi(e) : -1; return (v == 8 || v == 16 || v == 32 || v == 64) ? v : -1; }();
    // default: 32 members per wavefront while that still leaves at most two wavefronts per SIMD (the kernel's register occupancy) on the chip's 1024 SIMDs — the wavefront
    // pays for the union of 32 paths instead of 64 (config 5, 65 536 members: 21.1 -> 19.9 ms, same bits) —, 64 beyond (more wavefronts would queue)
    // (measured, RLC / ESDIRK34 with events, ms at 64 | 32 members per wavefront: 16 384 members 21.4 | 22.7, 32 768: 20.8 | 19.7, 49 152: 21.0 | 19.8, 65 536: 20.8 | 19.7 —
    // the kernel lasts as long as its slowest member's chain whatever the ensemble size; the halved wavefronts pay from half a wavefront per SIMD on)
    const int64_t simds = 4 * (int64_t)ctx->num_cu;
    const int auto_lanes = ((nb + 63) / 64 * 2 >= simds && (nb + 31) / 32 <= 2 * simds) ? 32 : 0;
    T.r.member_lanes = (T.r.o.group == 1 && !is_jit_model(model)) ? (member_lanes_env > 0 ? (member_lanes_env == 64 ? 0 : member_lanes_env) : auto_lanes) : 0;
  }
  T.r.eta_reset = std::pow(20.0, 1.25);
  T.r.eta_reset_ts = std::pow(100.0, 1.25);
  T.r.ls_steptol = std::pow(2.220446049250313e-16, 2.0 / 3.0);
  fill_tableau(method, T);
  if (steps) { T.steps_t_out = steps->t_out; T.steps_cap = (int)steps->cap; }
  if (sens) {
    T.sens_out = sens->out; T.sens_rtol = sens->rtol; T.sens_error_control = sens->natol > 0 ? 1 : 0;
    int64_t ns = 0, npar_ = 0, nroots_ = 0; int hm_ = 0;
    if (dsh_model_info(model, size, &ns, &npar_, &hm_, &nroots_) != DSH_OK) return DSH_E_INVALID;
    DSH_REQUIRE(sens->natol == 0 || sens->natol == 1 || sens->natol == ns, "sens_atol must have length 1 or nstates");
    bool uniform = true;
    for (int64_t i = 1; i < sens->natol; ++i) uniform = uniform && sens->atol_host[i] == sens->atol_host[0];
    DSH_REQUIRE(ns <= 4 || uniform, "device-resident sensitivities of models with more than 4 states take one sens_atol for every state");
    T.sens_pad = (ns > 4 || sens->natol <= 1) ? 1 : 0;  // 1: sens_atol[0] for every state
    for (int64_t i = 0; i < 4 && i < ns; ++i) T.sens_atol[i] = sens->natol == 0 ? 0.0 : (sens->natol == 1 ? sens->atol_host[0] : sens->atol_host[i]);
  }
  double* t_eval_dev = nullptr;
  unsigned long long* totals_dev = nullptr;
  SdirkConsts* consts_dev = nullptr;
  int rc = dsh_malloc(ctx, (int64_t)sizeof(SdirkConsts), 0, (void**)&consts_dev);
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipMemcpyAsync(consts_dev, &T, sizeof(SdirkConsts), hipMemcpyHostToDevice, ctx->stream));
  rc = dsh_malloc(ctx, (int64_t)(sizeof(double) * n_eval), 0, (void**)&t_eval_dev);
  if (rc != DSH_OK) return rc;
  rc = dsh_malloc(ctx, (int64_t)(sizeof(unsigned long long) * 8), 1, (void**)&totals_dev);
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipMemcpyAsync(t_eval_dev, t_eval_host, sizeof(double) * n_eval, hipMemcpyHostToDevice, ctx->stream));
  const bool ba = atol_nb == 1 && nb != 1;
  const bool wave = T.r.o.group == 64;
  const int per_wave = T.r.member_lanes > 0 ? T.r.member_lanes : 64;
  const dim3 grid((unsigned)((nb + per_wave - 1) / per_wave)), blk(64);
  DSH_HIP_CHECK(timing_begin(ctx));
  if (is_jit_model(model)) {
    const std::string name = std::string("dsh::k_sdirk_resident<dsh::JitModel, ") + (ba ? "true" : "false") + ", " + (wave ? "true" : "false") + ", " + (method == 1 ? "3" : "4") + (sens ? ", true>" : ">");
    rc = jit_launch(ctx, model, "dsh_sdirk_kernel.hpp", name, {name}, name, grid, blk, 0, nb, p, atol, (const SdirkConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats,
                    status, t_root, root_idx, ncols, totals_dev);
    if (rc != DSH_OK) return rc;
  } else if (!sens && T.r.o.deterministic_pow == 2 &&
             // the fast-arithmetic build (dsh_sdirk_fast.hip): static models with n <= 4; a model without it falls through to the exact kernel below
             sdirk_fast_launch(method, model, size, ba, wave, grid, ctx->stream, nb, p, atol, (const SdirkConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats, status,
                               t_root, root_idx, ncols, totals_dev)) {
  } else if (sens) {
    dispatch_static_model(model, size, [&](auto mdl) {
      using Mdl = decltype(mdl);
      if constexpr (sdirk_sens_ok<Mdl>()) {
#define DSH_SDIRK_SENS_LAUNCH(BA, WAVE, S)                                                                                                                              \
  hipLaunchKernelGGL((k_sdirk_resident<Mdl, BA, WAVE, S, true>), grid, blk, 0, ctx->stream, nb, p, atol, (const SdirkConsts*)consts_dev, (const double*)t_eval_dev, \
                     y_out, stats, status, t_root, root_idx, ncols, totals_dev)
#define DSH_SDIRK_SENS_LAUNCH_S(BA, WAVE) do { if (method == 1) DSH_SDIRK_SENS_LAUNCH(BA, WAVE, 3); else DSH_SDIRK_SENS_LAUNCH(BA, WAVE, 4); } while (0)
        if (ba) { if (wave) DSH_SDIRK_SENS_LAUNCH_S(true, true); else DSH_SDIRK_SENS_LAUNCH_S(true, false); }
        else { if (wave) DSH_SDIRK_SENS_LAUNCH_S(false, true); else DSH_SDIRK_SENS_LAUNCH_S(false, false); }
#undef DSH_SDIRK_SENS_LAUNCH_S
#undef DSH_SDIRK_SENS_LAUNCH
      }
    });
  } else
  dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    if constexpr (Mdl::N <= 4) {
#define DSH_SDIRK_LAUNCH(BA, WAVE, S)                                                                                                                         \
  hipLaunchKernelGGL((k_sdirk_resident<Mdl, BA, WAVE, S>), grid, blk, 0, ctx->stream, nb, p, atol, (const SdirkConsts*)consts_dev, (const double*)t_eval_dev, \
                     y_out, stats, status, t_root, root_idx, ncols, totals_dev)
#define DSH_SDIRK_LAUNCH_S(BA, WAVE) do { if (method == 1) DSH_SDIRK_LAUNCH(BA, WAVE, 3); else DSH_SDIRK_LAUNCH(BA, WAVE, 4); } while (0)
      if (ba) { if (wave) DSH_SDIRK_LAUNCH_S(true, true); else DSH_SDIRK_LAUNCH_S(true, false); }
      else { if (wave) DSH_SDIRK_LAUNCH_S(false, true); else DSH_SDIRK_LAUNCH_S(false, false); }
#undef DSH_SDIRK_LAUNCH_S
#undef DSH_SDIRK_LAUNCH
    }
  });
  DSH_HIP_CHECK(hipGetLastError());
  DSH_HIP_CHECK(timing_end(ctx));
  unsigned long long totals[8] = {0};
  DSH_HIP_CHECK(hipMemcpyAsync(totals, totals_dev, sizeof(unsigned long long) * 6, hipMemcpyDeviceToHost, ctx->stream));
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  DSH_HIP_CHECK(timing_collect(ctx));
  dsh_free(ctx, t_eval_dev);
  dsh_free(ctx, totals_dev);
  dsh_free(ctx, consts_dev);
  if (totals_host) for (int q = 0; q < 6; ++q) totals_host[q] = (int64_t)totals[q];
  return DSH_OK;
}
}  // namespace
