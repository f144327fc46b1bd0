// Device-resident, per-member adaptive BDF for ensembles of small systems (gfx950) — SURVEY §8(f) row 1.
//
// The trait-boundary path (host/bdf.hpp over dsh_fused.hip) advances the whole ensemble in lock-step: one (t, h, order) sequence, the
// max over all members in every convergence / error test, ~3 launches and ~1.3 host round trips per accepted step.  That is what the
// reference's Context/Vector/LinearSolver boundary implies for a batched backend, but it is NOT how diffsol's CPU path treats a parameter
// sweep: there every member is an independent IVP with its own step-size and order history.  This kernel restores exactly that semantics on
// the device: ONE launch integrates the whole ensemble, one lane per member, the complete solver state in registers —
//   difference array D (n x 8) and its swap partner, cached Jacobian, LU factors + pivots, Newton iterate, Convergence state (eta carry-over),
//   JacobianUpdate counters, PI-controller memory, statistics —
// and every scalar decision of Bdf::step (crates/diffsol/src/ode_solver/bdf.rs:1277-1589), NewtonNonlinearSolver / NoLineSearch / Convergence
// (crates/diffsol-nl/src/{newton,line_search,convergence}.rs), JacobianUpdate (jacobian_update.rs:12-79), set_step_size
// (state.rs:1209-1277), handle_tstop (bdf.rs:694-731) and solve_dense (method.rs:467-520) taken per lane.  No host round trip, no
// reduction, no HBM traffic besides parameters in and the requested save points out.  Lanes of a wavefront diverge where their members do
// (different Newton iteration counts, rejected steps, refactorisations); a wavefront is done when its slowest member is.
//
// Arithmetic is the oracle's, operation for operation (-ffp-contract=off); the one difference to a CPU run of the reference is libm: pow()
// in the step-size controller, the convergence-rate estimate and the initial step is ocml's here — results agree with independent CPU solves
// to rounding of h, not bitwise (tests/test_gpu_adaptive.py states the tolerance).
//
// Two control granularities (dsh_adaptive_options.group):
//   1  : every member its own (t, h, order) history — diffsol's CPU semantics for a sweep of independent IVPs; wavefronts diverge.
//   64 : the 64 members of a wavefront advance in lock-step, norms reduced with a max over the wavefront — the reference's batched semantics
//        (nbatch = 64) per group; no divergence at all, every control scalar is wavefront-uniform, no host round trip.
//
// Scope: static register models (n <= 4), with or without mass matrix (consistent initialisation on the device) and root functions
// (per-member event stops); shared pieces in dsh_resident.hpp, the (E)SDIRK counterpart in dsh_sdirk_resident.hip.
#include <cmath>
#include <cstdio>
#include <vector>

#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "dsh_internal.hpp"
#include "dsh_resident.hpp"

// DSH_EXPERIMENTS (compile time, `make EXPERIMENTS=1`; off in the shipped library): the variants that were built, verified bit-identical and MEASURED SLOWER than
// what ships — re-binning between segments of a per-member run (DSH_REBIN / DSH_REBIN_STEPS: 10.3 - 19 ms against 8.4 ms, profiles/r03_rebin.txt) and the
// phase-scheduled per-member kernel (DSH_MEMBER_SCHED: 10.75 against 9.23 ms).  Kept in the tree for the record and for whoever wants to take them further;
// dsh_experiments_enabled() tells a caller (the tests) whether this build has them.
#ifdef DSH_EXPERIMENTS
#include <hipcub/hipcub.hpp>
#endif
#include "dsh_adaptive_kernel.hpp"
#ifdef DSH_EXPERIMENTS
#include "dsh_member_sched_kernel.hpp"
#endif
#include "dsh_jit.hpp"

using namespace dsh;


extern "C" {

int dsh_experiments_enabled(void) {
#ifdef DSH_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

void dsh_adaptive_default_options(dsh_adaptive_options* o) {
  if (!o) return;
  // OdeSolverOptions defaults (problem.rs:132-152) + BdfConfig (config.rs:53-74)
  o->max_nonlinear_solver_iterations = 10;
  o->max_error_test_failures = 40;
  o->max_nonlinear_solver_failures = 50;
  o->nonlinear_solver_tolerance = 0.2;
  o->min_timestep = 1e-13;
  o->max_timestep_growth = 2.0;
  o->min_timestep_growth = 2.0;
  o->max_timestep_shrink = 0.9;
  o->min_timestep_shrink = 0.5;
  o->update_jacobian_after_steps = 20;
  o->update_rhs_jacobian_after_steps = 50;
  o->threshold_to_update_jacobian = 0.3;
  o->threshold_to_update_rhs_jacobian = 0.2;
  o->pi_control_proportional = 0.0;
  o->pi_control_integral = 0.5;
  o->ic_use_linesearch = 1;
  o->ic_max_linesearch_iterations = 10;
  o->ic_max_linear_solver_setups = 4;
  o->ic_max_newton_iterations = 10;
  o->ic_step_reduction_factor = 0.5;
  o->ic_armijo_constant = 1e-4;
  o->max_steps = 10000000;
  o->deterministic_pow = 1;
  o->group = 1;
}

int dsh_model_has_adaptive(int model, int64_t size) {
  if (is_jit_model(model)) {
    const JitInfo* ji = jit_info(model);
    if (!ji) return 0;
    return (ji->form == DSH_JIT_FORM_STATIC && ji->n <= 4) || (ji->form == DSH_JIT_FORM_STATIC_BANDED && ji->n <= 512) ? 1 : 0;
  }
  bool ok = false;
  dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    ok = Mdl::N <= 4;
  });
  return ok ? 1 : 0;
}

}  // extern "C"
namespace dsh {
bool adaptive_fast_launch(int model, int64_t size, bool ba, bool wave, dim3 grid, hipStream_t stream, int64_t nb, const double* p, const double* atol,
                          const AdaptiveConsts* consts, const double* t_eval, double* y_out, int32_t* stats, int32_t* status, double* t_root, int32_t* root_idx,
                          int32_t* ncols, unsigned long long* totals);  // dsh_adaptive_fast.hip
}
namespace {
struct SensSpec { double* out; double rtol; const double* atol_host; int64_t natol; };  // forward sensitivities of dsh_bdf_solve_adaptive_sens
// does the static model have a device-resident BDF with forward sensitivities?  (sens_mul / init_sens_mul, identity mass, no root functions, n <= 4)
template <class Mdl> constexpr bool adaptive_sens_ok() {
  if constexpr (model_has_sens<Mdl>::value) return Mdl::N <= 4 && !Mdl::HAS_MASS && Mdl::NROOTS == 0 && model_band_k<Mdl>::value == 0;
  else return false;
}
struct StepsSpec { double* t_out; int64_t cap; };  // OdeSolverMethod::solve: every accepted step out (AdaptiveConsts::steps_cap)
int bdf_solve_adaptive_impl(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                            double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats, int32_t* status,
                            double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host, const SensSpec* sens, const StepsSpec* steps = nullptr);
}  // namespace
extern "C" {
// hybrid models whose events are handled INSIDE dsh_bdf_solve_adaptive (move back to the root, apply the reset, restart at first order, go on to the last save
// point): register-resident form (n <= 4) or banded lane-per-member form, identity mass, with root functions
int dsh_model_has_adaptive_reset(int model, int64_t size) {
  if (is_jit_model(model)) {
    const JitInfo* ji = jit_info(model);
    if (!(ji && ji->has_reset && ji->nroots > 0)) return 0;
    // register-resident form (n <= 4; round 5: with a mass matrix too — hybrid DAEs, made consistent again after every reset), or the banded lane-per-member form
    // (k_bdf_lane_banded: the same event handling on per-lane memory; identity mass; BDF)
    return (ji->form == DSH_JIT_FORM_STATIC && ji->n <= 4) || (ji->form == DSH_JIT_FORM_STATIC_BANDED && !ji->has_mass) ? 1 : 0;
  }
  bool ok = false;
  dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    ok = model_has_reset<Mdl>::value && Mdl::N <= 4 && Mdl::NROOTS > 0 && model_band_k<Mdl>::value == 0;
  });
  return ok ? 1 : 0;
}
int dsh_model_has_adaptive_sens(int model, int64_t size) {
  if (is_jit_model(model)) {  // DiffSL / external models in the register-resident form with parameter derivatives (DSH_JIT_HAS_SENS)
    const JitInfo* ji = jit_info(model);
    if (!(ji && ji->has_sens && !ji->has_mass && ji->nroots == 0)) return 0;
    // register-resident form (n <= 4) or the banded lane-per-member form (the state and the sensitivity arrays in per-lane memory; k_bdf_adaptive's banded branch)
    return (ji->form == DSH_JIT_FORM_STATIC && ji->n <= 4) || ji->form == DSH_JIT_FORM_STATIC_BANDED ? 1 : 0;
  }
  bool ok = false;
  dispatch_static_model(model, size, [&](auto mdl) { ok = adaptive_sens_ok<decltype(mdl)>(); });
  return ok ? 1 : 0;
}
int dsh_bdf_solve_adaptive(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                           double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats, int32_t* status,
                           double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host) {
  DSH_ENTER(ctx);
  return bdf_solve_adaptive_impl(ctx, model, size, nb, p, atol, atol_nb, rtol, t0, h0, opts, t_eval_host, n_eval, y_out, stats, status, t_root, root_idx, ncols, totals_host, nullptr);
}
int dsh_bdf_solve_adaptive_sens(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double sens_rtol, const double* sens_atol_host,
                                int64_t nsens_atol, double* y_out, double* sens_out, int32_t* stats, int32_t* status, int64_t* totals_host) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(sens_out != nullptr, "sens_out is null");
  DSH_REQUIRE(nsens_atol == 0 || sens_atol_host != nullptr, "sens_atol is null");
  if (!dsh_model_has_adaptive_sens(model, size)) {
    set_error("dsh_bdf_solve_adaptive_sens: the model has no device-resident BDF with forward sensitivities (identity-mass ODE model with parameter derivatives and no root functions: register-resident n <= 4, or the banded lane-per-member form)");
    return DSH_E_UNSUPPORTED;
  }
  const SensSpec sp{sens_out, sens_rtol, sens_atol_host, nsens_atol};
  return bdf_solve_adaptive_impl(ctx, model, size, nb, p, atol, atol_nb, rtol, t0, h0, opts, t_eval_host, n_eval, y_out, stats, status, nullptr, nullptr, nullptr, totals_host, &sp);
}
// OdeSolverMethod::solve (method.rs:227-258 over :881-961) inside the launch: the register-resident kernels only (static models, built-in or run-time-compiled)
int dsh_model_has_adaptive_steps(int model, int64_t size) {
  if (!dsh_model_has_adaptive(model, size)) return 0;
  // static run-time-compiled models (k_bdf_adaptive) and — round 5 — the banded lane-per-member form (k_bdf_lane_banded; forward sensitivities excluded by the caller)
  if (is_jit_model(model)) { const JitInfo* ji = jit_info(model); return ji && (ji->form == DSH_JIT_FORM_STATIC || ji->form == DSH_JIT_FORM_STATIC_BANDED) ? 1 : 0; }
  return 1;
}
int dsh_bdf_solve_adaptive_steps(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                 double h0, const dsh_adaptive_options* opts, double t_final, int64_t max_cols, double* y_out, double* t_out, int32_t* stats,
                                 int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(max_cols >= 2 && max_cols <= 0x7fffffff && y_out != nullptr && t_out != nullptr && ncols != nullptr, "dsh_bdf_solve_adaptive_steps: max_cols >= 2, y_out, t_out and ncols are needed");
  if (!dsh_model_has_adaptive_steps(model, size)) {
    set_error("dsh_bdf_solve_adaptive_steps: the model has neither a register-resident BDF (static model, n <= 4) nor a banded lane-per-member form (the wavefront / workgroup forms have dsh_bdf_solve_wave_member_steps)");
    return DSH_E_UNSUPPORTED;
  }
  const StepsSpec st{t_out, max_cols};
  return bdf_solve_adaptive_impl(ctx, model, size, nb, p, atol, atol_nb, rtol, t0, h0, opts, &t_final, 1, y_out, stats, status, t_root, root_idx, ncols, totals_host, nullptr, &st);
}
}  // extern "C"
namespace {
int bdf_solve_adaptive_impl(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                            double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats, int32_t* status,
                            double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host, const SensSpec* sens, const StepsSpec* steps) {
  DSH_REQUIRE(ctx != nullptr, "ctx is null");
  DSH_REQUIRE(n_eval >= 1 && t_eval_host != nullptr, "t_eval must hold at least one time");
  DSH_REQUIRE(atol_nb == 1 || atol_nb == nb, "atol must be broadcast (nbatch 1) or per member");
  for (int64_t k = 0; k + 1 < n_eval; ++k) DSH_REQUIRE(t_eval_host[k] <= t_eval_host[k + 1], "t_eval must be increasing (InvalidTEval)");
  DSH_REQUIRE(t_eval_host[0] >= t0, "t_eval[0] before t0 (InvalidTEval)");
  if (!dsh_model_has_adaptive(model, size)) { set_error("dsh_bdf_solve_adaptive: model has no device-resident kernel (needs a static model, n <= 4)"); return DSH_E_UNSUPPORTED; }
  if (nb == 0) return DSH_OK;
  AdaptiveConsts C;
  std::memset((void*)&C, 0, sizeof C);  // the block is compared byte-wise with the cached copy below
  C.r.rtol = rtol; C.r.t0 = t0; C.r.h0 = h0; C.r.n_eval = (int)n_eval; C.r.member_lanes = 0;
  C.r.ls_steptol = std::pow(2.220446049250313e-16, 2.0 / 3.0);
  if (opts) C.r.o = *opts; else dsh_adaptive_default_options(&C.r.o);
  if (C.r.o.max_steps <= 0) C.r.o.max_steps = 10000000;
  DSH_REQUIRE(C.r.o.group == 1 || C.r.o.group == 64, "adaptive group must be 1 (per member) or 64 (wavefront lock-step)");
  {
    // per-member control of the register-resident kernel: DSH_MEMBER_LANES = 32 | 16 | 8 puts that many members on a wavefront (profiles/r05_member_lanes.md); 0 / unset: 64
    static const int member_lanes_env = [] { const char* e = std::getenv("DSH_MEMBER_LANES"); const int v = e && *e ? std::atoi(e) : 0; return (v == 8 || v == 16 || v == 32) ? v : 0; }();
    C.r.member_lanes = (C.r.o.group == 1 && !is_jit_model(model) && !sens && !steps) ? member_lanes_env : 0;
  }
  if (steps) { C.steps_t_out = steps->t_out; C.steps_cap = (int)steps->cap; }
  if (sens) {
    C.sens_out = sens->out; C.sens_rtol = sens->rtol; C.sens_error_control = sens->natol > 0 ? 1 : 0;
    int64_t ns = 0, npar_ = 0, nroots_ = 0; int hm_ = 0;
    if (dsh_model_info(model, size, &ns, &npar_, &hm_, &nroots_) != DSH_OK) return DSH_E_INVALID;
    DSH_REQUIRE(sens->natol == 0 || sens->natol == 1 || sens->natol == ns, "sens_atol must have length 1 or nstates");
    bool uniform = true;
    for (int64_t i = 1; i < sens->natol; ++i) uniform = uniform && sens->atol_host[i] == sens->atol_host[0];
    DSH_REQUIRE(ns <= 4 || uniform, "device-resident sensitivities of models with more than 4 states take one sens_atol for every state");
    C.sens_pad = (ns > 4 || sens->natol <= 1) ? 1 : 0;  // 1: sens_atol[0] for every state
    for (int64_t i = 0; i < 4 && i < ns; ++i) C.sens_atol[i] = sens->natol == 0 ? 0.0 : (sens->natol == 1 ? sens->atol_host[0] : sens->atol_host[i]);
  }
  {  // Bdf::_new tables (bdf.rs:286-306)
    const double kappa[6] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0};
    C.alpha[0] = 0.0; C.gamma[0] = 0.0; C.ec2[0] = 1.0;
    for (int i = 1; i <= kMaxOrder; ++i) {
      const double i_t = (double)i, one_over_i = 1.0 / i_t, one_over_i_plus_one = 1.0 / (i_t + 1.0);
      C.gamma[i] = C.gamma[i - 1] + one_over_i;
      C.alpha[i] = 1.0 / ((1.0 - kappa[i]) * C.gamma[i]);
      const double e = kappa[i] * C.gamma[i] + one_over_i_plus_one;
      C.ec2[i] = e * e;
    }
    C.r.eta_reset = std::pow(20.0, 1.25);
    C.r.eta_reset_ts = std::pow(100.0, 1.25);
    C.eta_reset_p08 = dsh_det_pow(C.r.eta_reset, 0.8);  // used by the kernel only with the deterministic pow (the same function there)
    C.eta_reset_ts_p08 = dsh_det_pow(C.r.eta_reset_ts, 0.8);
    for (int ord = 1; ord <= kMaxOrder; ++ord) {  // compute_r(order, 1.0) (bdf.rs:433-463), stored 6x6 column-major
      double* U = C.u[ord - 1];
      for (int k = 0; k < 36; ++k) U[k] = 0.0;
      for (int j = 0; j <= ord; ++j) U[j * 6 + 0] = 1.0;
      for (int j = 1; j <= ord; ++j)
        for (int i = 1; i <= ord; ++i) U[j * 6 + i] = U[j * 6 + i - 1] * ((double)i - 1.0 - 1.0 * (double)j) / (double)i;
    }
  }
  double* t_eval_dev = nullptr;
  unsigned long long* totals_dev = nullptr;
  AdaptiveConsts* consts_dev = nullptr;
  int rc = DSH_OK;
  // Repeated solves of one problem (a parameter study, the benchmark loop) pass the same constants and save points every time: they stay on the device
  // and are compared on the host instead of being uploaded again (two pageable host-to-device copies, ~15 us each on a 2.5 ms solve).  One small block per
  // context, released with it.
  const size_t cbytes = sizeof(AdaptiveConsts), tbytes = sizeof(double) * (size_t)n_eval, need = cbytes + tbytes;
  bool cached = false;
  if (tbytes <= 4096) {  // the block belongs to the context (one thread per context) and goes with it (dsh_ctx_destroy)
    if (!ctx->const_cache_dev && hipMalloc((void**)&ctx->const_cache_dev, cbytes + 4096) != hipSuccess) { (void)hipGetLastError(); ctx->const_cache_dev = nullptr; }
    if (ctx->const_cache_dev) {
      if (!ctx->const_cache_host) ctx->const_cache_host = new std::vector<unsigned char>();
      std::vector<unsigned char>& held = *ctx->const_cache_host;
      std::vector<unsigned char> now(need);
      std::memcpy(now.data(), &C, cbytes);
      std::memcpy(now.data() + cbytes, t_eval_host, tbytes);
      if (now != held) {
        held.swap(now);  // the staging vector must outlive the asynchronous copy: it is the cache itself
        DSH_HIP_CHECK(hipMemcpyAsync(ctx->const_cache_dev, held.data(), need, hipMemcpyHostToDevice, ctx->stream));
        DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      }
      consts_dev = (AdaptiveConsts*)ctx->const_cache_dev;
      t_eval_dev = (double*)(ctx->const_cache_dev + cbytes);
      cached = true;
    }
  }
  if (!cached) {
    rc = dsh_malloc(ctx, (int64_t)sizeof(AdaptiveConsts), 0, (void**)&consts_dev);
    if (rc != DSH_OK) return rc;
    DSH_HIP_CHECK(hipMemcpyAsync(consts_dev, &C, sizeof(AdaptiveConsts), hipMemcpyHostToDevice, ctx->stream));
    rc = dsh_malloc(ctx, (int64_t)(sizeof(double) * n_eval), 0, (void**)&t_eval_dev);
    if (rc != DSH_OK) return rc;
    DSH_HIP_CHECK(hipMemcpyAsync(t_eval_dev, t_eval_host, sizeof(double) * n_eval, hipMemcpyHostToDevice, ctx->stream));
  }
  rc = dsh_malloc(ctx, (int64_t)(sizeof(unsigned long long) * 8), 1, (void**)&totals_dev);
  if (rc != DSH_OK) { if (!cached) { dsh_free(ctx, t_eval_dev); dsh_free(ctx, consts_dev); } return rc; }
  const bool ba = atol_nb == 1 && nb != 1;
  const int per_wave = C.r.member_lanes > 0 ? C.r.member_lanes : 64;
  const dim3 grid((unsigned)((nb + per_wave - 1) / per_wave)), blk(64);
  bool launched = false;
  DSH_HIP_CHECK(timing_begin(ctx));
#ifdef DSH_EXPERIMENTS
  // ---- per-member control in SEGMENTS with re-binning between them (DSH_REBIN=k: a segment ends when a member has produced k more save points; 0 / unset:
  // one launch).  Members drift apart inside a wavefront — different orders, step sizes, distances to the next order selection — and a wavefront pays for the
  // union of its lanes' paths (profiles/r03_per_member.md: 61 % of the per-member time).  Between segments every member's integrator state is in memory and
  // the members are dealt to lanes again, sorted by (order, steps since the last change, step size).  The state is read back exactly, so results do not
  // depend on where the segments end: bit-identical to the single launch.  Register-resident static models only.
  const int rebin = [] { const char* e = std::getenv("DSH_REBIN"); return e && *e ? std::atoi(e) : 0; }();
  // DSH_REBIN_STEPS=k: a segment ends for a member after k trips of its step loop instead (all wavefronts of a segment then do the same number of trips: no
  // waiting for the slowest member of a segment); the host launches segments until no member is left, then one launch that only writes the results.
  const int rebin_steps = [] { const char* e = std::getenv("DSH_REBIN_STEPS"); return e && *e ? std::atoi(e) : 0; }();
  if (!sens && !steps && (rebin_steps > 0 || (rebin > 0 && n_eval > rebin)) && !is_jit_model(model) && C.r.o.group == 1 && nb >= 128) {
    double* seg_dbl = nullptr; int* seg_int = nullptr; unsigned long long *keys = nullptr, *keys_out = nullptr; int *idx_iota = nullptr, *lane_member = nullptr; void* cub_tmp = nullptr;
    AdaptiveConsts* seg_consts = nullptr; unsigned int* remaining = nullptr;
    size_t cub_bytes = 0;
    auto release = [&]() { (void)hipFree(seg_dbl); (void)hipFree(seg_int); (void)hipFree(keys); (void)hipFree(keys_out); (void)hipFree(idx_iota); (void)hipFree(lane_member); (void)hipFree(cub_tmp); (void)hipFree(seg_consts); (void)hipFree(remaining); };
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, keys, keys_out, idx_iota, lane_member, (int)nb, 0, 64, ctx->stream);
    const bool by_steps = rebin_steps > 0;
    const int nseg = by_steps ? 3 : (int)((n_eval + rebin - 1) / rebin);  // by steps: first / resumed / results-only
    bool okm = hipMalloc((void**)&seg_dbl, sizeof(double) * (size_t)kSegDbl * nb) == hipSuccess && hipMalloc((void**)&seg_int, sizeof(int) * (size_t)kSegInt * nb) == hipSuccess &&
               hipMalloc((void**)&keys, 8 * (size_t)nb) == hipSuccess && hipMalloc((void**)&keys_out, 8 * (size_t)nb) == hipSuccess && hipMalloc((void**)&idx_iota, 4 * (size_t)nb) == hipSuccess &&
               hipMalloc((void**)&lane_member, 4 * (size_t)nb) == hipSuccess && hipMalloc(&cub_tmp, cub_bytes ? cub_bytes : 8) == hipSuccess &&
               hipMalloc((void**)&seg_consts, sizeof(AdaptiveConsts) * (size_t)nseg) == hipSuccess && hipMalloc((void**)&remaining, 4) == hipSuccess;
    if (!okm) { (void)hipGetLastError(); release(); set_error("dsh_bdf_solve_adaptive (DSH_REBIN): out of device memory for the segment state"); dsh_free(ctx, totals_dev); if (!cached) { dsh_free(ctx, t_eval_dev); dsh_free(ctx, consts_dev); } return DSH_E_HIP; }
    {
      std::vector<int> iota((size_t)nb);
      for (int64_t k = 0; k < nb; ++k) iota[(size_t)k] = (int)k;
      DSH_HIP_CHECK(hipMemcpyAsync(idx_iota, iota.data(), 4 * (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
      DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    std::vector<AdaptiveConsts> segc((size_t)nseg, C);
    for (int sgi = 0; sgi < nseg; ++sgi) {
      AdaptiveConsts& S = segc[(size_t)sgi];
      S.seg_dbl = seg_dbl; S.seg_int = seg_int; S.seg_key = keys; S.seg_lane_member = sgi == 0 ? nullptr : lane_member;
      S.seg_fresh = sgi == 0; S.seg_last = sgi == nseg - 1; S.seg_remaining = nullptr; S.seg_step_budget = 0;
      S.seg_col_end = (int)std::min<int64_t>(n_eval, (int64_t)(sgi + 1) * rebin);
      if (by_steps) { S.seg_col_end = (int)n_eval + 1; S.seg_step_budget = rebin_steps; S.seg_remaining = remaining; if (S.seg_last) S.seg_lane_member = nullptr; }
    }
    DSH_HIP_CHECK(hipMemcpyAsync(seg_consts, segc.data(), sizeof(AdaptiveConsts) * (size_t)nseg, hipMemcpyHostToDevice, ctx->stream));
    auto launch_seg = [&](const AdaptiveConsts* cd) {
      return dispatch_static_model(model, size, [&](auto mdl) {
        using Mdl = decltype(mdl);
        if constexpr (Mdl::N <= 4) {
          if (ba) hipLaunchKernelGGL((k_bdf_adaptive<Mdl, true, false, true>), grid, blk, 0, ctx->stream, nb, p, atol, cd, (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev);
          else hipLaunchKernelGGL((k_bdf_adaptive<Mdl, false, false, true>), grid, blk, 0, ctx->stream, nb, p, atol, cd, (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev);
        }
      });
    };
    bool seg_ok = true;
    if (by_steps) {
      for (long sgi = 0; seg_ok; ++sgi) {
        DSH_HIP_CHECK(hipMemsetAsync(remaining, 0, 4, ctx->stream));
        seg_ok = launch_seg(seg_consts + (sgi == 0 ? 0 : 1));
        unsigned int left = 0;
        DSH_HIP_CHECK(hipMemcpyAsync(&left, remaining, 4, hipMemcpyDeviceToHost, ctx->stream));
        (void)hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, keys, keys_out, idx_iota, lane_member, (int)nb, 0, 64, ctx->stream);
        DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (left == 0) break;
      }
      if (seg_ok) seg_ok = launch_seg(seg_consts + 2);
    } else {
      for (int sgi = 0; sgi < nseg && seg_ok; ++sgi) {
        seg_ok = launch_seg(seg_consts + sgi);
        if (sgi + 1 < nseg)
          (void)hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, keys, keys_out, idx_iota, lane_member, (int)nb, 0, 64, ctx->stream);
      }
    }
    DSH_HIP_CHECK(hipGetLastError());
    DSH_HIP_CHECK(timing_end(ctx));
    unsigned long long totals_s[8] = {0};
    DSH_HIP_CHECK(hipMemcpyAsync(totals_s, totals_dev, sizeof(unsigned long long) * 6, hipMemcpyDeviceToHost, ctx->stream));
    DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    DSH_HIP_CHECK(timing_collect(ctx));
    release();
    if (!cached) { dsh_free(ctx, t_eval_dev); dsh_free(ctx, consts_dev); }
    dsh_free(ctx, totals_dev);
    if (totals_host) for (int k = 0; k < 6; ++k) totals_host[k] = (int64_t)totals_s[k];
    return DSH_OK;
  }
#endif  // DSH_EXPERIMENTS
  // per-member control of the register-resident models can run on the phase-scheduled kernel (dsh_member_sched_kernel.hpp; same bits) with
  // DSH_MEMBER_SCHED=1: opt-in (measured slower than the nested-loop kernel on C2: 10.75 vs 9.23 ms); read per call: tests run both kernels
#ifdef DSH_EXPERIMENTS
  const bool sched_env = [] { const char* e = std::getenv("DSH_MEMBER_SCHED"); return e && e[0] == '1'; }();
  const bool lane_v1_env = [] { const char* e = std::getenv("DSH_LANE_BANDED_V1"); return e && e[0] == '1'; }();  // the round-1 banded branch of k_bdf_adaptive (slower, same bits)
#else
  constexpr bool sched_env = false, lane_v1_env = false;
#endif
  if (is_jit_model(model)) {  // run-time-compiled model: the same kernel template, instantiated by hiprtc for the user's model
    const JitInfo* ji = jit_info(model);
    const bool sched = !sens && sched_env && C.r.o.group == 1 && ji && ji->form == DSH_JIT_FORM_STATIC && !ji->has_reset;  // the phase-scheduled kernel stops at events
    // banded lane-per-member form: the memory-streaming kernel (dsh_lane_banded_kernel.hpp; same bits); DSH_LANE_BANDED_V1=1 keeps k_bdf_adaptive's banded branch
    const bool lane_v2 = ji && ji->form == DSH_JIT_FORM_STATIC_BANDED && (ji->has_mass || !lane_v1_env) && !sens;  // forward sensitivities: k_bdf_adaptive's banded branch (the streaming kernel does not carry them)  // models with a mass matrix: k_bdf_lane_banded only
    const std::string tail = std::string(ba ? "true" : "false") + ", " + (C.r.o.group == 64 ? "true" : "false") + (sens ? ", false, true>" : ">");  // SENS: <.., SEG = false, SENS = true>
    const std::string name = sched ? std::string("dsh::k_bdf_member_sched<dsh::JitModel, ") + (ba ? "true" : "false") + ">"
                             : lane_v2 ? "dsh::k_bdf_lane_banded<dsh::JitModel, " + tail : "dsh::k_bdf_adaptive<dsh::JitModel, " + tail;
    // the banded lane kernel has two code objects: below four wavefronts per SIMD-quartet of the device (nb / (CUs x 64) < 4: one GPU's 32 768-member share of BASELINE
    // config 4 is 2) the small-ensemble build (dsh_jit.hip compile_module: one wavefront per SIMD, deeper unrolling, doubled chunks); DSH_LANE_BANDED_SMALL=0 / 1 forces
    static const int small_env = [] { const char* e = std::getenv("DSH_LANE_BANDED_SMALL"); return e && *e ? std::atoi(e) : -1; }();
    const bool small = lane_v2 && !sched && (small_env >= 0 ? small_env != 0 : nb < (int64_t)4 * 64 * ctx->num_cu);
    rc = jit_launch(ctx, model, sched ? "dsh_member_sched_kernel.hpp" : (lane_v2 ? "dsh_lane_banded_kernel.hpp" : "dsh_adaptive_kernel.hpp"), small ? name + "#small" : name, {name}, name, grid, blk, 0, nb, p, atol, (const AdaptiveConsts*)consts_dev, (const double*)t_eval_dev, y_out,
                    stats, status, t_root, root_idx, ncols, totals_dev);
    if (rc != DSH_OK) { if (!cached) { dsh_free(ctx, t_eval_dev); dsh_free(ctx, consts_dev); } dsh_free(ctx, totals_dev); return rc; }
    launched = true;
  } else
  if (!sens && C.r.o.deterministic_pow == 2 &&
      // the fast-arithmetic variant (dsh_adaptive_fast.hip): static models with n <= 4; everything else about the call is the same.  A model without that build
      // falls through to the exact kernel below (its `det` is "deterministic_pow != 0")
      adaptive_fast_launch(model, size, ba, C.r.o.group == 64, grid, ctx->stream, nb, p, atol, (const AdaptiveConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats,
                           status, t_root, root_idx, ncols, totals_dev)) {
    launched = true;
  } else if (sens) {
    launched = dispatch_static_model(model, size, [&](auto mdl) {
      using Mdl = decltype(mdl);
      if constexpr (adaptive_sens_ok<Mdl>()) {
#define DSH_ADAPTIVE_SENS_LAUNCH(BA, WAVE) \
  hipLaunchKernelGGL((k_bdf_adaptive<Mdl, BA, WAVE, false, true>), grid, blk, 0, ctx->stream, nb, p, atol, (const AdaptiveConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev)
        if (C.r.o.group == 64) { if (ba) DSH_ADAPTIVE_SENS_LAUNCH(true, true); else DSH_ADAPTIVE_SENS_LAUNCH(false, true); }
        else { if (ba) DSH_ADAPTIVE_SENS_LAUNCH(true, false); else DSH_ADAPTIVE_SENS_LAUNCH(false, false); }
#undef DSH_ADAPTIVE_SENS_LAUNCH
      }
    });
  } else
  launched = dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    if constexpr (Mdl::N <= 4) {
#define DSH_ADAPTIVE_LAUNCH(BA, WAVE) \
  hipLaunchKernelGGL((k_bdf_adaptive<Mdl, BA, WAVE>), grid, blk, 0, ctx->stream, nb, p, atol, (const AdaptiveConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev)
      if (C.r.o.group == 64) { if (ba) DSH_ADAPTIVE_LAUNCH(true, true); else DSH_ADAPTIVE_LAUNCH(false, true); }
#ifdef DSH_EXPERIMENTS
      else if (sched_env) {
        if (ba) hipLaunchKernelGGL((k_bdf_member_sched<Mdl, true>), grid, blk, 0, ctx->stream, nb, p, atol, (const AdaptiveConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev);
        else hipLaunchKernelGGL((k_bdf_member_sched<Mdl, false>), grid, blk, 0, ctx->stream, nb, p, atol, (const AdaptiveConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev);
      }
#endif
      else { if (ba) DSH_ADAPTIVE_LAUNCH(true, false); else DSH_ADAPTIVE_LAUNCH(false, false); }
#undef DSH_ADAPTIVE_LAUNCH
    }
  });
  (void)launched;
  DSH_HIP_CHECK(hipGetLastError());
  DSH_HIP_CHECK(timing_end(ctx));
  unsigned long long totals[8] = {0};
  DSH_HIP_CHECK(hipMemcpyAsync(totals, totals_dev, sizeof(unsigned long long) * 6, hipMemcpyDeviceToHost, ctx->stream));
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  DSH_HIP_CHECK(timing_collect(ctx));
  if (!cached) { dsh_free(ctx, t_eval_dev); dsh_free(ctx, consts_dev); }
  dsh_free(ctx, totals_dev);
  if (totals_host) for (int k = 0; k < 6; ++k) totals_host[k] = (int64_t)totals[k];
  return DSH_OK;
}
}  // namespace
