// Device-resident BDF for MEDIUM-sized systems (8 < n <= 64): one WAVEFRONT per ensemble member, lane = state component (gfx950).
// BASELINE config 4 (single-particle battery model, n = 42, 262 144 members, stop conditions on the terminal voltage) on the device.
//
// dsh_adaptive.hip keeps a whole member in one lane's registers, which ends at n = 4.  Here the member is spread over the lanes of one wavefront:
// lane i holds component i of the state, of the prediction, of psi, its row of the difference array D (8 doubles) and its row of M - cJ (the LU
// factors, NP doubles: the wavefront LU of dsh_lu_wave.hpp, in place); the cached Jacobian sits in LDS.  Every scalar of Bdf::step — t, h, order,
// Convergence, JacobianUpdate, PI memory — is wavefront-uniform, so the control flow of one member never diverges, and members do not wait for each
// other: per-member step sizes, orders and EVENT TIMES (the reference's CPU semantics for a sweep) with no host in the loop and no SIMT tax.
//   * model evaluation: x is exchanged through 512 B of LDS; component i of f and row i of J by dyn_component (dsh_models_dyn.hpp)
//   * norms: per-lane terms, summed in index order over v_readlane (the oracle's sequential sum: bit-identical)
//   * linear solves: wave_lu_solve_rows — the rows stay where the factorisation left them, no interchange of the right-hand side
// Arithmetic is the oracle's operation for operation; pow() and the model's tanh/asinh/exp are ocml's (tests state the tolerance).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dsh_internal.hpp"
#include "dsh_resident.hpp"
#include "dsh_wave_member_kernel.hpp"
#include "dsh_team_member_kernel.hpp"
#include "dsh_sdirk_wave_member_kernel.hpp"
#include "dsh_jit.hpp"

using namespace dsh;


extern "C" {

// 1: one wavefront per member (n <= 64; BDF and the SDIRK methods); 2: one workgroup per member (64 < n <= 320, identity mass; BDF — dsh_team_member_kernel.hpp); 0: neither
int dsh_model_has_wave_member(int model, int64_t size) {
  if (is_jit_model(model)) {  // run-time-sized DiffSL model: at most two stop conditions, one lane per component
    const JitInfo* ji = jit_info(model);
    if (!(ji && ji->form == DSH_JIT_FORM_DYNAMIC && ji->nroots <= 2 && ji->np <= 64)) return 0;  // (the workgroup form would take up to 128 / 192 parameters)
    // with a mass matrix (DAEs: consistent initialisation and M in the residual and in M - cJ) the kernel keeps M's rows in LDS next to J's: n <= 48
    if (ji->n <= (ji->has_mass ? 48 : 64)) return 1;
    return !ji->has_mass && ji->n <= kTeamMaxN ? 2 : 0;
  }
  if (!(model == DSH_MODEL_DYDT_Y2 || model == DSH_MODEL_GAUSSIAN_DECAY || model == DSH_MODEL_HEAT1D || model == DSH_MODEL_SPM ||
        (model == DSH_MODEL_ROBERTSON_ODE && size > 1)))
    return 0;
  int64_t n = 0;
  int has_mass = 0;
  if (dsh_model_info(model, size, &n, nullptr, &has_mass, nullptr) != DSH_OK) return 0;
  if (n < 1) return 0;
  if (n <= 64) return 1;
  return !has_mass && n <= kTeamMaxN ? 2 : 0;  // the workgroup form takes identity-mass models
}

// the register-resident LU of the workgroup-per-member BDF (64 < n <= 128) — on unless DSH_TEAM_REG_LU=0 (read at every launch: the tests compare the two forms)
static bool team_reg_lu_on() { const char* e = getenv("DSH_TEAM_REG_LU"); return !(e && e[0] == '0'); }
// identity-mass models above this size take the workgroup form (two wavefronts, the LU in registers) instead of the wavefront-per-member kernel; DSH_TEAM_SMALL_MIN overrides
static int team_small_min() { const char* e = getenv("DSH_TEAM_SMALL_MIN"); const int v = e ? atoi(e) : 32; return v < 8 ? 8 : (v > 64 ? 64 : v); }
struct WmSensSpec { double* out; double rtol; const double* atol_host; int64_t natol; };
struct WmStepsSpec { double* t_out; int64_t cap; };  // OdeSolverMethod::solve: every accepted step out (WaveMemberConsts::steps_cap)
static int bdf_solve_wave_member_impl(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                      double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                                      int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host, const WmSensSpec* sens,
                                      const WmStepsSpec* steps = nullptr);
// OdeSolverMethod::solve (method.rs:227-258 over :881-961) inside the launch of the wavefront- / workgroup-per-member BDF: the state after every accepted step of every
// member (y_out [max_cols][n][nb], t_out [max_cols][nb], ncols[b] = the columns member b produced; columns beyond max_cols are counted, not stored), to t_final
int dsh_bdf_solve_wave_member_steps(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                    double h0, const dsh_adaptive_options* opts, double t_final, int64_t max_cols, double* y_out, double* t_out, int32_t* stats,
                                    int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(max_cols >= 2 && max_cols <= 0x7fffffff && y_out != nullptr && t_out != nullptr && ncols != nullptr, "dsh_bdf_solve_wave_member_steps: max_cols >= 2, y_out, t_out and ncols are needed");
  const WmStepsSpec st{t_out, max_cols};
  return bdf_solve_wave_member_impl(ctx, model, size, nb, p, atol, atol_nb, rtol, t0, h0, opts, &t_final, 1, y_out, stats, status, t_root, root_idx, ncols, totals_host, nullptr, &st);
}
int dsh_bdf_solve_wave_member(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                              double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                              int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host) {
  DSH_ENTER(ctx);
  return bdf_solve_wave_member_impl(ctx, model, size, nb, p, atol, atol_nb, rtol, t0, h0, opts, t_eval_host, n_eval, y_out, stats, status, t_root, root_idx, ncols, totals_host, nullptr);
}
// hybrid models whose events are handled INSIDE the wavefront-per-member kernels (BDF, TR-BDF2, ESDIRK34: the reset applied at every event, then on to the last save
// point): run-time-compiled models with a reset operator, root functions and no mass matrix, n <= 64
int dsh_model_has_wave_member_reset(int model, int64_t size) {
  const int kind = is_jit_model(model) ? dsh_model_has_wave_member(model, size) : 0;  // 1: a wavefront per member (n <= 64); 2: a workgroup per member (64 < n <= 320; BDF)
  if (kind == 0) return 0;
  const JitInfo* ji = jit_info(model);
  return ji && ji->has_reset && !ji->has_mass && ji->nroots > 0 ? kind : 0;
}
// forward sensitivities in the wavefront-per-member BDF (k_bdf_wave_member<.., SENS>): run-time-compiled dense ODE models with parameter derivatives, n <= 64, at most
// kWmMaxSensParams parameters, no mass matrix, no root functions
int dsh_model_has_wave_member_sens(int model, int64_t size) {
  const int kind = is_jit_model(model) ? dsh_model_has_wave_member(model, size) : 0;  // 2: the workgroup-per-member BDF (64 < n <= 320)
  if (kind == 0) return 0;
  const JitInfo* ji = jit_info(model);
  return ji && ji->has_sens && !ji->has_mass && ji->nroots == 0 && ji->np <= kWmMaxSensParams ? kind : 0;
}
int dsh_bdf_solve_wave_member_sens(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                   double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double sens_rtol, const double* sens_atol_host,
                                   int64_t nsens_atol, double* y_out, double* sens_out, int32_t* stats, int32_t* status, int64_t* totals_host) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(sens_out != nullptr, "sens_out is null");
  DSH_REQUIRE(nsens_atol == 0 || sens_atol_host != nullptr, "sens_atol is null");
  if (!dsh_model_has_wave_member_sens(model, size)) {
    set_error("dsh_bdf_solve_wave_member_sens: needs a run-time-compiled ODE model with parameter derivatives, n <= 320, at most 16 parameters, no mass matrix, no root functions");
    return DSH_E_UNSUPPORTED;
  }
  for (int64_t i = 1; i < nsens_atol; ++i) DSH_REQUIRE(sens_atol_host[i] == sens_atol_host[0], "the wavefront-per-member kernel takes one sens_atol for every state");
  const WmSensSpec sp{sens_out, sens_rtol, sens_atol_host, nsens_atol};
  return bdf_solve_wave_member_impl(ctx, model, size, nb, p, atol, atol_nb, rtol, t0, h0, opts, t_eval_host, n_eval, y_out, stats, status, nullptr, nullptr, nullptr, totals_host, &sp);
}
static int bdf_solve_wave_member_impl(dsh_ctx* ctx, int model, int64_t size, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol, double t0,
                                      double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                                      int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host, const WmSensSpec* sens,
                                      const WmStepsSpec* steps) {
  DSH_REQUIRE(ctx != nullptr, "ctx is null");
  DSH_REQUIRE(n_eval >= 1 && t_eval_host != nullptr, "t_eval must hold at least one time");
  DSH_REQUIRE(atol_nb == 1 || atol_nb == nb, "atol must be broadcast (nbatch 1) or per member");
  for (int64_t q = 0; q + 1 < n_eval; ++q) DSH_REQUIRE(t_eval_host[q] <= t_eval_host[q + 1], "t_eval must be increasing (InvalidTEval)");
  DSH_REQUIRE(t_eval_host[0] >= t0, "t_eval[0] before t0 (InvalidTEval)");
  const int wm_kind = dsh_model_has_wave_member(model, size);
  if (!wm_kind) { set_error("dsh_bdf_solve_wave_member: needs a run-time-sized model (built-in or DiffSL) with n <= 64 (n <= 48 with a mass matrix; identity mass: n <= 320, one workgroup per member) and at most two stop conditions"); return DSH_E_UNSUPPORTED; }
  if (nb == 0) return DSH_OK;
  WaveMemberConsts C;
  int64_t n = 0, np = 0, nroots = 0;
  int rc = dsh_model_info(model, size, &n, &np, nullptr, &nroots);
  if (rc != DSH_OK) return rc;
  DSH_REQUIRE(np <= (wm_kind == 2 ? 64 * team_waves((int)n) : 64) && nroots <= 2, "wave-member kernel: at most 64 parameters (workgroup form: one per thread) and 2 root functions");
  C.model = model; C.n = (int)n; C.np = (int)np; C.nroots = (int)nroots;
  C.sens_out = sens ? sens->out : nullptr; C.sens_rtol = sens ? sens->rtol : 0.0; C.sens_atol = sens && sens->natol > 0 ? sens->atol_host[0] : 0.0;
  C.sens_error_control = sens && sens->natol > 0 ? 1 : 0; C.sens_pad = 1;
  C.steps_t_out = steps ? steps->t_out : nullptr; C.steps_cap = steps ? (int)steps->cap : 0; C.steps_pad = 0;
  C.r.rtol = rtol; C.r.t0 = t0; C.r.h0 = h0; C.r.n_eval = (int)n_eval; C.r.member_lanes = 0;
  C.r.ls_steptol = std::pow(2.220446049250313e-16, 2.0 / 3.0);
  if (opts) C.r.o = *opts; else dsh_adaptive_default_options(&C.r.o);
  if (C.r.o.max_steps <= 0) C.r.o.max_steps = 10000000;
  {
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
    for (int ord = 1; ord <= kMaxOrder; ++ord) {
      double* U = C.u[ord - 1];
      for (int k = 0; k < 36; ++k) U[k] = 0.0;
      for (int j = 0; j <= ord; ++j) U[j * 6 + 0] = 1.0;
      for (int j = 1; j <= ord; ++j)
        for (int i = 1; i <= ord; ++i) U[j * 6 + i] = U[j * 6 + i - 1] * ((double)i - 1.0 - 1.0 * (double)j) / (double)i;
    }
  }
  double* t_eval_dev = nullptr;
  unsigned long long* totals_dev = nullptr;
  WaveMemberConsts* consts_dev = nullptr;
  rc = dsh_malloc(ctx, (int64_t)sizeof(WaveMemberConsts), 0, (void**)&consts_dev);
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipMemcpyAsync(consts_dev, &C, sizeof(WaveMemberConsts), hipMemcpyHostToDevice, ctx->stream));
  rc = dsh_malloc(ctx, (int64_t)(sizeof(double) * n_eval), 0, (void**)&t_eval_dev);
  if (rc != DSH_OK) return rc;
  rc = dsh_malloc(ctx, (int64_t)(sizeof(unsigned long long) * 8), 1, (void**)&totals_dev);
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipMemcpyAsync(t_eval_dev, t_eval_host, sizeof(double) * n_eval, hipMemcpyHostToDevice, ctx->stream));
  int has_mass = 0;
  (void)dsh_model_info(model, size, nullptr, nullptr, &has_mass, nullptr);
  const int ab = atol_nb == 1 ? 1 : 0;
  double* jac_scratch = nullptr;
  // 32 < n <= 64, identity mass, no sensitivities: the workgroup form with the LU in registers too (faster from n = 33 on: 1024 members 11.9 -> 7.5 ms there, 13.9 -> 8.8 ms at n = 48) — the wavefront-per-member kernel's elimination (one lane per row, a
  // jump per pivot into straight-line code) needs more than the 512 registers a wavefront can have next to the integrator's state at NP = 64 (2.5 KB of scratch, most of
  // its traffic inside the elimination: 227 us per factorisation at n = 60 inside that kernel against 90 us here; profiles/r06_team_reg_lu.md)
  const bool small_team = wm_kind == 1 && n > team_small_min() && !has_mass && !sens && team_reg_lu_on();
  if (wm_kind == 2 || small_team) {
    // one workgroup per member (48 < n <= 320): the factors in registers (n <= 128) / LDS / global scratch, the cached Jacobians in global scratch (n^2 doubles per member)
    const int waves = team_waves((int)n);
    // 64 < n <= 128 without sensitivities: the factors in the registers of four wavefronts (dsh_team_reg_lu.hpp; DSH_TEAM_REG_LU=0: in LDS, two wavefronts, as before).
    // NL, the compile-time bound on n: n rounded up to 8 for a run-time-compiled model (its module is its own), five steps for the built-in ones
    // (120: the reference's own benchmark size, robertson_ode x 40).
    const bool reg_lu = !sens && n <= kTrgMaxN && team_reg_lu_on();
    const int NL = !reg_lu ? 0 : (is_jit_model(model) ? trg_nl((int)n) : (n <= 48 ? 48 : (n <= 64 ? 64 : (n <= 80 ? 80 : (n <= 96 ? 96 : (n <= 112 ? 112 : (n <= 120 ? 120 : 128)))))));
    const size_t lds_team = sizeof(double) * (reg_lu ? team_rl_lds_doubles(NL) : team_lds_doubles((int)n, waves));
    rc = dsh_malloc(ctx, (int64_t)(sizeof(double) * team_scratch_doubles((int)n, waves)) * nb, 0, (void**)&jac_scratch);
    if (rc != DSH_OK) { dsh_free(ctx, t_eval_dev); dsh_free(ctx, totals_dev); dsh_free(ctx, consts_dev); return rc; }
    DSH_HIP_CHECK(timing_begin(ctx));
    if (is_jit_model(model)) {
      const std::string name = reg_lu ? std::string("dsh::k_bdf_team_member_rl<") + std::to_string(NL) + ">"
                                      : std::string("dsh::k_bdf_team_member<") + std::to_string(waves) + (sens ? ", true>" : ">");
      rc = jit_launch(ctx, model, "dsh_jit_team_member.hpp", name, {name}, name, dim3((unsigned)nb), dim3(reg_lu ? trg_threads(NL) : 64 * waves), (unsigned)lds_team, nb, p, atol, ab,
                      (const WaveMemberConsts*)consts_dev, (const double*)t_eval_dev, jac_scratch, y_out, stats, status, t_root, root_idx, ncols, totals_dev);
      if (rc != DSH_OK) { dsh_free(ctx, jac_scratch); dsh_free(ctx, t_eval_dev); dsh_free(ctx, totals_dev); dsh_free(ctx, consts_dev); return rc; }
    } else if (reg_lu) {
      static bool attr_rl_dev[64] = {false};
      bool& attr = attr_rl_dev[ctx->device & 63];
      if (!attr) {
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_bdf_team_member_rl<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_bdf_team_member_rl<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_bdf_team_member_rl<80>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_bdf_team_member_rl<96>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_bdf_team_member_rl<112>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_bdf_team_member_rl<120>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_bdf_team_member_rl<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
      }
#define DSH_TMR_LAUNCH(NLV)                                                                                                                                              \
  hipLaunchKernelGGL((k_bdf_team_member_rl<NLV>), dim3((unsigned)nb), dim3(trg_threads(NLV)), lds_team, ctx->stream, nb, p, atol, ab, (const WaveMemberConsts*)consts_dev, \
                     (const double*)t_eval_dev, jac_scratch, y_out, stats, status, t_root, root_idx, ncols, totals_dev)
      if (getenv("DSH_TEAM_DEBUG")) {  // how many members share a CU (two when the workspace fits twice)
        int occ = 0;
        if (NL == 128) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bdf_team_member_rl<128>, kTrgThreads, lds_team);
        else if (NL == 120) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bdf_team_member_rl<120>, kTrgThreads, lds_team);
        else if (NL == 112) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bdf_team_member_rl<112>, kTrgThreads, lds_team);
        else if (NL == 96) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bdf_team_member_rl<96>, kTrgThreads, lds_team);
        else if (NL == 64) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bdf_team_member_rl<64>, trg_threads(64), lds_team);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bdf_team_member_rl<80>, kTrgThreads, lds_team);
        fprintf(stderr, "k_bdf_team_member_rl<%d>: n = %d, %zu bytes of LDS, %d workgroups per CU\n", NL, (int)n, lds_team, occ);
      }
      if (NL == 48) DSH_TMR_LAUNCH(48); else if (NL == 64) DSH_TMR_LAUNCH(64); else if (NL == 80) DSH_TMR_LAUNCH(80); else if (NL == 96) DSH_TMR_LAUNCH(96); else if (NL == 112) DSH_TMR_LAUNCH(112); else if (NL == 120) DSH_TMR_LAUNCH(120); else DSH_TMR_LAUNCH(128);
#undef DSH_TMR_LAUNCH
    } else {
      static bool attr_dev[64] = {false};
      bool& attr = attr_dev[ctx->device & 63];
      if (!attr) {
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_bdf_team_member<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_bdf_team_member<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
      }
#define DSH_TM_LAUNCH(WV)                                                                                                                                  \
  hipLaunchKernelGGL((k_bdf_team_member<WV>), dim3((unsigned)nb), dim3(64 * WV), lds_team, ctx->stream, nb, p, atol, ab, (const WaveMemberConsts*)consts_dev, \
                     (const double*)t_eval_dev, jac_scratch, y_out, stats, status, t_root, root_idx, ncols, totals_dev)
      if (waves == 2) DSH_TM_LAUNCH(2); else if (waves == 3) DSH_TM_LAUNCH(3); else if (waves == 4) DSH_TM_LAUNCH(4); else DSH_TM_LAUNCH(5);
#undef DSH_TM_LAUNCH
    }
  } else {
  int has_mass = 0;
  (void)dsh_model_info(model, size, nullptr, nullptr, &has_mass, nullptr);
  const size_t lds_bytes = sizeof(double) * (128 + (size_t)n * 64 + (has_mass ? (size_t)n * 64 + 64 : (sens ? 64 : 0)));  // xs | ps | sJ | (sM | xs2)  (sensitivities: xs2)
#define DSH_WM_LAUNCH(NPV)                                                                                                                              \
  hipLaunchKernelGGL((k_bdf_wave_member<NPV>), dim3((unsigned)nb), dim3(64), lds_bytes, ctx->stream, nb, p, atol, ab, (const WaveMemberConsts*)consts_dev, \
                     (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev)
  DSH_HIP_CHECK(timing_begin(ctx));
  if (is_jit_model(model)) {
    const std::string name = std::string("dsh::k_bdf_wave_member<") + (n <= 16 ? "16" : n <= 32 ? "32" : n <= 48 ? "48" : "64") + (sens ? ", true>" : ">");
    rc = jit_launch(ctx, model, "dsh_jit_wave_member.hpp", name, {name}, name, dim3((unsigned)nb), dim3(64), (unsigned)lds_bytes, nb, p, atol, ab,
                    (const WaveMemberConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev);
    if (rc != DSH_OK) { dsh_free(ctx, t_eval_dev); dsh_free(ctx, totals_dev); dsh_free(ctx, consts_dev); return rc; }
  } else if (n <= 16) DSH_WM_LAUNCH(16);
  else if (n <= 32) DSH_WM_LAUNCH(32);
  else if (n <= 48) DSH_WM_LAUNCH(48);
  else DSH_WM_LAUNCH(64);
  }
#undef DSH_WM_LAUNCH
  DSH_HIP_CHECK(hipGetLastError());
  DSH_HIP_CHECK(timing_end(ctx));
  unsigned long long totals[8] = {0};
  DSH_HIP_CHECK(hipMemcpyAsync(totals, totals_dev, sizeof(unsigned long long) * 6, hipMemcpyDeviceToHost, ctx->stream));
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  DSH_HIP_CHECK(timing_collect(ctx));
  if (jac_scratch) dsh_free(ctx, jac_scratch);
  dsh_free(ctx, t_eval_dev);
  dsh_free(ctx, totals_dev);
  dsh_free(ctx, consts_dev);
  if (totals_host) for (int q = 0; q < 6; ++q) totals_host[q] = (int64_t)totals[q];
  return DSH_OK;
}

// has_wave_member for the SDIRK methods: the wavefront-per-member models (n <= 64); the workgroup-per-member form is BDF only
int dsh_model_has_wave_member_sdirk(int model, int64_t size) { return dsh_model_has_wave_member(model, size); }  // 1: a wavefront per member; 2: a workgroup per member (k_sdirk_wave_member<.., TW>)

static int sdirk_solve_wave_member_impl(dsh_ctx* ctx, int model, int64_t size, int method, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                        double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                                        int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host, const WmSensSpec* sens,
                                        const WmStepsSpec* steps = nullptr);
// OdeSolverMethod::solve (method.rs:227-258 over :881-961) inside the launch of the wavefront- / workgroup-per-member TR-BDF2 / ESDIRK34: the arguments of
// dsh_bdf_solve_wave_member_steps plus the method
int dsh_sdirk_solve_wave_member_steps(dsh_ctx* ctx, int model, int64_t size, int method, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                      double t0, double h0, const dsh_adaptive_options* opts, double t_final, int64_t max_cols, double* y_out, double* t_out,
                                      int32_t* stats, int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(max_cols >= 2 && max_cols <= 0x7fffffff && y_out != nullptr && t_out != nullptr && ncols != nullptr, "dsh_sdirk_solve_wave_member_steps: max_cols >= 2, y_out, t_out and ncols are needed");
  const WmStepsSpec st{t_out, max_cols};
  return sdirk_solve_wave_member_impl(ctx, model, size, method, nb, p, atol, atol_nb, rtol, t0, h0, opts, &t_final, 1, y_out, stats, status, t_root, root_idx, ncols, totals_host, nullptr, &st);
}
int dsh_sdirk_solve_wave_member(dsh_ctx* ctx, int model, int64_t size, int method, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                                int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host) {
  DSH_ENTER(ctx);
  return sdirk_solve_wave_member_impl(ctx, model, size, method, nb, p, atol, atol_nb, rtol, t0, h0, opts, t_eval_host, n_eval, y_out, stats, status, t_root, root_idx, ncols, totals_host, nullptr);
}
// forward sensitivities in the wavefront-per-member TR-BDF2 / ESDIRK34 (k_sdirk_wave_member<.., SENS>): the models of dsh_model_has_wave_member_sens
int dsh_sdirk_solve_wave_member_sens(dsh_ctx* ctx, int model, int64_t size, int method, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                     double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double sens_rtol,
                                     const double* sens_atol_host, int64_t nsens_atol, double* y_out, double* sens_out, int32_t* stats, int32_t* status, int64_t* totals_host) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(sens_out != nullptr, "sens_out is null");
  DSH_REQUIRE(nsens_atol == 0 || sens_atol_host != nullptr, "sens_atol is null");
  if (!dsh_model_has_wave_member_sens(model, size)) {
    set_error("dsh_sdirk_solve_wave_member_sens: needs a run-time-compiled ODE model with parameter derivatives, n <= 320, at most 16 parameters, no mass matrix, no root functions");
    return DSH_E_UNSUPPORTED;
  }
  for (int64_t i = 1; i < nsens_atol; ++i) DSH_REQUIRE(sens_atol_host[i] == sens_atol_host[0], "the wavefront-per-member kernels take one sens_atol for every state");
  const WmSensSpec sp{sens_out, sens_rtol, sens_atol_host, nsens_atol};
  return sdirk_solve_wave_member_impl(ctx, model, size, method, nb, p, atol, atol_nb, rtol, t0, h0, opts, t_eval_host, n_eval, y_out, stats, status, nullptr, nullptr, nullptr, totals_host, &sp);
}
static int sdirk_solve_wave_member_impl(dsh_ctx* ctx, int model, int64_t size, int method, int64_t nb, const double* p, const double* atol, int64_t atol_nb, double rtol,
                                        double t0, double h0, const dsh_adaptive_options* opts, const double* t_eval_host, int64_t n_eval, double* y_out, int32_t* stats,
                                        int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols, int64_t* totals_host, const WmSensSpec* sens,
                                        const WmStepsSpec* steps) {
  DSH_REQUIRE(ctx != nullptr, "ctx is null");
  DSH_REQUIRE(method == 1 || method == 2, "method: 1 TR-BDF2, 2 ESDIRK34");
  DSH_REQUIRE(n_eval >= 1 && t_eval_host != nullptr, "t_eval must hold at least one time");
  DSH_REQUIRE(atol_nb == 1 || atol_nb == nb, "atol must be broadcast (nbatch 1) or per member");
  for (int64_t q = 0; q + 1 < n_eval; ++q) DSH_REQUIRE(t_eval_host[q] <= t_eval_host[q + 1], "t_eval must be increasing (InvalidTEval)");
  DSH_REQUIRE(t_eval_host[0] >= t0, "t_eval[0] before t0 (InvalidTEval)");
  const int wm_kind = dsh_model_has_wave_member_sdirk(model, size);  // 2: one workgroup per member (64 < n <= 320, identity mass)
  if (!wm_kind) { set_error("dsh_sdirk_solve_wave_member: needs a run-time-sized model (built-in or DiffSL) with n <= 64 (n <= 48 with a mass matrix; 64 < n <= 320 without one) and at most two stop conditions"); return DSH_E_UNSUPPORTED; }
  if (nb == 0) return DSH_OK;
  WaveSdirkConsts C;
  int64_t n = 0, np = 0, nroots = 0;
  int rc = dsh_model_info(model, size, &n, &np, nullptr, &nroots);
  if (rc != DSH_OK) return rc;
  C.model = model; C.n = (int)n; C.np = (int)np; C.nroots = (int)nroots;
  fill_tableau(method, C.T);
  if (steps) { C.T.steps_t_out = steps->t_out; C.T.steps_cap = (int)steps->cap; }
  C.T.sens_out = sens ? sens->out : nullptr; C.T.sens_rtol = sens ? sens->rtol : 0.0; C.T.sens_error_control = sens && sens->natol > 0 ? 1 : 0; C.T.sens_pad = 1;
  for (int q = 0; q < 4; ++q) C.T.sens_atol[q] = sens && sens->natol > 0 ? sens->atol_host[0] : 0.0;
  C.T.r.rtol = rtol; C.T.r.t0 = t0; C.T.r.h0 = h0; C.T.r.n_eval = (int)n_eval; C.T.r.member_lanes = 0;
  C.T.r.ls_steptol = std::pow(2.220446049250313e-16, 2.0 / 3.0);
  C.T.r.eta_reset = std::pow(20.0, 1.25);
  C.T.r.eta_reset_ts = std::pow(100.0, 1.25);
  if (opts) C.T.r.o = *opts; else dsh_adaptive_default_options(&C.T.r.o);
  if (C.T.r.o.max_steps <= 0) C.T.r.o.max_steps = 10000000;
  double* t_eval_dev = nullptr;
  unsigned long long* totals_dev = nullptr;
  WaveSdirkConsts* consts_dev = nullptr;
  rc = dsh_malloc(ctx, (int64_t)sizeof(WaveSdirkConsts), 0, (void**)&consts_dev);
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipMemcpyAsync(consts_dev, &C, sizeof(WaveSdirkConsts), hipMemcpyHostToDevice, ctx->stream));
  rc = dsh_malloc(ctx, (int64_t)(sizeof(double) * n_eval), 0, (void**)&t_eval_dev);
  if (rc != DSH_OK) return rc;
  rc = dsh_malloc(ctx, (int64_t)(sizeof(unsigned long long) * 8), 1, (void**)&totals_dev);
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipMemcpyAsync(t_eval_dev, t_eval_host, sizeof(double) * n_eval, hipMemcpyHostToDevice, ctx->stream));
  int has_mass = 0;
  (void)dsh_model_info(model, size, nullptr, nullptr, &has_mass, nullptr);
  const size_t lds_bytes = sizeof(double) * (128 + (size_t)n * 64 + (has_mass ? (size_t)n * 64 + 64 : (sens ? 64 : 0)));  // xs | ps | sJ | (sM | xs2)  (sensitivities: xs2)
  const int ab = atol_nb == 1 ? 1 : 0;
  const int S = C.T.s;
#define DSH_WS_LAUNCH(NPV, SV)                                                                                                                                  \
  hipLaunchKernelGGL((k_sdirk_wave_member<NPV, SV>), dim3((unsigned)nb), dim3(64), lds_bytes, ctx->stream, nb, p, atol, ab, (const WaveSdirkConsts*)consts_dev, \
                     (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev, (double*)nullptr)
#define DSH_WS_LAUNCH_S(NPV) do { if (S == 3) DSH_WS_LAUNCH(NPV, 3); else DSH_WS_LAUNCH(NPV, 4); } while (0)
  double* jac_scratch = nullptr;
  DSH_HIP_CHECK(timing_begin(ctx));
  if (wm_kind == 2) {
    // one workgroup per member (64 < n <= 320): the factors in LDS, the cached Jacobians in global scratch (n^2 doubles per member) — k_sdirk_wave_member<.., TW>
    const int waves = team_waves((int)n);
    const size_t lds_team = sizeof(double) * team_lds_doubles((int)n, waves);
    rc = dsh_malloc(ctx, (int64_t)(sizeof(double) * team_scratch_doubles((int)n, waves)) * nb, 0, (void**)&jac_scratch);
    if (rc != DSH_OK) { dsh_free(ctx, t_eval_dev); dsh_free(ctx, totals_dev); dsh_free(ctx, consts_dev); return rc; }
    if (is_jit_model(model)) {
      const std::string name = std::string("dsh::k_sdirk_wave_member<16, ") + std::to_string(S) + (sens ? ", true, " : ", false, ") + std::to_string(waves) + ">";
      rc = jit_launch(ctx, model, "dsh_jit_sdirk_wave_member.hpp", name, {name}, name, dim3((unsigned)nb), dim3(64 * waves), (unsigned)lds_team, nb, p, atol, ab,
                      (const WaveSdirkConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev, jac_scratch);
      if (rc != DSH_OK) { dsh_free(ctx, jac_scratch); dsh_free(ctx, t_eval_dev); dsh_free(ctx, totals_dev); dsh_free(ctx, consts_dev); return rc; }
    } else {
      static bool attr_dev[64] = {false};
      bool& attr = attr_dev[ctx->device & 63];
      if (!attr) {
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_sdirk_wave_member<16, 3, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_sdirk_wave_member<16, 3, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_sdirk_wave_member<16, 4, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_sdirk_wave_member<16, 4, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
      }
#define DSH_TS_LAUNCH(SV, TWV)                                                                                                                                          \
  hipLaunchKernelGGL((k_sdirk_wave_member<16, SV, false, TWV>), dim3((unsigned)nb), dim3(64 * TWV), lds_team, ctx->stream, nb, p, atol, ab, (const WaveSdirkConsts*)consts_dev, \
                     (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev, jac_scratch)
      if (S == 3) { if (waves == 2) DSH_TS_LAUNCH(3, 2); else if (waves == 3) DSH_TS_LAUNCH(3, 3); else if (waves == 4) DSH_TS_LAUNCH(3, 4); else DSH_TS_LAUNCH(3, 5); }
      else { if (waves == 2) DSH_TS_LAUNCH(4, 2); else if (waves == 3) DSH_TS_LAUNCH(4, 3); else if (waves == 4) DSH_TS_LAUNCH(4, 4); else DSH_TS_LAUNCH(4, 5); }
#undef DSH_TS_LAUNCH
    }
  } else
  if (is_jit_model(model)) {
    const std::string name = std::string("dsh::k_sdirk_wave_member<") + (n <= 16 ? "16" : n <= 32 ? "32" : n <= 48 ? "48" : "64") + ", " + std::to_string(S) + (sens ? ", true>" : ">");
    rc = jit_launch(ctx, model, "dsh_jit_sdirk_wave_member.hpp", name, {name}, name, dim3((unsigned)nb), dim3(64), (unsigned)lds_bytes, nb, p, atol, ab,
                    (const WaveSdirkConsts*)consts_dev, (const double*)t_eval_dev, y_out, stats, status, t_root, root_idx, ncols, totals_dev, (double*)nullptr);
    if (rc != DSH_OK) { dsh_free(ctx, t_eval_dev); dsh_free(ctx, totals_dev); dsh_free(ctx, consts_dev); return rc; }
  } else if (n <= 16) DSH_WS_LAUNCH_S(16);
  else if (n <= 32) DSH_WS_LAUNCH_S(32);
  else if (n <= 48) DSH_WS_LAUNCH_S(48);
  else DSH_WS_LAUNCH_S(64);
#undef DSH_WS_LAUNCH_S
#undef DSH_WS_LAUNCH
  DSH_HIP_CHECK(hipGetLastError());
  DSH_HIP_CHECK(timing_end(ctx));
  unsigned long long totals[8] = {0};
  DSH_HIP_CHECK(hipMemcpyAsync(totals, totals_dev, sizeof(unsigned long long) * 6, hipMemcpyDeviceToHost, ctx->stream));
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  DSH_HIP_CHECK(timing_collect(ctx));
  if (jac_scratch) dsh_free(ctx, jac_scratch);
  dsh_free(ctx, t_eval_dev);
  dsh_free(ctx, totals_dev);
  dsh_free(ctx, consts_dev);
  if (totals_host) for (int q = 0; q < 6; ++q) totals_host[q] = (int64_t)totals[q];
  return DSH_OK;
}

}  // extern "C"
