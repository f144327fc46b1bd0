// C ABI (include/diffsol_hip_solver.h) over the host-side integrators.  Error convention mirrors crates/diffsol-c/src/error_c.rs:12-121
// (thread-local last error).  No CPU fallback exists anywhere below: creating a solver without a HIP device fails.
#include <cmath>
#include "../../include/diffsol_hip_solver.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "bdf.hpp"
#include "diffsl.hpp"
#include "sdirk.hpp"

using namespace diffsol_hip;

struct dshs_solver {
  HipContext ctx;
  OdeSolverProblem problem;
  std::unique_ptr<OdeSolverMethod> solver;
  Bdf* bdf = nullptr;
  Sdirk* sdirk = nullptr;
  bool fused = false;
  bool kernel_timing = false;
  int method = 0;
  void make_solver() {
    solver.reset();
    bdf = nullptr;
    sdirk = nullptr;
    problem.eqn->rhs_statistics = OpStatistics();
    if (method == DSHS_METHOD_BDF) {
      auto b = std::make_unique<Bdf>(problem);
      bdf = b.get();
      if (kernel_timing) b->set_fuse_accept(false);
      fused = b->is_fused();
      solver = std::move(b);
    } else if (method == DSHS_METHOD_TR_BDF2 || method == DSHS_METHOD_ESDIRK34) {
      auto k = std::make_unique<Sdirk>(problem, method == DSHS_METHOD_TR_BDF2 ? Tableau::tr_bdf2() : Tableau::esdirk34());
      fused = k->is_fused();
      sdirk = k.get();
      solver = std::move(k);
    } else {
      throw LaError(DSH_E_INVALID, "unknown method");
    }
  }
  HipMat traj_y;
  std::vector<double> traj_t;
  // Which integrator solve_dense runs (include/diffsol_hip_solver.h DSHS_ENSEMBLE_*); -1 = auto.
  int ensemble_mode = -1;
  int last_mode = 0;              // mode the last solve_dense actually ran in
  int last_arith = 1;             // arithmetic of the last device-resident solve_dense: 1 exact, 2 fast (dshs_set_resident_arithmetic)
  int64_t last_totals[6] = {0, 0, 0, 0, 0, 0};
  std::vector<int32_t> scratch_status, scratch_ridx;
  std::vector<double> member_troot;  // root time of every member after a device-resident solve_dense (NaN: none)
  bool resident_roots_valid = false; // dshs_root_info reports the earliest of them instead of the (unstepped) host solver's
  // per-member device-resident solves run the ensemble sorted by parameters (member_order below); cached: the parameters are fixed at creation
  bool order_ready = false;
  void* perm_dev = nullptr;  // sorted position -> member
  void* inv_dev = nullptr;   // member -> sorted position
  void* p_sorted_dev = nullptr;
  ~dshs_solver() {
    for (void* q : {perm_dev, inv_dev, p_sorted_dev}) if (q) dsh_free(ctx.raw(), q);
  }
};

namespace {
thread_local std::string g_err;
template <class F>
int guarded(F&& f) {
  try {
    return f();
  } catch (const DiffsolError& e) {
    g_err = std::string("OdeSolverError::") + e.what();
    return -100 - (int)e.kind;
  } catch (const LaError& e) {
    g_err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
// [b][i] host <- device vector
void download(const HipVec& v, double* host) {
  std::vector<double> tmp = v.clone_as_vec();
  std::memcpy(host, tmp.data(), tmp.size() * sizeof(double));
}

// Which device-resident kernel (if any) integrates this problem (model x method) in the given control granularity.
struct ResidentPick {
  bool ok = false;
  bool wave_member = false;  // one wavefront per member (run-time-sized models, n <= 64; BDF also DiffSL models with a mass matrix, n <= 48; SDIRK identity mass) or, BDF with 64 < n <= 140, one workgroup per member
  int model = 0;
  int64_t size = 0;
  int method = 0;
};
// for_auto: the choice DSHS_ENSEMBLE_AUTO makes.  A built-in banded model with 64 < n <= 512 has a lane-per-member twin too: one lane walks the whole
// state, so the time is one member's chain whatever the ensemble size up to ~65 000 members (n = 512: BDF 0.16 - 0.24 s, TR-BDF2 1.0 - 1.6 s), while the
// host-driven lock-step path grows with the ensemble (BDF 0.30 s, TR-BDF2 0.49 s at 16 384 members; profiles/r02_heat_resident.jsonl).  AUTO takes the
// twin where it pays: BDF from 16 384 members on, the SDIRK methods from 65 536 on; an explicit per-member / wavefront-group request always gets it (it
// is the only per-member device path for such a model).
ResidentPick pick_resident(const dshs_solver* s, int group, bool for_auto = false) {
  ResidentPick r;
  r.method = s->method == DSHS_METHOD_BDF ? 0 : (s->method == DSHS_METHOD_TR_BDF2 ? 1 : 2);
  if (group != 1 && group != 64) return r;
  if (s->problem.sens) {
    // forward sensitivities: the register-resident BDF integrates them alongside (k_bdf_adaptive<.., SENS>; static ODE models, n <= 4, no root functions) on an
    // explicit request (dshs_solve_dense_adaptive_sens); dshs_solve_dense keeps the host-driven path, whose solver state dshs_interpolate_sens reads
    int m = 0; int64_t sz = 0;
    if (for_auto || s->problem.eqn->has_reset()) return r;
    if (s->problem.eqn->fused_model(&m, &sz) && dsh_model_has_adaptive_sens(m, sz)) { r.ok = true; r.model = m; r.size = sz; return r; }
    // run-time-sized model with a banded lane-per-member twin: state and sensitivity arrays in per-lane memory
    int twin = -1;
    if (s->problem.eqn->registry_model(&m, &sz) && (twin = dsh_model_lane_twin(m, sz)) >= 0 && dsh_model_has_adaptive_sens(twin, 0) && dsh_model_has_resident(r.method, twin, 0)) { r.ok = true; r.model = twin; r.size = 0; return r; }  // BDF, TR-BDF2, ESDIRK34
    // dense run-time-compiled model: one wavefront per member (per-member control)
    if (group == 1 && s->problem.eqn->registry_model(&m, &sz)) {
      if (dsh_model_has_wave_member_sens(m, sz)) { r.ok = true; r.wave_member = true; r.model = m; r.size = sz; }  // 1: a wavefront, 2 (64 < n <= 140): a workgroup per member; BDF, TR-BDF2, ESDIRK34
    }
    return r;
  }
  if (s->problem.eqn->has_reset()) {
    // hybrid models: the register-resident integrators apply the reset at every event inside the launch (dsh_model_has_adaptive_reset); every other form stops at an
    // event, so those models stay on the host-driven solve_dense, which applies the reset and continues
    int m = 0; int64_t sz = 0;
    // BDF, TR-BDF2 and ESDIRK34 alike — hybrid DAEs included (reset + mass matrix: the state is made consistent again after every reset, state.rs:279-306)
    if (s->problem.eqn->fused_model(&m, &sz) && dsh_model_has_adaptive_reset(m, sz)) { r.ok = true; r.model = m; r.size = sz; return r; }
    // run-time-sized banded model: its lane-per-member twin carries the reset through the events as well (k_bdf_lane_banded; k_sdirk_resident's banded branch)
    int tw = -1;
    if (s->problem.eqn->registry_model(&m, &sz) && (tw = dsh_model_lane_twin(m, sz)) >= 0 && dsh_model_has_adaptive_reset(tw, 0) && dsh_model_has_resident(r.method, tw, 0)) { r.ok = true; r.model = tw; r.size = 0; return r; }
    // dense run-time-compiled hybrid model: the wavefront-per-member kernels handle the events inside the launch as well (per member)
    if (group == 1 && s->problem.eqn->registry_model(&m, &sz)) {
      if (dsh_model_has_wave_member_reset(m, sz)) { r.ok = true; r.wave_member = true; r.model = m; r.size = sz; }  // 1: a wavefront, 2: a workgroup per member; BDF, TR-BDF2, ESDIRK34
    }
    return r;
  }
  int twin = -1;  // a run-time-sized model may carry its banded lane-per-member form: per-member / wavefront-group solves use it
  const char* lane_env = std::getenv("DSH_RESIDENT_LANE");  // "0": keep banded models on the wavefront-per-member kernel (testing / comparison)
  const bool lane_ok = !(lane_env && lane_env[0] == '0');
  int model = 0;
  int64_t size = 0;
  const bool big_twin_pays = !for_auto || s->problem.eqn->nstates() <= 64 || s->ctx.nbatch() >= (r.method == 0 ? 16384 : 65536);
  if (lane_ok && big_twin_pays && s->problem.eqn->registry_model(&model, &size) && (twin = dsh_model_lane_twin(model, size)) >= 0 && dsh_model_has_resident(r.method, twin, 0)) {
    r.ok = true; r.model = twin; r.size = 0;
  } else if (s->problem.eqn->fused_model(&model, &size) && dsh_model_has_resident(r.method, model, size)) {
    r.ok = true; r.model = model; r.size = size;
  } else if (group == 1 && s->problem.eqn->registry_model(&model, &size) && (r.method == 0 ? dsh_model_has_wave_member(model, size) != 0 : dsh_model_has_wave_member_sdirk(model, size) != 0) &&
             (!for_auto || s->problem.eqn->nstates() <= 140)) {
    // (the automatic mode takes the workgroup form up to the LDS sizes; 140 < n <= 320 — factors in global scratch, correct but slow — on explicit request only)
    r.ok = true; r.wave_member = true; r.model = model; r.size = size;
  } else if (group == 1 && !for_auto && s->problem.eqn->registry_model(&model, &size) && (twin = dsh_model_member_twin(model)) >= 0 &&
             (r.method == 0 ? dsh_model_has_wave_member(twin, 0) != 0 : dsh_model_has_wave_member_sdirk(twin, 0) != 0)) {
    // a static DiffSL model with 5 <= n <= 8: its run-time-sized twin, compiled by this first explicit per-member request (the automatic mode keeps such a model on
    // the host-driven fused kernels it was compiled for)
    r.ok = true; r.wave_member = true; r.model = twin; r.size = 0;
  }
  return r;
}
// DSHS_ENSEMBLE_AUTO: the device-resident integrators whenever the model has one — wavefront-sized lock-step groups (the reference's batched
// semantics, nbatch = 64 per group; for nbatch <= 64 that IS the lock-step ensemble, bit for bit) for models without root functions, one
// history per member (every member stops at its own event) for models with them — else the host-driven lock-step path over the trait operations.
int resolve_mode(const dshs_solver* s) {
  int mode = s->ensemble_mode;
  if (mode == DSHS_ENSEMBLE_AUTO) {
    static const int env_mode = [] {
      const char* e = std::getenv("DSH_ENSEMBLE_MODE");
      if (!e) return DSHS_ENSEMBLE_AUTO;
      const std::string v(e);
      if (v == "lockstep" || v == "0") return DSHS_ENSEMBLE_LOCKSTEP;
      if (v == "member" || v == "1") return DSHS_ENSEMBLE_PER_MEMBER;
      if (v == "wave" || v == "64") return DSHS_ENSEMBLE_WAVEFRONT;
      return DSHS_ENSEMBLE_AUTO;
    }();
    mode = env_mode;
  }
  if (mode == DSHS_ENSEMBLE_AUTO) {
    // the device-resident integrators start from the problem's (t0, y0): a solver that was stepped by hand continues on the host path
    if (s->solver->get_statistics().number_of_steps != 0 || s->solver->t() != s->problem.t0) return DSHS_ENSEMBLE_LOCKSTEP;
    const bool roots = s->problem.eqn->nroots() > 0;
    // a single IVP with a root function keeps the reference's solve_dense contract to the letter (output truncated at the root, solver state moved
    // to it): host-driven.  Ensembles with root functions are integrated per member — every member stops at ITS event (see dshs_root_info).
    if (roots && s->ctx.nbatch() == 1) return DSHS_ENSEMBLE_LOCKSTEP;
    if (!roots && pick_resident(s, 64, true).ok) return DSHS_ENSEMBLE_WAVEFRONT;
    if (pick_resident(s, 1, true).ok) return DSHS_ENSEMBLE_PER_MEMBER;
    return DSHS_ENSEMBLE_LOCKSTEP;
  }
  if (mode != DSHS_ENSEMBLE_LOCKSTEP && !pick_resident(s, mode).ok) return DSHS_ENSEMBLE_LOCKSTEP;
  return mode;
}

// Member order for the per-member kernels: members sorted along a Z-order (Morton) curve through parameter space — up to six parameters, ten bits each,
// every parameter scaled over its range in the ensemble (logarithmically when it is positive and spans more than two decades).  Neighbours on the curve
// have similar parameters, so the 64 members of a wavefront take similar numbers of steps and similar paths through the step logic.  Every member's result
// is independent of its position, so this changes no bit of any output: measured (MI355X) config 4, 262 144 battery members: 0.137 -> 0.069 s; config 2
// per member: 9.1 -> 8.5 ms.  DSH_MEMBER_SORT=0 keeps the caller's order.
bool prepare_member_order(dshs_solver* s) {
  const bool off = [] { const char* e = std::getenv("DSH_MEMBER_SORT"); return e && e[0] == '0'; }();  // read when a solver first needs it: tests compare both
  const int64_t nb = s->ctx.nbatch(), np = s->problem.eqn->nparams();
  if (off || nb < 1024 || np < 1) return false;
  if (s->order_ready) return s->perm_dev != nullptr;
  s->order_ready = true;
  dsh_ctx* c = s->ctx.raw();
  std::vector<double> p((size_t)(np * nb));  // batch-fastest: parameter k of member b at k * nb + b
  check(dsh_d2h(c, p.data(), s->problem.eqn->params().ptr(), (int64_t)sizeof(double) * np * nb), "member order (parameters)");
  const int nd = (int)std::min<int64_t>(np, 6), bits = 10;
  std::vector<uint64_t> key((size_t)nb, 0);
  for (int d = 0; d < nd; ++d) {
    const double* row = p.data() + (size_t)d * nb;
    double lo = row[0], hi = row[0];
    bool finite = true;
    for (int64_t b = 0; b < nb; ++b) { const double v = row[b]; if (!(v == v) || std::isinf(v)) { finite = false; break; } lo = std::min(lo, v); hi = std::max(hi, v); }
    if (!finite || !(hi > lo)) continue;  // constant (or unusable) parameter: contributes nothing to the order
    const bool logscale = lo > 0.0 && hi / lo > 100.0;
    const double a = logscale ? std::log(lo) : lo, w = (logscale ? std::log(hi) : hi) - a;
    for (int64_t b = 0; b < nb; ++b) {
      const double u = ((logscale ? std::log(row[b]) : row[b]) - a) / w;
      uint64_t q = (uint64_t)std::min(1023.0, std::max(0.0, u * 1024.0));
      uint64_t k = key[(size_t)b];
      for (int bit = 0; bit < bits; ++bit) k |= ((q >> bit) & 1ull) << (uint64_t)(bit * nd + d);
      key[(size_t)b] = k;
    }
  }
  // LSD radix sort of (key, member), 16 bits per pass, stable (ties keep the caller's order)
  std::vector<int32_t> perm((size_t)nb), tmp((size_t)nb);
  for (int64_t b = 0; b < nb; ++b) perm[(size_t)b] = (int32_t)b;
  const int total_bits = bits * nd;
  for (int shift = 0; shift < total_bits; shift += 16) {
    std::vector<int64_t> count(65537, 0);
    for (int64_t i = 0; i < nb; ++i) count[((key[(size_t)perm[(size_t)i]] >> shift) & 0xffffull) + 1]++;
    for (int i = 0; i < 65536; ++i) count[(size_t)i + 1] += count[(size_t)i];
    for (int64_t i = 0; i < nb; ++i) tmp[(size_t)count[(key[(size_t)perm[(size_t)i]] >> shift) & 0xffffull]++] = perm[(size_t)i];
    perm.swap(tmp);
  }
  bool identity = true;
  for (int64_t i = 0; i < nb && identity; ++i) identity = perm[(size_t)i] == (int32_t)i;
  if (identity) return false;  // already in order (or nothing to sort by)
  std::vector<int32_t> inv((size_t)nb);
  for (int64_t i = 0; i < nb; ++i) inv[(size_t)perm[(size_t)i]] = (int32_t)i;
  check(dsh_malloc(c, (int64_t)sizeof(int32_t) * nb, 0, &s->perm_dev), "member order");
  check(dsh_malloc(c, (int64_t)sizeof(int32_t) * nb, 0, &s->inv_dev), "member order");
  check(dsh_malloc(c, (int64_t)sizeof(double) * np * nb, 0, &s->p_sorted_dev), "member order");
  check(dsh_h2d(c, s->perm_dev, perm.data(), (int64_t)sizeof(int32_t) * nb), "member order");
  check(dsh_h2d(c, s->inv_dev, inv.data(), (int64_t)sizeof(int32_t) * nb), "member order");
  check(dsh_permute_members(c, np, nb, 8, s->problem.eqn->params().ptr(), (const int32_t*)s->perm_dev, s->p_sorted_dev), "member order");
  return true;
}

static int& resident_arith_flag() {
  static int mode = [] { const char* e = std::getenv("DSH_RESIDENT_ARITH"); return e && std::string(e) == "exact" ? DSHS_ARITH_EXACT : DSHS_ARITH_FAST; }();
  return mode;
}

// the solver's OdeSolverOptions / InitialConditionSolverOptions as the device-resident integrators take them
dsh_adaptive_options adaptive_options_of(const dshs_solver* s, int group, int deterministic_pow) {
  const OdeSolverOptions& oo = s->problem.ode_options;
  const InitialConditionSolverOptions& ic = s->problem.ic_options;
  dsh_adaptive_options o;
  dsh_adaptive_default_options(&o);
  o.max_nonlinear_solver_iterations = oo.max_nonlinear_solver_iterations;
  o.max_error_test_failures = oo.max_error_test_failures;
  o.max_nonlinear_solver_failures = oo.max_nonlinear_solver_failures;
  o.nonlinear_solver_tolerance = oo.nonlinear_solver_tolerance;
  o.min_timestep = oo.min_timestep;
  // BdfConfig / SdirkConfig defaults (config.rs:53-109) are identical
  o.max_timestep_growth = oo.max_timestep_growth.value_or(2.0);
  o.min_timestep_growth = oo.min_timestep_growth.value_or(2.0);
  o.max_timestep_shrink = oo.max_timestep_shrink.value_or(0.9);
  o.min_timestep_shrink = oo.min_timestep_shrink.value_or(0.5);
  o.update_jacobian_after_steps = oo.update_jacobian_after_steps;
  o.update_rhs_jacobian_after_steps = oo.update_rhs_jacobian_after_steps;
  o.threshold_to_update_jacobian = oo.threshold_to_update_jacobian;
  o.threshold_to_update_rhs_jacobian = oo.threshold_to_update_rhs_jacobian;
  o.pi_control_proportional = oo.pi_control_proportional;
  o.pi_control_integral = oo.pi_control_integral;
  o.ic_use_linesearch = ic.use_linesearch ? 1 : 0;
  o.ic_max_linesearch_iterations = ic.max_linesearch_iterations;
  o.ic_max_linear_solver_setups = ic.max_linear_solver_setups;
  o.ic_max_newton_iterations = ic.max_newton_iterations;
  o.ic_step_reduction_factor = ic.step_reduction_factor;
  o.ic_armijo_constant = ic.armijo_constant;
  o.group = group;
  o.deterministic_pow = deterministic_pow;
  return o;
}

// lazy: status_host is fetched only when a member failed (totals[5] != 0; it is all zero otherwise) and root_idx_host only for models with root
// functions (all -1 otherwise): the common case of dshs_solve_dense then moves no per-member bookkeeping over PCIe at all.
void run_resident(dshs_solver* s, const double* t_eval, int64_t nt, int group, int deterministic_pow, double* y_host, double* y_dev, int32_t* stats_host,
                  int32_t* status_host, double* t_root_host, int32_t* root_idx_host, int32_t* ncols_host, int64_t* totals, bool lazy = false, double* sens_host = nullptr) {
  const ResidentPick pk = pick_resident(s, group);
  if (s->problem.sens && (!pk.ok || !sens_host))
    throw LaError(DSH_E_UNSUPPORTED, "device-resident integration with forward sensitivities: dshs_solve_dense_adaptive_sens, BDF, static ODE models with parameter "
                                     "derivatives (n <= 4, no root functions); other problems integrate their sensitivities host-driven (dshs_solve_dense + dshs_interpolate_sens)");
  if (!pk.ok)
    throw LaError(DSH_E_UNSUPPORTED, "solve_dense_adaptive: no device-resident kernel for this model/method (static models with n <= 4 and banded lane-per-member forms: "
                                     "BDF/TR-BDF2/ESDIRK34; run-time-sized models with n <= 140 (n <= 48 with a mass matrix): per member (group 1), BDF/TR-BDF2/ESDIRK34; a static model "
                                     "with 5 <= n <= 8 states runs per member through its run-time-sized twin, dsh_model_set_member_twin_source)");
  const int model = pk.model;
  const int64_t size = pk.size;
  const int method = pk.method;
  const bool wave_member = pk.wave_member;
  const int64_t n = s->problem.eqn->nstates(), nb = s->ctx.nbatch();
  dsh_adaptive_options o = adaptive_options_of(s, group, deterministic_pow);
  dsh_ctx* c = s->ctx.raw();
  const bool sorted = group == 1 && prepare_member_order(s);
  const double* params_dev = sorted ? (const double*)s->p_sorted_dev : s->problem.eqn->params().ptr();
  double* out = y_dev;
  // every device buffer of this call is released when the function leaves, also by the exceptions check() throws between the allocations (an out-of-memory on
  // the second output buffer is exactly when leaking the first one would hurt)
  void *tmp_out = nullptr, *sorted_out = nullptr, *stats_dev = nullptr, *status_dev = nullptr, *troot_dev = nullptr, *ridx_dev = nullptr, *ncols_dev = nullptr, *sens_dev = nullptr,
       *sens_sorted = nullptr;
  struct Release {
    dsh_ctx* c; std::vector<void**> bufs;
    ~Release() { for (void** q : bufs) if (*q) dsh_free(c, *q); }
  } release{c, {&tmp_out, &sorted_out, &stats_dev, &status_dev, &troot_dev, &ridx_dev, &ncols_dev, &sens_dev, &sens_sorted}};
  if (!out) { check(dsh_malloc(c, (int64_t)sizeof(double) * nt * n * nb, 0, &tmp_out), "adaptive out"); out = (double*)tmp_out; }
  double* user_out = out;
  if (sorted) { check(dsh_malloc(c, (int64_t)sizeof(double) * nt * n * nb, 0, &sorted_out), "adaptive out (sorted)"); out = (double*)sorted_out; }
  if (stats_host) check(dsh_malloc(c, (int64_t)sizeof(int32_t) * 5 * nb, 0, &stats_dev), "adaptive stats");
  if (status_host) check(dsh_malloc(c, (int64_t)sizeof(int32_t) * nb, 0, &status_dev), "adaptive status");
  if (t_root_host) check(dsh_malloc(c, (int64_t)sizeof(double) * nb, 0, &troot_dev), "adaptive t_root");
  if (root_idx_host) check(dsh_malloc(c, (int64_t)sizeof(int32_t) * nb, 0, &ridx_dev), "adaptive root_idx");
  if (ncols_host) check(dsh_malloc(c, (int64_t)sizeof(int32_t) * nb, 0, &ncols_dev), "adaptive ncols");
  int rc;
  const int64_t npar = s->problem.eqn->nparams();
  if (s->problem.sens) {
    check(dsh_malloc(c, (int64_t)sizeof(double) * nt * npar * n * nb, 0, &sens_dev), "adaptive sens out");
    if (sorted) check(dsh_malloc(c, (int64_t)sizeof(double) * nt * npar * n * nb, 0, &sens_sorted), "adaptive sens out (sorted)");
    const std::vector<double> sa = s->problem.sens_error_control ? s->problem.sens_atol.clone_as_vec() : std::vector<double>();
    if (wave_member && method != 0)
      rc = dsh_sdirk_solve_wave_member_sens(c, model, size, method, nb, params_dev, s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o, t_eval, nt,
                                            s->problem.sens_rtol, sa.data(), (int64_t)sa.size(), out, (double*)(sorted ? sens_sorted : sens_dev), (int32_t*)stats_dev,
                                            (int32_t*)status_dev, totals);
    else if (wave_member)
      rc = dsh_bdf_solve_wave_member_sens(c, model, size, nb, params_dev, s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o, t_eval, nt,
                                          s->problem.sens_rtol, sa.data(), (int64_t)sa.size(), out, (double*)(sorted ? sens_sorted : sens_dev), (int32_t*)stats_dev,
                                          (int32_t*)status_dev, totals);
    else if (method == 0)
      rc = dsh_bdf_solve_adaptive_sens(c, model, size, nb, params_dev, s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o, t_eval, nt,
                                       s->problem.sens_rtol, sa.data(), (int64_t)sa.size(), out, (double*)(sorted ? sens_sorted : sens_dev), (int32_t*)stats_dev,
                                       (int32_t*)status_dev, totals);
    else
      rc = dsh_sdirk_solve_resident_sens(c, method, model, size, nb, params_dev, s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o, t_eval, nt,
                                         s->problem.sens_rtol, sa.data(), (int64_t)sa.size(), out, (double*)(sorted ? sens_sorted : sens_dev), (int32_t*)stats_dev,
                                         (int32_t*)status_dev, totals);
    if (rc == DSH_OK && sorted) rc = dsh_permute_members(c, nt * npar * n, nb, 8, sens_sorted, (const int32_t*)s->inv_dev, sens_dev);
    // [col][parameter][state][b] on the device -> [parameter][col][b][state] on the host (one y-shaped array per parameter: solve_dense_sensitivities' Vec<M>)
    for (int64_t k = 0; k < nt && rc == DSH_OK; ++k)
      for (int64_t q = 0; q < npar && rc == DSH_OK; ++q)
        rc = dsh_vec_download(c, n, nb, (const double*)sens_dev + (size_t)((k * npar + q) * n * nb), sens_host + (size_t)((q * nt + k) * n * nb));
  } else
  if (wave_member && method != 0)
    rc = dsh_sdirk_solve_wave_member(c, model, size, method, nb, params_dev, s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o,
                                     t_eval, nt, out, (int32_t*)stats_dev, (int32_t*)status_dev, (double*)troot_dev, (int32_t*)ridx_dev, (int32_t*)ncols_dev, totals);
  else if (wave_member)
    rc = dsh_bdf_solve_wave_member(c, model, size, nb, params_dev, s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o,
                                   t_eval, nt, out, (int32_t*)stats_dev, (int32_t*)status_dev, (double*)troot_dev, (int32_t*)ridx_dev, (int32_t*)ncols_dev, totals);
  else if (method == 0)
    rc = dsh_bdf_solve_adaptive(c, model, size, nb, params_dev, s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o,
                                t_eval, nt, out, (int32_t*)stats_dev, (int32_t*)status_dev, (double*)troot_dev, (int32_t*)ridx_dev, (int32_t*)ncols_dev, totals);
  else
    rc = dsh_sdirk_solve_resident(c, method, model, size, nb, params_dev, s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0,
                                  s->problem.h0, &o, t_eval, nt, out, (int32_t*)stats_dev, (int32_t*)status_dev, (double*)troot_dev, (int32_t*)ridx_dev,
                                  (int32_t*)ncols_dev, totals);
  if (sorted) {  // back into the caller's order: dst[., member] = src[., position of member]
    const int32_t* inv = (const int32_t*)s->inv_dev;
    auto back = [&](void*& buf, int64_t rows, int bytes) {
      if (rc != DSH_OK || !buf) return;
      void* t = nullptr;
      rc = dsh_malloc(c, (int64_t)bytes * rows * nb, 0, &t);
      if (rc == DSH_OK) rc = dsh_permute_members(c, rows, nb, bytes, buf, inv, t);
      if (rc == DSH_OK) { dsh_free(c, buf); buf = t; } else if (t) dsh_free(c, t);
    };
    if (rc == DSH_OK) rc = dsh_permute_members(c, nt * n, nb, 8, out, inv, user_out);
    out = user_out;
    back(stats_dev, 5, 4); back(status_dev, 1, 4); back(troot_dev, 1, 8); back(ridx_dev, 1, 4); back(ncols_dev, 1, 4);
  }
  if (rc == DSH_OK && y_host)
    for (int64_t k = 0; k < nt && rc == DSH_OK; ++k) rc = dsh_vec_download(c, n, nb, out + (size_t)(k * n * nb), y_host + (size_t)(k * n * nb));
  if (rc == DSH_OK && stats_host) rc = dsh_d2h(c, stats_host, stats_dev, (int64_t)sizeof(int32_t) * 5 * nb);
  const bool want_status = !lazy || totals[5] != 0, want_ridx = !lazy || s->problem.eqn->nroots() > 0;
  if (rc == DSH_OK && status_host) { if (want_status) rc = dsh_d2h(c, status_host, status_dev, (int64_t)sizeof(int32_t) * nb); }  // lazy and no failure: left untouched (the caller reads totals[5] first)
  if (rc == DSH_OK && t_root_host) rc = dsh_d2h(c, t_root_host, troot_dev, (int64_t)sizeof(double) * nb);
  if (rc == DSH_OK && root_idx_host) { if (want_ridx) rc = dsh_d2h(c, root_idx_host, ridx_dev, (int64_t)sizeof(int32_t) * nb); else std::fill(root_idx_host, root_idx_host + nb, -1); }
  if (rc == DSH_OK && ncols_host) rc = dsh_d2h(c, ncols_host, ncols_dev, (int64_t)sizeof(int32_t) * nb);
  check(rc, "solve_dense_adaptive");
}
}  // namespace

extern "C" {

const char* dshs_last_error(void) { return g_err.c_str(); }

static int diffsl_generate_impl(const char* code, int target, int model_index, char** source_out, int64_t* dims, double* defaults_out, int64_t defaults_cap) {
  return guarded([&] {
    if (!code || !source_out) throw LaError(DSH_E_INVALID, "dshs_diffsl_generate: null argument");
    if (target < 0 || target > 2) throw LaError(DSH_E_INVALID, "dshs_diffsl_generate: unknown target");
    if (model_index < 0) throw LaError(DSH_E_INVALID, "dshs_diffsl_generate: the model index is unsigned");
    diffsl::Compiled c = diffsl::compile(code, model_index);
    const diffsl::Target t = target == DSHS_DIFFSL_HIP_STATIC ? diffsl::Target::HipStatic : target == DSHS_DIFFSL_HIP_DYNAMIC ? diffsl::Target::HipDynamic : diffsl::Target::HostC;
    std::string src = diffsl::generate(c, t);
    char* out = (char*)std::malloc(src.size() + 1);
    if (!out) throw LaError(DSH_E_INVALID, "out of memory");
    std::memcpy(out, src.c_str(), src.size() + 1);
    *source_out = out;
    if (dims) { dims[0] = c.n; dims[1] = c.np; dims[2] = c.nroots; dims[3] = c.nout; dims[4] = c.has_mass ? 1 : 0; dims[5] = c.dummy_param ? 1 : 0; dims[6] = c.jac_kl; dims[7] = c.jac_ku; dims[8] = c.mass_kl; dims[9] = c.mass_ku; }
    if (defaults_out) for (int64_t k = 0; k < defaults_cap && k < (int64_t)c.input_defaults.size(); ++k) defaults_out[k] = c.input_defaults[k];
    return 0;
  });
}
int dshs_diffsl_generate(const char* code, int target, char** source_out, int64_t* dims, double* defaults_out, int64_t defaults_cap) {
  // consumes the one-shot index armed by dshs_diffsl_set_model_index (0 when none was armed): a later call never inherits it
  return diffsl_generate_impl(code, target, diffsl::take_pending_model_index(), source_out, dims, defaults_out, defaults_cap);
}
int dshs_diffsl_generate_indexed(const char* code, int target, int model_index, char** source_out, int64_t* dims, double* defaults_out, int64_t defaults_cap) {
  return diffsl_generate_impl(code, target, model_index, source_out, dims, defaults_out, defaults_cap);
}
void dshs_free_string(char* s) { std::free(s); }
int dshs_diffsl_set_model_index(int model_index) {
  if (model_index < 0) { g_err = "dshs_diffsl_set_model_index: the model index is unsigned"; return DSH_E_INVALID; }
  diffsl::set_model_index(model_index);
  return 0;
}

void dshs_default_options(dshs_options* o) {
  o->max_nonlinear_solver_iterations = 10;
  o->max_error_test_failures = 40;
  o->max_nonlinear_solver_failures = 50;
  o->nonlinear_solver_tolerance = 0.2;
  o->min_timestep = 1e-13;
  o->update_jacobian_after_steps = 20;
  o->update_rhs_jacobian_after_steps = 50;
  o->threshold_to_update_jacobian = 0.3;
  o->threshold_to_update_rhs_jacobian = 0.2;
  o->ic_use_linesearch = 1;
  o->use_fused_kernels = 1;
  o->block_threads = 0;
  o->ic_max_linesearch_iterations = 10;
  o->ic_max_linear_solver_setups = 4;
  o->ic_max_newton_iterations = 10;
  o->ic_step_reduction_factor = 0.5;
  o->ic_armijo_constant = 1e-4;
}

int dshs_create(int device, void* stream, int model, int64_t model_size, int64_t nbatch, const double* params, int64_t nparams_total, double rtol,
                const double* atol, int64_t natol, double t0, double h0, int method, const dshs_options* opts, dshs_solver** out) {
  return dshs_create_sens(device, stream, model, model_size, nbatch, params, nparams_total, rtol, atol, natol, t0, h0, method, opts, /*sens=*/0, 0.0, nullptr, 0, out);
}

int dshs_create_sens(int device, void* stream, int model, int64_t model_size, int64_t nbatch, const double* params, int64_t nparams_total, double rtol,
                     const double* atol, int64_t natol, double t0, double h0, int method, const dshs_options* opts, int sens, double sens_rtol,
                     const double* sens_atol, int64_t nsens_atol, dshs_solver** out) {
  return guarded([&]() {
    if (!out) throw LaError(DSH_E_INVALID, "out is null");
    dshs_options o;
    dshs_default_options(&o);
    if (opts) o = *opts;
    auto s = std::make_unique<dshs_solver>();
    s->ctx = HipContext(device, stream, nbatch);
    if (o.block_threads > 0) check(dsh_ctx_set_block(s->ctx.raw(), o.block_threads), "dsh_ctx_set_block");
    OdeSolverOptions oo;
    oo.max_nonlinear_solver_iterations = o.max_nonlinear_solver_iterations;
    oo.max_error_test_failures = o.max_error_test_failures;
    oo.max_nonlinear_solver_failures = o.max_nonlinear_solver_failures;
    oo.nonlinear_solver_tolerance = o.nonlinear_solver_tolerance;
    oo.min_timestep = o.min_timestep;
    oo.update_jacobian_after_steps = o.update_jacobian_after_steps;
    oo.update_rhs_jacobian_after_steps = o.update_rhs_jacobian_after_steps;
    oo.threshold_to_update_jacobian = o.threshold_to_update_jacobian;
    oo.threshold_to_update_rhs_jacobian = o.threshold_to_update_rhs_jacobian;
    InitialConditionSolverOptions ic;
    ic.use_linesearch = o.ic_use_linesearch != 0;
    ic.max_linesearch_iterations = o.ic_max_linesearch_iterations;
    ic.max_linear_solver_setups = o.ic_max_linear_solver_setups;
    ic.max_newton_iterations = o.ic_max_newton_iterations;
    ic.step_reduction_factor = o.ic_step_reduction_factor;
    ic.armijo_constant = o.ic_armijo_constant;
    std::vector<double> p(params, params + nparams_total), a(atol, atol + natol);
    OdeBuilder builder;
    builder.t0(t0).h0(h0).rtol(rtol).atol(a).context(s->ctx).use_fused_kernels(o.use_fused_kernels != 0).ode_options(oo).ic_options(ic);
    if (sens) {
      builder.sensitivities(true);
      if (nsens_atol > 0) builder.sens_tolerances(sens_rtol, std::vector<double>(sens_atol, sens_atol + nsens_atol));
    }
    s->problem = builder.build_model(model, model_size, p);
    s->method = method;
    s->make_solver();
    *out = s.release();
    return 0;
  });
}

void dshs_destroy(dshs_solver* s) { delete s; }

int dshs_reset(dshs_solver* s) {
  return guarded([&]() { s->resident_roots_valid = false; s->make_solver(); return 0; });
}
dsh_ctx* dshs_context(dshs_solver* s) { return s ? s->ctx.raw() : nullptr; }
int dshs_set_linear_solve_mode(dshs_solver* s, int mode) { return dsh_ctx_set_solve_mode(s->ctx.raw(), mode); }
int dshs_set_kernel_timing(dshs_solver* s, int enable) {
  // timed launches bracket the stand-alone Newton kernel: keep the accept launch separate while timing is on
  s->kernel_timing = enable != 0;
  if (s->bdf && s->kernel_timing) s->bdf->set_fuse_accept(false);
  return dsh_ctx_set_timing(s->ctx.raw(), enable);
}
int dshs_set_kernel_timing_target(dshs_solver* s, int target) { return dsh_ctx_set_timing_target(s->ctx.raw(), target); }
int dshs_get_kernel_timing(dshs_solver* s, int64_t* launches, double* total_ms) { return dsh_ctx_get_timing(s->ctx.raw(), launches, total_ms); }
int dshs_get_kernel_timing_overhead(dshs_solver* s, double* empty_bracket_ms, double* device_clock_total_ms) {
  return dsh_ctx_get_timing_overhead(s->ctx.raw(), empty_bracket_ms, device_clock_total_ms);
}

int64_t dshs_nstates(const dshs_solver* s) { return s->problem.eqn->nstates(); }
int64_t dshs_nbatch(const dshs_solver* s) { return s->ctx.nbatch(); }
int dshs_is_fused(const dshs_solver* s) { return s->fused ? 1 : 0; }

int dshs_step(dshs_solver* s, int* stop_reason) {
  return guarded([&]() {
    s->resident_roots_valid = false;  // the host solver moves again: dshs_root_info reports ITS events from here on
    OdeSolverStopReason r = s->solver->step();
    if (stop_reason) *stop_reason = (int)r;
    return 0;
  });
}
int dshs_set_stop_time(dshs_solver* s, double tstop) {
  return guarded([&]() { s->resident_roots_valid = false; s->solver->set_stop_time(tstop); return 0; });
}
int dshs_interpolate(dshs_solver* s, double t, double* y_host) {
  return guarded([&]() { HipVec y = s->solver->interpolate(t); download(y, y_host); return 0; });
}
int dshs_get_state(dshs_solver* s, double* t, double* h, int* order, double* y_host, double* dy_host) {
  return guarded([&]() {
    if (t) *t = s->solver->t();
    if (h) *h = s->solver->h();
    if (order) *order = s->solver->order();
    if (y_host) download(s->solver->y(), y_host);
    if (dy_host) download(s->solver->dy(), dy_host);
    return 0;
  });
}
int64_t dshs_nparams(const dshs_solver* s) { return s->problem.eqn->nparams(); }
int dshs_interpolate_sens(dshs_solver* s, double t, double* s_host) {
  return guarded([&]() {
    if ((!s->bdf && !s->sdirk) || !s->problem.sens) throw LaError(DSH_E_INVALID, "the solver was not created with forward sensitivities (dshs_create_sens)");
    const size_t len = (size_t)(s->problem.eqn->nstates() * s->ctx.nbatch());
    const std::vector<HipVec>& cur = s->bdf ? s->bdf->sens() : s->sdirk->sens();
    if (t != t) {  // NaN: state.s, the sensitivities at the current time
      for (size_t j = 0; j < cur.size(); ++j) download(cur[j], s_host + j * len);
      return 0;
    }
    std::vector<HipVec> out;
    if (s->bdf) s->bdf->interpolate_sens_inplace(t, out);
    else s->sdirk->interpolate_sens_inplace(t, out);
    for (size_t j = 0; j < out.size(); ++j) download(out[j], s_host + j * len);
    return 0;
  });
}
int dshs_root_info(dshs_solver* s, double* t_root, int* root_index) {
  if (s->resident_roots_valid) {  // the last solve_dense ran device-resident: the host solver was not stepped; report the EARLIEST member event
    // earliest = smallest distance from t0 along the direction of integration (sign of h0), not smallest |t|: t0 may be non-zero and times negative
    const double t0 = s->problem.t0, dir = s->problem.h0 < 0.0 ? -1.0 : 1.0;
    double best = 0.0;
    int idx = -1;
    for (size_t b = 0; b < s->member_troot.size(); ++b) {
      const double tr = s->member_troot[b];
      if (s->scratch_ridx[b] >= 0 && (idx < 0 || dir * (tr - t0) < dir * (best - t0))) { best = tr; idx = s->scratch_ridx[b]; }
    }
    *t_root = best;
    *root_index = idx;
    return 0;
  }
  *t_root = s->solver->root_time;
  *root_index = s->solver->root_index;
  return 0;
}
int dshs_bdf_get_diff(dshs_solver* s, double* diff_host) {
  return guarded([&]() {
    if (!s->bdf) throw LaError(DSH_E_INVALID, "not a BDF solver");
    std::vector<double> tmp = s->bdf->diff().clone_as_vec();
    std::memcpy(diff_host, tmp.data(), tmp.size() * sizeof(double));
    return 0;
  });
}
int dshs_stats(dshs_solver* s, int64_t* out) {
  const OdeSolverStatistics& st = s->solver->get_statistics();
  out[0] = st.number_of_linear_solver_setups; out[1] = st.number_of_steps; out[2] = st.number_of_error_test_failures;
  out[3] = st.number_of_nonlinear_solver_iterations; out[4] = st.number_of_nonlinear_solver_fails;
  out[5] = st.number_of_linear_solver_setups_from_checkpoint; out[6] = st.number_of_linear_solver_setups_from_first_convergence_fail;
  out[7] = st.number_of_linear_solver_setups_from_second_convergence_fail; out[8] = st.number_of_linear_solver_setups_from_error_test_fail;
  out[9] = st.number_of_linear_solver_setups_from_step_success;
  const OpStatistics& o = s->problem.eqn->rhs_statistics;
  out[10] = o.number_of_calls; out[11] = o.number_of_jac_muls; out[12] = o.number_of_matrix_evals;
  return 0;
}

int dshs_solve_to_points(dshs_solver* s, const double* t_points, int64_t npoints, double* y_host) {
  return guarded([&]() {
    s->resident_roots_valid = false;
    const size_t len = (size_t)(s->problem.eqn->nstates() * s->ctx.nbatch());
    for (int64_t k = 0; k < npoints; ++k) {
      while (std::fabs(s->solver->t()) < std::fabs(t_points[k])) {
        if (s->solver->step() == OdeSolverStopReason::RootFound) {
          HipVec y = s->solver->interpolate(s->solver->root_time);
          download(y, y_host + (size_t)k * len);
          return 1;
        }
      }
      HipVec y = s->solver->interpolate(t_points[k]);
      download(y, y_host + (size_t)k * len);
    }
    return 0;
  });
}

int dshs_solve(dshs_solver* s, double final_time, int keep_trajectory, double* y_final_host, int64_t* ncols, int* stop_reason) {
  return guarded([&]() {
    s->resident_roots_valid = false;
    OdeSolverStopReason r;
    int64_t cols = 1;
    if (keep_trajectory) {
      r = s->solver->solve(final_time, s->traj_y, s->traj_t);
      cols = (int64_t)s->traj_t.size();
    } else {
      s->solver->set_stop_time(final_time);
      while (true) {
        r = s->solver->step();
        cols++;
        if (r == OdeSolverStopReason::InternalTimestep) continue;
        if (r == OdeSolverStopReason::RootFound) s->solver->state_mut_back(s->solver->root_time);
        break;
      }
    }
    if (y_final_host) download(s->solver->y(), y_final_host);
    if (ncols) *ncols = cols;
    if (stop_reason) *stop_reason = (int)r;
    return 0;
  });
}
int dshs_trajectory(dshs_solver* s, double* t_host, double* y_host) {
  return guarded([&]() {
    if (s->traj_t.empty()) throw LaError(DSH_E_INVALID, "no trajectory stored (call dshs_solve with keep_trajectory=1)");
    std::memcpy(t_host, s->traj_t.data(), s->traj_t.size() * sizeof(double));
    // device layout [col][row][b] -> host [col][b][row]
    const int64_t n = s->traj_y.nrows(), nb = s->traj_y.nb();
    for (int64_t c = 0; c < s->traj_y.ncols(); ++c)
      check(dsh_vec_download(s->ctx.raw(), n, nb, s->traj_y.column(c).p, y_host + (size_t)(c * n * nb)), "trajectory download");
    return 0;
  });
}

int dshs_solve_dense(dshs_solver* s, const double* t_eval, int64_t nt, double* y_host, double* y_dev, int* stop_reason) {
  return guarded([&]() {
    const int mode = resolve_mode(s);
    s->last_mode = mode;
    if (mode != DSHS_ENSEMBLE_LOCKSTEP) {
      // OdeSolverMethod::solve_dense (method.rs:467-520) of the whole ensemble in one launch: the state never leaves the chip.
      const int64_t nb = s->ctx.nbatch();
      // Per-member bookkeeping on the host only when there is something to read: with no root functions and no failed member (the counters say so)
      // nothing of size nbatch is initialised, filled or scanned here — at 100 000 members that was ~0.1 ms per solve, 4 % of the headline solve.
      std::vector<int32_t>&status = s->scratch_status, &ridx = s->scratch_ridx;
      const bool has_roots = s->problem.eqn->nroots() > 0;
      status.resize((size_t)nb);
      if (has_roots) { ridx.resize((size_t)nb); s->member_troot.assign((size_t)nb, std::nan("")); }
      else { ridx.clear(); s->member_troot.clear(); }
      s->resident_roots_valid = false;
      // arithmetic: the fast build where it exists (BDF of a static model, no sensitivities: dsh_adaptive.hip falls back to the exact kernel otherwise), unless the
      // process asked for the exact kernel (dshs_set_resident_arithmetic / DSH_RESIDENT_ARITH=exact — what the bitwise parity tier pins)
      const int arith = resident_arith_flag() == DSHS_ARITH_FAST && !s->problem.sens ? 2 : 1;  // (2 where no fast build exists: the exact kernel with the portable pow, as 1)
      s->last_arith = arith;
      run_resident(s, t_eval, nt, mode, arith, y_host, y_dev, nullptr, status.data(), has_roots ? s->member_troot.data() : nullptr, has_roots ? ridx.data() : nullptr, nullptr,
                   s->last_totals, /*lazy=*/true);  // lazy: status is downloaded only if a member failed (else left untouched)
      s->resident_roots_valid = true;
      int64_t failed = 0, rooted = 0;
      int first_bad = 0;
      if (s->last_totals[5] != 0)
        for (int64_t b = 0; b < nb; ++b)
          if (status[(size_t)b] != 0) { if (!failed) first_bad = status[(size_t)b]; ++failed; }
      if (has_roots)
        for (int64_t b = 0; b < nb; ++b)
          if (ridx[(size_t)b] >= 0) ++rooted;
      if (failed) {
        char buf[256];
        std::snprintf(buf, sizeof buf, "solve_dense: %lld of %lld ensemble members failed (first status %d); dshs_solve_dense_adaptive returns the per-member status",
                      (long long)failed, (long long)nb, first_bad);
        if (first_bad >= 1 && first_bad <= (int)OdeSolverError::LinearSolveFailed) throw DiffsolError((OdeSolverError)first_bad, buf);
        throw LaError(first_bad == 20 ? DSH_E_BATCH_MISMATCH : DSH_E_INVALID, buf);
      }
      // hybrid models go on after their events: the solve ends at the last save point
      if (stop_reason) *stop_reason = (has_roots && rooted == nb && !s->problem.eqn->has_reset()) ? DSHS_STOP_ROOT_FOUND : DSHS_STOP_TSTOP_REACHED;
      return 0;
    }
    s->resident_roots_valid = false;
    std::vector<double> te(t_eval, t_eval + nt);
    HipMat ret;
    OdeSolverStopReason r = s->solver->solve_dense(te, ret);
    const int64_t n = ret.nrows(), nb = ret.nb();
    if (y_dev) check(dsh_d2d(s->ctx.raw(), y_dev, ret.ptr(), (int64_t)sizeof(double) * n * nb * ret.ncols()), "solve_dense d2d");
    if (y_host)
      for (int64_t c = 0; c < ret.ncols(); ++c) check(dsh_vec_download(s->ctx.raw(), n, nb, ret.column(c).p, y_host + (size_t)(c * n * nb)), "solve_dense download");
    s->ctx.sync();
    if (stop_reason) *stop_reason = (int)r;
    const OdeSolverStatistics& st = s->solver->get_statistics();
    s->last_totals[0] = st.number_of_steps * nb; s->last_totals[1] = st.number_of_nonlinear_solver_iterations * nb;
    s->last_totals[2] = st.number_of_linear_solver_setups * nb; s->last_totals[3] = st.number_of_error_test_failures * nb;
    s->last_totals[4] = st.number_of_nonlinear_solver_fails * nb; s->last_totals[5] = 0;
    return 0;
  });
}

// Arithmetic of the device-resident integrators that dshs_solve_dense launches in its default (non-lock-step) modes: DSHS_ARITH_FAST (default since round 6) = the
// builds of dsh_adaptive_fast.hip (BDF) / dsh_sdirk_fast.hip (TR-BDF2, ESDIRK34) where one exists (static models with n <= 4, no forward sensitivities: contracted
// multiply-adds, reciprocal-math division), else the exact kernel; DSHS_ARITH_EXACT = always the exact kernel (bit-identical to the CPU oracle).  Environment DSH_RESIDENT_ARITH=exact|fast sets the
// process default; the explicit entry points (dshs_solve_dense_adaptive, dshs_solve_adaptive, ...) take their arithmetic as an argument and ignore this.
int dshs_set_resident_arithmetic(int mode) {
  if (mode != DSHS_ARITH_EXACT && mode != DSHS_ARITH_FAST) return -1;
  resident_arith_flag() = mode;
  return 0;
}
int dshs_get_resident_arithmetic(void) { return resident_arith_flag(); }

int dshs_set_deterministic_pow(int on) {
  det_pow_flag() = on != 0;
  return 0;
}
int dshs_set_ensemble_mode(dshs_solver* s, int mode) {
  return guarded([&]() {
    if (mode != DSHS_ENSEMBLE_AUTO && mode != DSHS_ENSEMBLE_LOCKSTEP && mode != DSHS_ENSEMBLE_PER_MEMBER && mode != DSHS_ENSEMBLE_WAVEFRONT)
      throw LaError(DSH_E_INVALID, "dshs_set_ensemble_mode: mode must be DSHS_ENSEMBLE_AUTO, _LOCKSTEP, _PER_MEMBER or _WAVEFRONT");
    if ((mode == DSHS_ENSEMBLE_PER_MEMBER || mode == DSHS_ENSEMBLE_WAVEFRONT) && !pick_resident(s, mode).ok)
      throw LaError(DSH_E_UNSUPPORTED, "dshs_set_ensemble_mode: no device-resident kernel for this model/method in that mode");
    s->ensemble_mode = mode;
    return 0;
  });
}
int dshs_get_ensemble_mode(const dshs_solver* s, int* requested, int* resolved) {
  if (requested) *requested = s->ensemble_mode;
  if (resolved) *resolved = resolve_mode(s);
  return 0;
}
int dshs_last_solve_info(const dshs_solver* s, int* mode, int64_t* totals) {
  if (mode) *mode = s->last_mode;
  if (totals) for (int k = 0; k < 6; ++k) totals[k] = s->last_totals[k];
  return 0;
}

// Device-resident integration, whole ensemble solve in one launch (dsh_bdf_solve_adaptive / dsh_sdirk_solve_resident); the problem (model,
// parameters, tolerances, options, t0, h0) is the solver's OdeSolverProblem, nothing of the lock-step solver state is touched.
int dshs_solve_dense_adaptive(dshs_solver* s, const double* t_eval, int64_t nt, int group, int deterministic_pow, double* y_host, double* y_dev, int32_t* stats_host,
                              int32_t* status_host, double* t_root_host, int32_t* root_idx_host, int32_t* ncols_host, int64_t* totals) {
  return guarded([&]() {
    int64_t tot[6];
    s->resident_roots_valid = false;  // this call hands the per-member events to the caller's arrays; dshs_root_info falls back to the host solver's
    run_resident(s, t_eval, nt, group, deterministic_pow, y_host, y_dev, stats_host, status_host, t_root_host, root_idx_host, ncols_host, tot);
    for (int k = 0; k < 6; ++k) { s->last_totals[k] = tot[k]; if (totals) totals[k] = tot[k]; }
    s->last_mode = group;
    return 0;
  });
}

// OdeSolverMethod::solve (method.rs:227-258) on the device-resident BDF: every member's accepted steps out of one launch (dsh_bdf_solve_adaptive_steps), in the caller's
// member order; the problem is the solver's OdeSolverProblem, nothing of the lock-step solver state is touched.
int dshs_solve_adaptive(dshs_solver* s, double t_final, int64_t max_cols, int group, int deterministic_pow, double* y_host, double* t_host, int32_t* ncols_host,
                        int32_t* stats_host, int32_t* status_host, double* t_root_host, int32_t* root_idx_host, int64_t* totals) {
  return guarded([&]() {
    if (!y_host || !t_host || !ncols_host || max_cols < 2) throw LaError(DSH_E_INVALID, "dshs_solve_adaptive: y_host, t_host, ncols_host and max_cols >= 2 are needed");
    if (s->problem.sens) throw LaError(DSH_E_UNSUPPORTED, "dshs_solve_adaptive: without forward sensitivities (dshs_solve walks the host-driven path for the rest)");
    const ResidentPick pk = pick_resident(s, group);
    const bool sdirk = s->method != DSHS_METHOD_BDF;  // TR-BDF2 / ESDIRK34
    if (!pk.ok || (!sdirk && !pk.wave_member && !dsh_model_has_adaptive_steps(pk.model, pk.size)))
      throw LaError(DSH_E_UNSUPPORTED, "dshs_solve_adaptive: no device-resident integrator that writes every step for this model and method (register-resident static models, banded lane-per-member forms, "
                                       "wavefront / workgroup per member); dshs_solve returns every step of the host-driven lock-step solver");
    const int64_t n = s->problem.eqn->nstates(), nb = s->ctx.nbatch();
    const dsh_adaptive_options o = adaptive_options_of(s, group, deterministic_pow);
    dsh_ctx* c = s->ctx.raw();
    void *y_dev = nullptr, *t_dev = nullptr, *stats_dev = nullptr, *status_dev = nullptr, *troot_dev = nullptr, *ridx_dev = nullptr, *ncols_dev = nullptr;
    struct Release {
      dsh_ctx* c; std::vector<void**> bufs;
      ~Release() { for (void** q : bufs) if (*q) dsh_free(c, *q); }
    } release{c, {&y_dev, &t_dev, &stats_dev, &status_dev, &troot_dev, &ridx_dev, &ncols_dev}};
    check(dsh_malloc(c, (int64_t)sizeof(double) * max_cols * n * nb, 0, &y_dev), "solve_adaptive out");
    check(dsh_malloc(c, (int64_t)sizeof(double) * max_cols * nb, 0, &t_dev), "solve_adaptive times");
    check(dsh_malloc(c, (int64_t)sizeof(int32_t) * nb, 0, &ncols_dev), "solve_adaptive ncols");
    if (stats_host) check(dsh_malloc(c, (int64_t)sizeof(int32_t) * 5 * nb, 0, &stats_dev), "solve_adaptive stats");
    if (status_host) check(dsh_malloc(c, (int64_t)sizeof(int32_t) * nb, 0, &status_dev), "solve_adaptive status");
    if (t_root_host) check(dsh_malloc(c, (int64_t)sizeof(double) * nb, 0, &troot_dev), "solve_adaptive t_root");
    if (root_idx_host) check(dsh_malloc(c, (int64_t)sizeof(int32_t) * nb, 0, &ridx_dev), "solve_adaptive root_idx");
    int64_t tot[6] = {0};
    if (sdirk && pk.wave_member)
      check(dsh_sdirk_solve_wave_member_steps(c, pk.model, pk.size, pk.method, nb, s->problem.eqn->params().ptr(), s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o,
                                              t_final, max_cols, (double*)y_dev, (double*)t_dev, (int32_t*)stats_dev, (int32_t*)status_dev, (double*)troot_dev, (int32_t*)ridx_dev,
                                              (int32_t*)ncols_dev, tot), "dsh_sdirk_solve_wave_member_steps");
    else
    if (sdirk)
      check(dsh_sdirk_solve_resident_steps(c, pk.method, pk.model, pk.size, nb, s->problem.eqn->params().ptr(), s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o,
                                           t_final, max_cols, (double*)y_dev, (double*)t_dev, (int32_t*)stats_dev, (int32_t*)status_dev, (double*)troot_dev, (int32_t*)ridx_dev,
                                           (int32_t*)ncols_dev, tot), "dsh_sdirk_solve_resident_steps");
    else
    if (pk.wave_member)  // dense run-time-sized models: one wavefront / workgroup per member (always per-member control)
      check(dsh_bdf_solve_wave_member_steps(c, pk.model, pk.size, nb, s->problem.eqn->params().ptr(), s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o, t_final,
                                            max_cols, (double*)y_dev, (double*)t_dev, (int32_t*)stats_dev, (int32_t*)status_dev, (double*)troot_dev, (int32_t*)ridx_dev,
                                            (int32_t*)ncols_dev, tot), "dsh_bdf_solve_wave_member_steps");
    else
    check(dsh_bdf_solve_adaptive_steps(c, pk.model, pk.size, nb, s->problem.eqn->params().ptr(), s->problem.atol.ptr(), 1, s->problem.rtol, s->problem.t0, s->problem.h0, &o, t_final,
                                       max_cols, (double*)y_dev, (double*)t_dev, (int32_t*)stats_dev, (int32_t*)status_dev, (double*)troot_dev, (int32_t*)ridx_dev,
                                       (int32_t*)ncols_dev, tot), "dsh_bdf_solve_adaptive_steps");
    check(dsh_d2h(c, ncols_host, ncols_dev, (int64_t)sizeof(int32_t) * nb), "solve_adaptive ncols");
    int32_t most = 0;
    for (int64_t b = 0; b < nb; ++b) most = std::max(most, ncols_host[b]);
    const int64_t used = std::min<int64_t>(most, max_cols);  // columns nobody wrote stay untouched on the host
    for (int64_t k = 0; k < used; ++k) check(dsh_vec_download(c, n, nb, (const double*)y_dev + (size_t)(k * n * nb), y_host + (size_t)(k * n * nb)), "solve_adaptive download");
    check(dsh_d2h(c, t_host, t_dev, (int64_t)sizeof(double) * used * nb), "solve_adaptive times");
    if (stats_host) check(dsh_d2h(c, stats_host, stats_dev, (int64_t)sizeof(int32_t) * 5 * nb), "solve_adaptive stats");
    if (status_host) check(dsh_d2h(c, status_host, status_dev, (int64_t)sizeof(int32_t) * nb), "solve_adaptive status");
    if (t_root_host) check(dsh_d2h(c, t_root_host, troot_dev, (int64_t)sizeof(double) * nb), "solve_adaptive t_root");
    if (root_idx_host) check(dsh_d2h(c, root_idx_host, ridx_dev, (int64_t)sizeof(int32_t) * nb), "solve_adaptive root_idx");
    s->resident_roots_valid = false;
    for (int k = 0; k < 6; ++k) { s->last_totals[k] = tot[k]; if (totals) totals[k] = tot[k]; }
    s->last_mode = group;
    return 0;
  });
}

int dshs_solve_dense_adaptive_sens(dshs_solver* s, const double* t_eval, int64_t nt, int group, int deterministic_pow, double* y_host, double* sens_host,
                                   int32_t* stats_host, int32_t* status_host, int64_t* totals) {
  return guarded([&]() {
    if (!s->problem.sens) throw LaError(DSH_E_INVALID, "the solver was not created with forward sensitivities (dshs_create_sens)");
    if (!sens_host) throw LaError(DSH_E_INVALID, "dshs_solve_dense_adaptive_sens: sens_host is null");
    int64_t tot[6];
    s->resident_roots_valid = false;
    run_resident(s, t_eval, nt, group, deterministic_pow, y_host, nullptr, stats_host, status_host, nullptr, nullptr, nullptr, tot, false, sens_host);
    for (int k = 0; k < 6; ++k) { s->last_totals[k] = tot[k]; if (totals) totals[k] = tot[k]; }
    s->last_mode = group;
    return 0;
  });
}

}  // extern "C"
