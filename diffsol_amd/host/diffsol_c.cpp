// include/diffsol_c_hip.h — the reference's runtime-typed C API (crates/diffsol-c/src/*_c.rs) over the HIP backend.
// A thin layer: DiffSL text -> dshs_diffsl_generate + dsh_model_compile; every solve builds a dshs_solver (OdeBuilder ... .bdf()/.tr_bdf2()/.esdirk34())
// for the parameter sets it was handed, like OdeWrapper::solve does in the reference (crates/diffsol-c/src/ode.rs:457-499, solve.rs).
#include "../../include/diffsol_c_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../../include/diffsol_hip.h"
#include "../../include/diffsol_hip_solver.h"
#include "hip_la.hpp"
#include "diffsl.hpp"
#include <optional>

using namespace diffsol_hip;

namespace {

struct LastError {
  bool set = false;
  std::string message, file;
  uint32_t line = 0;
};
thread_local LastError g_last;

int32_t record(const std::string& msg, const char* file, uint32_t line, int32_t code) {
  g_last.set = true; g_last.message = msg; g_last.file = file; g_last.line = line;
  return code;
}
#define C_ERROR(msg) record((msg), __FILE__, __LINE__, DIFFSOL_ERR)
#define C_INVALID_ARG(msg) record((msg), __FILE__, __LINE__, DIFFSOL_BAD_ARG)

struct Settings {  // shared by the ode and the option handles it hands out (the reference shares them behind Arc<Mutex<..>>)
  dshs_options o;
};

}  // namespace

struct diffsol_host_array {
  std::vector<double> data;
  std::vector<size_t> shape, strides;  // strides in bytes
};
struct diffsol_ode_solver_options { std::shared_ptr<Settings> s; };
struct diffsol_ic_solver_options { std::shared_ptr<Settings> s; };
struct diffsol_ode_wrapper {
  std::string code;
  int model = -1, lane_model = -1;
  int64_t n = 0, np = 0, nroots = 0, nout = 0;
  bool has_mass = false, no_inputs = false;
  std::vector<double> defaults;
  int32_t linear_solver = DIFFSOL_LINEAR_SOLVER_DEFAULT, ode_solver = DIFFSOL_ODE_SOLVER_BDF, ensemble_mode = DIFFSOL_ENSEMBLE_AUTO;
  double rtol = 1e-6, t0 = 0.0, h0 = 1.0;
  std::vector<double> atol{1e-6};
  std::optional<double> sens_rtol, sens_atol;  // None: the sensitivities stay out of the error control (ode.rs get/set_sens_rtol / _atol)
  // ode.rs:319-388: stored and returned like the reference's; integrating the outputs and the adjoint tolerances are not implemented by this backend,
  // so a solve of a wrapper with integrate_out set fails with an explicit error instead of silently ignoring it
  bool integrate_out = false;
  std::optional<double> out_rtol, out_atol, param_rtol, param_atol;
  bool out_is_state = false;                   // out_i { u_i }: the outputs are the states
  std::shared_ptr<Settings> settings;
  ~diffsol_ode_wrapper() {
    if (model >= 0) dsh_model_release(model);
    if (lane_model >= 0) dsh_model_release(lane_model);
  }
};
struct diffsol_solution_wrapper {
  int64_t nrows = 0, ncols = 0, nb = 1;
  std::vector<std::vector<double>> sens;  // per parameter: [col][row][b] like ys (solve_fwd_sens only)
  std::vector<double> ys;  // [col][row][b]
  std::vector<double> ts;
  std::vector<int32_t> status, root_index, member_cols;
  std::vector<double> t_root;
};

namespace {

HostArray* vector_array(std::vector<double>&& v) {
  auto* a = new diffsol_host_array();
  a->shape = {v.size()};
  a->strides = {sizeof(double)};
  a->data = std::move(v);
  return a;
}

int method_of(int32_t ode_solver) {
  switch (ode_solver) {
    case DIFFSOL_ODE_SOLVER_BDF: return DSHS_METHOD_BDF;
    case DIFFSOL_ODE_SOLVER_ESDIRK34: return DSHS_METHOD_ESDIRK34;
    case DIFFSOL_ODE_SOLVER_TR_BDF2: return DSHS_METHOD_TR_BDF2;
    default: return -1;
  }
}

// number of parameter sets in a params buffer, or -1
int64_t batch_of(const OdeWrapper* ode, size_t params_len) {
  if (ode->no_inputs) return params_len == 0 ? 1 : -1;
  if (params_len == 0 || params_len % (size_t)ode->np != 0) return -1;
  return (int64_t)(params_len / (size_t)ode->np);
}
std::vector<double> params_of(const OdeWrapper* ode, const double* p, size_t len, int64_t nb) {
  if (ode->no_inputs) return std::vector<double>((size_t)nb, 0.0);  // the placeholder parameter of a model without inputs
  return std::vector<double>(p, p + len);
}

struct SolverGuard {
  dshs_solver* s = nullptr;
  ~SolverGuard() { if (s) dshs_destroy(s); }
};

int32_t make_solver(const OdeWrapper* ode, const double* params, size_t params_len, SolverGuard* g, int64_t* nb_out) {
  const int64_t nb = batch_of(ode, params_len);
  if (nb < 1) return C_ERROR("expected " + std::to_string(ode->no_inputs ? 0 : ode->np) + " parameters per member, got " + std::to_string(params_len));
  const int method = method_of(ode->ode_solver);
  if (method < 0) return C_ERROR("ode solver type is not available in the HIP backend");
  std::vector<double> p = params_of(ode, params, params_len, nb);
  int rc = dshs_create(0, nullptr, ode->model, 0, nb, p.data(), (int64_t)p.size(), ode->rtol, ode->atol.data(), (int64_t)ode->atol.size(), ode->t0, ode->h0, method,
                       &ode->settings->o, &g->s);
  if (rc != 0) return C_ERROR(std::string(dshs_last_error()));
  *nb_out = nb;
  return DIFFSOL_OK;
}

// rows of `ys`: out_i if the model has one, else the state.  states: [col][i][b] on the host -> out: [col][k][b]
int32_t apply_out(const OdeWrapper* ode, int64_t nb, const std::vector<double>& params, const std::vector<double>& ts, std::vector<double>& ys) {
  if (ode->nout == 0) return DIFFSOL_OK;
  try {
    HipContext ctx(0, nullptr, nb);
    HipVec p = HipVec::from_vec(params, ctx);
    HipVec x = HipVec::zeros(ode->n, ctx), g = HipVec::zeros(ode->nout, ctx);
    std::vector<double> out((size_t)ode->nout * nb * ts.size());
    const size_t in_col = (size_t)ode->n * nb, out_col = (size_t)ode->nout * nb;
    for (size_t c = 0; c < ts.size(); ++c) {
      check(dsh_h2d(ctx.raw(), x.ptr(), ys.data() + c * in_col, (int64_t)(in_col * sizeof(double))), "h2d");
      check(dsh_model_out(ctx.raw(), ode->model, 0, nb, ts[c], x.ptr(), p.ptr(), g.ptr()), "out");
      check(dsh_d2h(ctx.raw(), out.data() + c * out_col, g.ptr(), (int64_t)(out_col * sizeof(double))), "d2h");
      // columns past a member's own stop hold NaN states: so do its outputs (min / max in an out_i expression would turn NaN into a number)
      for (int64_t b = 0; b < nb; ++b)
        if (std::isnan(ys[c * in_col + (size_t)b]))
          for (int64_t k = 0; k < ode->nout; ++k) out[c * out_col + (size_t)(k * nb + b)] = std::numeric_limits<double>::quiet_NaN();
    }
    ys.swap(out);
  } catch (const std::exception& e) {
    return C_ERROR(e.what());
  }
  return DIFFSOL_OK;
}

// [col][b][i] (batch-major host layout of the dshs_* API) -> [col][i][b]
void to_batch_fastest(std::vector<double>& y, int64_t ncols, int64_t nb, int64_t n) {
  if (nb == 1) return;
  std::vector<double> out(y.size());
  for (int64_t c = 0; c < ncols; ++c)
    for (int64_t b = 0; b < nb; ++b)
      for (int64_t i = 0; i < n; ++i) out[(size_t)((c * n + i) * nb + b)] = y[(size_t)((c * nb + b) * n + i)];
  y.swap(out);
}

int32_t eval_op(OdeWrapper* ode, int which, const double* params, size_t params_len, double t, const double* y, size_t y_len, const double* v, size_t v_len,
                HostArray** out_array) {
  const int64_t nb = batch_of(ode, params_len);
  if (nb < 1) return C_ERROR("expected " + std::to_string(ode->no_inputs ? 0 : ode->np) + " parameters per member, got " + std::to_string(params_len));
  auto expand = [&](const double* a, size_t len, std::vector<double>& dst) -> bool {  // one vector for all members, or one per member
    if (len == (size_t)ode->n) { dst.resize((size_t)(ode->n * nb)); for (int64_t b = 0; b < nb; ++b) std::memcpy(dst.data() + b * ode->n, a, sizeof(double) * ode->n); return true; }
    if (len == (size_t)(ode->n * nb)) { dst.assign(a, a + len); return true; }
    return false;
  };
  try {
    HipContext ctx(0, nullptr, nb);
    HipVec p = HipVec::from_vec(params_of(ode, params, params_len, nb), ctx);
    HipVec out = HipVec::zeros(ode->n, ctx);
    if (which == 0) {
      check(dsh_model_init(ctx.raw(), ode->model, 0, nb, ode->t0, p.ptr(), out.ptr()), "init");
    } else {
      std::vector<double> yv, vv;
      if (!expand(y, y_len, yv)) return C_ERROR("y has " + std::to_string(y_len) + " entries, expected " + std::to_string(ode->n) + " (or nbatch times that)");
      HipVec x = HipVec::from_vec(yv, ctx);
      if (which == 1) {
        check(dsh_model_rhs(ctx.raw(), ode->model, 0, nb, t, x.ptr(), p.ptr(), out.ptr()), "rhs");
      } else {
        if (!expand(v, v_len, vv)) return C_ERROR("v has " + std::to_string(v_len) + " entries, expected " + std::to_string(ode->n) + " (or nbatch times that)");
        HipVec vd = HipVec::from_vec(vv, ctx);
        check(dsh_model_jac_mul(ctx.raw(), ode->model, 0, nb, t, x.ptr(), p.ptr(), vd.ptr(), out.ptr()), "jac_mul");
      }
    }
    *out_array = vector_array(out.clone_as_vec());
  } catch (const std::exception& e) {
    return C_ERROR(e.what());
  }
  return DIFFSOL_OK;
}

struct EnumInfo { const char* name; bool valid; };
const EnumInfo kMatrix[] = {{"nalgebra_dense", false}, {"faer_dense", false}, {"faer_sparse", false}, {"hip_dense", true}};
const EnumInfo kLinear[] = {{"default", true}, {"lu", true}, {"klu", false}};
const EnumInfo kOdeSolver[] = {{"bdf", true}, {"esdirk34", true}, {"tr_bdf2", true}, {"tsit45", false}};
const EnumInfo kScalar[] = {{"f32", false}, {"f64", true}};
const EnumInfo kJit[] = {{"cranelift", false}, {"llvm", false}, {"hiprtc", true}};
template <size_t N> bool enum_valid(const EnumInfo (&t)[N], int32_t v) { return v >= 0 && (size_t)v < N && t[v].valid; }
template <size_t N> const char* enum_name(const EnumInfo (&t)[N], int32_t v, const char* what) {
  if (v < 0 || (size_t)v >= N) { C_INVALID_ARG(std::string("invalid ") + what); return nullptr; }
  return t[v].name;
}

}  // namespace

extern "C" {

// ------------------------------------------------------------------ error_c.rs
int32_t diffsol_error_code(void) { return g_last.set ? 1 : 0; }
const char* diffsol_error(void) { return g_last.set ? g_last.message.c_str() : nullptr; }
const char* diffsol_last_error_message(void) { return g_last.set ? g_last.message.c_str() : nullptr; }
const char* diffsol_last_error_file(void) { return g_last.set ? g_last.file.c_str() : nullptr; }
uint32_t diffsol_last_error_line(void) { return g_last.set ? g_last.line : 0; }
void diffsol_clear_last_error(void) { g_last = LastError(); }

// ------------------------------------------------------------------ host_array_c.rs
HostArray* diffsol_host_array_alloc_vector(size_t len, int32_t dtype) {
  if (dtype != DIFFSOL_SCALAR_F64) { C_INVALID_ARG("invalid dtype"); return nullptr; }
  return vector_array(std::vector<double>(len, 0.0));
}
void diffsol_host_array_free(HostArray* array) {
  if (!array) { C_INVALID_ARG("host array is null"); return; }
  delete array;
}
const uint8_t* diffsol_host_array_ptr(const HostArray* array) {
  if (!array) { C_INVALID_ARG("host array is null"); return nullptr; }
  return reinterpret_cast<const uint8_t*>(array->data.data());
}
size_t diffsol_host_array_ndim(const HostArray* array) {
  if (!array) { C_INVALID_ARG("host array is null"); return 0; }
  return array->shape.size();
}
size_t diffsol_host_array_dim(const HostArray* array, size_t index) {
  if (!array || index >= array->shape.size()) { C_INVALID_ARG("host array is null or index out of range"); return 0; }
  return array->shape[index];
}
size_t diffsol_host_array_stride(const HostArray* array, size_t index) {
  if (!array || index >= array->strides.size()) { C_INVALID_ARG("host array is null or index out of range"); return 0; }
  return array->strides[index];
}
int32_t diffsol_host_array_dtype(const HostArray* array) {
  if (!array) { C_INVALID_ARG("host array is null"); return -1; }
  return DIFFSOL_SCALAR_F64;
}

// ------------------------------------------------------------------ runtime enums
size_t diffsol_matrix_type_count(void) { return 4; }
int32_t diffsol_matrix_type_is_valid(int32_t v) { return enum_valid(kMatrix, v) ? 1 : 0; }
const char* diffsol_matrix_type_name(int32_t v) { return enum_name(kMatrix, v, "matrix_type"); }
size_t diffsol_linear_solver_type_count(void) { return 3; }
int32_t diffsol_linear_solver_type_is_valid(int32_t v) { return enum_valid(kLinear, v) ? 1 : 0; }
const char* diffsol_linear_solver_type_name(int32_t v) { return enum_name(kLinear, v, "linear_solver_type"); }
size_t diffsol_ode_solver_type_count(void) { return 4; }
int32_t diffsol_ode_solver_type_is_valid(int32_t v) { return enum_valid(kOdeSolver, v) ? 1 : 0; }
const char* diffsol_ode_solver_type_name(int32_t v) { return enum_name(kOdeSolver, v, "ode_solver_type"); }
size_t diffsol_scalar_type_count(void) { return 2; }
int32_t diffsol_scalar_type_is_valid(int32_t v) { return enum_valid(kScalar, v) ? 1 : 0; }
const char* diffsol_scalar_type_name(int32_t v) { return enum_name(kScalar, v, "scalar_type"); }
size_t diffsol_jit_backend_type_count(void) { return 3; }
int32_t diffsol_jit_backend_type_is_valid(int32_t v) { return enum_valid(kJit, v) ? 1 : 0; }
const char* diffsol_jit_backend_type_name(int32_t v) { return enum_name(kJit, v, "jit_backend_type"); }

// ------------------------------------------------------------------ ode_c.rs
OdeWrapper* diffsol_ode_new_jit(const char* code, int32_t jit_backend, int32_t matrix_type, int32_t linear_solver, int32_t ode_solver) {
  if (!code) { C_INVALID_ARG("code is null"); return nullptr; }
  if (!enum_valid(kMatrix, matrix_type)) { C_INVALID_ARG("invalid matrix_type (this library provides hip_dense)"); return nullptr; }
  if (!enum_valid(kLinear, linear_solver)) { C_INVALID_ARG("invalid linear_solver_type"); return nullptr; }
  if (!enum_valid(kOdeSolver, ode_solver)) { C_INVALID_ARG("invalid ode_solver_type"); return nullptr; }
  if (!enum_valid(kJit, jit_backend)) { C_INVALID_ARG("invalid jit_backend_type (this library provides hiprtc)"); return nullptr; }
  auto ode = std::make_unique<diffsol_ode_wrapper>();
  ode->code = code;
  ode->linear_solver = linear_solver;
  ode->ode_solver = ode_solver;
  ode->settings = std::make_shared<Settings>();
  dshs_default_options(&ode->settings->o);
  // ONE parse / differentiation of the text (it takes seconds for the 290 000-line DFN model): dimensions, input defaults, the out_i shape and every device form
  // are read off the same diffsl::Compiled
  int id = -1;
  try {
    const diffsl::Compiled c = diffsl::compile(code, diffsl::take_pending_model_index());  // the index armed for THIS call, 0 otherwise
    const int64_t dims[10] = {c.n, c.np, c.nroots, c.nout, c.has_mass ? 1 : 0, c.dummy_param ? 1 : 0, c.jac_kl, c.jac_ku, c.mass_kl, c.mass_ku};
    const bool is_static = dims[0] <= 8 && dims[2] <= 1;
    ode->n = dims[0]; ode->np = dims[1]; ode->nroots = dims[2]; ode->nout = dims[3]; ode->has_mass = dims[4] != 0; ode->no_inputs = dims[5] != 0;
    ode->defaults.assign((size_t)ode->np, 0.0);
    for (size_t k = 0; k < ode->defaults.size() && k < c.input_defaults.size(); ++k) ode->defaults[k] = c.input_defaults[k];
    const std::string src = diffsl::generate(c, is_static ? diffsl::Target::HipStatic : diffsl::Target::HipDynamic);
    const int rc = dsh_model_compile(src.c_str(), is_static ? DSH_JIT_FORM_STATIC : DSH_JIT_FORM_DYNAMIC, ode->n, ode->np, ode->nroots, ode->nout, ode->has_mass ? 1 : 0, &id);
    if (rc != 0) { C_ERROR(std::string(dsh_last_error())); return nullptr; }
    ode->model = id;
    dsh_model_set_band(id, (int)dims[6], (int)dims[7], (int)dims[8], (int)dims[9]);
    if (is_static && ode->n >= 5) {  // per-member device solves of a static model with 5 <= n <= 8 run on its run-time-sized form, compiled at the first such request
      const std::string dyn = diffsl::generate(c, diffsl::Target::HipDynamic);
      (void)dsh_model_set_member_twin_source(id, dyn.c_str(), ode->n, ode->np, ode->nroots, ode->nout);
    }
    // out_i { u_i } (the outputs are the states, component by component): solve_fwd_sens then needs no output derivatives
    ode->out_is_state = (int64_t)c.out.size() == ode->n;
    for (size_t i = 0; i < c.out.size() && ode->out_is_state; ++i) {
      const diffsl::Node& nd = c.g.at(c.out[i]);
      ode->out_is_state = nd.k == diffsl::NK::State && nd.i == (int)i;
    }
    // banded run-time-sized model: the same text once more in the lane-per-member form, which per-member / wavefront device-resident solves run on
    if (!is_static && ode->n <= 64 && (!ode->has_mass || (dims[8] == 0 && dims[9] == 0)) && ode->nroots <= 8 && std::max(dims[6], dims[7]) <= 4) {
      int lane = -1;
      std::string lane_src;
      bool have = true;
      try { lane_src = diffsl::generate(c, diffsl::Target::HipStatic); } catch (...) { have = false; }
      if (have && dsh_model_compile(lane_src.c_str(), DSH_JIT_FORM_STATIC_BANDED, ode->n, ode->np, ode->nroots, ode->nout, ode->has_mass ? 1 : 0, &lane) == 0 && dsh_model_set_twin(id, lane) == 0)
        ode->lane_model = lane;
    }
  } catch (const std::exception& e) { C_ERROR(std::string(e.what())); return nullptr; }
  return ode.release();
}

namespace {
// `#define NAME <unsigned>` in a user's external HIP source
bool macro_value(const std::string& text, const char* name, int64_t* out) {
  const std::string key = std::string("#define ") + name;
  size_t p = text.find(key);
  while (p != std::string::npos) {
    size_t q = p + key.size();
    if (q < text.size() && (text[q] == ' ' || text[q] == '\t')) {
      while (q < text.size() && (text[q] == ' ' || text[q] == '\t')) ++q;
      char* end = nullptr;
      const long long v = std::strtoll(text.c_str() + q, &end, 10);
      if (end != text.c_str() + q && v >= 0) { *out = v; return true; }
    }
    p = text.find(key, p + 1);
  }
  return false;
}
}  // namespace

// diffsol_ode_new_external (ode_c.rs:181-230): a model whose functions are LINKED INTO the library.  Nothing is linked into this one — device code is
// compiled at run time — so the constructor exists for binding compatibility and reports that; diffsol_ode_new_external_dynamic takes the model as a file.
OdeWrapper* diffsol_ode_new_external(int32_t matrix_type, int32_t linear_solver, int32_t ode_solver, const DiffsolDepPair* rhs_state_deps_ptr, size_t rhs_state_deps_len,
                                     const DiffsolDepPair* rhs_input_deps_ptr, size_t rhs_input_deps_len, const DiffsolDepPair* mass_state_deps_ptr,
                                     size_t mass_state_deps_len) {
  (void)matrix_type; (void)linear_solver; (void)ode_solver; (void)rhs_state_deps_ptr; (void)rhs_state_deps_len; (void)rhs_input_deps_ptr; (void)rhs_input_deps_len;
  (void)mass_state_deps_ptr; (void)mass_state_deps_len;
  C_ERROR("diffsol_ode_new_external: no model is statically linked into the HIP backend (device code is compiled at run time); "
          "use diffsol_ode_new_external_dynamic with the path of a HIP source that defines the external functions");
  return nullptr;
}

// diffsol_ode_new_external_dynamic (ode_c.rs:232-281).  The reference loads a shared library that exports the external model ABI (set_u0, rhs, rhs_grad,
// mass, calc_out, calc_stop, set_inputs, get_dims, ...: crates/diffsol-c/tests/external-dynamic-logistic/src/lib.rs).  A device backend cannot call host
// code, so `path` names a HIP SOURCE file instead that defines the same functions, same names and argument orders, as device functions
// (`DIFFSOL_DEVICE void rhs(double time, const double* u, double* data, double* rr, uint32_t thread_id, uint32_t thread_dim)` ...), and states what
// get_dims would return as macros: DIFFSOL_EXTERNAL_STATES, _INPUTS, _OUTPUTS, _DATA, _STOP, _HAS_MASS.  Required: set_inputs(const double* inputs,
// double* data), set_u0(u, data, tid, tdim), rhs, rhs_grad(time, u, du, data, ddata, rr, drr, tid, tdim); with _HAS_MASS: mass(time, v, data, mv, tid, tdim);
// with _STOP > 0: calc_stop(time, u, data, root, tid, tdim); with _OUTPUTS > 0: calc_out(time, u, data, out, tid, tdim) (else the outputs are the states).
// thread_id = 0, thread_dim = 1: one ensemble member per lane.  Register-resident form: at most 8 states and one stop condition.  The dependency
// lists (sparsity) are accepted and not needed: Jacobians are assembled dense from rhs_grad.
OdeWrapper* diffsol_ode_new_external_dynamic(const char* path, int32_t matrix_type, int32_t linear_solver, int32_t ode_solver, const DiffsolDepPair* rhs_state_deps_ptr,
                                             size_t rhs_state_deps_len, const DiffsolDepPair* rhs_input_deps_ptr, size_t rhs_input_deps_len,
                                             const DiffsolDepPair* mass_state_deps_ptr, size_t mass_state_deps_len) {
  if (!path) { C_INVALID_ARG("path is null"); return nullptr; }
  if ((!rhs_state_deps_ptr && rhs_state_deps_len) || (!rhs_input_deps_ptr && rhs_input_deps_len) || (!mass_state_deps_ptr && mass_state_deps_len)) {
    C_INVALID_ARG("dependency pointer is null with a non-zero length"); return nullptr;
  }
  if (!enum_valid(kMatrix, matrix_type)) { C_INVALID_ARG("invalid matrix_type (this library provides hip_dense)"); return nullptr; }
  if (!enum_valid(kLinear, linear_solver)) { C_INVALID_ARG("invalid linear_solver_type"); return nullptr; }
  if (!enum_valid(kOdeSolver, ode_solver)) { C_INVALID_ARG("invalid ode_solver_type"); return nullptr; }
  std::string text;
  {
    FILE* f = std::fopen(path, "rb");
    if (!f) { C_ERROR(std::string("diffsol_ode_new_external_dynamic: cannot open ") + path); return nullptr; }
    char buf[4096];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    std::fclose(f);
  }
  int64_t ns = -1, ni = -1, no = 0, nd = 0, nstop = 0, hm = 0;
  if (!macro_value(text, "DIFFSOL_EXTERNAL_STATES", &ns) || !macro_value(text, "DIFFSOL_EXTERNAL_INPUTS", &ni)) {
    C_ERROR("diffsol_ode_new_external_dynamic: the source must #define DIFFSOL_EXTERNAL_STATES and DIFFSOL_EXTERNAL_INPUTS (what get_dims returns)"); return nullptr;
  }
  (void)macro_value(text, "DIFFSOL_EXTERNAL_OUTPUTS", &no); (void)macro_value(text, "DIFFSOL_EXTERNAL_DATA", &nd);
  (void)macro_value(text, "DIFFSOL_EXTERNAL_STOP", &nstop); (void)macro_value(text, "DIFFSOL_EXTERNAL_HAS_MASS", &hm);
  if (ns < 1 || ns > 8 || nstop > 1) { C_ERROR("diffsol_ode_new_external_dynamic: external models run in the register-resident form: 1..8 states, at most one stop condition"); return nullptr; }
  const bool no_inputs = ni == 0;
  const int64_t np = no_inputs ? 1 : ni, nout = no > 0 ? no : ns;
  std::string src = "// adapter generated by diffsol_ode_new_external_dynamic around a user's external model (the reference's external ABI as device functions)\n"
                    "#include <stdint.h>\n#define DIFFSOL_DEVICE __device__ static inline\nnamespace dsh_ext {\n" + text + "\n}  // namespace dsh_ext\nnamespace dsh {\nstruct JitModel {\n";
  src += "  static constexpr int N = " + std::to_string(ns) + ", NP = " + std::to_string(np) + ", NROOTS = " + std::to_string(nstop) + ", NOUT = " + std::to_string(nout) + ";\n";
  src += std::string("  static constexpr bool HAS_MASS = ") + (hm ? "true" : "false") + ";\n  static constexpr int ND = " + std::to_string(nd > 0 ? nd : 1) + ";\n";
  src += "  __device__ static void load(const double (&p)[NP], double (&data)[ND]) {\n    for (int k = 0; k < ND; ++k) data[k] = 0.0;\n";
  src += no_inputs ? "    (void)p;\n  }\n" : "    dsh_ext::set_inputs(p, data);\n  }\n";
  src += "  __device__ static void rhs(double t, const double (&x)[N], const double (&p)[NP], double (&y)[N]) { double data[ND]; load(p, data); dsh_ext::rhs(t, x, data, y, 0u, 1u); }\n";
  src += "  __device__ static void jac_mul(double t, const double (&x)[N], const double (&p)[NP], const double (&v)[N], double (&y)[N]) {\n"
         "    double data[ND], ddata[ND], rr[N];\n    load(p, data);\n    for (int k = 0; k < ND; ++k) ddata[k] = 0.0;\n"
         "    dsh_ext::rhs(t, x, data, rr, 0u, 1u);\n    dsh_ext::rhs_grad(t, x, v, data, ddata, rr, y, 0u, 1u);\n  }\n";
  src += "  __device__ static void mass_gemv(double t, const double (&x)[N], const double (&p)[NP], double beta, double (&y)[N]) {\n";
  src += hm ? "    double data[ND], mv[N];\n    load(p, data);\n    dsh_ext::mass(t, x, data, mv, 0u, 1u);\n    for (int i = 0; i < N; ++i) y[i] = mv[i] + beta * y[i];\n  }\n"
            : "    (void)t; (void)p;\n    for (int i = 0; i < N; ++i) y[i] = x[i] + beta * y[i];\n  }\n";
  src += "  __device__ static void init(double t, const double (&p)[NP], double (&y)[N]) { (void)t; double data[ND]; load(p, data); dsh_ext::set_u0(y, data, 0u, 1u); }\n";
  src += std::string("  __device__ static void root(double t, const double (&x)[N], const double (&p)[NP], double (&g)[") + std::to_string(nstop > 0 ? nstop : 1) + "]) {\n";
  src += nstop > 0 ? "    double data[ND]; load(p, data); dsh_ext::calc_stop(t, x, data, g, 0u, 1u);\n  }\n" : "    (void)t; (void)x; (void)p; g[0] = 1.0;\n  }\n";
  src += "  __device__ static void out(double t, const double (&x)[N], const double (&p)[NP], double (&g)[NOUT]) {\n";
  src += no > 0 ? "    double data[ND]; load(p, data); dsh_ext::calc_out(t, x, data, g, 0u, 1u);\n  }\n" : "    (void)t; (void)p;\n    for (int i = 0; i < N; ++i) g[i] = x[i];\n  }\n";
  src += "};\n}  // namespace dsh\n";
  auto ode = std::make_unique<diffsol_ode_wrapper>();
  ode->code = src;
  ode->linear_solver = linear_solver;
  ode->ode_solver = ode_solver;
  ode->settings = std::make_shared<Settings>();
  dshs_default_options(&ode->settings->o);
  ode->n = ns; ode->np = np; ode->nroots = nstop; ode->nout = nout; ode->has_mass = hm != 0; ode->no_inputs = no_inputs;
  ode->defaults.assign((size_t)np, 0.0);
  ode->out_is_state = no == 0;
  int id = -1;
  if (dsh_model_compile(src.c_str(), DSH_JIT_FORM_STATIC, ode->n, ode->np, ode->nroots, ode->nout, ode->has_mass ? 1 : 0, &id) != 0) { C_ERROR(std::string(dsh_last_error())); return nullptr; }
  ode->model = id;
  return ode.release();
}
void diffsol_ode_free(OdeWrapper* ode) {
  if (!ode) { C_INVALID_ARG("ode is null"); return; }
  delete ode;
}
int32_t diffsol_ode_get_options(const OdeWrapper* ode, OdeSolverOptions** out_options) {
  if (!ode || !out_options) return C_INVALID_ARG("invalid arguments to diffsol_ode_get_options");
  *out_options = new diffsol_ode_solver_options{ode->settings};
  return DIFFSOL_OK;
}
int32_t diffsol_ode_get_ic_options(const OdeWrapper* ode, InitialConditionSolverOptions** out_options) {
  if (!ode || !out_options) return C_INVALID_ARG("invalid arguments to diffsol_ode_get_ic_options");
  *out_options = new diffsol_ic_solver_options{ode->settings};
  return DIFFSOL_OK;
}
int32_t diffsol_ode_y0(OdeWrapper* ode, const double* params_ptr, size_t params_len, HostArray** out_array) {
  if (!ode || !out_array || (!params_ptr && params_len)) return C_INVALID_ARG("invalid arguments to diffsol_ode_y0");
  return eval_op(ode, 0, params_ptr, params_len, 0.0, nullptr, 0, nullptr, 0, out_array);
}
int32_t diffsol_ode_rhs(OdeWrapper* ode, const double* params_ptr, size_t params_len, double t, const double* y_ptr, size_t y_len, HostArray** out_array) {
  if (!ode || !out_array || (!params_ptr && params_len) || (!y_ptr && y_len)) return C_INVALID_ARG("invalid arguments to diffsol_ode_rhs");
  return eval_op(ode, 1, params_ptr, params_len, t, y_ptr, y_len, nullptr, 0, out_array);
}
int32_t diffsol_ode_rhs_jac_mul(OdeWrapper* ode, const double* params_ptr, size_t params_len, double t, const double* y_ptr, size_t y_len, const double* v_ptr,
                                size_t v_len, HostArray** out_array) {
  if (!ode || !out_array || (!params_ptr && params_len) || (!y_ptr && y_len) || (!v_ptr && v_len)) return C_INVALID_ARG("invalid arguments to diffsol_ode_rhs_jac_mul");
  return eval_op(ode, 2, params_ptr, params_len, t, y_ptr, y_len, v_ptr, v_len, out_array);
}

int32_t diffsol_ode_solve(OdeWrapper* ode, const double* params_ptr, size_t params_len, double final_time, SolutionWrapper** out_solution) {
  if (!ode || !out_solution || (!params_ptr && params_len)) return C_INVALID_ARG("invalid arguments to diffsol_ode_solve");
  if (ode->integrate_out) return C_ERROR("integrate_out is set: integrating the output equations alongside the states is not implemented by the HIP backend (clear it with diffsol_ode_set_integrate_out)");
  if (ode->ensemble_mode != DIFFSOL_ENSEMBLE_LOCKSTEP && ode->ensemble_mode != DIFFSOL_ENSEMBLE_AUTO)
    return C_ERROR("solve() returns every internal step, which only exists for the lock-step ensemble; use solve_dense with the per-member / wavefront modes");
  SolverGuard g;
  int64_t nb = 0;
  int32_t rc = make_solver(ode, params_ptr, params_len, &g, &nb);
  if (rc != DIFFSOL_OK) return rc;
  auto sol = std::make_unique<diffsol_solution_wrapper>();
  int64_t ncols = 0;
  int reason = 0;
  std::vector<double> y_final((size_t)(ode->n * nb));
  if (dshs_solve(g.s, final_time, 1, y_final.data(), &ncols, &reason) != 0) return C_ERROR(std::string(dshs_last_error()));
  sol->ts.resize((size_t)ncols);
  sol->ys.resize((size_t)(ncols * nb * ode->n));
  if (dshs_trajectory(g.s, sol->ts.data(), sol->ys.data()) != 0) return C_ERROR(std::string(dshs_last_error()));
  to_batch_fastest(sol->ys, ncols, nb, ode->n);
  sol->nb = nb; sol->ncols = ncols; sol->nrows = ode->nout ? ode->nout : ode->n;
  double t_root = std::numeric_limits<double>::quiet_NaN();
  int root_index = -1;
  if (reason == DSHS_STOP_ROOT_FOUND) dshs_root_info(g.s, &t_root, &root_index);
  sol->status.assign((size_t)nb, 0); sol->t_root.assign((size_t)nb, t_root); sol->root_index.assign((size_t)nb, root_index); sol->member_cols.assign((size_t)nb, (int32_t)ncols);
  rc = apply_out(ode, nb, params_of(ode, params_ptr, params_len, nb), sol->ts, sol->ys);
  if (rc != DIFFSOL_OK) return rc;
  *out_solution = sol.release();
  return DIFFSOL_OK;
}

int32_t diffsol_ode_solve_dense(OdeWrapper* ode, const double* params_ptr, size_t params_len, const double* t_eval_ptr, size_t t_eval_len,
                                SolutionWrapper** out_solution) {
  if (!ode || !out_solution || (!params_ptr && params_len) || !t_eval_ptr || t_eval_len == 0) return C_INVALID_ARG("invalid arguments to diffsol_ode_solve_dense");
  if (ode->integrate_out) return C_ERROR("integrate_out is set: integrating the output equations alongside the states is not implemented by the HIP backend (clear it with diffsol_ode_set_integrate_out)");
  SolverGuard g;
  int64_t nb = 0;
  int32_t rc = make_solver(ode, params_ptr, params_len, &g, &nb);
  if (rc != DIFFSOL_OK) return rc;
  auto sol = std::make_unique<diffsol_solution_wrapper>();
  const int64_t nt = (int64_t)t_eval_len;
  sol->ys.resize((size_t)(nt * nb * ode->n));
  sol->nb = nb; sol->nrows = ode->nout ? ode->nout : ode->n;
  sol->status.assign((size_t)nb, 0);
  sol->t_root.assign((size_t)nb, std::numeric_limits<double>::quiet_NaN());
  sol->root_index.assign((size_t)nb, -1);
  sol->member_cols.assign((size_t)nb, (int32_t)nt);
  int64_t ncols = nt;
  // DIFFSOL_ENSEMBLE_AUTO: what dshs_solve_dense would pick (device-resident whenever the model has such a kernel, include/diffsol_hip_solver.h)
  int mode = ode->ensemble_mode;
  if (mode == DIFFSOL_ENSEMBLE_AUTO) { int requested = 0; dshs_get_ensemble_mode(g.s, &requested, &mode); }
  if (mode == DIFFSOL_ENSEMBLE_LOCKSTEP) {
    int reason = 0;
    if (dshs_set_ensemble_mode(g.s, DSHS_ENSEMBLE_LOCKSTEP) != 0) return C_ERROR(std::string(dshs_last_error()));
    if (dshs_solve_dense(g.s, t_eval_ptr, nt, sol->ys.data(), nullptr, &reason) != 0) return C_ERROR(std::string(dshs_last_error()));
    if (reason == DSHS_STOP_ROOT_FOUND) {  // solve_dense stops at the root: the column after the last t_eval <= t_root holds the state at the root (method.rs:498-516)
      double t_root = 0.0; int idx = -1;
      dshs_root_info(g.s, &t_root, &idx);
      int64_t k = 0;
      while (k < nt && t_eval_ptr[k] <= t_root) ++k;
      ncols = k < nt ? k + 1 : nt;
      sol->t_root.assign((size_t)nb, t_root); sol->root_index.assign((size_t)nb, idx); sol->member_cols.assign((size_t)nb, (int32_t)ncols);
      sol->ts.assign(t_eval_ptr, t_eval_ptr + ncols);
      if (k < nt) sol->ts[(size_t)k] = t_root;
      sol->ys.resize((size_t)(ncols * nb * ode->n));
    } else {
      sol->ts.assign(t_eval_ptr, t_eval_ptr + nt);
    }
  } else {
    int64_t totals[6];
    if (dshs_solve_dense_adaptive(g.s, t_eval_ptr, nt, mode, 1, sol->ys.data(), nullptr, nullptr, sol->status.data(), sol->t_root.data(), sol->root_index.data(),
                                  sol->member_cols.data(), totals) != 0)
      return C_ERROR(std::string(dshs_last_error()));
    sol->ts.assign(t_eval_ptr, t_eval_ptr + nt);
  }
  to_batch_fastest(sol->ys, ncols, nb, ode->n);
  sol->ncols = ncols;
  rc = apply_out(ode, nb, params_of(ode, params_ptr, params_len, nb), sol->ts, sol->ys);
  if (rc != DIFFSOL_OK) return rc;
  *out_solution = sol.release();
  return DIFFSOL_OK;
}

// ode_c.rs:586-617 over ode.rs:514-528 over OdeSolverMethod::solve_dense_sensitivities (ode_solver/sensitivities.rs:114-261): the states (outputs) and the
// forward sensitivities d(out)/dp_j at t_eval, one parameter set or a lock-step batch of them.  BDF / TR-BDF2 / ESDIRK34, ODEs and DAEs; host-driven over the
// device operators (the resident kernels integrate the state equations only).  Limits of this backend: out_i other than the states themselves, stop
// conditions that fire, and reset_i are refused here.
int32_t diffsol_ode_solve_fwd_sens(OdeWrapper* ode, const double* params_ptr, size_t params_len, const double* t_eval_ptr, size_t t_eval_len,
                                   SolutionWrapper** out_solution) {
  if (!ode || !out_solution || (!params_ptr && params_len) || (!t_eval_ptr && t_eval_len)) return C_INVALID_ARG("invalid arguments to diffsol_ode_solve_fwd_sens");
  if (t_eval_len == 0) return C_ERROR("t_eval must not be empty");
  if (ode->integrate_out) return C_ERROR("integrate_out is set: integrating the output equations alongside the states is not implemented by the HIP backend (clear it with diffsol_ode_set_integrate_out)");
  if (ode->no_inputs) return C_ERROR("the model has no inputs: nothing to differentiate with respect to");
  if (ode->nout != 0 && !ode->out_is_state) return C_ERROR("solve_fwd_sens with an out_i other than the states is not supported by the HIP backend (omit out_i or use out_i { u_i })");
  for (size_t k = 0; k + 1 < t_eval_len; ++k) if (t_eval_ptr[k] > t_eval_ptr[k + 1]) return C_ERROR("t_eval must be increasing");
  if (t_eval_ptr[0] < ode->t0) return C_ERROR("t_eval must not start before t0");
  const int64_t nb = batch_of(ode, params_len);
  if (nb < 1) return C_ERROR("expected " + std::to_string(ode->np) + " parameters per member, got " + std::to_string(params_len));
  const int method = method_of(ode->ode_solver);
  if (method < 0) return C_ERROR("ode solver type is not available in the HIP backend");
  SolverGuard g;
  const double sa = ode->sens_atol.value_or(0.0);
  // with either tolerance missing the sensitivities stay out of the error control (builder.rs:1501-1505 / sens_equations.rs:283-285)
  const bool ec = ode->sens_rtol.has_value() && ode->sens_atol.has_value();
  if (dshs_create_sens(0, nullptr, ode->model, 0, nb, params_ptr, (int64_t)params_len, ode->rtol, ode->atol.data(), (int64_t)ode->atol.size(), ode->t0, ode->h0, method,
                       &ode->settings->o, 1, ode->sens_rtol.value_or(0.0), &sa, ec ? 1 : 0, &g.s) != 0)
    return C_ERROR(std::string(dshs_last_error()));
  auto sol = std::make_unique<diffsol_solution_wrapper>();
  const int64_t nt = (int64_t)t_eval_len, n = ode->n, np = ode->np;
  sol->nb = nb; sol->nrows = n; sol->ncols = nt;
  sol->ys.resize((size_t)(nt * nb * n));
  sol->sens.assign((size_t)np, std::vector<double>((size_t)(nt * nb * n)));
  sol->ts.assign(t_eval_ptr, t_eval_ptr + nt);
  sol->status.assign((size_t)nb, 0);
  sol->t_root.assign((size_t)nb, std::numeric_limits<double>::quiet_NaN());
  sol->root_index.assign((size_t)nb, -1);
  sol->member_cols.assign((size_t)nb, (int32_t)nt);
  std::vector<double> ybuf((size_t)(nb * n)), sbuf((size_t)(np * nb * n));
  int64_t col = 0;
  auto drain = [&](double upto) -> bool {
    while (col < nt && t_eval_ptr[col] <= upto) {
      if (dshs_interpolate(g.s, t_eval_ptr[col], ybuf.data()) != 0 || dshs_interpolate_sens(g.s, t_eval_ptr[col], sbuf.data()) != 0) return false;
      // host buffers are [b][i]; the wrapper stores [col][i][b]
      for (int64_t b = 0; b < nb; ++b)
        for (int64_t i = 0; i < n; ++i) {
          sol->ys[(size_t)((col * n + i) * nb + b)] = ybuf[(size_t)(b * n + i)];
          for (int64_t j = 0; j < np; ++j) sol->sens[(size_t)j][(size_t)((col * n + i) * nb + b)] = sbuf[(size_t)((j * nb + b) * n + i)];
        }
      ++col;
    }
    return true;
  };
  double t_now = ode->t0, h_now = 0.0;
  int order_now = 0;
  if (!drain(t_now)) return C_ERROR(std::string(dshs_last_error()));
  if (col < nt) {
    if (dshs_set_stop_time(g.s, t_eval_ptr[nt - 1]) != 0) return C_ERROR(std::string(dshs_last_error()));
    while (true) {
      int reason = 0;
      if (dshs_step(g.s, &reason) != 0) return C_ERROR(std::string(dshs_last_error()));
      if (reason == DSHS_STOP_ROOT_FOUND) return C_ERROR("solve_fwd_sens: a stop condition fired; events with forward sensitivities are not supported by the HIP backend");
      dshs_get_state(g.s, &t_now, &h_now, &order_now, nullptr, nullptr);
      if (!drain(t_now)) return C_ERROR(std::string(dshs_last_error()));
      if (reason == DSHS_STOP_TSTOP_REACHED) break;
    }
  }
  *out_solution = sol.release();
  return DIFFSOL_OK;
}
int32_t diffsol_ode_get_sens_rtol(const OdeWrapper* ode, int32_t* out_is_some, double* out_value) {
  if (!ode || !out_is_some || !out_value) return C_INVALID_ARG("invalid arguments to diffsol_ode_get_sens_rtol");
  *out_is_some = ode->sens_rtol.has_value() ? 1 : 0; *out_value = ode->sens_rtol.value_or(0.0);
  return DIFFSOL_OK;
}
int32_t diffsol_ode_set_sens_rtol(OdeWrapper* ode, int32_t value_is_some, double value) {
  if (!ode) return C_INVALID_ARG("ode is null");
  if (value_is_some) ode->sens_rtol = value; else ode->sens_rtol.reset();
  return DIFFSOL_OK;
}
int32_t diffsol_ode_get_sens_atol(const OdeWrapper* ode, int32_t* out_is_some, double* out_value) {
  if (!ode || !out_is_some || !out_value) return C_INVALID_ARG("invalid arguments to diffsol_ode_get_sens_atol");
  *out_is_some = ode->sens_atol.has_value() ? 1 : 0; *out_value = ode->sens_atol.value_or(0.0);
  return DIFFSOL_OK;
}
int32_t diffsol_ode_set_sens_atol(OdeWrapper* ode, int32_t value_is_some, double value) {
  if (!ode) return C_INVALID_ARG("ode is null");
  if (value_is_some) ode->sens_atol = value; else ode->sens_atol.reset();
  return DIFFSOL_OK;
}


// ode_c.rs:893-1190: integrate_out and the output / parameter tolerances of the adjoint equations — kept and returned; see diffsol_ode_wrapper
int32_t diffsol_ode_get_integrate_out(const OdeWrapper* ode, int32_t* out_value) {
  if (!ode || !out_value) return C_INVALID_ARG("invalid arguments to diffsol_ode_get_integrate_out");
  *out_value = ode->integrate_out ? 1 : 0;
  return DIFFSOL_OK;
}
int32_t diffsol_ode_set_integrate_out(OdeWrapper* ode, int32_t value) {
  if (!ode) return C_INVALID_ARG("ode is null");
  ode->integrate_out = value != 0;
  return DIFFSOL_OK;
}
#define DIFFSOL_OPTIONAL_F64(field)                                                                                         \
  int32_t diffsol_ode_get_##field(const OdeWrapper* ode, int32_t* out_is_some, double* out_value) {                          \
    if (!ode || !out_is_some || !out_value) return C_INVALID_ARG("invalid arguments to diffsol_ode_get_" #field);            \
    *out_is_some = ode->field.has_value() ? 1 : 0; *out_value = ode->field.value_or(0.0);                                   \
    return DIFFSOL_OK;                                                                                                      \
  }                                                                                                                         \
  int32_t diffsol_ode_set_##field(OdeWrapper* ode, int32_t value_is_some, double value) {                                    \
    if (!ode) return C_INVALID_ARG("ode is null");                                                                           \
    if (value_is_some) ode->field = value; else ode->field.reset();                                                         \
    return DIFFSOL_OK;                                                                                                      \
  }
DIFFSOL_OPTIONAL_F64(out_rtol)
DIFFSOL_OPTIONAL_F64(out_atol)
DIFFSOL_OPTIONAL_F64(param_rtol)
DIFFSOL_OPTIONAL_F64(param_atol)
#undef DIFFSOL_OPTIONAL_F64

// string_c.rs:11-78: memory the caller fills and hands to the library (bindings build their strings in it)
char* diffsol_alloc_string(size_t size) { return size == 0 ? nullptr : (char*)std::calloc(size, 1); }
void diffsol_free_string(char* ptr, size_t size) { if (ptr && size) std::free(ptr); }
uint8_t* diffsol_alloc(size_t size, size_t align) {
  if (size == 0) return nullptr;
  if (align == 0) align = 1;
  if (align & (align - 1)) return nullptr;  // Layout::from_size_align refuses alignments that are not powers of two
  void* p = nullptr;
  if (align <= alignof(max_align_t)) return (uint8_t*)std::malloc(size);
  if (posix_memalign(&p, align < sizeof(void*) ? sizeof(void*) : align, size) != 0) return nullptr;
  return (uint8_t*)p;
}
void diffsol_free(uint8_t* ptr, size_t size, size_t align) { (void)align; if (ptr && size) std::free(ptr); }

int32_t diffsol_ode_get_matrix_type(const OdeWrapper* ode) { if (!ode) { C_INVALID_ARG("ode is null"); return -1; } return DIFFSOL_MATRIX_HIP_DENSE; }
int32_t diffsol_ode_get_ode_solver(const OdeWrapper* ode) { if (!ode) { C_INVALID_ARG("ode is null"); return -1; } return ode->ode_solver; }
int32_t diffsol_ode_set_ode_solver(OdeWrapper* ode, int32_t value) {
  if (!ode) return C_INVALID_ARG("ode is null");
  if (!enum_valid(kOdeSolver, value)) return C_INVALID_ARG("invalid ode_solver_type");
  ode->ode_solver = value;
  return DIFFSOL_OK;
}
int32_t diffsol_ode_get_linear_solver(const OdeWrapper* ode) { if (!ode) { C_INVALID_ARG("ode is null"); return -1; } return ode->linear_solver; }
int32_t diffsol_ode_set_linear_solver(OdeWrapper* ode, int32_t value) {
  if (!ode) return C_INVALID_ARG("ode is null");
  if (!enum_valid(kLinear, value)) return C_INVALID_ARG("invalid linear_solver_type");
  ode->linear_solver = value;
  return DIFFSOL_OK;
}
#define DIFFSOL_SCALAR_ACCESSORS(field, expr_get, stmt_set)                                                            \
  int32_t diffsol_ode_get_##field(const OdeWrapper* ode, double* out_value) {                                         \
    if (!ode || !out_value) return C_INVALID_ARG("invalid arguments to diffsol_ode_get_" #field);                      \
    *out_value = expr_get;                                                                                             \
    return DIFFSOL_OK;                                                                                                 \
  }                                                                                                                    \
  int32_t diffsol_ode_set_##field(OdeWrapper* ode, double value) {                                                     \
    if (!ode) return C_INVALID_ARG("invalid arguments to diffsol_ode_set_" #field);                                    \
    stmt_set;                                                                                                          \
    return DIFFSOL_OK;                                                                                                 \
  }
DIFFSOL_SCALAR_ACCESSORS(rtol, ode->rtol, ode->rtol = value)
DIFFSOL_SCALAR_ACCESSORS(atol, ode->atol[0], ode->atol.assign(1, value))
DIFFSOL_SCALAR_ACCESSORS(t0, ode->t0, ode->t0 = value)
DIFFSOL_SCALAR_ACCESSORS(h0, ode->h0, ode->h0 = value)
#undef DIFFSOL_SCALAR_ACCESSORS

int32_t diffsol_ode_get_ensemble_mode(const OdeWrapper* ode) { if (!ode) { C_INVALID_ARG("ode is null"); return -1; } return ode->ensemble_mode; }
int32_t diffsol_ode_set_ensemble_mode(OdeWrapper* ode, int32_t mode) {
  if (!ode) return C_INVALID_ARG("ode is null");
  if (mode != DIFFSOL_ENSEMBLE_AUTO && mode != DIFFSOL_ENSEMBLE_LOCKSTEP && mode != DIFFSOL_ENSEMBLE_PER_MEMBER && mode != DIFFSOL_ENSEMBLE_WAVEFRONT)
    return C_INVALID_ARG("invalid ensemble mode");
  ode->ensemble_mode = mode;
  return DIFFSOL_OK;
}
int32_t diffsol_ode_get_dims(const OdeWrapper* ode, size_t* nstates, size_t* nparams, size_t* nout, size_t* nroots) {
  if (!ode) return C_INVALID_ARG("ode is null");
  if (nstates) *nstates = (size_t)ode->n;
  if (nparams) *nparams = ode->no_inputs ? 0 : (size_t)ode->np;
  if (nout) *nout = (size_t)ode->nout;
  if (nroots) *nroots = (size_t)ode->nroots;
  return DIFFSOL_OK;
}
int32_t diffsol_ode_set_atol_vector(OdeWrapper* ode, const double* atol_ptr, size_t atol_len) {
  if (!ode || !atol_ptr) return C_INVALID_ARG("invalid arguments to diffsol_ode_set_atol_vector");
  if (atol_len != 1 && atol_len != (size_t)ode->n) return C_ERROR("atol must have 1 or nstates entries");
  ode->atol.assign(atol_ptr, atol_ptr + atol_len);
  return DIFFSOL_OK;
}

// ------------------------------------------------------------------ options
void diffsol_ode_options_free(OdeSolverOptions* options) { if (!options) { C_INVALID_ARG("options is null"); return; } delete options; }
void diffsol_ic_options_free(InitialConditionSolverOptions* options) { if (!options) { C_INVALID_ARG("options is null"); return; } delete options; }
#define DIFFSOL_DEFINE_OPTION(prefix, type, ctype, field, member, cast)                                                \
  int32_t prefix##_get_##field(const type* options, ctype* out_value) {                                                \
    if (!options || !out_value) return C_INVALID_ARG("invalid arguments to " #prefix "_get_" #field);                  \
    *out_value = (ctype)options->s->o.member;                                                                          \
    return DIFFSOL_OK;                                                                                                 \
  }                                                                                                                    \
  int32_t prefix##_set_##field(type* options, ctype value) {                                                           \
    if (!options) return C_INVALID_ARG("invalid arguments to " #prefix "_set_" #field);                                \
    options->s->o.member = (cast)value;                                                                                \
    return DIFFSOL_OK;                                                                                                 \
  }
DIFFSOL_DEFINE_OPTION(diffsol_ode_options, OdeSolverOptions, size_t, max_nonlinear_solver_iterations, max_nonlinear_solver_iterations, int)
DIFFSOL_DEFINE_OPTION(diffsol_ode_options, OdeSolverOptions, size_t, max_error_test_failures, max_error_test_failures, int)
DIFFSOL_DEFINE_OPTION(diffsol_ode_options, OdeSolverOptions, size_t, update_jacobian_after_steps, update_jacobian_after_steps, int)
DIFFSOL_DEFINE_OPTION(diffsol_ode_options, OdeSolverOptions, size_t, update_rhs_jacobian_after_steps, update_rhs_jacobian_after_steps, int)
DIFFSOL_DEFINE_OPTION(diffsol_ode_options, OdeSolverOptions, double, threshold_to_update_jacobian, threshold_to_update_jacobian, double)
DIFFSOL_DEFINE_OPTION(diffsol_ode_options, OdeSolverOptions, double, threshold_to_update_rhs_jacobian, threshold_to_update_rhs_jacobian, double)
DIFFSOL_DEFINE_OPTION(diffsol_ode_options, OdeSolverOptions, double, min_timestep, min_timestep, double)
DIFFSOL_DEFINE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, int32_t, use_linesearch, ic_use_linesearch, int)
DIFFSOL_DEFINE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, size_t, max_linesearch_iterations, ic_max_linesearch_iterations, int)
DIFFSOL_DEFINE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, size_t, max_newton_iterations, ic_max_newton_iterations, int)
DIFFSOL_DEFINE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, size_t, max_linear_solver_setups, ic_max_linear_solver_setups, int)
DIFFSOL_DEFINE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, double, step_reduction_factor, ic_step_reduction_factor, double)
DIFFSOL_DEFINE_OPTION(diffsol_ic_options, InitialConditionSolverOptions, double, armijo_constant, ic_armijo_constant, double)
#undef DIFFSOL_DEFINE_OPTION

// ------------------------------------------------------------------ solution_wrapper_c.rs
void diffsol_solution_wrapper_free(SolutionWrapper* solution) { if (!solution) { C_INVALID_ARG("solution wrapper is null"); return; } delete solution; }
int32_t diffsol_solution_wrapper_get_ys(const SolutionWrapper* solution, HostArray** out_array) {
  if (!solution || !out_array) return C_INVALID_ARG("invalid arguments to diffsol_solution_wrapper_get_ys");
  auto* a = new diffsol_host_array();
  a->data = solution->ys;
  const size_t nb = (size_t)solution->nb, nr = (size_t)solution->nrows, nc = (size_t)solution->ncols, e = sizeof(double);
  if (nb == 1) { a->shape = {nr, nc}; a->strides = {e, e * nr}; }
  else { a->shape = {nr, nc, nb}; a->strides = {e * nb, e * nb * nr, e}; }
  *out_array = a;
  return DIFFSOL_OK;
}
int32_t diffsol_solution_wrapper_get_ts(const SolutionWrapper* solution, HostArray** out_array) {
  if (!solution || !out_array) return C_INVALID_ARG("invalid arguments to diffsol_solution_wrapper_get_ts");
  *out_array = vector_array(std::vector<double>(solution->ts));
  return DIFFSOL_OK;
}
// solution_wrapper_c.rs:101-125: one array per parameter, shaped like ys; the list (not the arrays) is released with diffsol_host_array_list_free (ode_c.rs:163-171)
int32_t diffsol_solution_wrapper_get_sens(const SolutionWrapper* solution, HostArray*** out_sens, size_t* out_sens_len) {
  if (!solution || !out_sens || !out_sens_len) return C_INVALID_ARG("invalid arguments to diffsol_solution_wrapper_get_sens");
  const size_t np = solution->sens.size();
  HostArray** list = np ? (HostArray**)std::malloc(np * sizeof(HostArray*)) : nullptr;
  const size_t nb = (size_t)solution->nb, nr = (size_t)solution->nrows, nc = (size_t)solution->ncols, e = sizeof(double);
  for (size_t j = 0; j < np; ++j) {
    auto* a = new diffsol_host_array();
    a->data = solution->sens[j];
    if (nb == 1) { a->shape = {nr, nc}; a->strides = {e, e * nr}; }
    else { a->shape = {nr, nc, nb}; a->strides = {e * nb, e * nb * nr, e}; }
    list[j] = a;
  }
  *out_sens = list;
  *out_sens_len = np;
  return DIFFSOL_OK;
}
void diffsol_host_array_list_free(HostArray** list, size_t len) {  // ode_c.rs:163-171: the list only — its arrays are freed one by one
  (void)len;
  if (!list) { C_INVALID_ARG("host array list is null"); return; }
  std::free(list);
}
int64_t diffsol_solution_wrapper_get_member_info(const SolutionWrapper* solution, int32_t* status, double* t_root, int32_t* root_index, int32_t* ncols) {
  if (!solution) { C_INVALID_ARG("solution wrapper is null"); return -1; }
  const size_t nb = (size_t)solution->nb;
  if (status) std::memcpy(status, solution->status.data(), nb * sizeof(int32_t));
  if (t_root) std::memcpy(t_root, solution->t_root.data(), nb * sizeof(double));
  if (root_index) std::memcpy(root_index, solution->root_index.data(), nb * sizeof(int32_t));
  if (ncols) std::memcpy(ncols, solution->member_cols.data(), nb * sizeof(int32_t));
  return solution->nb;
}

}  // extern "C"
