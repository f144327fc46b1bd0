// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the reference's test/bench models as batched OdeEquations.
// Each per-system function cites the reference closure it follows (paths relative to /root/reference).
// The equations object mirrors what OdeBuilder::build produces (crates/diffsol/src/ode_solver/builder.rs:1784-1893):
// rhs (+ jac_mul), optional mass gemv, init, optional root; parameters `p` are batch-major
// (np per batch member, crates/diffsol/src/ode_equations/test_models/exponential_decay.rs:300-311).
// Jacobian and mass matrices are assembled column by column from jac_mul / gemv with unit vectors,
// exactly like the default `_default_jacobian_inplace` / `_default_matrix_inplace`
// (crates/diffsol/src/op/nonlinear_op.rs:211-219, crates/diffsol/src/op/linear_op.rs:41-50), so the
// OpStatistics counters (number_of_calls / jac_muls / matrix_evals, crates/diffsol/src/op/mod.rs:95-128)
// can be compared with the reference's insta snapshots.
#pragma once
#include "oracle_la.hpp"
#include <memory>

namespace orc {

struct OpStats { long calls = 0, jac_muls = 0, matrix_evals = 0; };

enum ModelId : int {
  MODEL_EXPONENTIAL_DECAY = 0,            // n=2, p=[k,y0]
  MODEL_EXPONENTIAL_DECAY_ALGEBRAIC = 1,  // n=3, p=[k], M=diag(1,1,0), init (1,1,0)  (non-batched variant)
  MODEL_EXPONENTIAL_DECAY_ALGEBRAIC_BATCHED = 2,  // same, init (1,1,1) (batched variant)
  MODEL_ROBERTSON_ODE = 3,                // n=3*ngroups, p=[k1,k2,k3]
  MODEL_ROBERTSON_DAE = 4,                // n=3, M=diag(1,1,0)
  MODEL_DYDT_Y2 = 5,                      // n=size, y0=-200
  MODEL_GAUSSIAN_DECAY = 6,               // n=size, p=[a]*size
  MODEL_HEAT1D = 7,                       // n=size(mgrid+1), p=[D], finite differences, triangle IC
  MODEL_RLC = 8,                          // n=4 DAE, p=[R,L,C,V0,omega,ithresh]
  MODEL_EXPONENTIAL_DECAY_ROOT = 9,       // exponential decay with root x0-0.6
};

struct Model {
  int n = 0, np = 0, nroots = 0;
  bool has_mass = false;
  virtual ~Model() = default;
  virtual void rhs(const double* x, const double* p, double t, double* y) const = 0;
  virtual void jac_mul(const double* x, const double* p, double t, const double* v, double* y) const = 0;
  // y = M x + beta y
  virtual void mass(const double*, const double*, double, double, double*) const {}
  virtual void init(const double* p, double t, double* y) const = 0;
  virtual void root(const double*, const double*, double, double*) const {}
};

// crates/diffsol/src/ode_equations/test_models/exponential_decay.rs:14-21 (rhs), :54-61 (jac), :72-81 (init), :98-100 (root)
struct ExponentialDecay : Model {
  explicit ExponentialDecay(bool with_root = false) { n = 2; np = 2; nroots = with_root ? 1 : 0; }
  void rhs(const double* x, const double* p, double, double* y) const override { for (int i = 0; i < n; ++i) y[i] = x[i] * (-p[0]); }
  void jac_mul(const double*, const double* p, double, const double* v, double* y) const override { for (int i = 0; i < n; ++i) y[i] = v[i] * (-p[0]); }
  void init(const double* p, double, double* y) const override { for (int i = 0; i < n; ++i) y[i] = p[1]; }
  void root(const double* x, const double*, double, double* g) const override { g[0] = x[0] - 0.6; }
};

// crates/diffsol/src/ode_equations/test_models/exponential_decay_with_algebraic.rs:18-23 (rhs), :64-75 (jac),
// :94-105 (mass), :122-126 (init), :267-276 (batched init)
struct ExponentialDecayAlgebraic : Model {
  bool batched_init;
  explicit ExponentialDecayAlgebraic(bool batched_init_) : batched_init(batched_init_) { n = 3; np = 1; has_mass = true; }
  void rhs(const double* x, const double* p, double, double* y) const override {
    for (int i = 0; i < n; ++i) y[i] = x[i] * (-p[0]);
    y[n - 1] = x[n - 1] - x[n - 2];
  }
  void jac_mul(const double*, const double* p, double, const double* v, double* y) const override {
    for (int i = 0; i < n; ++i) y[i] = v[i] * (-p[0]);
    y[n - 1] = v[n - 1] - v[n - 2];
  }
  void mass(const double* x, const double*, double, double beta, double* y) const override {
    double yn = beta * y[n - 1];
    for (int i = 0; i < n; ++i) y[i] = 1.0 * x[i] + beta * y[i];
    y[n - 1] = yn;
  }
  void init(const double*, double, double* y) const override { y[0] = 1.0; y[1] = 1.0; y[2] = batched_init ? 1.0 : 0.0; }
};

// crates/diffsol/src/ode_equations/test_models/robertson_ode.rs:71-90 (rhs, jac_mul), :92-101 (init)
struct RobertsonOde : Model {
  int ngroups;
  explicit RobertsonOde(int ngroups_) : ngroups(ngroups_) { n = 3 * ngroups_; np = 3; }
  void rhs(const double* x, const double* p, double, double* y) const override {
    for (int ig = 0; ig < ngroups; ++ig) {
      int i = ig * 3;
      y[i] = -p[0] * x[i] + p[1] * x[i + 1] * x[i + 2];
      y[i + 1] = p[0] * x[i] - p[1] * x[i + 1] * x[i + 2] - p[2] * x[i + 1] * x[i + 1];
      y[i + 2] = p[2] * x[i + 1] * x[i + 1];
    }
  }
  void jac_mul(const double* x, const double* p, double, const double* v, double* y) const override {
    for (int ig = 0; ig < ngroups; ++ig) {
      int i = ig * 3;
      y[i] = -p[0] * v[i] + p[1] * v[i + 1] * x[i + 2] + p[1] * x[i + 1] * v[i + 2];
      y[i + 1] = p[0] * v[i] - p[1] * v[i + 1] * x[i + 2] - p[1] * x[i + 1] * v[i + 2] - 2.0 * p[2] * x[i + 1] * v[i + 1];
      y[i + 2] = 2.0 * p[2] * x[i + 1] * v[i + 1];
    }
  }
  void init(const double*, double, double* y) const override {
    for (int ig = 0; ig < ngroups; ++ig) { y[3 * ig] = 1.0; y[3 * ig + 1] = 0.0; y[3 * ig + 2] = 0.0; }
  }
};

// crates/diffsol/src/ode_equations/test_models/robertson.rs:60-94
struct RobertsonDae : Model {
  RobertsonDae() { n = 3; np = 3; has_mass = true; }
  void rhs(const double* x, const double* p, double, double* y) const override {
    y[0] = -p[0] * x[0] + p[1] * x[1] * x[2];
    y[1] = p[0] * x[0] - p[1] * x[1] * x[2] - p[2] * x[1] * x[1];
    y[2] = x[0] + x[1] + x[2] - 1.0;
  }
  void jac_mul(const double* x, const double* p, double, const double* v, double* y) const override {
    y[0] = -p[0] * v[0] + p[1] * v[1] * x[2] + p[1] * x[1] * v[2];
    y[1] = p[0] * v[0] - p[1] * v[1] * x[2] - p[1] * x[1] * v[2] - 2.0 * p[2] * x[1] * v[1];
    y[2] = v[0] + v[1] + v[2];
  }
  void mass(const double* x, const double*, double, double beta, double* y) const override {
    y[0] = x[0] + beta * y[0];
    y[1] = x[1] + beta * y[1];
    y[2] = beta * y[2];
  }
  void init(const double*, double, double* y) const override { y[0] = 1.0; y[1] = 0.0; y[2] = 0.0; }
};

// crates/diffsol/src/ode_equations/test_models/dydt_y2.rs:9-19
struct DydtY2 : Model {
  explicit DydtY2(int size) { n = size; np = 0; }
  void rhs(const double* x, const double*, double, double* y) const override { for (int i = 0; i < n; ++i) y[i] = x[i] * x[i]; }
  void jac_mul(const double* x, const double*, double, const double* v, double* y) const override { for (int i = 0; i < n; ++i) y[i] = v[i] * x[i] * 2.0; }
  void init(const double*, double, double* y) const override { for (int i = 0; i < n; ++i) y[i] = -200.0; }
};

// crates/diffsol/src/ode_equations/test_models/gaussian_decay.rs:12-23
struct GaussianDecay : Model {
  explicit GaussianDecay(int size) { n = size; np = size; }
  void rhs(const double* x, const double* p, double t, double* y) const override { for (int i = 0; i < n; ++i) y[i] = x[i] * p[i] * (-t); }
  void jac_mul(const double*, const double* p, double t, const double* v, double* y) const override { for (int i = 0; i < n; ++i) y[i] = v[i] * p[i] * (-t); }
  void init(const double*, double, double* y) const override { for (int i = 0; i < n; ++i) y[i] = 1.0; }
};

// 1-D heat equation by second-order finite differences on n interior points, Dirichlet 0 boundaries,
// h = 1/(n+1), triangle initial condition, F = D*(A u)/h^2 with A = tridiag(1,-2,1).
// crates/diffsol/src/ode_equations/test_models/heat1d.rs:16-52 (DiffSL text: h, A_ij, u_i, F_i), examples/pde-heat/src/main.rs:15-40.
struct Heat1d : Model {
  double h;
  explicit Heat1d(int size) { n = size; np = 1; h = 1.0 / (double)(size + 1); }
  void stencil(const double* u, const double* p, double* y) const {
    for (int i = 0; i < n; ++i) {
      double left = i > 0 ? u[i - 1] : 0.0, right = i + 1 < n ? u[i + 1] : 0.0;
      double heat = left + (-2.0) * u[i] + right;
      y[i] = p[0] * heat / (h * h);
    }
  }
  void rhs(const double* x, const double* p, double, double* y) const override { stencil(x, p, y); }
  void jac_mul(const double*, const double* p, double, const double* v, double* y) const override { stencil(v, p, y); }
  void init(const double*, double, double* y) const override {
    for (int i = 0; i < n; ++i) { double x = (double)(i + 1) * h; y[i] = x < 0.5 ? 2.0 * x : 2.0 * (1.0 - x); }
  }
};

// Series RLC circuit DAE, examples/electrical-circuits/src/main.rs:10-41:
// u=(iR,iL,iC,V), M=diag(0,1,0,1), F=(V-R*iR, (Vs-V)/L, iL-iR-iC, iC/C), Vs=V0*sin(omega*t).
// p=[R,L,C,V0,omega,ithresh]; optional root g = iR - ithresh (SURVEY §8(d) C5 adds it to exercise RootFinder).
struct Rlc : Model {
  explicit Rlc(bool with_root) { n = 4; np = 6; has_mass = true; nroots = with_root ? 1 : 0; }
  void rhs(const double* x, const double* p, double t, double* y) const override {
    double vs = p[3] * std::sin(p[4] * t);
    y[0] = x[3] - p[0] * x[0];
    y[1] = (vs - x[3]) / p[1];
    y[2] = x[1] - x[0] - x[2];
    y[3] = x[2] / p[2];
  }
  void jac_mul(const double*, const double* p, double, const double* v, double* y) const override {
    y[0] = v[3] - p[0] * v[0];
    y[1] = (-v[3]) / p[1];
    y[2] = v[1] - v[0] - v[2];
    y[3] = v[2] / p[2];
  }
  void mass(const double* x, const double*, double, double beta, double* y) const override {
    y[0] = beta * y[0];
    y[1] = x[1] + beta * y[1];
    y[2] = beta * y[2];
    y[3] = x[3] + beta * y[3];
  }
  void init(const double*, double, double* y) const override { y[0] = y[1] = y[2] = y[3] = 0.0; }
  void root(const double* x, const double* p, double, double* g) const override { g[0] = x[0] - p[5]; }
};

inline std::unique_ptr<Model> make_model(int id, int size) {
  switch (id) {
    case MODEL_EXPONENTIAL_DECAY: return std::make_unique<ExponentialDecay>(false);
    case MODEL_EXPONENTIAL_DECAY_ROOT: return std::make_unique<ExponentialDecay>(true);
    case MODEL_EXPONENTIAL_DECAY_ALGEBRAIC: return std::make_unique<ExponentialDecayAlgebraic>(false);
    case MODEL_EXPONENTIAL_DECAY_ALGEBRAIC_BATCHED: return std::make_unique<ExponentialDecayAlgebraic>(true);
    case MODEL_ROBERTSON_ODE: return std::make_unique<RobertsonOde>(size <= 0 ? 1 : size);
    case MODEL_ROBERTSON_DAE: return std::make_unique<RobertsonDae>();
    case MODEL_DYDT_Y2: return std::make_unique<DydtY2>(size);
    case MODEL_GAUSSIAN_DECAY: return std::make_unique<GaussianDecay>(size);
    case MODEL_HEAT1D: return std::make_unique<Heat1d>(size);
    case MODEL_RLC: return std::make_unique<Rlc>(size != 0);
    default: throw std::runtime_error("oracle: unknown model id");
  }
}

// Batched OdeEquations (rhs/jac/mass/init/root applied per batch member with that member's parameters).
struct Eqn {
  std::unique_ptr<Model> model;
  int nb = 1;
  V p;  // np x nb
  mutable OpStats rhs_stats;
  Eqn(std::unique_ptr<Model> m, int nb_, const std::vector<double>& p_) : model(std::move(m)), nb(nb_), p(model->np, nb_) {
    if ((int)p_.size() != model->np * nb_) throw std::runtime_error("oracle: parameter vector has wrong length");
    p.d = p_;
  }
  int n() const { return model->n; }
  bool has_mass() const { return model->has_mass; }
  const double* pb(int b) const { return p.d.data() + (size_t)b * model->np; }
  void rhs(const V& x, double t, V& y) const {
    rhs_stats.calls++;
    for (int b = 0; b < nb; ++b) model->rhs(&x.d[(size_t)b * x.n], pb(b), t, &y.d[(size_t)b * y.n]);
  }
  void jac_mul(const V& x, double t, const V& v, V& y) const {
    rhs_stats.jac_muls++;
    for (int b = 0; b < nb; ++b) model->jac_mul(&x.d[(size_t)b * x.n], pb(b), t, &v.d[(size_t)b * v.n], &y.d[(size_t)b * y.n]);
  }
  void jacobian(const V& x, double t, M& J) const {  // op/nonlinear_op.rs:211-219 via Closure (counts a matrix eval, op/closure.rs:140-146)
    rhs_stats.matrix_evals++;
    int n_ = n();
    V v(n_, nb), col(n_, nb);
    for (int j = 0; j < n_; ++j) {
      for (int b = 0; b < nb; ++b) v.at(b, j) = 1.0;
      jac_mul(x, t, v, col);
      J.set_column(j, col);
      for (int b = 0; b < nb; ++b) v.at(b, j) = 0.0;
    }
  }
  void mass_gemv(const V& x, double t, double beta, V& y) const {
    for (int b = 0; b < nb; ++b) model->mass(&x.d[(size_t)b * x.n], pb(b), t, beta, &y.d[(size_t)b * y.n]);
  }
  void mass_matrix(double t, M& Mm) const {  // op/linear_op.rs:41-50
    int n_ = n();
    V v(n_, nb), col(n_, nb);
    for (int j = 0; j < n_; ++j) {
      for (int b = 0; b < nb; ++b) v.at(b, j) = 1.0;
      mass_gemv(v, t, 0.0, col);
      Mm.set_column(j, col);
      for (int b = 0; b < nb; ++b) v.at(b, j) = 0.0;
    }
  }
  void init(double t, V& y) const {
    for (int b = 0; b < nb; ++b) model->init(pb(b), t, &y.d[(size_t)b * y.n]);
  }
  void root(const V& x, double t, V& g) const {
    for (int b = 0; b < nb; ++b) model->root(&x.d[(size_t)b * x.n], pb(b), t, &g.d[(size_t)b * g.n]);
  }
};

}  // namespace orc
