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
  MODEL_SPM = 10,                         // single-particle battery model, n=2+2*size (size shells per particle, 20 in spm.ds), p=[I]
  MODEL_HEAT2D = 11,                      // 2-D heat equation on a size x size grid, n=size^2, DAE (boundary rows algebraic), p=[diffusion scale] (1 = the reference), band size
  MODEL_FOODWEB = 12,                     // predator-prey food web on a size x size grid, n=2 size^2, DAE (predators algebraic), p=[alpha, beta] ((50, 1000) = the reference), band 2 size
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
  // Reset operator of hybrid models (OdeEquations::reset, DiffSL reset_i): the state after an event, y_new = reset(y, t)
  bool has_reset = false;
  virtual void reset(const double*, const double*, double, double*) const { throw std::runtime_error("oracle: model has no reset operator"); }
  // forward sensitivities (OdeEquationsImplicitSens): y = (df/dp)(x, p, t) v  and  y = (dy0/dp)(p, t) v, v of length np
  // (NonLinearOpSens::sens_mul_inplace op/nonlinear_op.rs:51-53, ConstantOpSens::sens_mul_inplace)
  bool has_sens = false;
  virtual void sens_mul(const double*, const double*, double, const double*, double*) const { throw std::runtime_error("oracle: model has no parameter sensitivities"); }
  virtual void init_sens_mul(const double*, double, const double*, double*) const { throw std::runtime_error("oracle: model has no parameter sensitivities"); }
};

// crates/diffsol/src/ode_equations/test_models/exponential_decay.rs:14-21 (rhs), :54-61 (jac), :72-81 (init), :98-100 (root)
struct ExponentialDecay : Model {
  explicit ExponentialDecay(bool with_root = false) { n = 2; np = 2; nroots = with_root ? 1 : 0; has_sens = true; }
  // exponential_decay.rs:33-36 (df/dp v = -x v_k) and :90-93 (dy0/dp v = (v_y0, v_y0))
  void sens_mul(const double* x, const double*, double, const double* v, double* y) const override { for (int i = 0; i < n; ++i) y[i] = x[i] * (-v[0]); }
  void init_sens_mul(const double*, double, const double* v, double* y) const override { y[0] = v[1]; y[1] = v[1]; }
  void rhs(const double* x, const double* p, double, double* y) const override { for (int i = 0; i < n; ++i) y[i] = x[i] * (-p[0]); }
  void jac_mul(const double*, const double* p, double, const double* v, double* y) const override { for (int i = 0; i < n; ++i) y[i] = v[i] * (-p[0]); }
  void init(const double* p, double, double* y) const override { for (int i = 0; i < n; ++i) y[i] = p[1]; }
  void root(const double* x, const double*, double, double* g) const override { g[0] = x[0] - 0.6; }
};

// crates/diffsol/src/ode_equations/test_models/exponential_decay_with_algebraic.rs:18-23 (rhs), :64-75 (jac),
// :94-105 (mass), :122-126 (init), :267-276 (batched init)
struct ExponentialDecayAlgebraic : Model {
  bool batched_init;
  explicit ExponentialDecayAlgebraic(bool batched_init_) : batched_init(batched_init_) { n = 3; np = 1; has_mass = true; has_sens = true; }
  // :33-44 (sens: y = x * (-v[0]), last = 0), :128-135 (init_sens: zeros)
  void sens_mul(const double* x, const double*, double, const double* v, double* y) const override {
    for (int i = 0; i < n; ++i) y[i] = x[i] * (-v[0]);
    y[n - 1] = 0.0;
  }
  void init_sens_mul(const double*, double, const double*, double* y) const override { for (int i = 0; i < n; ++i) y[i] = 0.0; }
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
  explicit RobertsonOde(int ngroups_) : ngroups(ngroups_) { n = 3 * ngroups_; np = 3; has_sens = true; }
  // robertson_ode_with_sens.rs:38-42 (df/dp v) and :50 (dy0/dp v = 0)
  void sens_mul(const double* x, const double*, double, const double* v, double* y) const override {
    for (int ig = 0; ig < ngroups; ++ig) {
      int i = ig * 3;
      y[i] = -v[0] * x[i] + v[1] * x[i + 1] * x[i + 2];
      y[i + 1] = v[0] * x[i] - v[1] * x[i + 1] * x[i + 2] - v[2] * x[i + 1] * x[i + 1];
      y[i + 2] = v[2] * x[i + 1] * x[i + 1];
    }
  }
  void init_sens_mul(const double*, double, const double*, double* y) const override { for (int i = 0; i < n; ++i) y[i] = 0.0; }
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
  RobertsonDae() { n = 3; np = 3; has_mass = true; has_sens = true; }
  // robertson.rs:73-77 (sens_mul), :91-93 (init_sens: zeros)
  void sens_mul(const double* x, const double*, double, const double* v, double* y) const override {
    y[0] = -v[0] * x[0] + v[1] * x[1] * x[2];
    y[1] = v[0] * x[0] - v[1] * x[1] * x[2] - v[2] * x[1] * x[1];
    y[2] = 0.0;
  }
  void init_sens_mul(const double*, double, const double*, double* y) const override { for (int i = 0; i < n; ++i) y[i] = 0.0; }
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
    double vs = p[3] * dsh_det_sin(p[4] * t);
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

// Single-particle lithium-ion model of book/src/primer/src/spm.ds (used by examples/physics-based-battery-simulation/src/main.rs:12-45):
//   u = (discharge capacity, throughput capacity, x_neg[0..m), x_pos[0..m))   m = 20 radial finite volumes per particle
//   F = (I/3600, |I|/3600, A_neg x_neg + flux_neg(I) e_{m-1}, A_pos x_pos + flux_pos(I) e_{m-1}),  identity mass,
//   stop = (V - 3.105, 4.1 - V),  V(u, I) = terminal voltage from the surface concentrations (linear extrapolation of the two outer shells).
// spm.ds lists the discretisation as literal sparse tensors (constant6_ij, constant7_ij, ...); they are the spherical finite-volume
// Laplacian on m uniform shells scaled by D/R^2 and are generated here from that formula (lower_i = 3 i^2 / ((i+1)^3 - i^3) / dr^2,
// upper_i = 3 (i+1)^2 / ((i+1)^3 - i^3) / dr^2, no outer flux) — equal to the file's literals to rounding (<= 2 ulp).  The scalar
// coefficients (flux scalings, exchange-current prefactors, open-circuit-potential fits, voltage cut-offs) are the file's values.
struct SpmCoeffs {
  static constexpr double kInvHour = 0.0002777777777777778;          // constant4 / F_i rows 0,1
  static constexpr double kDiffNeg = 0.39e-3, kDiffPos = 1.0e-3;      // D/R^2 of constant7_ij / constant6_ij
  static constexpr double kFluxNeg = 3.2835305549534856e-12 * -520607810.21082705;  // constant0_i * (constant3 * I)
  static constexpr double kFluxPos = 4.106800547504748e-12 * 243644455.17866704;    // constant1_i * (constant2 * I)
  static constexpr double kSurfIn = -0.4999999999999983, kSurfOutPos = 1.4999999999999982, kSurfOutNeg = 1.4999999999999984;  // constant5_ij / constant9_ij
  static constexpr double kCmaxPos = 51217.9257309275, kCmaxNeg = 24983.2619938437;
  static constexpr double kThermal2 = 0.05138515824298745;            // 2RT/F
};
template <class T> inline T spm_clamp(T v, T lo, T hi) { return v < hi ? (v > lo ? v : lo) : hi; }  // max(min(v, hi), lo)
// open-circuit potentials (the tanh fits written out in spm.ds out_i)
inline double spm_ocp_pos(double s) {
  return 2.16216 + 0.07645 * dsh_det_tanh(30.834 - 57.858397200000006 * s) + 2.1581 * dsh_det_tanh(52.294 - 53.412228 * s) - 0.14169 * dsh_det_tanh(11.0923 - 21.0852666 * s) +
         0.2051 * dsh_det_tanh(1.4684 - 5.829105600000001 * s) + 0.2531 * dsh_det_tanh(4.291641337386018 - 8.069908814589667 * s) - 0.02167 * dsh_det_tanh(-87.5 + 177.0 * s) +
         1e-06 * (1.0 / s + 1.0 / (-1.0 + s));
}
inline double spm_ocp_neg(double s) {
  return 0.194 + 1.5 * dsh_det_exp(-120.0 * s) + 0.0351 * dsh_det_tanh(-3.44578313253012 + 12.048192771084336 * s) - 0.0045 * dsh_det_tanh(-7.1344537815126055 + 8.403361344537815 * s) -
         0.035 * dsh_det_tanh(-18.466 + 20.0 * s) - 0.0147 * dsh_det_tanh(-14.705882352941176 + 29.41176470588235 * s) - 0.102 * dsh_det_tanh(-1.3661971830985917 + 7.042253521126761 * s) -
         0.022 * dsh_det_tanh(-54.8780487804878 + 60.975609756097555 * s) - 0.011 * dsh_det_tanh(-5.486725663716814 + 44.24778761061947 * s) +
         0.0155 * dsh_det_tanh(-3.6206896551724133 + 34.48275862068965 * s) + 1e-06 * (1.0 / s + 1.0 / (-1.0 + s));
}
// terminal voltage from the two outer shells of each particle (spm.ds varying2..5 and out_i)
inline double spm_voltage(double neg_in, double neg_out, double pos_in, double pos_out, double current) {
  using C = SpmCoeffs;
  const double sp = C::kSurfIn * pos_in + C::kSurfOutPos * pos_out, sn = C::kSurfIn * neg_in + C::kSurfOutNeg * neg_out;
  const double cp = spm_clamp(-25608.96286546366 * pos_in + 76826.88859639116 * pos_out, 0.000512179257309275, 51217.92521874824);  // constant10_ij
  const double cn = spm_clamp(-12491.630996921805 * neg_in + 37474.892990765504 * neg_out, 0.000249832619938437, 24983.261744011077);  // constant8_ij
  const double stp = spm_clamp(sp, 1e-10, 0.9999999999), stn = spm_clamp(sn, 1e-10, 0.9999999999);
  const double eta_p = C::kThermal2 * dsh_det_asinh((-2.3508116177110145 * current) / (2.0 * ((1.8973665961010275e-05 * std::sqrt(cp)) * std::sqrt(C::kCmaxPos - cp))));
  const double eta_n = C::kThermal2 * dsh_det_asinh((1.9590096814258458 * current) / (2.0 * ((0.0006324555320336759 * std::sqrt(cn)) * std::sqrt(C::kCmaxNeg - cn))));
  return (eta_p + spm_ocp_pos(stp)) - (eta_n + spm_ocp_neg(stn));
}
struct Spm : Model {
  int m;
  explicit Spm(int shells) : m(shells <= 0 ? 20 : shells) { n = 2 + 2 * m; np = 1; nroots = 2; }
  // row i of the spherical finite-volume Laplacian applied to x (m shells), scaled by s
  double diffusion(const double* x, int i, double s) const {
    const double dr = 1.0 / (double)m, i0 = (double)i, i1 = (double)(i + 1);
    const double vol = i1 * i1 * i1 - i0 * i0 * i0;
    const double lower = 3.0 * i0 * i0 / vol / (dr * dr) * s, upper = i + 1 < m ? 3.0 * i1 * i1 / vol / (dr * dr) * s : 0.0;
    double acc = (-(lower + upper)) * x[i];
    if (i > 0) acc += lower * x[i - 1];
    if (i + 1 < m) acc += upper * x[i + 1];
    return acc;
  }
  void apply(const double* x, const double* p, bool jac, double* y) const {
    using C = SpmCoeffs;
    y[0] = jac ? 0.0 : C::kInvHour * p[0];
    y[1] = jac ? 0.0 : C::kInvHour * std::fabs(p[0]);
    for (int i = 0; i < m; ++i) {
      double fn = diffusion(x + 2, i, C::kDiffNeg), fp = diffusion(x + 2 + m, i, C::kDiffPos);
      if (!jac && i == m - 1) { fn += C::kFluxNeg * p[0]; fp += C::kFluxPos * p[0]; }
      y[2 + i] = fn;
      y[2 + m + i] = fp;
    }
  }
  void rhs(const double* x, const double* p, double, double* y) const override { apply(x, p, false, y); }
  void jac_mul(const double*, const double* p, double, const double* v, double* y) const override { apply(v, p, true, y); }
  void init(const double*, double, double* y) const override {
    y[0] = y[1] = 0.0;
    for (int i = 0; i < m; ++i) { y[2 + i] = 0.8000000000000016; y[2 + m + i] = 0.6000000000000001; }
  }
  void root(const double* x, const double* p, double, double* g) const override {
    const double v = spm_voltage(x[2 + m - 2], x[2 + m - 1], x[2 + 2 * m - 2], x[2 + 2 * m - 1], p[0]);
    g[0] = -3.105 + v;
    g[1] = 4.1 - v;
  }
};

// 2-D heat equation, 5-point differences on an m x m grid of the unit square, Dirichlet 0 boundary values kept as algebraic equations (res_i = u_i):
// crates/diffsol/src/ode_equations/test_models/heat2d.rs:105-123 (rhs), :125-149 (jac_mul), :151-183 (init), :185-205 (mass).  The reference has no parameter;
// p[0] scales the diffusion coefficient (p[0] = 1: p[0] / (dx dx) is the reference's 1 / (dx dx) to the bit) so that an ensemble has distinct members.
// Half-bandwidth of f_y: m (the 5-point stencil's +-m neighbours) — the `Sunmatrix_Band` case of book/src/benchmarks/sundials.md:27-28.
struct Heat2d : Model {
  int m;
  explicit Heat2d(int mgrid) : m(mgrid) { if (m < 3) throw std::runtime_error("oracle: heat2d needs a grid of at least 3 x 3"); n = m * m; np = 1; has_mass = true; }
  void stencil(const double* u, const double* p, double* y) const {
    for (int i = 0; i < n; ++i) y[i] = u[i];  // y.copy_from(x): the boundary equations
    const double mm = (double)m, four = 4.0;
    const double dx = 1.0 / (mm - 1.0);
    const double coeff = p[0] / (dx * dx);
    for (int j = 1; j < m - 1; ++j) {
      const int offset = m * j;
      for (int i = 1; i < m - 1; ++i) {
        const int loc = offset + i;
        y[loc] = coeff * (u[loc - 1] + u[loc + 1] + u[loc - m] + u[loc + m] - four * u[loc]);
      }
    }
  }
  void rhs(const double* x, const double* p, double, double* y) const override { stencil(x, p, y); }
  void jac_mul(const double*, const double* p, double, const double* v, double* y) const override { stencil(v, p, y); }
  void mass(const double* x, const double*, double, double beta, double* y) const override {
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < m; ++i) {
        const int loc = m * j + i;
        if (j == 0 || j == m - 1 || i == 0 || i == m - 1) y[loc] *= beta;
        else y[loc] = x[loc] + beta * y[loc];
      }
  }
  void init(const double*, double, double* uu) const override {
    const double mm = (double)m, one = 1.0, sixteen = 16.0;
    const double dx = one / (mm - one);
    for (int j = 0; j < m; ++j) {
      const double yfact = dx * (double)j;
      for (int i = 0; i < m; ++i) {
        const double xfact = dx * (double)i;
        const int loc = m * j + i;
        uu[loc] = (j == 0 || j == m - 1 || i == 0 || i == m - 1) ? 0.0 : sixteen * xfact * (one - xfact) * yfact * (one - yfact);
      }
    }
  }
  // heat2d.rs:200-205: out = (||x||_2 dx)^2 — not part of the integration (the tests below form it from the states)
};

// Food web: predator-prey interaction with diffusion on the unit square (the Sundials idaFoodWeb example), one prey and one predator species, nx x nx grid,
// species interleaved (loc = 2 jx + 2 nx jy), homogeneous Neumann boundaries by mirror points, predators algebraic:
// crates/diffsol/src/ode_equations/test_models/foodweb.rs:9-22 (constants), :232-273 (coefficients), :342-366 (init), :419-494 (rhs), :502-582 (jac_mul), :639-655 (mass).
// p = (ALPHA, BETA) of the growth-rate field b(x, y) = 1 + alpha x y + beta sin(4 pi x) sin(4 pi y) ((50, 1000) in the reference).  The sines use the portable
// dsh_det_sin of include/diffsol_detpow.h (the device has no libm; within 1 ulp of it).  Half-bandwidth of f_y: 2 nx.
struct Foodweb : Model {
  int nx;
  static constexpr int NS = 2;
  double acoef[2][2], bcoef[2], cox[2], coy[2];
  explicit Foodweb(int nx_) : nx(nx_) {
    if (nx < 2) throw std::runtime_error("oracle: foodweb needs a grid of at least 2 x 2");
    n = NS * nx * nx; np = 2; has_mass = true;
    const double AA = 1.0, EE = 10000.0, GG = 0.5e-6, BB = 1.0, DPREY = 1.0, DPRED = 0.05;
    const double DX = 1.0 / ((double)nx - 1.0), DY = 1.0 / ((double)nx - 1.0);
    acoef[0][1] = -GG; acoef[1][0] = EE; acoef[0][0] = -AA; acoef[1][1] = -AA;
    bcoef[0] = BB; bcoef[1] = -BB;
    cox[0] = DPREY / (DX * DX); cox[1] = DPRED / (DX * DX);
    coy[0] = DPREY / (DY * DY); coy[1] = DPRED / (DY * DY);
  }
  void apply(const double* x, const double* p, const double* v, double* y) const {  // v == nullptr: f(x); else J(x) v
    const int nsmx = NS * nx;
    const double dx = 1.0 / ((double)nx - 1.0), dy = 1.0 / ((double)nx - 1.0);
    const double* u = v ? v : x;
    for (int jy = 0; jy < nx; ++jy) {
      const double yy = (double)jy * dy;
      const int idyu = jy != nx - 1 ? nsmx : -nsmx, idyl = jy != 0 ? nsmx : -nsmx;
      for (int jx = 0; jx < nx; ++jx) {
        const double xx = (double)jx * dx;
        const int idxu = jx != nx - 1 ? NS : -NS, idxl = jx != 0 ? NS : -NS;
        const int loc = NS * jx + nsmx * jy, locxu = loc + idxu, locxl = loc - idxl, locyu = loc + idyu, locyl = loc - idyl;
        double rates[NS], drates[NS];
        for (int is = 0; is < NS; ++is) {
          double dp = 0.0, ddp = 0.0;
          for (int js = 0; js < NS; ++js) { dp += acoef[is][js] * x[loc + js]; if (v) ddp += acoef[is][js] * v[loc + js]; }
          rates[is] = dp; drates[is] = ddp;
        }
        const double fac = 1.0 + p[0] * xx * yy + p[1] * dsh_det_sin(4.0 * 3.14159265358979323846 * xx) * dsh_det_sin(4.0 * 3.14159265358979323846 * yy);
        for (int is = 0; is < NS; ++is) {
          if (v) drates[is] = x[loc + is] * drates[is] + v[loc + is] * (bcoef[is] * fac + rates[is]);
          else rates[is] = x[loc + is] * (bcoef[is] * fac + rates[is]);
        }
        for (int is = 0; is < NS; ++is) {
          const double dcyli = u[loc + is] - u[locyl + is], dcyui = u[locyu + is] - u[loc + is];
          const double dcxli = u[loc + is] - u[locxl + is], dcxui = u[locxu + is] - u[loc + is];
          y[loc + is] = coy[is] * (dcyui - dcyli) + cox[is] * (dcxui - dcxli) + (v ? drates[is] : rates[is]);
        }
      }
    }
  }
  void rhs(const double* x, const double* p, double, double* y) const override { apply(x, p, nullptr, y); }
  void jac_mul(const double* x, const double* p, double, const double* v, double* y) const override { apply(x, p, v, y); }
  void mass(const double* x, const double*, double, double beta, double* y) const override {
    for (int i = 0; i < n; ++i) y[i] = (i % NS) < 1 ? x[i] + beta * y[i] : beta * y[i];
  }
  void init(const double*, double, double* y) const override {
    const double dx = 1.0 / ((double)nx - 1.0), dy = 1.0 / ((double)nx - 1.0);
    for (int jy = 0; jy < nx; ++jy) {
      const double yy = (double)jy * dy;
      for (int jx = 0; jx < nx; ++jx) {
        const double xx = (double)jx * dx;
        double xyfactor = 16.0 * xx * (1.0 - xx) * yy * (1.0 - yy);
        xyfactor = xyfactor * xyfactor;
        const int loc = NS * nx * jy + NS * jx;
        y[loc] = 10.0 + 1.0 * xyfactor;
        y[loc + 1] = 1.0e5;
      }
    }
  }
};

// A user model compiled to a shared library with the external-model C ABI (dsl_dims, dsl_rhs, dsl_jac_mul, dsl_mass_gemv, dsl_init, dsl_root):
// the CPU twin of a run-time-compiled device model, so the oracle can integrate exactly the model the GPU integrates.  This is the role of the
// reference's compiled DiffSL module behind DiffSl<M, CG> (crates/diffsol/src/ode_equations/diffsl.rs: rhs, rhs_grad, mass, set_u0, calc_stop).
struct ExternalFns {
  void (*dims)(int*, int*, int*, int*, int*) = nullptr;
  void (*rhs)(double, const double*, const double*, double*) = nullptr;
  void (*jac_mul)(double, const double*, const double*, const double*, double*) = nullptr;
  void (*mass_gemv)(double, const double*, const double*, double, double*) = nullptr;
  void (*init)(double, const double*, double*) = nullptr;
  void (*root)(double, const double*, const double*, double*) = nullptr;
  void (*out)(double, const double*, const double*, double*) = nullptr;
  // optional (models with inputs): (dF/dp) v and (du0/dp) v — rhs_sgrad / set_u0_sgrad of the reference's compiled DiffSL module
  void (*sens_mul)(double, const double*, const double*, const double*, double*) = nullptr;
  void (*init_sens_mul)(double, const double*, const double*, double*) = nullptr;
  void (*reset)(double, const double*, const double*, double*) = nullptr;  // optional: reset_i
};
constexpr int MODEL_EXTERNAL_BASE = 1000;
inline std::vector<ExternalFns>& external_models() {
  static std::vector<ExternalFns> v;
  return v;
}
struct ExternalModel : Model {
  ExternalFns f;
  int nout = 0;
  explicit ExternalModel(const ExternalFns& fns) : f(fns) {
    int hm = 0;
    f.dims(&n, &np, &nroots, &nout, &hm);
    has_mass = hm != 0;
    has_sens = f.sens_mul != nullptr && f.init_sens_mul != nullptr;
    has_reset = f.reset != nullptr;
  }
  void reset(const double* x, const double* p, double t, double* y) const override { f.reset(t, x, p, y); }
  void sens_mul(const double* x, const double* p, double t, const double* v, double* y) const override { f.sens_mul(t, x, p, v, y); }
  void init_sens_mul(const double* p, double t, const double* v, double* y) const override { f.init_sens_mul(t, p, v, y); }
  void rhs(const double* x, const double* p, double t, double* y) const override { f.rhs(t, x, p, y); }
  void jac_mul(const double* x, const double* p, double t, const double* v, double* y) const override { f.jac_mul(t, x, p, v, y); }
  void mass(const double* x, const double* p, double t, double beta, double* y) const override { f.mass_gemv(t, x, p, beta, y); }
  void init(const double* p, double t, double* y) const override { f.init(t, p, y); }
  void root(const double* x, const double* p, double t, double* g) const override { f.root(t, x, p, g); }
};

inline std::unique_ptr<Model> make_model(int id, int size) {
  if (id >= MODEL_EXTERNAL_BASE) {
    const size_t k = (size_t)(id - MODEL_EXTERNAL_BASE);
    if (k >= external_models().size()) throw std::runtime_error("oracle: unknown external model id");
    return std::make_unique<ExternalModel>(external_models()[k]);
  }
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
    case MODEL_SPM: return std::make_unique<Spm>(size);
    case MODEL_HEAT2D: return std::make_unique<Heat2d>(size <= 0 ? 10 : size);
    case MODEL_FOODWEB: return std::make_unique<Foodweb>(size <= 0 ? 10 : size);
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
  // df/dp as an n x np matrix, column by column from sens_mul with unit vectors (NonLinearOpSens::_default_sens_inplace, op/nonlinear_op.rs:72-81;
  // closure_with_sens.rs does not count these calls in OpStatistics)
  void rhs_sens(const V& x, double t, M& S) const {
    const int n_ = n(), np_ = model->np;
    std::vector<double> v((size_t)np_, 0.0);
    V col(n_, nb);
    for (int j = 0; j < np_; ++j) {
      v[(size_t)j] = 1.0;
      for (int b = 0; b < nb; ++b) model->sens_mul(&x.d[(size_t)b * x.n], pb(b), t, v.data(), &col.d[(size_t)b * n_]);
      S.set_column(j, col);
      v[(size_t)j] = 0.0;
    }
  }
  // SensInit::call (ode_equations/sens_equations.rs:62-70): s_j(t0) = (dy0/dp) e_j
  void init_sens(double t, int j, V& s) const {
    std::vector<double> v((size_t)model->np, 0.0);
    v[(size_t)j] = 1.0;
    for (int b = 0; b < nb; ++b) model->init_sens_mul(pb(b), t, v.data(), &s.d[(size_t)b * s.n]);
  }
  void root(const V& x, double t, V& g) const {
    for (int b = 0; b < nb; ++b) model->root(&x.d[(size_t)b * x.n], pb(b), t, &g.d[(size_t)b * g.n]);
  }
  void reset(const V& x, double t, V& y) const {
    for (int b = 0; b < nb; ++b) model->reset(&x.d[(size_t)b * x.n], pb(b), t, &y.d[(size_t)b * y.n]);
  }
};

}  // namespace orc
