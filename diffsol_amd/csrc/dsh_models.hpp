// Device-side model definitions (the OdeEquations plug-in boundary) for libdiffsol_hip.so (gfx950).
//
// A "static" model has compile-time size N <= 8 and lives entirely in registers: one lane integrates one system.  Each provides
//   rhs(t,x,p,y)           NonLinearOp::call_inplace          (diffsol/src/op/nonlinear_op.rs:10-21)
//   jac_mul(t,x,p,v,y)     NonLinearOpJacobian::jac_mul_inplace (:175-178)
//   mass_gemv(t,x,p,beta,y) y = M x + beta y                  LinearOp::gemv_inplace (diffsol/src/op/linear_op.rs:9-18)
//   init(t,p,y), root(t,x,p,g)
// The dense Jacobian / mass matrix are assembled column by column from jac_mul / mass_gemv with unit vectors — the reference's default
// `_default_jacobian_inplace` / `_default_matrix_inplace` (op/nonlinear_op.rs:211-219, op/linear_op.rs:41-50) — fully unrolled in
// registers, so the entries are bit-identical to the reference's assembly.  Expressions follow the reference closures literally
// (cited per model); the build uses -ffp-contract=off so no FMA contraction changes their rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/diffsol_hip.h"
#include "../../include/diffsol_detpow.h"

namespace dsh {

// test_models/exponential_decay.rs:14-21 (rhs), :54-61 (jac), :72-81 (init), :98-100 (root)
template <bool WITH_ROOT>
struct ExponentialDecayT {
  static constexpr int N = 2, NP = 2, NROOTS = WITH_ROOT ? 1 : 0;
  static constexpr bool HAS_MASS = false;
  __device__ static void rhs(double, const double (&x)[N], const double (&p)[NP], double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] * (-p[0]);
  }
  __device__ static void jac_mul(double, const double (&)[N], const double (&p)[NP], const double (&v)[N], double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = v[i] * (-p[0]);
  }
  __device__ static void mass_gemv(double, const double (&x)[N], const double (&)[NP], double beta, double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = 1.0 * x[i] + beta * y[i];
  }
  __device__ static void init(double, const double (&p)[NP], double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = p[1];
  }
  __device__ static void root(double, const double (&x)[N], const double (&)[NP], double (&g)[1]) { g[0] = x[0] - 0.6; }
  // forward sensitivities: (df/dp) v = -x v_k  (exponential_decay.rs:33-36), (dy0/dp) v = (v_y0, v_y0)  (:90-93)
  __device__ static void sens_mul(double, const double (&x)[N], const double (&)[NP], const double (&v)[NP], double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] * (-v[0]);
  }
  __device__ static void init_sens_mul(double, const double (&)[NP], const double (&v)[NP], double (&y)[N]) { y[0] = v[1]; y[1] = v[1]; }
};

// test_models/exponential_decay_with_algebraic.rs:18-23 (rhs), :64-75 (jac), :94-105 (mass), :122-126 / :267-276 (init)
template <bool BATCHED_INIT>
struct ExponentialDecayAlgebraicT {
  static constexpr int N = 3, NP = 1, NROOTS = 0;
  static constexpr bool HAS_MASS = true;
  __device__ static void rhs(double, const double (&x)[N], const double (&p)[NP], double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] * (-p[0]);
    y[N - 1] = x[N - 1] - x[N - 2];
  }
  __device__ static void jac_mul(double, const double (&)[N], const double (&p)[NP], const double (&v)[N], double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = v[i] * (-p[0]);
    y[N - 1] = v[N - 1] - v[N - 2];
  }
  __device__ static void mass_gemv(double, const double (&x)[N], const double (&)[NP], double beta, double (&y)[N]) {
    double yn = beta * y[N - 1];
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = 1.0 * x[i] + beta * y[i];
    y[N - 1] = yn;
  }
  __device__ static void init(double, const double (&)[NP], double (&y)[N]) { y[0] = 1.0; y[1] = 1.0; y[2] = BATCHED_INIT ? 1.0 : 0.0; }
  __device__ static void root(double, const double (&)[N], const double (&)[NP], double (&)[1]) {}
  // exponential_decay_with_algebraic.rs:33-44 (sens: y = x * (-v[0]), last = 0), :128-135 (init_sens: zeros)
  __device__ static void sens_mul(double, const double (&x)[N], const double (&)[NP], const double (&v)[NP], double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = x[i] * (-v[0]);
    y[N - 1] = 0.0;
  }
  __device__ static void init_sens_mul(double, const double (&)[NP], const double (&)[NP], double (&y)[N]) { y[0] = 0.0; y[1] = 0.0; y[2] = 0.0; }
};

// test_models/robertson_ode.rs:71-90 (rhs, jac_mul), :92-101 (init); one group per system (ensembles vary p, not ngroups)
struct RobertsonOde1 {
  static constexpr int N = 3, NP = 3, NROOTS = 0;
  static constexpr bool HAS_MASS = false;
  __device__ static void rhs(double, const double (&x)[N], const double (&p)[NP], double (&y)[N]) {
    y[0] = -p[0] * x[0] + p[1] * x[1] * x[2];
    y[1] = p[0] * x[0] - p[1] * x[1] * x[2] - p[2] * x[1] * x[1];
    y[2] = p[2] * x[1] * x[1];
  }
  __device__ static void jac_mul(double, const double (&x)[N], const double (&p)[NP], const double (&v)[N], double (&y)[N]) {
    y[0] = -p[0] * v[0] + p[1] * v[1] * x[2] + p[1] * x[1] * v[2];
    y[1] = p[0] * v[0] - p[1] * v[1] * x[2] - p[1] * x[1] * v[2] - 2.0 * p[2] * x[1] * v[1];
    y[2] = 2.0 * p[2] * x[1] * v[1];
  }
  __device__ static void mass_gemv(double, const double (&x)[N], const double (&)[NP], double beta, double (&y)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = 1.0 * x[i] + beta * y[i];
  }
  __device__ static void init(double, const double (&)[NP], double (&y)[N]) { y[0] = 1.0; y[1] = 0.0; y[2] = 0.0; }
  __device__ static void root(double, const double (&)[N], const double (&)[NP], double (&)[1]) {}
  // forward sensitivities: test_models/robertson_ode_with_sens.rs:38-42 (df/dp v), :50 (dy0/dp v = 0)
  __device__ static void sens_mul(double, const double (&x)[N], const double (&)[NP], const double (&v)[NP], double (&y)[N]) {
    y[0] = -v[0] * x[0] + v[1] * x[1] * x[2];
    y[1] = v[0] * x[0] - v[1] * x[1] * x[2] - v[2] * x[1] * x[1];
    y[2] = v[2] * x[1] * x[1];
  }
  __device__ static void init_sens_mul(double, const double (&)[NP], const double (&)[NP], double (&y)[N]) { y[0] = 0.0; y[1] = 0.0; y[2] = 0.0; }
};

// test_models/robertson.rs:60-94
struct RobertsonDae {
  static constexpr int N = 3, NP = 3, NROOTS = 0;
  static constexpr bool HAS_MASS = true;
  __device__ static void rhs(double, const double (&x)[N], const double (&p)[NP], double (&y)[N]) {
    y[0] = -p[0] * x[0] + p[1] * x[1] * x[2];
    y[1] = p[0] * x[0] - p[1] * x[1] * x[2] - p[2] * x[1] * x[1];
    y[2] = x[0] + x[1] + x[2] - 1.0;
  }
  __device__ static void jac_mul(double, const double (&x)[N], const double (&p)[NP], const double (&v)[N], double (&y)[N]) {
    y[0] = -p[0] * v[0] + p[1] * v[1] * x[2] + p[1] * x[1] * v[2];
    y[1] = p[0] * v[0] - p[1] * v[1] * x[2] - p[1] * x[1] * v[2] - 2.0 * p[2] * x[1] * v[1];
    y[2] = v[0] + v[1] + v[2];
  }
  __device__ static void mass_gemv(double, const double (&x)[N], const double (&)[NP], double beta, double (&y)[N]) {
    y[0] = x[0] + beta * y[0];
    y[1] = x[1] + beta * y[1];
    y[2] = beta * y[2];
  }
  __device__ static void init(double, const double (&)[NP], double (&y)[N]) { y[0] = 1.0; y[1] = 0.0; y[2] = 0.0; }
  __device__ static void root(double, const double (&)[N], const double (&)[NP], double (&)[1]) {}
  // robertson.rs:73-77 (sens_mul), :91-93 (init_sens: zeros)
  __device__ static void sens_mul(double, const double (&x)[N], const double (&)[NP], const double (&v)[NP], double (&y)[N]) {
    y[0] = -v[0] * x[0] + v[1] * x[1] * x[2];
    y[1] = v[0] * x[0] - v[1] * x[1] * x[2] - v[2] * x[1] * x[1];
    y[2] = 0.0;
  }
  __device__ static void init_sens_mul(double, const double (&)[NP], const double (&)[NP], double (&y)[N]) { y[0] = 0.0; y[1] = 0.0; y[2] = 0.0; }
};

// examples/electrical-circuits/src/main.rs:10-41 — u=(iR,iL,iC,V), M=diag(0,1,0,1), p=[R,L,C,V0,omega,ithresh]
template <bool WITH_ROOT>
struct RlcT {
  static constexpr int N = 4, NP = 6, NROOTS = WITH_ROOT ? 1 : 0;
  static constexpr bool HAS_MASS = true;
  __device__ static void rhs(double t, const double (&x)[N], const double (&p)[NP], double (&y)[N]) {
    double vs = p[3] * dsh_det_sin(p[4] * t);
    y[0] = x[3] - p[0] * x[0];
    y[1] = (vs - x[3]) / p[1];
    y[2] = x[1] - x[0] - x[2];
    y[3] = x[2] / p[2];
  }
  __device__ static void jac_mul(double, const double (&)[N], const double (&p)[NP], const double (&v)[N], double (&y)[N]) {
    y[0] = v[3] - p[0] * v[0];
    y[1] = (-v[3]) / p[1];
    y[2] = v[1] - v[0] - v[2];
    y[3] = v[2] / p[2];
  }
  __device__ static void mass_gemv(double, const double (&x)[N], const double (&)[NP], double beta, double (&y)[N]) {
    y[0] = beta * y[0];
    y[1] = x[1] + beta * y[1];
    y[2] = beta * y[2];
    y[3] = x[3] + beta * y[3];
  }
  __device__ static void init(double, const double (&)[NP], double (&y)[N]) { y[0] = 0.0; y[1] = 0.0; y[2] = 0.0; y[3] = 0.0; }
  __device__ static void root(double, const double (&x)[N], const double (&p)[NP], double (&g)[1]) { g[0] = x[0] - p[5]; }
};

// number of outputs of a model (out_i of DiffSL models; the built-in registry models return their state)
template <class...> using model_void_t = void;
template <class M, class = void> struct model_nout { static constexpr int value = 0; };
template <class M> struct model_nout<M, model_void_t<decltype(M::NOUT)>> { static constexpr int value = M::NOUT; };

// a model with parameter sensitivities provides sens_mul ((df/dp) v) and init_sens_mul ((dy0/dp) v), v of length NP (NonLinearOpSens / ConstantOpSens)
template <class M, class = void> struct model_has_sens { static constexpr bool value = false; };
template <class M> struct model_has_sens<M, model_void_t<decltype(&M::sens_mul)>> { static constexpr bool value = true; };
// a hybrid model provides reset (the state after an event; OdeEquations::reset, DiffSL reset_i)
template <class M, class = void> struct model_has_reset { static constexpr bool value = false; };
template <class M> struct model_has_reset<M, model_void_t<decltype(&M::reset)>> { static constexpr bool value = true; };

// a model may declare that its Jacobian is banded (BAND_K = max(kl, ku) <= 4) and provide the band directly (jac_band): the device-resident BDF then keeps
// the state in per-lane memory and factors the band only, which lifts its size limit from the register budget (n <= 4) to n <= 64
template <class M, class = void> struct model_band_k { static constexpr int value = 0; };
template <class M> struct model_band_k<M, model_void_t<decltype(M::BAND_K)>> { static constexpr int value = M::BAND_K; };
// a banded model with a DIAGONAL mass matrix may state that M x == diag(M 1) x bit for bit (rows: 0, x_i, or coefficient * x_i; the DiffSL front end checks it)
// bandwidth of the mass matrix of a banded model (0: diagonal); the lane-per-member BDF then keeps M's band next to the Jacobian's
template <class M, class = void> struct model_mass_band_k { static constexpr int value = 0; };
template <class M> struct model_mass_band_k<M, model_void_t<decltype(M::MASS_BAND_K)>> { static constexpr int value = M::MASS_BAND_K; };
template <class M, class = void> struct model_mass_rows_scaled { static constexpr bool value = false; };
template <class M> struct model_mass_rows_scaled<M, model_void_t<decltype(M::MASS_ROWS_SCALED)>> { static constexpr bool value = M::MASS_ROWS_SCALED; };

// Column-by-column dense assembly from jac_mul / mass_gemv with unit vectors (see header comment).  Column-major A[j*N+i].
template <class Mdl>
__device__ __forceinline__ void assemble_jacobian(double t, const double (&x)[Mdl::N], const double (&p)[Mdl::NP], double (&J)[Mdl::N * Mdl::N]) {
  constexpr int N = Mdl::N;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double v[N], col[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (i == j) ? 1.0 : 0.0;
    Mdl::jac_mul(t, x, p, v, col);
#pragma unroll
    for (int i = 0; i < N; ++i) J[j * N + i] = col[i];
  }
}
template <class Mdl>
__device__ __forceinline__ void assemble_mass(double t, const double (&p)[Mdl::NP], double (&Mm)[Mdl::N * Mdl::N]) {
  constexpr int N = Mdl::N;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double v[N], col[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { v[i] = (i == j) ? 1.0 : 0.0; col[i] = 0.0; }
    Mdl::mass_gemv(t, v, p, 0.0, col);
#pragma unroll
    for (int i = 0; i < N; ++i) Mm[j * N + i] = col[i];
  }
}

// Dispatch a functor templated on the static model type; returns false if `model` has no static specialisation for this size.
template <class F>
inline bool dispatch_static_model(int model, int64_t size, F&& f) {
  switch (model) {
    case DSH_MODEL_EXPONENTIAL_DECAY: f(ExponentialDecayT<false>{}); return true;
    case DSH_MODEL_EXPONENTIAL_DECAY_ROOT: f(ExponentialDecayT<true>{}); return true;
    case DSH_MODEL_EXPONENTIAL_DECAY_ALGEBRAIC: f(ExponentialDecayAlgebraicT<false>{}); return true;
    case DSH_MODEL_EXPONENTIAL_DECAY_ALGEBRAIC_BATCHED: f(ExponentialDecayAlgebraicT<true>{}); return true;
    case DSH_MODEL_ROBERTSON_ODE: if (size <= 1) { f(RobertsonOde1{}); return true; } return false;
    case DSH_MODEL_ROBERTSON_DAE: f(RobertsonDae{}); return true;
    case DSH_MODEL_RLC: if (size != 0) f(RlcT<true>{}); else f(RlcT<false>{}); return true;
    default: return false;
  }
}

}  // namespace dsh
