// The run-time-sized registry models (dsh_models_dyn.hpp) in the static, banded form of the lane-per-member device-resident BDF (dsh_adaptive_kernel.hpp,
// Mdl::BAND_K / jac_band): instantiated by hiprtc for the size at hand (dsh_model_lane_twin, dsh_jit.hip), e.g. DynLane<DSH_MODEL_SPM, 42, 1, 2, 1> for the
// single-particle battery model with 20 shells.  Every entry is the registry's own per-component function, so the arithmetic — and with it the parity with
// the CPU oracle, whose Jacobian is J e_j column by column — is that of the other kernels.
#pragma once
#include "dsh_device.hpp"
#include "dsh_models_dyn.hpp"

namespace dsh {

template <int MODEL, int SIZE_N, int NPAR, int NROOT, int K>
struct DynLane {
  static constexpr int N = SIZE_N, NP = NPAR, NROOTS = NROOT, NOUT = 0, BAND_K = K;
  static constexpr bool HAS_MASS = false;
  // f is evaluated in every Newton iteration of the lane-per-member integrators, whose vectors live in per-lane memory: inlined (scratch addressing
  // instead of generic pointers) and fully unrolled (the component index is a constant: the model's case distinctions fold away and all loads of x can be
  // in flight together — rolled, every component waits for its own loads, ~n serialised memory round trips per call)
  __device__ __forceinline__ static void rhs(double t, const double (&x)[N], const double (&p)[NP], double (&y)[N]) {
    auto X = [&](int64_t k) { return x[k]; };
    auto V = [&](int64_t) { return 0.0; };
    auto P = [&](int64_t k) { return p[k]; };
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = dyn_component(MODEL, (int64_t)N, t, (int64_t)i, X, V, P, false);
  }
  __device__ static void jac_mul(double t, const double (&x)[N], const double (&p)[NP], const double (&v)[N], double (&y)[N]) {
    auto X = [&](int64_t k) { return x[k]; };
    auto V = [&](int64_t k) { return v[k]; };
    auto P = [&](int64_t k) { return p[k]; };
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) y[i] = dyn_component(MODEL, (int64_t)N, t, (int64_t)i, X, V, P, true);
  }
  // d f_i / d u_col at [(col - i + K) * N + i]: row i of J e_col
  __device__ static void jac_band(double t, const double (&x)[N], const double (&p)[NP], double (&Jb)[(2 * K + 1) * N]) {
    auto X = [&](int64_t k) { return x[k]; };
    auto P = [&](int64_t k) { return p[k]; };
#pragma unroll
    for (int d = 0; d < 2 * K + 1; ++d)
DSH_UNROLL_N
      for (int i = 0; i < N; ++i) {
        const int col = i + d - K;
        auto E = [&](int64_t k) { return k == col ? 1.0 : 0.0; };
        Jb[d * N + i] = (col >= 0 && col < N) ? dyn_component(MODEL, (int64_t)N, t, (int64_t)i, X, E, P, true) : 0.0;
      }
  }
  __device__ static void mass_gemv(double, const double (&x)[N], const double (&)[NP], double beta, double (&y)[N]) {
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) y[i] = 1.0 * x[i] + beta * y[i];
  }
  __device__ static void init(double, const double (&)[NP], double (&y)[N]) {
DSH_UNROLL_N
    for (int i = 0; i < N; ++i) y[i] = dyn_init_value(MODEL, (int64_t)N, (int64_t)i);
  }
  __device__ static void root(double t, const double (&x)[N], const double (&p)[NP], double (&g)[NROOT > 0 ? NROOT : 1]) {
    auto X = [&](int64_t k) { return x[k]; };
    auto P = [&](int64_t k) { return p[k]; };
    double gg[2] = {0.0, 0.0};
    dyn_root_values(MODEL, (int64_t)N, t, X, P, gg);
#pragma unroll
    for (int r = 0; r < NROOT; ++r) g[r] = gg[r];
  }
};

}  // namespace dsh
