// One-launch-per-operator kernels of the OdeEquations boundary for the register-resident ("static") models (launch code: dsh_models.hip).
// In a header so that run-time-compiled model modules (dsh_jit.hip) instantiate the same kernel for user models.
#pragma once
#include "dsh_device.hpp"
#include "dsh_lu_dev.hpp"
#include "dsh_models.hpp"

namespace dsh {

enum class Op { Rhs, JacMul, Jacobian, MassGemv, MassMatrix, Init, Root, Out, RhsSens, InitSens, Reset };

template <class Mdl, Op OP>
__global__ void k_static_model(int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, const double* __restrict__ v,
                               double beta, double* __restrict__ y) {
  constexpr int N = Mdl::N, NP = Mdl::NP;
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  double pp[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) pp[k] = p[(int64_t)k * nb + b];
  if constexpr (OP == Op::Rhs) {
    double xr[N], yr[N];
    load_vec<N>(x, nb, b, xr);
    Mdl::rhs(t, xr, pp, yr);
    store_vec<N>(y, nb, b, yr);
  } else if constexpr (OP == Op::JacMul) {
    double xr[N], vr[N], yr[N];
    load_vec<N>(x, nb, b, xr);
    load_vec<N>(v, nb, b, vr);
    Mdl::jac_mul(t, xr, pp, vr, yr);
    store_vec<N>(y, nb, b, yr);
  } else if constexpr (OP == Op::Jacobian) {
    double xr[N], J[N * N];
    load_vec<N>(x, nb, b, xr);
    assemble_jacobian<Mdl>(t, xr, pp, J);
    store_mat<N>(y, nb, b, J);
  } else if constexpr (OP == Op::MassGemv) {
    double xr[N], yr[N];
    load_vec<N>(x, nb, b, xr);
    load_vec<N>(y, nb, b, yr);
    Mdl::mass_gemv(t, xr, pp, beta, yr);
    store_vec<N>(y, nb, b, yr);
  } else if constexpr (OP == Op::MassMatrix) {
    double Mm[N * N];
    assemble_mass<Mdl>(t, pp, Mm);
    store_mat<N>(y, nb, b, Mm);
  } else if constexpr (OP == Op::Init) {
    double yr[N];
    Mdl::init(t, pp, yr);
    store_vec<N>(y, nb, b, yr);
  } else if constexpr (OP == Op::Root) {
    if constexpr (Mdl::NROOTS > 0) {
      double xr[N], g[Mdl::NROOTS];
      load_vec<N>(x, nb, b, xr);
      Mdl::root(t, xr, pp, g);
#pragma unroll
      for (int k = 0; k < Mdl::NROOTS; ++k) y[(int64_t)k * nb + b] = g[k];
    }
  } else if constexpr (OP == Op::RhsSens || OP == Op::InitSens) {
    // df/dp (n x np) resp. dy0/dp (n x np), column by column from sens_mul / init_sens_mul with unit vectors: NonLinearOpSens::_default_sens_inplace
    // (op/nonlinear_op.rs:72-81) and SensInit (ode_equations/sens_equations.rs:62-70) for every parameter at once; column j at (j*N + i)*nb + b
    if constexpr (model_has_sens<Mdl>::value) {
      double xr[N];
      if constexpr (OP == Op::RhsSens) load_vec<N>(x, nb, b, xr);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        double e[NP], col[N];
#pragma unroll
        for (int k = 0; k < NP; ++k) e[k] = (k == j) ? 1.0 : 0.0;
        if constexpr (OP == Op::RhsSens) Mdl::sens_mul(t, xr, pp, e, col);
        else Mdl::init_sens_mul(t, pp, e, col);
#pragma unroll
        for (int i = 0; i < N; ++i) y[((int64_t)j * N + i) * nb + b] = col[i];
      }
    }
  } else if constexpr (OP == Op::Reset) {  // reset_i of a hybrid model: the state after an event
    if constexpr (model_has_reset<Mdl>::value) {
      double xr[N], yr[N];
      load_vec<N>(x, nb, b, xr);
      Mdl::reset(t, xr, pp, yr);
      store_vec<N>(y, nb, b, yr);
    }
  } else if constexpr (OP == Op::Out) {  // out_i of a DiffSL model (calc_out): nout x nb, batch-fastest
    constexpr int NO = model_nout<Mdl>::value;
    if constexpr (NO > 0) {
      double xr[N], g[NO];
      load_vec<N>(x, nb, b, xr);
      Mdl::out(t, xr, pp, g);
#pragma unroll
      for (int k = 0; k < NO; ++k) y[(int64_t)k * nb + b] = g[k];
    }
  }
}

}  // namespace dsh
