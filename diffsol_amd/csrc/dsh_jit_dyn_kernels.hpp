// Element-wise kernels of a run-time-compiled, run-time-sized model: one thread per (component, system), the layout of the built-in dynamic
// models (dsh_models.hip k_dyn_*).  Included ONLY by the hiprtc translation unit of a model, after the generated jit_* functions
// (diffsol_amd/host/diffsl.hpp, Target::HipDynamic).  Batch-fastest layout: element (i, b) at i * nb + b, so a wavefront shares `i` and the
// generated switch (i) does not diverge.
#pragma once
#include "dsh_device.hpp"

namespace dsh {

// The accessors the generated functions read their operands through.  Concrete types shared by every kernel (not one closure type per kernel): an
// outlined model (one __noinline__ function per component, diffsl.hpp emit_switch) is then instantiated once, not once per kernel.
struct JitVec {  // element k of this thread's system
  const double* p; int64_t nb, b;
  __device__ __forceinline__ double operator()(int64_t k) const { return p[k * nb + b]; }
};
struct JitDir {  // a direction: a vector of the batch, or the unit vector e_unit (Jacobian / mass-matrix columns)
  const double* p; int64_t nb, b, unit;
  __device__ __forceinline__ double operator()(int64_t k) const { return unit >= 0 ? (k == unit ? 1.0 : 0.0) : p[k * nb + b]; }
};

extern "C" __global__ void k_jit_dyn_rhs(int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, const double* __restrict__ v,
                                         double* __restrict__ y) {
  const int64_t total = (int64_t)kJitN * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / nb, b = idx % nb;
    const JitVec X{x, nb, b};
    const JitDir V{v, nb, b, -1};
    const JitVec P{p, nb, b};
    y[idx] = jit_component(t, (long)i, X, V, P, v != nullptr);
  }
}
// dense Jacobian entry (i, j) = component i of J e_j: the arithmetic of jac_mul with a unit vector, like the built-in models
extern "C" __global__ void k_jit_dyn_jacobian(int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, double* __restrict__ jac) {
  const int64_t total = (int64_t)kJitN * kJitN * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / nb, b = idx % nb;
    const int64_t i = e % kJitN, j = e / kJitN;
    const JitVec X{x, nb, b};
    const JitDir V{nullptr, nb, b, j};
    const JitVec P{p, nb, b};
    jac[idx] = jit_component(t, (long)i, X, V, P, true);
  }
}
// the same entries for the structural nonzeros of f_y only (kJitJacRow / kJitJacCol, written by the front end for large sparse models); the caller zeroes `jac` first
extern "C" __global__ void k_jit_dyn_jacobian_sparse(int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, double* __restrict__ jac) {
  const int64_t total = (int64_t)kJitJacNnz * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / nb, b = idx % nb;
    const int64_t i = kJitJacRow[e], j = kJitJacCol[e];
    const JitVec X{x, nb, b};
    const JitDir V{nullptr, nb, b, j};
    const JitVec P{p, nb, b};
    jac[(j * kJitN + i) * nb + b] = jit_component(t, (long)i, X, V, P, true);
  }
}
extern "C" __global__ void k_jit_dyn_init(int64_t nb, double t, const double* __restrict__ p, double* __restrict__ y) {
  const int64_t total = (int64_t)kJitN * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / nb, b = idx % nb;
    const JitVec P{p, nb, b};
    y[idx] = jit_init_component(t, (long)i, P);
  }
}
// y = M x + beta y
extern "C" __global__ void k_jit_dyn_mass_gemv(int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, double beta, double* __restrict__ y) {
  const int64_t total = (int64_t)kJitN * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / nb, b = idx % nb;
    const JitVec X{x, nb, b};
    const JitVec P{p, nb, b};
    y[idx] = jit_mass_component(t, (long)i, X, P) + beta * y[idx];
  }
}
// dense mass matrix entry (i, j) = component i of M e_j
extern "C" __global__ void k_jit_dyn_mass_matrix(int64_t nb, double t, const double* __restrict__ p, double* __restrict__ mass) {
  const int64_t total = (int64_t)kJitN * kJitN * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / nb, b = idx % nb;
    const int64_t i = e % kJitN, j = e / kJitN;
    const JitDir X{nullptr, nb, b, j};
    const JitVec P{p, nb, b};
    mass[idx] = jit_mass_component(t, (long)i, X, P);
  }
}
// df/dp (init = 0) or du0/dp (init = 1) as an n x np matrix, column j at (j * n + i) * nb + b: component i of sens_mul / init_sens_mul with the unit vector e_j
// (NonLinearOpSens::_default_sens_inplace, op/nonlinear_op.rs:72-81), like k_static_model<.., Op::RhsSens> of the register-resident forms
extern "C" __global__ void k_jit_dyn_sens(int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, int init, double* __restrict__ s) {
  const int64_t total = (int64_t)kJitN * kJitNP * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / nb, b = idx % nb;
    const int64_t i = e % kJitN, j = e / kJitN;
    const JitVec X{x, nb, b};
    const JitDir V{nullptr, nb, b, j};
    const JitVec P{p, nb, b};
    s[idx] = jit_sens_component(t, (long)i, X, V, P, init != 0);
  }
}
// y = reset(x): the state after an event of a hybrid model
extern "C" __global__ void k_jit_dyn_reset(int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, double* __restrict__ y) {
  const int64_t total = (int64_t)kJitN * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / nb, b = idx % nb;
    const JitVec X{x, nb, b};
    const JitVec P{p, nb, b};
    y[idx] = jit_reset_component(t, (long)i, X, P);
  }
}
// which = 0: stop_i (roots), 1: out_i; g is count x nb, batch-fastest
extern "C" __global__ void k_jit_dyn_root_out(int64_t nb, double t, const double* __restrict__ x, const double* __restrict__ p, int which, double* __restrict__ g) {
  const int64_t count = which == 0 ? kJitNRoots : kJitNOut, total = count * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / nb, b = idx % nb;
    const JitVec X{x, nb, b};
    const JitVec P{p, nb, b};
    g[idx] = which == 0 ? jit_root_component(t, (long)i, X, P) : jit_out_component(t, (long)i, X, P);
  }
}

}  // namespace dsh
