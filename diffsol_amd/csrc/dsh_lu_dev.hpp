// Device-side dense LU building blocks shared by dsh_lu.hip and dsh_fused.hip (gfx950).
//
// Algorithm = LAPACK-style partial pivoting as done by nalgebra 0.35 `LU::new` / `LU::solve_mut`, which is what the reference's
// CPU oracle path runs (diffsol-la/src/linear_solver/nalgebra/lu.rs:36,50) and equivalent to cuSOLVER getrf/getrs used by CudaLU
// (diffsol-la/src/linear_solver/cuda/lu.rs:84,132):
//   step k: pivot = first max |a_rk|, r >= k; swap rows; l_rk = a_rk * (1/pivot); a_rc = (-a_kc)*l_rk + a_rc.
//   solve: apply the row swaps to b, forward substitute with unit L (column oriented), back substitute with U (x_k = b_k / u_kk).
// The register versions keep one whole system per lane with every index a compile-time constant (pivot row selection is done with
// predicated swaps), so nothing spills to scratch; with -ffp-contract=off the arithmetic is bit-identical to the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dsh {

template <int N>
__device__ __forceinline__ void lu_factor_reg(double (&A)[N * N], int (&P)[N], bool& singular) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int p = i;
    double best = fabs(A[i * N + i]);
#pragma unroll
    for (int r = i + 1; r < N; ++r) {
      double v = fabs(A[i * N + r]);
      if (v > best) { best = v; p = r; }
    }
    double diag = A[i * N + i];
#pragma unroll
    for (int r = i + 1; r < N; ++r) if (r == p) diag = A[i * N + r];
    if (diag == 0.0) { P[i] = i; singular = true; }
    else {
      P[i] = p;
      // row swap i <-> p written as value selects (a conditional store would be turned into a dynamically indexed scratch access)
#pragma unroll
      for (int c = 0; c < N; ++c) {
        const double ai = A[c * N + i];
        double picked = ai;
#pragma unroll
        for (int r = i + 1; r < N; ++r) {
          const bool s = (r == p);
          const double ar = A[c * N + r];
          picked = s ? ar : picked;
          A[c * N + r] = s ? ai : ar;
        }
        A[c * N + i] = picked;
      }
      double inv_diag = 1.0 / diag;
#pragma unroll
      for (int r = i + 1; r < N; ++r) A[i * N + r] = A[i * N + r] * inv_diag;
#pragma unroll
      for (int c = i + 1; c < N; ++c) {
        double pr = A[c * N + i];
#pragma unroll
        for (int r = i + 1; r < N; ++r) A[c * N + r] = (-pr) * A[i * N + r] + A[c * N + r];
      }
    }
  }
}

// returns false if a zero diagonal of U was met (LuSolveFailed)
template <int N>
__device__ __forceinline__ bool lu_solve_reg(const double (&A)[N * N], const int (&P)[N], double (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double vi = v[i];
    double picked = vi;
#pragma unroll
    for (int r = i + 1; r < N; ++r) {
      const bool s = (r == P[i]);
      const double vr = v[r];
      picked = s ? vr : picked;
      v[r] = s ? vi : vr;
    }
    v[i] = picked;
  }
#pragma unroll
  for (int i = 0; i + 1 < N; ++i) {
    double coeff = v[i];
#pragma unroll
    for (int r = i + 1; r < N; ++r) v[r] = (-coeff) * A[i * N + r] + v[r];
  }
  bool ok = true;
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double diag = A[i * N + i];
    if (diag == 0.0) ok = false;
    double coeff = v[i] / diag;
    v[i] = coeff;
#pragma unroll
    for (int r = 0; r < i; ++r) v[r] = (-coeff) * A[i * N + r] + v[r];
  }
  return ok;
}

// ---- coalesced load/store of one system's factors between HBM (batch-fastest) and registers
template <int N>
__device__ __forceinline__ void load_mat(const double* __restrict__ p, int64_t nb, int64_t b, double (&A)[N * N]) {
#pragma unroll
  for (int e = 0; e < N * N; ++e) A[e] = p[(int64_t)e * nb + b];
}
template <int N>
__device__ __forceinline__ void store_mat(double* __restrict__ p, int64_t nb, int64_t b, const double (&A)[N * N]) {
#pragma unroll
  for (int e = 0; e < N * N; ++e) p[(int64_t)e * nb + b] = A[e];
}
template <int N>
__device__ __forceinline__ void load_piv(const int32_t* __restrict__ p, int64_t nb, int64_t b, int (&P)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) P[k] = p[(int64_t)k * nb + b];
}
template <int N>
__device__ __forceinline__ void store_piv(int32_t* __restrict__ p, int64_t nb, int64_t b, const int (&P)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) p[(int64_t)k * nb + b] = P[k];
}
template <int N>
__device__ __forceinline__ void load_vec(const double* __restrict__ p, int64_t nb, int64_t b, double (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = p[(int64_t)k * nb + b];
}
template <int N>
__device__ __forceinline__ void store_vec(double* __restrict__ p, int64_t nb, int64_t b, const double (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) p[(int64_t)k * nb + b] = v[k];
}


}  // namespace dsh
