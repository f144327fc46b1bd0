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
// lu_solve_reg with the reciprocals of U's diagonal supplied (dinv[i] = 1 / A[i * N + i], computed once per factorisation): the back substitution multiplies instead
// of dividing.  NOT the arithmetic of the reference's solve — used by the opt-in fast variant of the device-resident BDF only (dsh_adaptive_fast.hip).
template <int N>
__device__ __forceinline__ bool lu_solve_reg_inv(const double (&A)[N * N], const double (&dinv)[N], const int (&P)[N], double (&v)[N]) {
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
    if (A[i * N + i] == 0.0) ok = false;
    double coeff = v[i] * dinv[i];
    v[i] = coeff;
#pragma unroll
    for (int r = 0; r < i; ++r) v[r] = (-coeff) * A[i * N + r] + v[r];
  }
  return ok;
}

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


// ---- banded LU on per-lane arrays (device-resident integrators for banded models): the algorithm of dsh_lu_band.hpp — LAPACK dgbtrf-style partial
// pivoting with a (K+1) x (2K+1) register window, the non-trivial operations of the dense elimination in the same order — on the matrix
// A = Jb * (-c) + I, where Jb holds the band of the Jacobian: entry (i, col) at Jb[(col - i + K) * N + i].  Factors: U(r, r+d) at Uf[d*N + r] (d <= 2K),
// multiplier of row j+r at step j at Lf[(r-1)*N + j].  band_factor_lane_fn takes the entries of the matrix from a function instead (models with a mass
// matrix: J (-c) + M; the InitOp Jacobian of the consistent initialisation).
template <int N, int K, class ENTRY>
__device__ __forceinline__ void band_factor_lane_fn(ENTRY&& in_band, double* Lf, double* Uf, int* P, bool& singular) {
  constexpr int R = K + 1, C = 2 * K + 1;
  auto in = [&](int i, int col) -> double {  // in_band(i, col): entry (i, col) of the matrix, asked for |i - col| <= K inside the matrix only
    if (i >= N || col >= N) return 0.0;
    return in_band(i, col);
  };
  double W[R][C];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < C; ++q) W[r][q] = (q - r <= K && r - q <= K) ? in(r, q) : 0.0;
  for (int j = 0; j < N; ++j) {
    int p = 0;
    double best = fabs(W[0][0]);
#pragma unroll
    for (int r = 1; r < R; ++r) {
      const double v = fabs(W[r][0]);
      if (v > best) { best = v; p = r; }
    }
    double diag = W[0][0];
#pragma unroll
    for (int r = 1; r < R; ++r) diag = (r == p) ? W[r][0] : diag;
    if (diag == 0.0) {
      P[j] = j;
      singular = true;
    } else {
      P[j] = j + p;
#pragma unroll
      for (int q = 0; q < C; ++q) {
        const double top = W[0][q];
        double picked = top;
#pragma unroll
        for (int r = 1; r < R; ++r) {
          const bool sel = (r == p);
          const double cur = W[r][q];
          picked = sel ? cur : picked;
          W[r][q] = sel ? top : cur;
        }
        W[0][q] = picked;
      }
      const double inv_diag = 1.0 / diag;
#pragma unroll
      for (int r = 1; r < R; ++r) W[r][0] = W[r][0] * inv_diag;
#pragma unroll
      for (int q = 1; q < C; ++q) {
        const double pr = W[0][q];
#pragma unroll
        for (int r = 1; r < R; ++r) W[r][q] = (-pr) * W[r][0] + W[r][q];
      }
    }
#pragma unroll
    for (int q = 0; q < C; ++q) Uf[q * N + j] = W[0][q];
#pragma unroll
    for (int r = 1; r < R; ++r) Lf[(r - 1) * N + j] = W[r][0];
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) {
#pragma unroll
      for (int q = 0; q + 1 < C; ++q) W[r][q] = W[r + 1][q + 1];
      W[r][C - 1] = 0.0;
    }
    const int i = j + 1 + K;
#pragma unroll
    for (int q = 0; q < C; ++q) W[R - 1][q] = in(i, i - K + q);
  }
}

template <int N, int K>
__device__ __forceinline__ void band_factor_lane(const double* Jb, double c, double* Lf, double* Uf, int* P, bool& singular) {
  // J * (-c) + M with M = from_diagonal(ones) (op/bdf.rs:138-141, :273-300)
  band_factor_lane_fn<N, K>([&](int i, int col) -> double { return Jb[(col - i + K) * N + i] * (-c) + (i == col ? 1.0 : 0.0); }, Lf, Uf, P, singular);
}

// returns false if a zero diagonal of U was met (LuSolveFailed)
template <int N, int K>
__device__ __forceinline__ bool band_solve_lane(const double* Lf, const double* Uf, const int* P, double* v) {
  constexpr int R = K + 1, C = 2 * K + 1;
  double w[R];
#pragma unroll
  for (int r = 0; r < R; ++r) w[r] = r < N ? v[r] : 0.0;
  for (int j = 0; j < N; ++j) {
    const int pv = P[j] - j;
    const double top = w[0];
    double x = top;
#pragma unroll
    for (int r = 1; r < R; ++r) {
      const bool sel = (r == pv);
      const double cur = w[r];
      x = sel ? cur : x;
      w[r] = sel ? top : cur;
    }
    v[j] = x;
#pragma unroll
    for (int r = 1; r < R; ++r) w[r] = (-x) * Lf[(r - 1) * N + j] + w[r];
#pragma unroll
    for (int r = 0; r + 1 < R; ++r) w[r] = w[r + 1];
    w[R - 1] = j + 1 + K < N ? v[j + 1 + K] : 0.0;
  }
  bool ok = true;
  double u[C];
#pragma unroll
  for (int q = 0; q < C; ++q) { const int r = N - 1 - (C - 1) + q; u[q] = r >= 0 ? v[r] : 0.0; }
  for (int i = N - 1; i >= 0; --i) {
    const double diag = Uf[i];
    if (diag == 0.0) ok = false;
    const double x = u[C - 1] / diag;
    v[i] = x;
#pragma unroll
    for (int d = 1; d < C; ++d) u[C - 1 - d] = (i - d >= 0) ? (-x) * Uf[d * N + (i - d)] + u[C - 1 - d] : u[C - 1 - d];
#pragma unroll
    for (int q = C - 1; q > 0; --q) u[q] = u[q - 1];
    u[0] = i - C >= 0 ? v[i - C] : 0.0;
  }
  return ok;
}

}  // namespace dsh
