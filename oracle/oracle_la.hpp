// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the linear-algebra semantics diffsol's BDF/SDIRK hot path relies on.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// Follows (paths relative to /root/reference):
//   crates/diffsol-la/src/vector/nalgebra_serial.rs:395-408   squared_norm (mean of squares)
//   crates/diffsol-la/src/vector/cuda.rs:1421-1432             batched reduction = max over batches
//   crates/diffsol-la/src/vector/nalgebra_serial.rs:484-504    root_finding
//   crates/diffsol-la/src/matrix/dense_nalgebra_serial.rs:325-329  scale_add_and_assign = y*beta + x
//   crates/diffsol-la/src/linear_solver/nalgebra/lu.rs:30-64   NalgebraLU (clone + nalgebra::LU)
// Third-party arithmetic not present in the reference tree: nalgebra 0.35 `LU::new` / `LU::solve_mut`
// (Cargo.toml semver requirement, no lockfile).  Its published algorithm is restated in `DenseLU`:
// partial pivoting by first-max |a|, multipliers formed as a*(1/pivot), rank-1 update
// a_rk = (-a_ik)*l_ri + a_rk, column-oriented triangular solves.  Compiled with -ffp-contract=off
// so that the operation order below *is* the arithmetic.
#pragma once
#include <cmath>
#include "../include/diffsol_detpow.h"
#include <cstddef>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace orc {

// Batched vector, batch-major like the reference API ([b0 states..., b1 states...],
// crates/diffsol-la/src/vector/cuda.rs:119-125).  nb==1 operands broadcast
// (crates/diffsol-la/src/context/mod.rs:24-26).
struct V {
  int n = 0;
  int nb = 1;
  std::vector<double> d;
  V() = default;
  V(int n_, int nb_, double val = 0.0) : n(n_), nb(nb_), d((size_t)n_ * nb_, val) {}
  double& at(int b, int i) { return d[(size_t)b * n + i]; }
  double at(int b, int i) const { return d[(size_t)(nb == 1 ? 0 : b) * n + i]; }
  size_t size() const { return d.size(); }
};

inline void check_same(const V& a, const V& b) {
  if (a.n != b.n || a.nb != b.nb) throw std::runtime_error("oracle: vector shape mismatch");
}

inline void copy_from(V& y, const V& x) { check_same(y, x); y.d = x.d; }
inline void fill(V& y, double v) { for (auto& e : y.d) e = v; }
inline void add_assign(V& y, const V& x) { check_same(y, x); for (size_t i = 0; i < y.d.size(); ++i) y.d[i] = y.d[i] + x.d[i]; }
inline void sub_assign(V& y, const V& x) { check_same(y, x); for (size_t i = 0; i < y.d.size(); ++i) y.d[i] = y.d[i] - x.d[i]; }
inline void mul_assign(V& y, double s) { for (auto& e : y.d) e = e * s; }
// y = alpha*x + beta*y   (nalgebra axcpy with c=1: alpha*x*1 + beta*y; beta==0 -> alpha*x)
inline void axpy(V& y, double alpha, const V& x, double beta) {
  check_same(y, x);
  if (beta == 0.0) { for (size_t i = 0; i < y.d.size(); ++i) y.d[i] = alpha * x.d[i]; }
  else { for (size_t i = 0; i < y.d.size(); ++i) y.d[i] = alpha * x.d[i] + beta * y.d[i]; }
}

// max over batches of mean_i (x_i / (|y_i| rtol + atol_i))^2 ; NaN propagates (deliberate: the
// reference's CUDA max drops NaN lanes, its CPU path (nb==1) returns NaN).
inline double squared_norm(const V& x, const V& y, const V& atol, double rtol) {
  if (x.n != y.n || x.n != atol.n) throw std::runtime_error("oracle: squared_norm length mismatch");
  if (x.n == 0) return 0.0;
  double mx = 0.0;
  for (int b = 0; b < x.nb; ++b) {
    double acc = 0.0;
    for (int i = 0; i < x.n; ++i) {
      double term = x.at(b, i) / (std::fabs(y.at(b, i)) * rtol + atol.at(b, i));
      acc += term * term;
    }
    double nrm = acc / (double)x.n;
    if (nrm > mx || nrm != nrm) mx = nrm;
  }
  return mx;
}

// Batched dense matrix, column-major per batch then batch-contiguous
// (crates/diffsol-la/src/matrix/cuda.rs:20-31).
struct M {
  int nr = 0, nc = 0, nb = 1;
  std::vector<double> d;
  M() = default;
  M(int nr_, int nc_, int nb_) : nr(nr_), nc(nc_), nb(nb_), d((size_t)nr_ * nc_ * nb_, 0.0) {}
  double& at(int b, int i, int j) { return d[((size_t)b * nc + j) * nr + i]; }
  double at(int b, int i, int j) const { return d[((size_t)(nb == 1 ? 0 : b) * nc + j) * nr + i]; }
  V column(int j) const {
    V v(nr, nb);
    for (int b = 0; b < nb; ++b) for (int i = 0; i < nr; ++i) v.at(b, i) = at(b, i, j);
    return v;
  }
  void set_column(int j, const V& v) {
    if (v.n != nr || v.nb != nb) throw std::runtime_error("oracle: set_column shape mismatch");
    for (int b = 0; b < nb; ++b) for (int i = 0; i < nr; ++i) at(b, i, j) = v.at(b, i);
  }
  // column i += alpha * column j  (dense_nalgebra_serial.rs:398-416)
  void column_axpy(double alpha, int j, int i) {
    for (int b = 0; b < nb; ++b) for (int k = 0; k < nr; ++k) at(b, k, i) = at(b, k, i) + alpha * at(b, k, j);
  }
  static M identity(int n, int nb) {
    M m(n, n, nb);
    for (int b = 0; b < nb; ++b) for (int i = 0; i < n; ++i) m.at(b, i, i) = 1.0;
    return m;
  }
};

// self = y*beta + x  (dense_nalgebra_serial.rs:325-329: copy y, mul by beta, add x)
inline void scale_add_and_assign(M& self, const M& x, double beta, const M& y) {
  for (size_t k = 0; k < self.d.size(); ++k) self.d[k] = y.d[k] * beta + x.d[k];
}

// nalgebra small-matrix gemm path (blas.rs gemm -> per-column gemv -> axcpy), beta = 0, alpha = 1:
// out[:,j] = A[:,0]*B[0,j]; then out[:,j] = A[:,k]*B[k,j] + out[:,j] for k=1..
// B is a broadcast (nb==1) matrix.  Used for D[:,0..k+1] <- D[:,0..k+1]*(R*U) (bdf.rs:568-577).
inline void gemm_cols(M& out, const M& a, int ka, const M& bm) {
  for (int b = 0; b < a.nb; ++b)
    for (int j = 0; j < bm.nc; ++j)
      for (int i = 0; i < a.nr; ++i) {
        double acc = a.at(b, i, 0) * bm.at(0, 0, j);
        for (int k = 1; k < ka; ++k) acc = a.at(b, i, k) * bm.at(0, k, j) + acc;
        out.at(b, i, j) = acc;
      }
}
inline M mat_mul_small(const M& a, const M& b) {  // nb == 1 both
  M out(a.nr, b.nc, 1);
  gemm_cols(out, a, a.nc, b);
  return out;
}

// nalgebra 0.35 LU::new / solve_mut restated (see header comment), one system per batch member.
struct DenseLU {
  int n = 0, nb = 0;
  std::vector<double> lu;                  // [b][col-major n*n]
  std::vector<int> piv;                    // [b][n]  row swapped with row k at step k (k if none)
  void factor(const M& a) {
    n = a.nr; nb = a.nb;
    lu = a.d;
    piv.assign((size_t)n * nb, 0);
    for (int b = 0; b < nb; ++b) {
      double* A = lu.data() + (size_t)b * n * n;
      int* P = piv.data() + (size_t)b * n;
      for (int i = 0; i < n; ++i) {
        int p = i; double best = std::fabs(A[i * n + i]);
        for (int r = i + 1; r < n; ++r) { double v = std::fabs(A[i * n + r]); if (v > best) { best = v; p = r; } }
        P[i] = p;
        double diag = A[i * n + p];
        if (diag == 0.0) { P[i] = i; continue; }    // nalgebra: no non-zero entry in this column, skip
        if (p != i) for (int c = 0; c < n; ++c) std::swap(A[c * n + i], A[c * n + p]);
        double inv_diag = 1.0 / diag;
        for (int r = i + 1; r < n; ++r) A[i * n + r] = A[i * n + r] * inv_diag;
        for (int c = i + 1; c < n; ++c) {
          double pr = A[c * n + i];
          for (int r = i + 1; r < n; ++r) A[c * n + r] = (-pr) * A[i * n + r] + A[c * n + r];
        }
      }
    }
  }
  // returns false if a zero diagonal is met (nalgebra solve_mut -> LuSolveFailed, lu.rs:36-40)
  bool solve(V& x) const {
    if (x.n != n || x.nb != nb) throw std::runtime_error("oracle: LU solve shape mismatch");
    bool ok = true;
    for (int b = 0; b < nb; ++b) {
      const double* A = lu.data() + (size_t)b * n * n;
      const int* P = piv.data() + (size_t)b * n;
      double* v = x.d.data() + (size_t)b * n;
      for (int i = 0; i < n; ++i) if (P[i] != i) std::swap(v[i], v[P[i]]);
      for (int i = 0; i + 1 < n; ++i) {
        double coeff = v[i];
        for (int r = i + 1; r < n; ++r) v[r] = (-coeff) * A[i * n + r] + v[r];
      }
      for (int i = n - 1; i >= 0; --i) {
        double diag = A[i * n + i];
        if (diag == 0.0) { ok = false; break; }
        double coeff = v[i] / diag;
        v[i] = coeff;
        for (int r = 0; r < i; ++r) v[r] = (-coeff) * A[i * n + r] + v[r];
      }
    }
    return ok;
  }
};

// The reference's other CPU solver, FaerLU (diffsol-la/src/linear_solver/faer/lu.rs:12-56), wraps faer's `FullPivLu` (faer is a Cargo dependency, absent
// from /root/reference; faer 0.24 per the workspace Cargo.toml:29): Gaussian elimination with COMPLETE pivoting, P A Q = L U, the pivot of step k being the
// entry of largest magnitude of the trailing block.  This is that published algorithm, unblocked.  faer's own evaluation order (blocked rank updates, SIMD
// reductions inside its kernels) is not restated — **parity unpinned** for this variant beyond the reference's own known answer (lu.rs:60-73, a diagonal
// operator) and agreement to rounding with the partial-pivoting restatement above; nothing on the HIP path depends on it (CudaLU pivots by column).
struct FullPivLU {
  int n = 0, nb = 0;
  std::vector<double> lu;        // [b][col-major n*n]
  std::vector<int> rperm, cperm; // [b][n]: row / column swapped with k at step k
  void factor(const M& a) {
    n = a.nr; nb = a.nb;
    lu = a.d;
    rperm.assign((size_t)n * nb, 0); cperm.assign((size_t)n * nb, 0);
    for (int b = 0; b < nb; ++b) {
      double* A = lu.data() + (size_t)b * n * n;
      int* RP = rperm.data() + (size_t)b * n; int* CP = cperm.data() + (size_t)b * n;
      for (int k = 0; k < n; ++k) {
        int pr = k, pc = k; double best = -1.0;
        for (int c = k; c < n; ++c) for (int r = k; r < n; ++r) { const double v = std::fabs(A[c * n + r]); if (v > best) { best = v; pr = r; pc = c; } }
        RP[k] = pr; CP[k] = pc;
        if (best == 0.0) continue;  // the trailing block is zero: rank-deficient, U gets zero diagonals from here on
        if (pr != k) for (int c = 0; c < n; ++c) std::swap(A[c * n + k], A[c * n + pr]);
        if (pc != k) for (int r = 0; r < n; ++r) std::swap(A[k * n + r], A[pc * n + r]);
        const double inv = 1.0 / A[k * n + k];
        for (int r = k + 1; r < n; ++r) A[k * n + r] *= inv;
        for (int c = k + 1; c < n; ++c) { const double u = A[c * n + k]; for (int r = k + 1; r < n; ++r) A[c * n + r] -= A[k * n + r] * u; }
      }
    }
  }
  bool solve(V& x) const {
    bool ok = true;
    for (int b = 0; b < nb; ++b) {
      const double* A = lu.data() + (size_t)b * n * n;
      const int* RP = rperm.data() + (size_t)b * n; const int* CP = cperm.data() + (size_t)b * n;
      double* v = x.d.data() + (size_t)b * n;
      for (int k = 0; k < n; ++k) if (RP[k] != k) std::swap(v[k], v[RP[k]]);
      for (int k = 0; k + 1 < n; ++k) for (int r = k + 1; r < n; ++r) v[r] -= A[k * n + r] * v[k];
      for (int k = n - 1; k >= 0; --k) {
        if (A[k * n + k] == 0.0) { ok = false; break; }
        v[k] /= A[k * n + k];
        for (int r = 0; r < k; ++r) v[r] -= A[k * n + r] * v[k];
      }
      for (int k = n - 1; k >= 0; --k) if (CP[k] != k) std::swap(v[k], v[CP[k]]);  // x = Q y: undo the column interchanges, last first
    }
    return ok;
  }
};

// pow as the integrators use it: libm's (what Rust's f64::powf calls), or — for bit-for-bit comparison with the device-resident kernels, which
// cannot call libm — the deterministic pow of include/diffsol_detpow.h on both sides (orc_set_det_pow).  Constants (20^1.25, eps^(2/3), ...) always
// come from libm: the device receives them from the host.
inline bool& det_pow_flag() { static bool f = false; return f; }
// event counters (diagnostic: how often a BDF run takes each of its paths; scripts/phase_frequencies.py reads them through orc_event_counts)
struct EventCounts { long pow_calls = 0, pow_first_iter = 0, pow_first_iter_eta_reset = 0, pow_first_iter_eta_reset_ts = 0, pow_rate = 0, step_size_updates = 0, order_selections = 0, powi_calls = 0; };
inline EventCounts& event_counts() { static thread_local EventCounts c; return c; }  // per thread: the multi-threaded baseline must not share a cache line
inline double rpow(double x, double y) { event_counts().pow_calls++; return det_pow_flag() ? dsh_det_pow(x, y) : std::pow(x, y); }

// compiler-rt __powidf2 (what Rust's f64::powi lowers to) — convergence.rs:85 uses `rate.pow(i32)`.
inline double powi(double a, int b) {
  const bool recip = b < 0;
  double r = 1.0;
  while (true) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return recip ? 1.0 / r : r;
}

}  // namespace orc
