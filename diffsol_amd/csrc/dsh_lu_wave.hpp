// Dense LU for 8 < n <= 64: one system per group of GW lanes of a wavefront, lane = matrix row, the row held in registers (gfx950).
//
// Between the one-lane-per-system register kernels (n <= 8, dsh_lu_dev.hpp) and the LDS-resident workgroup kernels (dsh_lu_coop.hpp)
// there is the size range where a whole system still fits the register file of ONE wavefront: 64 lanes x NP doubles.  Lane r keeps row r
// (NP = n rounded up to 16/32/48/64 doubles); everything a pivot step needs from another row — the pivot row, the row it is swapped with —
// comes through lane broadcasts (v_readlane when the group is the whole wave, ds_bpermute when several systems share a wave), so the
// factorisation needs no LDS, no workgroup barrier and no scratch, and the factors cross HBM exactly once in each direction.
//
// The elimination loop over k stays rolled; each step jumps (k is wave-uniform) to straight-line code specialised for its column index, so
// that every register index is static and only the columns right of the pivot are touched; column k of the current step is carried in its own
// register (`colk`, refreshed from column k+1 at the end of the previous step).  Rows are never moved between lanes: an interchange swaps two
// position numbers, and each row is stored at its final position at the end.  Per-element arithmetic is that of lu_factor_reg / the oracle:
// l = a * (1/pivot); a_rc = (-u_kc) * l_rk + a_rc; first maximum wins the pivot search — bit-identical factors.
//
// Factors are system-major (b*n*n + c*n + r), in place: the caller transposes the batch-fastest operand once (k_soa_to_aos).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dsh {

constexpr int kWaveLuThreads = 512;  // 8 wavefronts: the 8 systems whose batch-fastest rhs entries share one 64-byte line

template <int GW>
__device__ __forceinline__ double group_bcast(double x, int src) {
  if constexpr (GW == 64) {  // src is wave-uniform: scalar broadcast, no LDS crossbar
    const int s = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), s);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), s);
    return __hiloint2double(hi, lo);
  } else {
    return __shfl(x, src, GW);
  }
}
template <int GW>
__device__ __forceinline__ int group_bcast_i(int x, int src) {
  if constexpr (GW == 64) return __builtin_amdgcn_readlane(x, __builtin_amdgcn_readfirstlane(src));
  else return __shfl(x, src, GW);
}

// Elimination step with a compile-time column index K: rows below the pivot get l in column K and the rank-1 update in columns K+1.. .
// P is the lane that holds the pivot row.
template <int K, int NP, int GW>
__device__ __forceinline__ void elim_step(double (&a)[NP], double l, bool below, int P, double& nextcol) {
  // Broadcasts stay outside predicated code: reading a lane that is masked off (the pivot lane is never `below`) is undefined.
  if (below) a[K] = l;
  if constexpr (GW == 64) {
    // chunks of 8 columns: the pivot-row entries of a chunk are read into scalar registers with all lanes active, then the rows below the pivot
    // update under ONE predicate per chunk (2 v_readlane + v_mul + v_add per element instead of + 2 v_cndmask)
    constexpr int CH = 8;
#pragma unroll
    for (int c0 = K + 1; c0 < NP; c0 += CH) {
      double u[CH];
#pragma unroll
      for (int q = 0; q < CH; ++q) u[q] = (c0 + q < NP) ? group_bcast<GW>(a[(c0 + q < NP) ? c0 + q : K], P) : 0.0;
      if (below) {
#pragma unroll
        for (int q = 0; q < CH; ++q) if (c0 + q < NP) a[(c0 + q < NP) ? c0 + q : K] = (-u[q]) * l + a[(c0 + q < NP) ? c0 + q : K];
      }
    }
  } else {
#pragma unroll
    for (int c = K + 1; c < NP; ++c) {
      const double u = group_bcast<GW>(a[c], P);
      const double upd = (-u) * l + a[c];
      a[c] = below ? upd : a[c];
    }
  }
  if constexpr (K + 1 < NP) nextcol = a[K + 1];
}

// Rows never move between lanes: `pos` is the position the lane's row currently has in the (conceptually) interchanged matrix.  A pivot step
// picks the candidate with the largest |a| — smallest position on ties, i.e. the first maximum of the sequential scan — and exchanges two
// position numbers instead of two rows; a row's final position is the step at which it became the pivot row, and that is where it is stored.
// The factorisation itself, on rows already in registers (used by the kernel below and by the wavefront-per-member integrator): on return lane L's
// row holds its L and U entries, `pos` the row's final position (P A = L U with row pos^-1(k) of A at position k), `mypiv` the LAPACK-style pivot
// record of position `pos`.  `live` / `rowlive`: this group holds a system / this lane holds one of its rows.
template <int NP, int GW>
__device__ __forceinline__ void wave_lu_factor_rows(double (&a)[NP], int n, bool live, bool rowlive, int gl, int gbase, int& pos, int& mypiv, bool& singular) {
  double colk = a[0];
  pos = gl; mypiv = gl;
  for (int k = 0; k < n; ++k) {
    double best = -1.0;
    int p = n;
    if (rowlive && pos >= k) { const double v = fabs(colk); if (v > best) { best = v; p = pos; } }
    group_argmax(best, p, GW);
    if (p >= n) p = k;  // NaN column: keep the diagonal like the sequential scan
    const unsigned long long holders = __ballot(rowlive && pos == p) >> gbase;
    const int P = __ffsll((unsigned long long)(GW == 64 ? holders : (holders & ((1ull << (GW & 63)) - 1ull)))) - 1;  // lane (in group) holding position p
    const double diag = group_bcast<GW>(colk, P);
    const bool elim = live && diag != 0.0;  // a zero pivot leaves the rows where they are (lu_factor_reg does the same)
    if (elim) {
      if (pos == k) pos = p;  // the row that sat at position k moves to p ...
      if (gl == P) pos = k;   // ... and the pivot row takes position k
    } else {
      singular = true;
      p = k;
    }
    if (pos == k) mypiv = p;
    const bool below = elim && rowlive && pos > k;
    const double l = colk * (1.0 / diag);
    double nextcol = 0.0;
#define DSH_ELIM_CASE(K) case K: if constexpr (K < NP) elim_step<K, NP, GW>(a, l, below, P, nextcol); break;
    switch (k) {
      DSH_ELIM_CASE(0) DSH_ELIM_CASE(1) DSH_ELIM_CASE(2) DSH_ELIM_CASE(3) DSH_ELIM_CASE(4) DSH_ELIM_CASE(5) DSH_ELIM_CASE(6) DSH_ELIM_CASE(7)
      DSH_ELIM_CASE(8) DSH_ELIM_CASE(9) DSH_ELIM_CASE(10) DSH_ELIM_CASE(11) DSH_ELIM_CASE(12) DSH_ELIM_CASE(13) DSH_ELIM_CASE(14) DSH_ELIM_CASE(15)
      DSH_ELIM_CASE(16) DSH_ELIM_CASE(17) DSH_ELIM_CASE(18) DSH_ELIM_CASE(19) DSH_ELIM_CASE(20) DSH_ELIM_CASE(21) DSH_ELIM_CASE(22) DSH_ELIM_CASE(23)
      DSH_ELIM_CASE(24) DSH_ELIM_CASE(25) DSH_ELIM_CASE(26) DSH_ELIM_CASE(27) DSH_ELIM_CASE(28) DSH_ELIM_CASE(29) DSH_ELIM_CASE(30) DSH_ELIM_CASE(31)
      DSH_ELIM_CASE(32) DSH_ELIM_CASE(33) DSH_ELIM_CASE(34) DSH_ELIM_CASE(35) DSH_ELIM_CASE(36) DSH_ELIM_CASE(37) DSH_ELIM_CASE(38) DSH_ELIM_CASE(39)
      DSH_ELIM_CASE(40) DSH_ELIM_CASE(41) DSH_ELIM_CASE(42) DSH_ELIM_CASE(43) DSH_ELIM_CASE(44) DSH_ELIM_CASE(45) DSH_ELIM_CASE(46) DSH_ELIM_CASE(47)
      DSH_ELIM_CASE(48) DSH_ELIM_CASE(49) DSH_ELIM_CASE(50) DSH_ELIM_CASE(51) DSH_ELIM_CASE(52) DSH_ELIM_CASE(53) DSH_ELIM_CASE(54) DSH_ELIM_CASE(55)
      DSH_ELIM_CASE(56) DSH_ELIM_CASE(57) DSH_ELIM_CASE(58) DSH_ELIM_CASE(59) DSH_ELIM_CASE(60) DSH_ELIM_CASE(61) DSH_ELIM_CASE(62) DSH_ELIM_CASE(63)
      default: break;
    }
#undef DSH_ELIM_CASE
    colk = nextcol;
  }
}

// Solve with the rows where the factorisation left them (whole-wavefront groups): lane L keeps ITS right-hand-side entry — it travels with the
// lane's row, i.e. it sits at position pos[L] of P b without any data movement.  Positions are eliminated in order; the lane holding position k is
// found with a ballot and read with v_readlane.  On return lane L holds the unknown number pos[L]; `unknown_of_lane` moves x_i to lane i.
// Arithmetic and order are those of lu_solve_reg / the oracle (column-oriented substitutions).
template <int NP>
__device__ __forceinline__ bool wave_lu_solve_rows(const double (&a)[NP], int n, bool rowlive, int pos, double& v) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < NP; ++k) {  // L y = P b (unit lower triangle)
    if (k + 1 < n) {
      const int holder = __ffsll((unsigned long long)__ballot(rowlive && pos == k)) - 1;
      const double coeff = group_bcast<64>(v, holder);
      if (rowlive && pos > k) v = (-coeff) * a[k] + v;
    }
  }
#pragma unroll
  for (int k = NP - 1; k >= 0; --k) {  // U x = y
    if (k < n) {
      const int holder = __ffsll((unsigned long long)__ballot(rowlive && pos == k)) - 1;
      const double diag = group_bcast<64>(a[k], holder);
      if (diag == 0.0) ok = false;
      const double coeff = group_bcast<64>(v, holder) / diag;
      if (rowlive && pos == k) v = coeff;
      else if (rowlive && pos < k) v = (-coeff) * a[k] + v;
    }
  }
  return ok;
}

template <int NP, int GW>
__global__ __launch_bounds__(kWaveLuThreads) void k_lu_factor_wave(int n, int64_t nb, double* __restrict__ f_aos, int32_t* __restrict__ piv_aos,
                                                                   unsigned long long* singular_word, unsigned int epoch) {
  const int gl = threadIdx.x % GW;
  const int gbase = (threadIdx.x & 63) - gl;  // first lane of this group inside the wavefront
  const int64_t b = ((int64_t)blockIdx.x * kWaveLuThreads + threadIdx.x) / GW;
  const bool live = b < nb, rowlive = live && gl < n;
  double* F = f_aos + (size_t)(live ? b : 0) * n * n;
  double a[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) a[c] = (rowlive && c < n) ? F[(size_t)c * n + gl] : 0.0;
  int pos, mypiv;
  bool singular = false;
  wave_lu_factor_rows<NP, GW>(a, n, live, rowlive, gl, gbase, pos, mypiv, singular);
  if (rowlive) {
#pragma unroll
    for (int c = 0; c < NP; ++c) if (c < n) F[(size_t)c * n + pos] = a[c];
    piv_aos[(size_t)b * n + pos] = mypiv;
  }
  if (singular && live && gl == 0) publish_singular(singular_word, 1ull, epoch);
}

// Solve in place: rhs is batch-fastest, factors / pivots system-major.
template <int NP, int GW>
__global__ __launch_bounds__(kWaveLuThreads) void k_lu_solve_wave(int n, int64_t nb, const double* __restrict__ f_aos, const int32_t* __restrict__ piv_aos,
                                                                  double* __restrict__ rhs, unsigned long long* rec, unsigned int seq) {
  const int gl = threadIdx.x % GW;
  const int64_t b = ((int64_t)blockIdx.x * kWaveLuThreads + threadIdx.x) / GW;
  const bool live = b < nb, rowlive = live && gl < n;
  const double* F = f_aos + (size_t)(live ? b : 0) * n * n;
  double a[NP];
#pragma unroll
  for (int c = 0; c < NP; ++c) a[c] = (rowlive && c < n) ? F[(size_t)c * n + gl] : 0.0;
  double v = rowlive ? rhs[(int64_t)gl * nb + b] : 0.0;
  const int mypiv = rowlive ? piv_aos[(size_t)b * n + gl] : gl;
  for (int i = 0; i < n; ++i) {  // the recorded interchanges, in order
    const int p = group_bcast_i<GW>(mypiv, i);
    if (p != i) {
      const double vi = group_bcast<GW>(v, i), vp = group_bcast<GW>(v, p);
      if (gl == i) v = vp; else if (gl == p) v = vi;
    }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) {  // L y = P b  (unit lower triangle)
    if (k + 1 < n) {
      const double coeff = group_bcast<GW>(v, k);
      if (gl > k) v = (-coeff) * a[k] + v;
    }
  }
  bool ok = true;
#pragma unroll
  for (int k = NP - 1; k >= 0; --k) {  // U x = y
    if (k < n) {
      const double diag = group_bcast<GW>(a[k], k);
      if (diag == 0.0) ok = false;
      const double coeff = group_bcast<GW>(v, k) / diag;
      if (gl == k) v = coeff; else if (gl < k) v = (-coeff) * a[k] + v;
    }
  }
  if (rowlive) rhs[(int64_t)gl * nb + b] = v;
  block_publish(0ull, 0ull, (live && gl == 0 && !ok) ? 1ull : 0ull, rec, seq);
}

}  // namespace dsh
