// OPT-IN reordered banded solve for small ensembles (K = 1: tridiagonal systems with partial pivoting), selected by dsh_ctx_set_solve_mode(ctx,
// DSH_SOLVE_REORDERED) — never the default.  VERDICT r3 item 4.
//
// The default solve (k_lu_band_solve_team, dsh_lu_band_team.hpp) keeps the sequential order of operations of the reference's getrs, so that its solutions are
// bit-identical to the CPU path; at n = 512 x 4096 systems it is exactly as long as ONE wavefront's dependent instruction stream over 2 x 512 steps
// (57 us, 24 % of HBM; profiles/r03_band_solve.md: the floor of the bit-identical chain is ~47 us).  Going lower needs another summation order.  Here each
// sweep is cut into 16 chunks per system and every chunk is first solved as an AFFINE MAP of what enters it:
//   forward  (L, unit lower bidiagonal, interchanges interleaved):  the value carried into step j is the only state; a step maps it to
//            carried' = b[j+1] - l_j carried  (no interchange)   or   carried' = carried - l_j b[j+1]  (interchange)  — affine either way, so a chunk is
//            (A, B): carried_out = A + B carried_in   (2 multiply-adds per step);
//   backward (U with two super-diagonals):  x_i = (y_i - u1_i x_{i+1} - u2_i x_{i+2}) / d_i: the state is (x_{i+1}, x_{i+2}); a chunk maps it to
//            (x_first, x_first+1) = M (x_{last+1}, x_{last+2}) + off   (x_i = p_i + q_i s1 + r_i s2: three coupled recurrences);
// then the 16 maps of a system are composed (15 short steps, through LDS), and every chunk runs its steps again with the true incoming values.  Twice the
// arithmetic, 1/8 of the dependent chain, one lane per (system, chunk): 65 536 lanes at 512 x 4096 instead of 4096 — the kernel becomes a stream over the
// factors (HBM-bound) instead of a latency chain.  Results differ from the default solve in the last bits (another association of the same sums; the division
// is a multiplication by the reciprocal): tests hold it to a relative 1e-12 on diagonally dominant systems and C3's solution to 1e-9 of the oracle.
// Layouts: k_lu_band_solve's (dsh_lu_band.hpp): U(i-d, i) at fac[(d n + i - d) nb + b], d = 0..2; multipliers at fac[(3 n + j) nb + b]; pivots piv[j nb + b].
#pragma once
#include "dsh_lu_band.hpp"

namespace dsh {

constexpr int kAffSys = 16, kAffChunks = 16, kAffThreads = kAffSys * kAffChunks;

template <int CLMAX>
__global__ __launch_bounds__(kAffThreads) void k_lu_band_solve_affine(int64_t n, int64_t nb, const double* __restrict__ fac, const int32_t* __restrict__ piv,
                                                                      double* __restrict__ rhs, unsigned long long* rec, unsigned int seq) {
  constexpr int S = kAffSys, NC = kAffChunks;
  __shared__ double sA[NC][S], sB[NC][S];
  __shared__ double sM[NC][6][S];
  __shared__ int sBad[S];
  const int tid = threadIdx.x, s = tid % S, c = tid / S;
  const int64_t b0i = (int64_t)blockIdx.x * S + s;
  const bool valid = b0i < nb;
  const int64_t b = valid ? b0i : nb - 1;  // lanes past the ensemble shadow the last system (no stores)
  const int ni = (int)n;
  const int CL = CLMAX;                    // rows per chunk: the launch code picks the smallest instantiation with 16 CLMAX >= n; rows past n are identity rows
  const int j0 = c * CL;
  const int len = ni - j0 < 0 ? 0 : (ni - j0 < CL ? ni - j0 : CL);
  const double* __restrict__ lfac = fac + 3 * n * nb;
  const double* __restrict__ ud = fac;                 // U(i, i)
  const double* __restrict__ u1b = fac + 1 * n * nb;   // U(i, i+1) at u1b[i nb + b]
  const double* __restrict__ u2b = fac + 2 * n * nb;   // U(i, i+2) at u2b[i nb + b]
  if (tid < S) sBad[tid] = 0;

  // Rows past the end of the matrix (the tail of the last chunk, chunks beyond n) are run as IDENTITY rows — multiplier 0, right-hand side 0, diagonal 1, no
  // interchange — formed with arithmetic masks from unconditional, clamped loads: a `cond ? load : 0` is compiled into a branch around the load with an
  // s_waitcnt vmcnt(0) behind it, i.e. one memory round trip per row (measured: 49 us for this kernel, all of it those serialised loads).
  (void)len;
  // ---------------------------------------------------------------- forward
  double bv[CLMAX + 1];   // b[j0 + t]; after phase 3: y[j0 + t]
  double lv[CLMAX];
  bool sw[CLMAX];
  int pr[CLMAX];
#pragma unroll
  for (int t = 0; t <= CLMAX; ++t) {
    const int j = j0 + t;
    const int jc = j < ni ? j : ni - 1;
    bv[t] = rhs[(int64_t)jc * nb + b];
  }
#pragma unroll
  for (int t = 0; t < CLMAX; ++t) {
    const int j = j0 + t;
    const int jc = j < ni ? j : ni - 1;
    lv[t] = lfac[(int64_t)jc * nb + b];
    pr[t] = piv[(int64_t)jc * nb + b];
  }
  // every load of the sweep is issued before its first use: without the fence the scheduler threads the recurrence between the loads and the wavefront runs
  // with a handful of loads in flight — at one wavefront per SIMD that is the whole memory-level parallelism of the kernel
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t <= CLMAX; ++t) bv[t] = bv[t] * (j0 + t < ni ? 1.0 : 0.0);
#pragma unroll
  for (int t = 0; t < CLMAX; ++t) {
    const int j = j0 + t;
    lv[t] = lv[t] * (j < ni ? 1.0 : 0.0);
    sw[t] = (pr[t] != (j < ni ? j : ni - 1)) & (j < ni);
  }
  {
    double a = c == 0 ? bv[0] : 0.0, bb = c == 0 ? 0.0 : 1.0;
#pragma unroll
    for (int t = 0; t < CLMAX; ++t) {
      const double L = lv[t], nx = bv[t + 1];
      const double a_ns = (-a) * L + nx, b_ns = (-bb) * L, a_sw = (-nx) * L + a;
      a = sw[t] ? a_sw : a_ns;
      bb = sw[t] ? bb : b_ns;
    }
    sA[c][s] = a;
    sB[c][s] = bb;
  }
  __syncthreads();
  {
    double carried = bv[0];
    if (c > 0) {
      carried = sA[0][s];
      for (int cc = 1; cc < c; ++cc) carried = sB[cc][s] * carried + sA[cc][s];
    }
#pragma unroll
    for (int t = 0; t < CLMAX; ++t) {
      const double L = lv[t], nx = bv[t + 1];
      const double x = sw[t] ? nx : carried;
      carried = (-x) * L + (sw[t] ? carried : nx);
      bv[t] = x;  // y[j0 + t]
    }
  }

  // ---------------------------------------------------------------- backward
  double inv[CLMAX], u1v[CLMAX], u2v[CLMAX];
  bool bad = false;
#pragma unroll
  for (int t = 0; t < CLMAX; ++t) {
    const int i = j0 + t;
    const int ic = i < ni ? i : ni - 1;
    inv[t] = ud[(int64_t)ic * nb + b];
    u1v[t] = u1b[(int64_t)ic * nb + b];
    u2v[t] = u2b[(int64_t)ic * nb + b];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < CLMAX; ++t) {
    const int i = j0 + t;
    const double in = i < ni ? 1.0 : 0.0;
    const double d = inv[t] * in + (1.0 - in);
    u1v[t] = u1v[t] * (i + 1 < ni ? 1.0 : 0.0);
    u2v[t] = u2v[t] * (i + 2 < ni ? 1.0 : 0.0);
    bad = bad | (d == 0.0);
    inv[t] = 1.0 / d;
  }
  {
    double p1 = 0.0, q1 = 1.0, r1 = 0.0, p2 = 0.0, q2 = 0.0, r2 = 1.0;
#pragma unroll
    for (int t = CLMAX - 1; t >= 0; --t) {
      const double u1 = u1v[t], u2 = u2v[t], iv = inv[t];
      const double p = ((-u2) * p2 + ((-u1) * p1 + bv[t])) * iv;
      const double q = ((-u2) * q2 + (-u1) * q1) * iv;
      const double r = ((-u2) * r2 + (-u1) * r1) * iv;
      p2 = p1; q2 = q1; r2 = r1;
      p1 = p; q1 = q; r1 = r;
    }
    sM[c][0][s] = p1; sM[c][1][s] = q1; sM[c][2][s] = r1;
    sM[c][3][s] = p2; sM[c][4][s] = q2; sM[c][5][s] = r2;
    if (bad) sBad[s] = 1;
  }
  __syncthreads();
  {
    double s1 = 0.0, s2 = 0.0;  // x beyond the last row
    for (int cc = NC - 1; cc > c; --cc) {
      const double n1 = sM[cc][1][s] * s1 + (sM[cc][2][s] * s2 + sM[cc][0][s]);
      const double n2 = sM[cc][4][s] * s1 + (sM[cc][5][s] * s2 + sM[cc][3][s]);
      s1 = n1; s2 = n2;
    }
#pragma unroll
    for (int t = CLMAX - 1; t >= 0; --t) {
      const int i = j0 + t;
      const double x = ((-u2v[t]) * s2 + ((-u1v[t]) * s1 + bv[t])) * inv[t];
      s2 = s1; s1 = x;
      if (valid && i < ni) rhs[(int64_t)i * nb + b] = x;
    }
  }
  block_publish(0ull, 0ull, (c == 0 && valid && sBad[s] != 0) ? 1ull : 0ull, rec, seq);
}

}  // namespace dsh
