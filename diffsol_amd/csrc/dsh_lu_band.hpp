// Banded batched LU (gfx950): one lane per system, the active part of the elimination in registers.
//
// Why it exists: the matrices of PDE-type and compartment models (heat equation, single-particle battery model) are dense CONTAINERS of banded
// matrices.  A dense LU spends its time multiplying by zeros; here only the band is read, eliminated and stored.  What makes this a drop-in and not an
// approximation: partial pivoting on a banded matrix only ever touches the band widened by kl (LAPACK dgbtrf), and every operation the dense
// elimination performs outside of it has an exact-zero operand ((-0)*l + a = a, (-u)*0 + a = a), so the banded factorisation below performs exactly
// the non-trivial operations of the dense kernels (dsh_lu_dev.hpp: first-max pivot, l = a * (1/pivot), a = (-u) * l + a) in the same order:
// solutions are BIT-IDENTICAL to the dense path's and to the oracle's dense LU (tests/test_gpu_lu_models.py).  Row interchanges are kept in LAPACK
// band form (not applied to earlier L columns) and applied to the right-hand side interleaved with the forward substitution — the same multiplier
// meets the same right-hand-side element in the same order as with the fully permuted dense factors.
//
// K = max(kl, ku) is a template parameter: sliding window of (K+1) x (2K+1) entries per lane (rows j..j+K, columns j..j+2K of the partially
// eliminated matrix), compile-time indices only.  Layouts (batch-fastest, coalesced across lanes): input = the dense matrix (j*n + i)*nb + b;
// U(r, r+d) at ((d)*n + r)*nb + b for d = 0..2K; multiplier of row j+r at step j at ((2K+1 + r-1)*n + j)*nb + b; pivots k*nb + b.
#pragma once
#include <type_traits>
#include "dsh_device.hpp"

namespace dsh {

// largest |i-j| below / above the diagonal with a non-zero entry, over all systems: probe[0] = kl, probe[1] = ku (atomicMax; caller zeroes)
__global__ void k_band_probe(int64_t n, int64_t nb, const double* __restrict__ a, int* __restrict__ probe) {
  const int64_t total = n * n * nb;
  int kl = 0, ku = 0;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const double v = a[idx];
    if (v != 0.0) {  // NaN counts as an entry
      const int64_t e = idx / nb;
      const int d = (int)(e % n) - (int)(e / n);  // i - j
      kl = max(kl, d);
      ku = max(ku, -d);
    }
  }
  const int wkl = (int)wave_max_u64((unsigned long long)kl), wku = (int)wave_max_u64((unsigned long long)ku);
  if ((threadIdx.x & 63) == 0) {
    if (wkl) atomicMax(&probe[0], wkl);
    if (wku) atomicMax(&probe[1], wku);
  }
}

template <int K>
__global__ __launch_bounds__(64) void k_lu_band_factor(int64_t n, int64_t nb, const double* __restrict__ a, double* __restrict__ fac, int32_t* __restrict__ piv,
                                                       unsigned long long* singular_count, unsigned int epoch) {
  constexpr int R = K + 1, C = 2 * K + 1;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long sing = 0ull;
  if (b < nb) {
    double W[R][C];
    // loads are unconditional (clamped address, value selected afterwards): a conditional load becomes a branch and the compiler then waits for
    // every load separately instead of keeping a whole chunk in flight
    auto in = [&](int64_t i, int64_t c) -> double {
      const double v = a[(min(c, n - 1) * n + min(i, n - 1)) * nb + b];
      return (i < n && c < n) ? v : 0.0;
    };
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < C; ++q) W[r][q] = (q - r <= K && r - q <= K) ? in(r, q) : 0.0;
    // the rows entering the window are fetched kPf steps ahead, so the loads of a chunk are in flight together instead of one memory latency per step
    constexpr int kPf = K == 1 ? 32 : (K == 2 ? 16 : 8);
    for (int64_t j0 = 0; j0 < n; j0 += kPf) {
      double next_row[kPf][C];
#pragma unroll
      for (int s = 0; s < kPf; ++s) {
        const int64_t i = j0 + s + 1 + K;  // row entering the window after step j0 + s; its window columns are i-K .. i+K, all inside the band
#pragma unroll
        for (int q = 0; q < C; ++q) next_row[s][q] = in(i, i - K + q);
      }
#pragma unroll
      for (int s = 0; s < kPf; ++s) {
        const int64_t j = j0 + s;
        if (j < n) {
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
            piv[j * nb + b] = (int32_t)j;
            sing = 1ull;
          } else {
            piv[j * nb + b] = (int32_t)(j + p);
#pragma unroll
            for (int q = 0; q < C; ++q) {  // interchange rows j and j+p (value selects, no dynamic register indexing)
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
          for (int q = 0; q < C; ++q) fac[((int64_t)q * n + j) * nb + b] = W[0][q];                  // U(j, j+q)
#pragma unroll
          for (int r = 1; r < R; ++r) fac[((int64_t)(C + r - 1) * n + j) * nb + b] = W[r][0];        // multiplier of row j+r
#pragma unroll
          for (int r = 0; r + 1 < R; ++r) {
#pragma unroll
            for (int q = 0; q + 1 < C; ++q) W[r][q] = W[r + 1][q + 1];
            W[r][C - 1] = 0.0;  // column j+1+2K of rows above j+1+K: outside the original band
          }
#pragma unroll
          for (int q = 0; q < C; ++q) W[R - 1][q] = next_row[s][q];
        }
      }
    }
  }
  sing = wave_sum_u64(sing);
  if ((threadIdx.x & 63) == 0 && sing) publish_singular(singular_count, sing, epoch);
}

template <int K>
__global__ __launch_bounds__(64) void k_lu_band_solve(int64_t n, int64_t nb, const double* __restrict__ fac, const int32_t* __restrict__ piv, double* __restrict__ rhs,
                                                      unsigned long long* rec, unsigned int seq) {
  constexpr int R = K + 1, C = 2 * K + 1;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad = 0ull;
  if (b < nb) {
    constexpr int kPf = K == 1 ? 32 : (K == 2 ? 16 : 8);  // chunk of steps whose operands are loaded together (one memory latency per chunk, not per step)
    // ---- forward: interchanges interleaved with the unit-lower-triangular solve; window v[0..K] = entries j..j+K
    double v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const double t = rhs[min((int64_t)r, n - 1) * nb + b]; v[r] = r < n ? t : 0.0; }
    for (int64_t j0 = 0; j0 < n; j0 += kPf) {
      double l[kPf][K], next_b[kPf];
      int pv[kPf];
#pragma unroll
      for (int s = 0; s < kPf; ++s) {
        const int64_t j = min(j0 + s, n - 1);  // clamped, unconditional loads (see k_lu_band_factor); steps beyond n are skipped below
        pv[s] = piv[j * nb + b] - (int)j;
#pragma unroll
        for (int r = 0; r < K; ++r) l[s][r] = fac[((int64_t)(C + r) * n + j) * nb + b];
        const double t = rhs[min(j0 + s + 1 + K, n - 1) * nb + b];
        next_b[s] = j0 + s + 1 + K < n ? t : 0.0;
      }
#pragma unroll
      for (int s = 0; s < kPf; ++s) {
        const int64_t j = j0 + s;
        if (j < n) {
          const double top = v[0];
          double x = top;
#pragma unroll
          for (int r = 1; r < R; ++r) {
            const bool sel = (r == pv[s]);
            const double cur = v[r];
            x = sel ? cur : x;
            v[r] = sel ? top : cur;
          }
          rhs[j * nb + b] = x;
#pragma unroll
          for (int r = 1; r < R; ++r) v[r] = (-x) * l[s][r - 1] + v[r];
#pragma unroll
          for (int r = 0; r + 1 < R; ++r) v[r] = v[r + 1];
          v[R - 1] = next_b[s];
        }
      }
    }
    // ---- backward with U (bandwidth 2K): window w[0..2K] = entries i-2K..i, column oriented like the dense solve
    double w[C];
#pragma unroll
    for (int q = 0; q < C; ++q) { const int64_t r = n - 1 - (C - 1) + q; const double t = rhs[max(r, (int64_t)0) * nb + b]; w[q] = r >= 0 ? t : 0.0; }
    for (int64_t i0 = n - 1; i0 >= 0; i0 -= kPf) {
      double u[kPf][C], next_y[kPf];
#pragma unroll
      for (int s = 0; s < kPf; ++s) {
        const int64_t i = i0 - s;
#pragma unroll
        for (int d = 0; d < C; ++d) {  // U(i-d, i)
          const double t = fac[((int64_t)d * n + max(i - d, (int64_t)0)) * nb + b];
          u[s][d] = i - d >= 0 ? t : 0.0;
        }
        const double t = rhs[max(i - C, (int64_t)0) * nb + b];
        next_y[s] = i - C >= 0 ? t : 0.0;
      }
#pragma unroll
      for (int s = 0; s < kPf; ++s) {
        const int64_t i = i0 - s;
        if (i >= 0) {
          const double diag = u[s][0];
          if (diag == 0.0) bad = 1ull;
          const double x = w[C - 1] / diag;
          rhs[i * nb + b] = x;
#pragma unroll
          for (int d = 1; d < C; ++d) w[C - 1 - d] = (-x) * u[s][d] + w[C - 1 - d];
#pragma unroll
          for (int q = C - 1; q > 0; --q) w[q] = w[q - 1];
          w[0] = next_y[s];
        }
      }
    }
  }
  block_publish(0ull, 0ull, bad, rec, seq);
}


// Banded solve for SMALL ensembles (a few thousand systems): same factors, same arithmetic in the same order as k_lu_band_solve — bit-identical
// solutions — with the memory latency taken off the sequential chain.
// One lane per system gives nb / 64 wavefronts; at BASELINE config 3's shape (n = 512, 4096 systems) that is 64 wavefronts, each walking a 2 x 512-step
// dependent chain whose operands arrive 32 steps at a time: 64 memory round trips of ~3 us per solve, 204 us, 0.53 TB/s (profiles/r01_lu_bench.md).
// Here a wavefront owns S = 8 systems.  Its 64 lanes are 8 row groups x 8 systems: every load instruction fetches one operand of 8 consecutive
// steps (64-byte segments of 8 neighbouring systems), several chunks of 16 steps ahead of the chain (up to ~40 loads, ~100 steps, in flight per
// wavefront; 8x as many wavefronts), lands in registers, and is handed through LDS to the 8 lanes that run the chain.  Results go back through LDS
// and are stored by all 64 lanes.  No data is shared between systems and no operation is reordered.
template <int K, int S>
__global__ __launch_bounds__(64) void k_lu_band_solve_wide(int64_t n, int64_t nb, const double* __restrict__ fac, const int32_t* __restrict__ piv, double* __restrict__ rhs,
                                                           unsigned long long* rec, unsigned int seq) {
  constexpr int G = 64 / S, R = K + 1, C = 2 * K + 1;
  constexpr int CHK = 16;           // steps per chunk
  constexpr int Q = CHK / G;        // load instructions per operand and chunk
  constexpr int FO = K + 2;         // forward operands per step: K multipliers, pivot offset, the entry that enters the window
  constexpr int BO = C + 1;         // backward: C entries of U, the entry that enters the window
  constexpr int MO = BO > FO ? BO : FO;
  constexpr int DF = (40 / (FO * Q)) < 2 ? 2 : (40 / (FO * Q));  // chunks in flight
  constexpr int DB = (40 / (BO * Q)) < 2 ? 2 : (40 / (BO * Q));
  static_assert(CHK % G == 0, "chunk must be a multiple of the row groups");
  __shared__ double sOp[MO][CHK][S];
  __shared__ double sOut[CHK][S];
  const int lane = threadIdx.x, s = lane % S, g = lane / S;
  const int64_t b0 = (int64_t)blockIdx.x * S + s;
  const bool valid = b0 < nb;
  const int64_t b = valid ? b0 : nb - 1;  // lanes past the ensemble shadow the last system (no stores)
  const bool chain = g == 0;
  const int64_t nch = (n + CHK - 1) / CHK;
  unsigned long long bad = 0ull;

  // ---------------------------------------------------------------- forward: interchanges interleaved with the unit-lower-triangular solve
  {
    double pf[DF][FO][Q];
    int pp[DF][Q];  // the pivot rows stay integers until they land (a conversion at issue would wait for the load)
    auto issue = [&](int64_t c, double (&pd)[FO][Q], int (&pi)[Q]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int64_t j = min(c * CHK + q * G + g, n - 1);  // clamped, unconditional loads; steps beyond n are skipped by the chain
#pragma unroll
        for (int r = 0; r < K; ++r) pd[r][q] = fac[((int64_t)(C + r) * n + j) * nb + b];
        pi[q] = piv[j * nb + b];
        pd[K + 1][q] = rhs[min(j + 1 + K, n - 1) * nb + b];
      }
    };
    double v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const double t = rhs[min((int64_t)r, n - 1) * nb + b]; v[r] = r < n ? t : 0.0; }
    // the window is loop-carried state: its loads complete here, before the prefetch starts — otherwise every first use inside the loop is given a wait
    // that drains the prefetch queue
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(v[r]));
#pragma unroll
    for (int d = 0; d < DF; ++d) issue(d, pf[d], pp[d]);
    // every trip runs all DF stages and every stage issues its loads (clamped rows past the end): with a fixed number of younger loads behind each
    // landing the compiler's s_waitcnt leaves the prefetch in flight; a conditional issue would make it wait for everything
    for (int64_t c0 = 0; c0 < nch; c0 += DF) {
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const int64_t c = c0 + d;
        {
#pragma unroll
          for (int q = 0; q < Q; ++q) {
#pragma unroll
            for (int o = 0; o < FO; ++o) if (o != K) sOp[o][q * G + g][s] = pf[d][o][q];
            sOp[K][q * G + g][s] = (double)(pp[d][q] - (int)min(c * CHK + q * G + g, n - 1));  // pivot row - step
          }
          issue(c + DF, pf[d], pp[d]);
          __builtin_amdgcn_wave_barrier();
          if (chain) {
            double l[CHK][K], pvd[CHK], nxt[CHK];
#pragma unroll
            for (int t = 0; t < CHK; ++t) {
#pragma unroll
              for (int r = 0; r < K; ++r) l[t][r] = sOp[r][t][s];
              pvd[t] = sOp[K][t][s];
              nxt[t] = sOp[K + 1][t][s];
            }
            // full chunks run without a branch per step (a scalar branch on a vector compare costs more than the step's arithmetic)
            auto steps = [&](auto full) __attribute__((always_inline)) {
              const int jb = (int)c * CHK, ni = (int)n;
#pragma unroll
              for (int t = 0; t < CHK; ++t) {
                const int j = jb + t;
                if (decltype(full)::value || j < ni) {
                  const int pv = (int)pvd[t];
                  const double top = v[0];
                  double x = top;
#pragma unroll
                  for (int r = 1; r < R; ++r) {
                    const bool sel = (r == pv);
                    const double cur = v[r];
                    x = sel ? cur : x;
                    v[r] = sel ? top : cur;
                  }
                  sOut[t][s] = x;
#pragma unroll
                  for (int r = 1; r < R; ++r) v[r] = (-x) * l[t][r - 1] + v[r];
#pragma unroll
                  for (int r = 0; r + 1 < R; ++r) v[r] = v[r + 1];
                  v[R - 1] = j + 1 + K < ni ? nxt[t] : 0.0;
                }
              }
            };
            if ((c + 1) * CHK <= n) steps(std::true_type{}); else steps(std::false_type{});
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const int64_t j = c * CHK + q * G + g;
            if (valid && j < n) rhs[j * nb + b0] = sOut[q * G + g][s];
          }
        }
      }
    }
  }
  __threadfence();  // the backward sweep reads, from other lanes, what the forward sweep stored
  // ---------------------------------------------------------------- backward with U (bandwidth 2K): chunk c covers rows n-1 - (c*CHK + t)
  {
    double pb[DB][BO][Q];
    auto issue = [&](int64_t c, double (&pd)[BO][Q]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int64_t i = max(n - 1 - (c * CHK + q * G + g), (int64_t)0);
#pragma unroll
        for (int d = 0; d < C; ++d) pd[d][q] = fac[((int64_t)d * n + max(i - d, (int64_t)0)) * nb + b];  // U(i-d, i)
        pd[C][q] = rhs[max(i - C, (int64_t)0) * nb + b];
      }
    };
    double w[C];
#pragma unroll
    for (int q = 0; q < C; ++q) { const int64_t r = n - 1 - (C - 1) + q; const double t = rhs[max(r, (int64_t)0) * nb + b]; w[q] = r >= 0 ? t : 0.0; }
#pragma unroll
    for (int q = 0; q < C; ++q) asm volatile("" : "+v"(w[q]));
#pragma unroll
    for (int d = 0; d < DB; ++d) issue(d, pb[d]);
    for (int64_t c0 = 0; c0 < nch; c0 += DB) {
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const int64_t c = c0 + d;
        {
#pragma unroll
          for (int o = 0; o < BO; ++o)
#pragma unroll
            for (int q = 0; q < Q; ++q) sOp[o][q * G + g][s] = pb[d][o][q];
          issue(c + DB, pb[d]);
          __builtin_amdgcn_wave_barrier();
          if (chain) {
            double u[CHK][C], nxt[CHK];
#pragma unroll
            for (int t = 0; t < CHK; ++t) {
#pragma unroll
              for (int d2 = 0; d2 < C; ++d2) u[t][d2] = sOp[d2][t][s];
              nxt[t] = sOp[C][t][s];
            }
            auto steps = [&](auto full) __attribute__((always_inline)) {
              const int ib = (int)n - 1 - (int)c * CHK;
#pragma unroll
              for (int t = 0; t < CHK; ++t) {
                const int i = ib - t;
                if (decltype(full)::value || i >= 0) {
                  const double diag = u[t][0];
                  if (diag == 0.0) bad = 1ull;
                  const double x = w[C - 1] / diag;
                  sOut[t][s] = x;
#pragma unroll
                  for (int d2 = 1; d2 < C; ++d2) { const double uu = i - d2 >= 0 ? u[t][d2] : 0.0; w[C - 1 - d2] = (-x) * uu + w[C - 1 - d2]; }  // as k_lu_band_solve: entries above row 0 are zeros
#pragma unroll
                  for (int q = C - 1; q > 0; --q) w[q] = w[q - 1];
                  w[0] = i - C >= 0 ? nxt[t] : 0.0;
                }
              }
            };
            if ((c + 1) * CHK <= n) steps(std::true_type{}); else steps(std::false_type{});
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const int64_t i = n - 1 - (c * CHK + q * G + g);
            if (valid && i >= 0) rhs[i * nb + b0] = sOut[q * G + g][s];
          }
        }
      }
    }
  }
  block_publish(0ull, 0ull, (chain && valid) ? bad : 0ull, rec, seq);
}

}  // namespace dsh
