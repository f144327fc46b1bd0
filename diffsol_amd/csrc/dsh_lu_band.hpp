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

// PACKED: the operand is a band container (diffsol_hip.h dsh_mat_band_*: entry (i, j), -kl <= j - i <= ku, at ((j - i + kl) * n + i) * nb + b) instead of a
// dense one — same entries, same eliminations
template <int K, bool PACKED = false>
__global__ __launch_bounds__(64) void k_lu_band_factor(int64_t n, int64_t nb, const double* __restrict__ a, double* __restrict__ fac, int32_t* __restrict__ piv,
                                                       unsigned long long* singular_count, unsigned int epoch, int pkl = 0, int pku = 0) {
  constexpr int R = K + 1, C = 2 * K + 1;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long sing = 0ull;
  if (b < nb) {
    double W[R][C];
    // loads are unconditional (clamped address, value selected afterwards): a conditional load becomes a branch and the compiler then waits for
    // every load separately instead of keeping a whole chunk in flight
    auto in = [&](int64_t i, int64_t c) -> double {
      if constexpr (PACKED) {
        const int64_t d = c - i, dc = min(max(d, (int64_t)-pkl), (int64_t)pku);
        const double v = a[((dc + pkl) * n + min(i, n - 1)) * nb + b];
        return (i < n && c < n && d == dc) ? v : 0.0;
      } else {
        const double v = a[(min(c, n - 1) * n + min(i, n - 1)) * nb + b];
        return (i < n && c < n) ? v : 0.0;
      }
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
// Measured at n = 512 x 4096 (scripts/ubench/band_wide_bench.hip, HIP events): one lane per system 212 us; this kernel 126 us as first written; 86 us with
// the stripped interior chunks (no end-of-matrix / interchange selects, the division's denominator half off the chain); 76 us with 32-bit row offsets
// on wavefront-uniform bases.  512 systems (one wavefront per CU) take 61 us: that is the chain itself, ~2 x (7 dependent FP64 instructions per row).
template <int K, int S>
__global__ __launch_bounds__(64) void k_lu_band_solve_wide(int64_t n, int64_t nb, const double* __restrict__ fac, const int32_t* __restrict__ piv, double* __restrict__ rhs,
                                                           unsigned long long* rec, unsigned int seq) {
  constexpr int G = 64 / S, R = K + 1, C = 2 * K + 1;
  constexpr int CHK = 16;           // steps per chunk (8 and 32 measure the same: the cost is per row)
  constexpr int Q = CHK / G;        // load instructions per operand and chunk
  constexpr int FO = K + 2;         // forward operands per step: K multipliers, pivot offset, the entry that enters the window
  constexpr int BO = C + 1;         // backward: C entries of U, the entry that enters the window
  constexpr int MO = BO > FO ? BO : FO;
  constexpr int DF = (40 / (FO * Q)) < 2 ? 2 : (40 / (FO * Q));  // chunks in flight
  constexpr int DB = (40 / (BO * Q)) < 2 ? 2 : (40 / (BO * Q));
  static_assert(CHK % G == 0, "chunk must be a multiple of the row groups");
  __shared__ double sOp[MO][CHK][S];
  __shared__ double sOut[CHK][S];
  __shared__ double sInv[CHK][S];  // backward sweep: the denominator half of the division by U's diagonal (div_refined_rcp), made by the staging lanes
  const int lane = threadIdx.x, s = lane % S, g = lane / S;
  const int64_t b0 = (int64_t)blockIdx.x * S + s;
  const bool valid = b0 < nb;
  const int64_t b = valid ? b0 : nb - 1;  // lanes past the ensemble shadow the last system (no stores)
  const bool chain = g == 0;
  const int64_t nch = (n + CHK - 1) / CHK;
  unsigned long long bad = 0ull;
  const uint32_t nb8 = (uint32_t)nb * 8u, b8 = (uint32_t)b * 8u, last8 = (uint32_t)(n - 1) * nb8 + b8;  // byte offsets: row stride, this system, its last row
  auto ld_f64 = [](const double* base, uint32_t off) __attribute__((always_inline)) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off); };
  auto ld_i32 = [](const int32_t* base, uint32_t off) __attribute__((always_inline)) { return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(base) + off); };
  auto st_f64 = [](double* base, uint32_t off, double v) __attribute__((always_inline)) { *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) = v; };

  // The wavefront is alone on its SIMD and runs one instruction stream: what a row costs is the number of instructions the chain lanes spend on it
  // (~5 cycles each, ~10 when dependent).  So every chunk that can — all of them but the ends of the matrix — runs a stripped variant of the step:
  // no end-of-matrix selects; forward, no interchange selects when no system of the wavefront interchanges in the chunk (diagonally dominant
  // matrices never do); backward, the division by the diagonal reduced to its three-instruction numerator half (dsh_device.hpp).  Same operations on
  // the same operands in the same order as the general variant, which the remaining chunks use.

  // ---------------------------------------------------------------- forward: interchanges interleaved with the unit-lower-triangular solve
  {
    double pf[DF][FO][Q];
    int pp[DF][Q];  // the pivot rows stay integers until they land (a conversion at issue would wait for the load)
    // Addresses: a wavefront-uniform base per operand plus ONE 32-bit byte offset per row (global_load saddr + voffset), advanced by a constant per
    // chunk — a 64-bit index product per load cost more instructions than the chunk's arithmetic.  The caller guarantees n * nb < 2^28.
    const double* lbase[K];
#pragma unroll
    for (int r = 0; r < K; ++r) lbase[r] = fac + (int64_t)(C + r) * n * nb;
    uint32_t offq[Q];  // byte offset of row c*CHK + q*G + g of this lane's system, for the chunk issued next
#pragma unroll
    for (int q = 0; q < Q; ++q) offq[q] = (uint32_t)(q * G + g) * nb8 + b8;
    auto issue = [&](double (&pd)[FO][Q], int (&pi)[Q]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const uint32_t o = min(offq[q], last8);  // clamped, unconditional loads; steps beyond n are skipped by the chain
#pragma unroll
        for (int r = 0; r < K; ++r) pd[r][q] = ld_f64(lbase[r], o);
        pi[q] = ld_i32(piv, o >> 1);
        pd[K + 1][q] = ld_f64(rhs, min(offq[q] + (uint32_t)(1 + K) * nb8, last8));
        offq[q] += (uint32_t)CHK * nb8;
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
    for (int d = 0; d < DF; ++d) issue(pf[d], pp[d]);
    // every trip runs all DF stages and every stage issues its loads (clamped rows past the end): with a fixed number of younger loads behind each
    // landing the compiler's s_waitcnt leaves the prefetch in flight; a conditional issue would make it wait for everything
    for (int64_t c0 = 0; c0 < nch; c0 += DF) {
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const int64_t c = c0 + d;
        {
          bool moved = false;
#pragma unroll
          for (int q = 0; q < Q; ++q) {
#pragma unroll
            for (int o = 0; o < FO; ++o) if (o != K) sOp[o][q * G + g][s] = pf[d][o][q];
            const int off = pp[d][q] - min((int)c * CHK + q * G + g, (int)n - 1);  // pivot row - step
            sOp[K][q * G + g][s] = (double)off;
            moved = moved | (off != 0);
          }
          const bool stripped = (c + 1) * CHK + K < n && __builtin_amdgcn_ballot_w64(moved) == 0ull;  // wavefront-uniform
          issue(pf[d], pp[d]);
          __builtin_amdgcn_wave_barrier();
          if (chain) {
            double l[CHK][K], nxt[CHK];
#pragma unroll
            for (int t = 0; t < CHK; ++t) {
#pragma unroll
              for (int r = 0; r < K; ++r) l[t][r] = sOp[r][t][s];
              nxt[t] = sOp[K + 1][t][s];
            }
            if (stripped) {
#pragma unroll
              for (int t = 0; t < CHK; ++t) {
                const double x = v[0];
                sOut[t][s] = x;
#pragma unroll
                for (int r = 1; r < R; ++r) v[r] = (-x) * l[t][r - 1] + v[r];
#pragma unroll
                for (int r = 0; r + 1 < R; ++r) v[r] = v[r + 1];
                v[R - 1] = nxt[t];
              }
            } else {
              const int jb = (int)c * CHK, ni = (int)n;
#pragma unroll
              for (int t = 0; t < CHK; ++t) {
                const int j = jb + t;
                if (j < ni) {
                  const int pv = (int)sOp[K][t][s];
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
            }
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const int j = (int)c * CHK + q * G + g;
            if (valid && j < (int)n) st_f64(rhs, (uint32_t)j * nb8 + b8, sOut[q * G + g][s]);
          }
        }
      }
    }
  }
  // the backward sweep reads, from other lanes of this wavefront, what the forward sweep stored: a WORKGROUP-scope fence (stores acknowledged, same L1).
  // __threadfence() is an agent-scope fence — on this multi-XCD part an L2 write-back and invalidate per wavefront, 20 us of the kernel at 512 x 4096
  // (scripts/ubench/band_wide_bench.hip timeline, profiles/r03_band_solve.md)
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  // ---------------------------------------------------------------- backward with U (bandwidth 2K): chunk c covers rows n-1 - (c*CHK + t)
  {
    double pb[DB][BO][Q];
    // U(i-d, i) lives at fac[(d*n + i-d)*nb + b] = (fac + (d*n - d)*nb)[i*nb + b]: one offset per row serves every diagonal; a row above the matrix
    // (i - d < 0, its value is never used) is clamped to row 0 of its diagonal by a single max
    const double* ubase[C];
    int ulo[C];
#pragma unroll
    for (int d = 0; d < C; ++d) { ubase[d] = fac + ((int64_t)d * n - d) * nb; ulo[d] = (int)((uint32_t)d * nb8 + b8); }
    int offi[Q];  // byte offset of row i = n-1 - (c*CHK + q*G + g), for the chunk issued next; negative past the top of the matrix
#pragma unroll
    for (int q = 0; q < Q; ++q) offi[q] = ((int)n - 1 - (q * G + g)) * (int)nb8 + (int)b8;
    auto issue = [&](double (&pd)[BO][Q]) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int d = 0; d < C; ++d) pd[d][q] = ld_f64(ubase[d], (uint32_t)max(offi[q], ulo[d]));
        pd[C][q] = ld_f64(rhs, (uint32_t)max(offi[q] - C * (int)nb8, (int)b8));
        offi[q] -= CHK * (int)nb8;
      }
    };
    double w[C];
#pragma unroll
    for (int q = 0; q < C; ++q) { const int64_t r = n - 1 - (C - 1) + q; const double t = rhs[max(r, (int64_t)0) * nb + b]; w[q] = r >= 0 ? t : 0.0; }
#pragma unroll
    for (int q = 0; q < C; ++q) asm volatile("" : "+v"(w[q]));
#pragma unroll
    for (int d = 0; d < DB; ++d) issue(pb[d]);
    for (int64_t c0 = 0; c0 < nch; c0 += DB) {
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const int64_t c = c0 + d;
        {
          bool den_ok = true;
#pragma unroll
          for (int o = 0; o < BO; ++o)
#pragma unroll
            for (int q = 0; q < Q; ++q) sOp[o][q * G + g][s] = pb[d][o][q];
#pragma unroll
          for (int q = 0; q < Q; ++q) { sInv[q * G + g][s] = div_refined_rcp(pb[d][0][q]); den_ok = den_ok & div_den_ok(pb[d][0][q]); }
          const bool stripped = n - 1 - (c * CHK + CHK - 1) - C >= 0;  // every row of the chunk has its whole band and a successor entering the window
          issue(pb[d]);
          __builtin_amdgcn_wave_barrier();
          double w0[C];
          // the general step; also the second run of a stripped chunk whose quotients could not be vouched for
          auto general = [&]() __attribute__((always_inline)) {
            const int ib = (int)n - 1 - (int)c * CHK;
#pragma unroll
            for (int t = 0; t < CHK; ++t) {
              const int i = ib - t;
              if (i >= 0) {
                const double diag = sOp[0][t][s];
                if (diag == 0.0) bad = 1ull;
                const double x = w[C - 1] / diag;
                sOut[t][s] = x;
#pragma unroll
                for (int d2 = 1; d2 < C; ++d2) { const double uu = i - d2 >= 0 ? sOp[d2][t][s] : 0.0; w[C - 1 - d2] = (-x) * uu + w[C - 1 - d2]; }  // as k_lu_band_solve: entries above row 0 are zeros
#pragma unroll
                for (int q = C - 1; q > 0; --q) w[q] = w[q - 1];
                w[0] = i - C >= 0 ? sOp[C][t][s] : 0.0;
              }
            }
          };
          if (chain) {
            if (stripped) {
              double u[CHK][C], nxt[CHK], iv[CHK];
#pragma unroll
              for (int t = 0; t < CHK; ++t) {
#pragma unroll
                for (int d2 = 0; d2 < C; ++d2) u[t][d2] = sOp[d2][t][s];
                nxt[t] = sOp[C][t][s];
                iv[t] = sInv[t][s];
              }
#pragma unroll
              for (int q = 0; q < C; ++q) w0[q] = w[q];
#pragma unroll
              for (int t = 0; t < CHK; ++t) {
                const double x = div_by_refined(w[C - 1], u[t][0], iv[t]);
                sOut[t][s] = x;
#pragma unroll
                for (int d2 = 1; d2 < C; ++d2) w[C - 1 - d2] = (-x) * u[t][d2] + w[C - 1 - d2];
#pragma unroll
                for (int q = C - 1; q > 0; --q) w[q] = w[q - 1];
                w[0] = nxt[t];
              }
            } else {
              general();
            }
          }
          __builtin_amdgcn_wave_barrier();
          double xq[Q];
#pragma unroll
          for (int q = 0; q < Q; ++q) xq[q] = sOut[q * G + g][s];
          if (stripped) {
            // were the quotients quotients?  Checked here by the lanes that store them, off the chain: diagonal and quotient in the ranges that imply
            // a numerator in range.  Otherwise (zero / tiny / huge entries: rare) the system's chain lane runs the chunk again from its saved window
            bool ok = den_ok;
#pragma unroll
            for (int q = 0; q < Q; ++q) ok = ok & div_quot_ok(xq[q]);
            unsigned long long m = __builtin_amdgcn_ballot_w64(!ok);
            if (m != 0ull) {
              m |= m >> 32; m |= m >> 16; m |= m >> 8;  // bit s: some row of system s failed (lane = g*S + s, S = 8)
              static_assert(S == 8, "the fold above assumes 8 systems per wavefront");
              if (chain && ((m >> s) & 1ull)) {
#pragma unroll
                for (int q = 0; q < C; ++q) w[q] = w0[q];
                general();
              }
              __builtin_amdgcn_wave_barrier();
#pragma unroll
              for (int q = 0; q < Q; ++q) xq[q] = sOut[q * G + g][s];
            }
          }
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const int i = (int)n - 1 - ((int)c * CHK + q * G + g);
            if (valid && i >= 0) st_f64(rhs, (uint32_t)i * nb8 + b8, xq[q]);
          }
        }
      }
    }
  }
  block_publish(0ull, 0ull, (chain && valid) ? bad : 0ull, rec, seq);
}

}  // namespace dsh
