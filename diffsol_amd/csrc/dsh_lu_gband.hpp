// General-bandwidth batched banded LU (gfx950): ONE WAVEFRONT PER SYSTEM, arbitrary (kl, ku) up to 64 each — the `Sunmatrix_Band` analogue the reference's
// benchmark notes miss (book/src/benchmarks/sundials.md:27-28): the Jacobians of 2-D PDE discretisations (test_models/heat2d.rs: n = m^2, half-bandwidth m;
// foodweb.rs: n = 2 nx^2, half-bandwidth 2 nx) are dense CONTAINERS of banded matrices that the register kernels of dsh_lu_band.hpp (K <= 4, one lane per
// system, a (K+1) x (2K+1) window in registers) cannot hold.
//
// What it computes: LAPACK dgbtrf-shaped partial pivoting — at step j the pivot is searched in rows j .. j+kl, the pivot row (which reaches kl+ku columns past
// the diagonal after interchanges: the kl fill-in diagonals) eliminates rows j+1 .. j+kl — with the arithmetic of the dense kernels and of the oracle's dense LU
// (dsh_lu_dev.hpp / oracle_la.hpp DenseLU: first-max pivot, l = a * (1 / pivot), a = (-u) * l + a, column-oriented substitutions, x = v / u_ii) in the same order
// per element.  Every operation the dense elimination performs outside the widened band has an exact-zero operand, so for finite matrices factors and solutions
// are BIT-IDENTICAL to the dense path's (tests/test_gpu_lu_gband.py).  A zero pivot leaves its column as it is and is reported like the dense kernels do.
//
// Factorisation: the active window — rows j .. j+kl, columns j .. j+kl+ku of the partially eliminated matrix — lives in LDS as a ring ((kl+1) x (kl+ku+1)
// doubles per system: 1.8 KB at k = 10, 6.9 KB at k = 20, 67 KB at k = 64); lanes own COLUMNS of the window (up to three each), so the pivot row sits in
// registers during the rank-1 update and every LDS access of the update is conflict-free (consecutive lanes, consecutive words).  The operand is first STAGED
// (k_gband_stage: a tiled transpose, coalesced on both sides) from its batch-fastest container into a system-major, row-major copy of the band — read in place,
// the neighbours of a row lie n nb 8 bytes apart (3.3 MB at n = 100 x 4096 members): one DRAM page and one TLB entry per lane and step, 10 us per elimination
// step (measured, profiles/r06_band_general.md).  Row j+kl+1 is fetched from the staged copy one step ahead.  One wavefront only ever talks to itself through
// LDS: the synchronisation is a wave-level fence, no s_barrier.
// Solves: the right-hand side stays in REGISTERS, element i in lane i % 64 (register i / 64); the pivot element of a step is broadcast with v_readlane; the
// multipliers / U columns of a run of steps are one contiguous piece of the factor storage, fetched with full-width loads while the previous run's chain
// executes and handed to the lanes through LDS.
//
// Factor layout (system-major, S = (2 kl + ku + 1) n doubles per system):  L: multiplier of row j+r at step j at  j * kl + (r - 1), r = 1 .. kl;
// U BY COLUMNS: U(i - d, i) at  n * kl + i * (kl + ku + 1) + d, d = 0 .. kl+ku (d = 0: the diagonal) — what the backward substitution reads per step is one
// contiguous run.  Pivots system-major: piv[b * n + j] = row interchanged with row j at step j.
#pragma once
#include "dsh_device.hpp"

namespace dsh {

constexpr int kGbMaxK = 64;      // kl, ku <= 64: the window's columns are at most 3 per lane
constexpr int kGbSolveWaves = 4; // systems per workgroup of the solve (one per wavefront): nb / 4 reduction records

__host__ __device__ inline int64_t gband_factor_doubles(int64_t n, int kl, int ku) { return n * (int64_t)(2 * kl + ku + 1); }
__host__ __device__ inline size_t gband_window_bytes(int kl, int ku) { return sizeof(double) * (size_t)(kl + 1) * (size_t)(kl + ku + 1); }
// wavefronts (= systems) per workgroup of the factorisation: as many as 60 KB of LDS hold, at most 4
inline int gband_factor_waves(int kl, int ku) {
  const size_t w = gband_window_bytes(kl, ku);
  int k = (int)((size_t)60 * 1024 / (w ? w : 1));
  return k < 1 ? 1 : (k > 4 ? 4 : k);
}

__device__ __forceinline__ void gband_wave_sync() {  // LDS traffic of ONE wavefront: in order in the hardware; this keeps the compiler from reordering across it
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double gband_readlane(double v, int lane) {
  return __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(v), lane));
}
__device__ __forceinline__ double gband_writelane(double old, double val, int lane) {  // old with lane `lane` replaced by the (wavefront-uniform) val
  return (int)(threadIdx.x & 63) == lane ? val : old;
}

// The band of every system from its batch-fastest container — dense ((c n + i) nb + b) or (PACKED) a band container of bandwidths (pkl, pku) (entry (i, c) at
// ((c - i + pkl) n + i) nb + b) — into w[b][i][cc], cc = c - i + kl in [0, kl + ku + 1): zeros where the column falls outside the matrix (or the container's band).
// 32 systems x 32 entries per workgroup through LDS: 256-byte runs on the read side, 256-byte runs on the write side.  Grid: (ceil(nb / 32), ceil(n Wb / 32)).
template <bool PACKED>
__global__ __launch_bounds__(256) void k_gband_stage(int n, int64_t nb, int kl, int ku, const double* __restrict__ a, int pkl, int pku, double* __restrict__ w) {
  __shared__ double t[32][33];
  const int Wb = kl + ku + 1;
  const int64_t E = (int64_t)n * Wb, b0 = (int64_t)blockIdx.x * 32, e0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int64_t e = e0 + r, b = b0 + tx;
    double val = 0.0;
    if (e < E && b < nb) {
      const int i = (int)(e / Wb), cc = (int)(e % Wb), d = cc - kl, c = i + d;
      if (c >= 0 && c < n) {
        if constexpr (PACKED) { if (d <= pku && -d <= pkl) val = a[((int64_t)(d + pkl) * n + i) * nb + b]; }
        else val = a[((int64_t)c * n + i) * nb + b];
      }
    }
    t[r][tx] = val;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t b = b0 + r, e = e0 + tx;
    if (b < nb && e < E) w[b * E + e] = t[tx][r];
  }
}

// CPL = columns of the window per lane (ceil((kl + ku + 1) / 64)); w = the staged band (k_gband_stage).  Launch: ceil(nb / waves) workgroups of 64 * waves
// threads, waves * window bytes of LDS.
template <int CPL>
__global__ __launch_bounds__(256) void k_lu_gband_factor(int n, int64_t nb, int kl, int ku, const double* __restrict__ w, double* __restrict__ fac, int32_t* __restrict__ piv,
                                                         unsigned long long* singular_count, unsigned int epoch) {
  extern __shared__ double gb_lds[];
  const int Wc = kl + ku + 1, R = kl + 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, waves = blockDim.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * waves + wave;
  const bool live = b < nb;
  unsigned long long sing = 0ull;
  if (live) {
    double* win = gb_lds + (size_t)wave * R * Wc;
    double* Lc = fac + (size_t)b * gband_factor_doubles(n, kl, ku);
    double* Uc = Lc + (size_t)n * kl;
    int32_t* pv = piv + (size_t)b * n;
    const double* wb = w + (size_t)b * n * Wc;  // row i at wb + i * Wc: columns i - kl .. i + ku
    // rows 0 .. kl enter the window: columns 0 .. Wc - 1 (entry (i, c) at staged index c - i + kl)
    for (int i = 0; i < R; ++i) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const int cc = lane + 64 * q;
        if (cc < Wc) { const int sidx = cc - i + kl; win[i * Wc + cc] = (i < n && sidx >= 0 && sidx < Wc) ? wb[(size_t)i * Wc + sidx] : 0.0; }
      }
    }
    gband_wave_sync();
    int srow_j = 0, scol_j = 0;  // j % R, j % Wc
    for (int j = 0; j < n; ++j) {
      // the row that enters after this step: row j + R, columns j + 1 .. j + Wc (fetched now, stored at the end of the step)
      double nxt[CPL];
#pragma unroll
      for (int q = 0; q < CPL; ++q) { const int cc = lane + 64 * q; nxt[q] = (j + R < n && cc < Wc) ? wb[(size_t)(j + R) * Wc + cc] : 0.0; }  // its columns j + 1 .. j + Wc ARE its band
      // ---- pivot search over rows j .. j + rows_here (first maximum of |a_rj|, like the sequential scan)
      const int rows_here = min(kl, n - 1 - j);
      double vs = 0.0, vs64 = 0.0, best = -1.0;
      int p = 0x7fffffff;
      if (lane <= rows_here) {  // (rows_here = 64 has one candidate more than the wavefront has lanes: lane 0 takes it below)
        int sr = srow_j + lane; if (sr >= R) sr -= R;
        vs = win[sr * Wc + scol_j];
        best = fabs(vs);
        p = lane;
        if (!(best >= 0.0)) { best = lane == 0 ? 1.0e308 * 10.0 : -1.0; }  // NaN: on the diagonal it stays the pivot (the scan starts from it), elsewhere it never wins
      }
      if (rows_here == 64 && lane == 0) {
        int sr = srow_j + 64; if (sr >= R) sr -= R;
        vs64 = win[sr * Wc + scol_j];
        if (fabs(vs64) > best) { best = fabs(vs64); p = 64; }
      }
      group_argmax(best, p, 64);
      p = __builtin_amdgcn_readfirstlane(p);
      if (p > rows_here) p = 0;
      const double diag = p == 64 ? gband_readlane(vs64, 0) : gband_readlane(vs, p);
      const bool elim = diag != 0.0;
      if (!elim) { p = 0; sing = 1ull; }
      if (lane == 0) pv[j] = j + p;
      // ---- the pivot row into registers; the old row j takes its place at position j + p
      int srow_p = srow_j + p; if (srow_p >= R) srow_p -= R;
      double u[CPL];
#pragma unroll
      for (int q = 0; q < CPL; ++q) {
        const int cc = lane + 64 * q;
        u[q] = 0.0;
        if (cc < Wc) {
          int sc = scol_j + cc; if (sc >= Wc) sc -= Wc;
          u[q] = win[srow_p * Wc + sc];
          if (p != 0) win[srow_p * Wc + sc] = win[srow_j * Wc + sc];
          if (j + cc < n) Uc[(size_t)(j + cc) * Wc + cc] = u[q];  // U(j, j + cc), stored by columns
        }
      }
      gband_wave_sync();
      // ---- multipliers (lane r - 1 = row j + r) and the rank-1 update (lanes = columns)
      double l = 0.0;
      if (lane < rows_here) {
        int sr = srow_j + lane + 1; if (sr >= R) sr -= R;
        const double e = win[sr * Wc + scol_j];
        l = elim ? e * (1.0 / diag) : e;  // a zero pivot leaves the column as it is (the dense kernels do the same)
        win[sr * Wc + scol_j] = 0.0;      // this slot is column j + Wc of the row from the next step on: outside its band
        Lc[(size_t)j * kl + lane] = l;
      } else if (lane < kl) {
        Lc[(size_t)j * kl + lane] = 0.0;  // rows beyond the matrix
      }
      if (elim) {
        for (int r = 1; r <= rows_here; ++r) {
          const double lr = gband_readlane(l, r - 1);
          int sr = srow_j + r; if (sr >= R) sr -= R;
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            const int cc = lane + 64 * q;
            if (cc >= 1 && cc < Wc) {
              int sc = scol_j + cc; if (sc >= Wc) sc -= Wc;
              const double cur = win[sr * Wc + sc];
              win[sr * Wc + sc] = (-u[q]) * lr + cur;
            }
          }
        }
      }
      // ---- row j + R into the slot row j leaves (columns j + 1 .. j + Wc: every slot of the ring row)
      if (j + R < n) {
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          const int cc = lane + 64 * q;
          if (cc < Wc) {
            int sc = scol_j + 1 + cc; if (sc >= Wc) sc -= Wc;
            win[srow_j * Wc + sc] = nxt[q];
          }
        }
      }
      gband_wave_sync();
      if (++srow_j == R) srow_j = 0;
      if (++scol_j == Wc) scol_j = 0;
    }
  }
  sing = wave_sum_u64(live && lane == 0 ? sing : 0ull);
  if (lane == 0 && sing) publish_singular(singular_count, sing, epoch);
}

// x <- A^-1 x for every system; MB = registers of the right-hand side per lane (n <= 64 MB); W3: kl + ku > 64 (a column of U reaches three registers of the
// right-hand side).  Launch: ceil(nb / kGbSolveWaves) workgroups of 64 * kGbSolveWaves threads, gband_solve_lds_bytes of LDS.  The record of a workgroup carries the
// number of its systems that met a zero diagonal (LuSolveFailed).
// Operand traffic: the multipliers of SB consecutive steps (and the U columns of SB consecutive steps) are ONE contiguous run of the factor storage.  A wavefront
// fetches the next run with full-width loads (every lane 8 bytes, <= GQ instructions) while the chain of the current run executes, parks it in LDS, and the lanes
// pick their operands of four steps at a time from there.  (First version: every lane fetched its own operand of every step from global memory — 10 - 20 live
// lanes per load instruction, 600 load instructions per system at n = 100: bound by the issue rate of the vector memory pipeline at 1.0 - 1.5 TB/s.)
template <bool W3> struct gband_solve_cfg { static constexpr int GQ = W3 ? 9 : 5; };  // registers per lane of one staged run: 576 / 320 doubles
__host__ __device__ inline int gband_solve_sb(int width, int cap_doubles) { int sb = (cap_doubles / (width > 0 ? width : 1)) / 4 * 4; return sb < 4 ? 4 : (sb > 32 ? 32 : sb); }
inline size_t gband_solve_lds_bytes(bool w3) { return sizeof(double) * 64 * (size_t)(w3 ? 9 : 5) * kGbSolveWaves; }
template <int MB, bool W3>
__global__ __launch_bounds__(64 * kGbSolveWaves) void k_lu_gband_solve(int n, int64_t nb, int kl, int ku, const double* __restrict__ fac, const int32_t* __restrict__ piv,
                                                                      double* __restrict__ rhs, unsigned long long* rec, unsigned int seq) {
  constexpr int CH = 4;                         // steps whose operands are read from LDS together
  constexpr int GQ = gband_solve_cfg<W3>::GQ;   // a staged run is at most 64 GQ doubles
  extern __shared__ double gb_stage_all[];
  const int Wc = kl + ku + 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * kGbSolveWaves + wave;
  double* stage = gb_stage_all + (size_t)wave * 64 * GQ;
  unsigned long long bad = 0ull;
  if (b < nb) {
    const double* Lc = fac + (size_t)b * gband_factor_doubles(n, kl, ku);
    const double* Uc = Lc + (size_t)n * kl;
    const int32_t* pvp = piv + (size_t)b * n;
    double v[MB];
    int pv[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const int row = lane + 64 * m;
      v[m] = row < n ? rhs[(int64_t)row * nb + b] : 0.0;
      pv[m] = row < n ? pvp[row] : row;
    }
    double G[GQ];
    auto gload = [&](const double* src, int len) {  // the next run into registers: lane, lane + 64, ... (len <= 64 GQ; len <= 0: nothing)
#pragma unroll
      for (int q = 0; q < GQ; ++q) { const int e = lane + 64 * q; G[q] = e < len ? src[e] : 0.0; }
    };
    auto gpark = [&](int len) {  // ... and from the registers into this wavefront's LDS
#pragma unroll
      for (int q = 0; q < GQ; ++q) { const int e = lane + 64 * q; if (e < len) stage[e] = G[q]; }
    };
    // ---- forward: interchange j <-> piv[j], then rows j+1 .. j+kl take (-x_j) l + v
    if (kl > 0) {
      const int SB = gband_solve_sb(kl, 64 * (W3 ? 8 : 4));  // steps per staged run of multipliers (kl <= 64: at least 4)
      gload(Lc, min(SB, n) * kl);
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        const int jbase = 64 * m;
        if (jbase < n) {
          const int jend = min(64, n - jbase);
          for (int c0 = 0; c0 < jend; c0 += SB) {  // run: steps jbase + c0 .. jbase + c1 - 1
            const int c1 = min(c0 + SB, jend);
            gband_wave_sync();  // the previous run's readers are done
            gpark((c1 - c0) * kl);
            gband_wave_sync();
            {  // the run after this one (it may belong to the next block of 64 steps)
              const int j_next = jbase + c1;
              const int cnt = c1 < jend ? min(SB, jend - c1) : (j_next < n ? min(SB, min(64, n - j_next)) : 0);
              gload(Lc + (size_t)j_next * kl, cnt * kl);
            }
            for (int jj0 = c0; jj0 < c1; jj0 += CH) {
              double la[CH], lb[CH];
#pragma unroll
              for (int s = 0; s < CH; ++s) {
                const int jj = jj0 + s, j = jbase + jj;
                const int ra = lane - jj, rb = lane + 64 - jj;  // row - j for this lane's element of register m / m + 1
                const bool oka = jj < c1 && ra >= 1 && ra <= kl && j + ra < n, okb = jj < c1 && rb <= kl && j + rb < n;
                la[s] = oka ? stage[(jj - c0) * kl + ra - 1] : 0.0;
                lb[s] = okb ? stage[(jj - c0) * kl + rb - 1] : 0.0;
              }
#pragma unroll
              for (int s = 0; s < CH; ++s) {
                const int jj = jj0 + s, j = jbase + jj;
                if (jj < c1) {
                  const int p = __builtin_amdgcn_readlane(pv[m], jj);
                  if (p != j) {
                    const double vj = gband_readlane(v[m], jj);
                    if ((p >> 6) == m) {
                      const double vp = gband_readlane(v[m], p & 63);
                      v[m] = gband_writelane(v[m], vp, jj);
                      v[m] = gband_writelane(v[m], vj, p & 63);
                    } else if (m + 1 < MB) {
                      const double vp = gband_readlane(v[m + 1 < MB ? m + 1 : m], p & 63);
                      v[m] = gband_writelane(v[m], vp, jj);
                      v[m + 1 < MB ? m + 1 : m] = gband_writelane(v[m + 1 < MB ? m + 1 : m], vj, p & 63);
                    }
                  }
                  const double x = gband_readlane(v[m], jj);
                  const int ra = lane - jj, rb = lane + 64 - jj;
                  if (ra >= 1 && ra <= kl && j + ra < n) v[m] = (-x) * la[s] + v[m];
                  if (m + 1 < MB) { if (rb <= kl && j + rb < n) v[m + 1 < MB ? m + 1 : m] = (-x) * lb[s] + v[m + 1 < MB ? m + 1 : m]; }
                }
              }
            }
          }
        }
      }
    }  // (kl = 0: no multipliers and no interchanges — piv[j] = j)
    // ---- backward, column oriented like the dense solve: x_i = v_i / u_ii, then rows i-1 .. i-(kl+ku) take (-x_i) u_ri + v
    {
      const int SB = gband_solve_sb(Wc, 64 * GQ);  // steps per staged run of U columns
      bool stop = false;
      {  // first run: the top steps of the last block of 64
        const int mtop = (n - 1) >> 6, jtop = (n - 1) - 64 * mtop, c_lo = max(jtop - SB + 1, 0);
        gload(Uc + (size_t)(64 * mtop + c_lo) * Wc, (jtop - c_lo + 1) * Wc);
      }
#pragma unroll
      for (int m = MB - 1; m >= 0; --m) {
        const int jbase = 64 * m;
        if (jbase < n && !stop) {
          const int jtop = min(63, n - 1 - jbase);
          for (int c_hi = jtop; c_hi >= 0 && !stop; c_hi -= SB) {  // run: steps jbase + c_hi down to jbase + c_lo
            const int c_lo = max(c_hi - SB + 1, 0);
            gband_wave_sync();
            gpark((c_hi - c_lo + 1) * Wc);
            gband_wave_sync();
            {  // the run after this one
              int nhi, nlo, nbase;
              if (c_lo > 0) { nbase = jbase; nhi = c_lo - 1; nlo = max(nhi - SB + 1, 0); }
              else { nbase = jbase - 64; nhi = 63; nlo = max(nhi - SB + 1, 0); }
              gload(Uc + (size_t)(nbase >= 0 ? nbase + nlo : 0) * Wc, nbase >= 0 ? (nhi - nlo + 1) * Wc : 0);
            }
            for (int jj0 = c_hi; jj0 >= c_lo && !stop; jj0 -= CH) {
              double ua[CH], ub[CH], uc[W3 ? CH : 1], dg[CH];
#pragma unroll
              for (int s = 0; s < CH; ++s) {
                const int jj = jj0 - s;
                const int da = jj - lane, db = jj + 64 - lane, dc = jj + 128 - lane;  // i - row for this lane's element of register m / m-1 / m-2
                const bool in = jj >= c_lo;
                const double* col = stage + (in ? (jj - c_lo) * Wc : 0);
                ua[s] = (in && da >= 1 && da < Wc) ? col[da] : 0.0;
                ub[s] = (in && m >= 1 && db < Wc) ? col[db] : 0.0;
                if constexpr (W3) uc[s] = (in && m >= 2 && dc < Wc) ? col[dc] : 0.0;
                dg[s] = in ? col[0] : 1.0;
              }
#pragma unroll
              for (int s = 0; s < CH; ++s) {
                const int jj = jj0 - s;
                if (jj >= c_lo && !stop) {
                  const double diag = dg[s];
                  if (diag == 0.0) { bad = 1ull; stop = true; }  // the dense solve breaks here as well (LuSolveFailed)
                  else {
                    const double x = gband_readlane(v[m], jj) / diag;
                    v[m] = gband_writelane(v[m], x, jj);
                    const int da = jj - lane, db = jj + 64 - lane, dc = jj + 128 - lane;
                    if (da >= 1 && da < Wc) v[m] = (-x) * ua[s] + v[m];
                    if (m >= 1) { if (db < Wc) v[m >= 1 ? m - 1 : 0] = (-x) * ub[s] + v[m >= 1 ? m - 1 : 0]; }
                    if constexpr (W3) { if (m >= 2) { if (dc < Wc) v[m >= 2 ? m - 2 : 0] = (-x) * uc[s] + v[m >= 2 ? m - 2 : 0]; } }
                  }
                }
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const int row = lane + 64 * m;
      if (row < n) rhs[(int64_t)row * nb + b] = v[m];
    }
  }
  block_publish(0ull, 0ull, lane == 0 ? bad : 0ull, rec, seq);
}

}  // namespace dsh
