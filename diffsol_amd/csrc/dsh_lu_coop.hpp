// Workgroup-per-system dense LU for n > 64 (gfx950).
//
// For n <= 8 a whole system lives in one lane's registers (dsh_lu_dev.hpp), for n <= 64 in the registers of one wavefront (dsh_lu_wave.hpp).
// Beyond that a system gets a whole 256-thread workgroup.  The LU object owns its factor storage and keeps it SYSTEM-MAJOR (a contiguous
// n*n column-major block per system: coalesced for a workgroup working on one system); the public operands (A, rhs) stay batch-fastest and
// are transposed on the way in (k_soa_to_aos).
//
//   * k_lu_factor_blocked<NB>: right-looking blocked factorisation, panel in LDS, register-tiled trailing update (see below).
//   * k_lu_factor_global_coop: unblocked, factors updated in place in HBM/L2 — fallback for n beyond the LDS panel budget (n > ~2000).
//   * k_lu_solve_global_coop: rhs in LDS, factors streamed once from HBM.
//
// Arithmetic per element is identical to lu_factor_reg / the oracle (l = a*(1/pivot); a_rc = (-a_kc)*l_rk + a_rc; first-max pivot), each
// element being updated by exactly one thread per step, so results stay bit-identical to the CPU path regardless of the thread mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dsh {

constexpr int kCoopThreads = 256;

// batch-fastest -> system-major copy of the matrices (one thread per element, tiled through LDS by the generic transpose in dsh_ctx.hip is
// not needed: reads are coalesced along b, writes along e for a 32x32 tile)
__global__ void k_soa_to_aos(int64_t nelem, int64_t nb, const double* __restrict__ soa, double* __restrict__ aos) {
  __shared__ double tile[32][33];
  const int64_t e0 = (int64_t)blockIdx.y * 32, b0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) { int64_t e = e0 + k, b = b0 + tx; if (e < nelem && b < nb) tile[k][tx] = soa[e * nb + b]; }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) { int64_t b = b0 + k, e = e0 + tx; if (e < nelem && b < nb) aos[b * nelem + e] = tile[tx][k]; }
}

// One workgroup per system, factors in place in global memory (system-major).  Any n.
__global__ void k_lu_factor_global_coop(int n, int64_t nb, double* __restrict__ f_aos, int32_t* __restrict__ piv_aos, unsigned long long* singular_word,
                                        unsigned int epoch) {
  __shared__ double s_best[kCoopThreads / 64];
  __shared__ int s_row[kCoopThreads / 64];
  __shared__ int s_piv;
  __shared__ double s_diag;
  const int64_t b = blockIdx.x;
  double* A = f_aos + (size_t)b * n * n;  // column-major, ld = n
  const int tid = threadIdx.x;
  bool singular = false;
  for (int k = 0; k < n; ++k) {
    double best = -1.0;
    int prow = n;
    for (int r = k + tid; r < n; r += kCoopThreads) { double v = fabs(A[(size_t)k * n + r]); if (v > best) { best = v; prow = r; } }
    group_argmax(best, prow, 64);
    if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_row[tid >> 6] = prow; }
    __syncthreads();
    if (tid == 0) {
      double bb = s_best[0]; int rr = s_row[0];
      for (int w = 1; w < kCoopThreads / 64; ++w) if (s_best[w] > bb || (s_best[w] == bb && s_row[w] < rr)) { bb = s_best[w]; rr = s_row[w]; }
      if (rr >= n) rr = k;
      s_piv = rr;
      s_diag = A[(size_t)k * n + rr];
    }
    __syncthreads();
    const int p = s_piv;
    const double diag = s_diag;
    const bool zero = diag == 0.0;
    if (zero) singular = true;
    if (tid == 0) piv_aos[(size_t)b * n + k] = zero ? k : p;
    if (!zero && p != k)
      for (int c = tid; c < n; c += kCoopThreads) { double tmp = A[(size_t)c * n + k]; A[(size_t)c * n + k] = A[(size_t)c * n + p]; A[(size_t)c * n + p] = tmp; }
    __syncthreads();
    if (!zero) {
      const double inv_diag = 1.0 / diag;
      for (int r = k + 1 + tid; r < n; r += kCoopThreads) A[(size_t)k * n + r] = A[(size_t)k * n + r] * inv_diag;
    }
    __syncthreads();
    if (!zero) {
      const int m = n - k - 1;
      for (int64_t idx = tid; idx < (int64_t)m * m; idx += kCoopThreads) {
        const int r = k + 1 + (int)(idx % m), c = k + 1 + (int)(idx / m);
        A[(size_t)c * n + r] = (-A[(size_t)c * n + k]) * A[(size_t)k * n + r] + A[(size_t)c * n + r];
      }
    }
    __syncthreads();
  }
  if (singular && tid == 0) publish_singular(singular_word, 1ull, epoch);
}

// One workgroup per system, rhs staged in LDS (8n bytes), factors streamed from global memory.
__global__ void k_lu_solve_global_coop(int n, int64_t nb, const double* __restrict__ f_aos, const int32_t* __restrict__ piv_aos, double* __restrict__ rhs,
                                       unsigned long long* rec, unsigned int seq) {
  extern __shared__ double v[];
  const int64_t b = blockIdx.x;
  const double* A = f_aos + (size_t)b * n * n;
  const int tid = threadIdx.x;
  for (int r = tid; r < n; r += kCoopThreads) v[r] = rhs[(int64_t)r * nb + b];
  __syncthreads();
  if (tid == 0) {
    const int32_t* P = piv_aos + (size_t)b * n;
    for (int i = 0; i < n; ++i) { int p = P[i]; if (p != i) { double tmp = v[i]; v[i] = v[p]; v[p] = tmp; } }
  }
  __syncthreads();
  for (int i = 0; i + 1 < n; ++i) {
    const double coeff = v[i];
    __syncthreads();
    for (int r = i + 1 + tid; r < n; r += kCoopThreads) v[r] = (-coeff) * A[(size_t)i * n + r] + v[r];
    __syncthreads();
  }
  bool ok = true;
  for (int i = n - 1; i >= 0; --i) {
    const double diag = A[(size_t)i * n + i];
    if (diag == 0.0) ok = false;
    const double coeff = v[i] / diag;
    __syncthreads();
    if (tid == 0) v[i] = coeff;
    for (int r = tid; r < i; r += kCoopThreads) v[r] = (-coeff) * A[(size_t)i * n + r] + v[r];
    __syncthreads();
  }
  for (int r = tid; r < n; r += kCoopThreads) rhs[(int64_t)r * nb + b] = v[r];
  block_publish(0ull, 0ull, (tid == 0 && !ok) ? 1ull : 0ull, rec, seq);
}

// Blocked right-looking LU, one workgroup per system, for systems too large for LDS (n > ~137).  Panels of NB columns are factored in LDS
// (partial pivoting over the whole remaining column, as the unblocked algorithm does); the panel's row swaps are then applied to the other
// columns, U12 = L11^-1 A12 is formed one column per thread, and the trailing matrix gets its NB rank-1 updates in one pass:
// thread = row (coalesced column-major access), its NB multipliers l_rk in registers, the U12 block broadcast from LDS.  THREADS = 256 or 512:
// 62 % of the wave cycles of the 256-thread version are s_waitcnt stalls (LDS broadcast / global latency with ONE wavefront per SIMD,
// profiles/r01_lu_bench.md), so larger systems run two wavefronts per SIMD.
// Every element still receives exactly the updates a_rc = (-u_kc) * l_rk + a_rc for k ascending, each as a separate multiply and add, so
// the factors are bit-identical to the unblocked kernels and the oracle; only the number of passes over the trailing matrix changes
// (n/NB instead of n): HBM/L2 traffic per system ~ 16 n^3 / (3 NB) bytes instead of 16 n^3 / 3.
// MFMA = true (opt-in, DSH_LU_MFMA=1, n a multiple of 16): the trailing update runs on the FP64 matrix cores, v_mfma_f64_16x16x4_f64, one 16 x 16 tile of
// A22 per wavefront and step, A22 -= L21 U12 as eight chained 16x16x4 products.  Operand layout found by experiment (scripts/ubench/mfma_f64_layout.hip):
// a(l) = A[l % 16][l / 16], b(l) = B[l / 16][l % 16], d(l, r) = D[4 r + l / 16][l % 16].  The product is formed TRANSPOSED — MFMA row index = column of
// A22, MFMA column index = row of A22 — so that every register of the accumulator holds 16 consecutive rows of one column (128-byte segments of the
// column-major factor storage) and the L21 operand of a row tile is loaded once for all its column tiles.  The matrix cores fuse each multiply-add and
// sum the four products of an instruction in their own order: results differ from the bit-exact path in the last bits (and a near-tie between pivot
// candidates may then resolve differently), which is why this is opt-in and tested to a tolerance (tests/test_gpu_lu_models.py), not bitwise.
template <int NB, int THREADS, bool MFMA = false>
__global__ __launch_bounds__(THREADS) void k_lu_factor_blocked(int n, int64_t nb, double* __restrict__ f_aos, int32_t* __restrict__ piv_aos,
                                                                   unsigned long long* singular_word, unsigned int epoch, unsigned long long* phase_clocks) {
  // optional phase profile (DSH_LU_PHASE_PROFILE=1): workgroup 0 accumulates the 100 MHz wall clock per phase
  const bool prof = phase_clocks != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long tprev = prof ? wall_clock64() : 0ull;
  auto mark = [&](int phase) {
    if (prof) { const unsigned long long now = wall_clock64(); phase_clocks[phase] += now - tprev; tprev = now; }
  };
  extern __shared__ double sh[];  // panel (ldp x NB) during the panel factorisation, then the U12 block (mc x LDU)
  __shared__ double s_best[THREADS / 64];
  __shared__ int s_row[THREADS / 64];
  __shared__ int s_piv[NB];
  __shared__ double s_l11[NB * (NB + 1)];
  constexpr int LDU = NB + 2;
  const int tid = threadIdx.x;
  double* A = f_aos + (size_t)blockIdx.x * n * n;  // column-major, ld = n
  int32_t* PIV = piv_aos + (size_t)blockIdx.x * n;
  const int ldp = n | 1;
  bool singular = false;
  for (int jb = 0; jb < n; jb += NB) {
    const int w = (n - jb) < NB ? (n - jb) : NB;
    const int m = n - jb;
    // ---- 1. panel -> LDS
    for (int c = 0; c < w; ++c)
      for (int r = tid; r < m; r += THREADS) sh[c * ldp + r] = A[(size_t)(jb + c) * n + jb + r];
    __syncthreads();
    mark(0);
    // ---- 2. unblocked factorisation of the m x w panel
    for (int k = 0; k < w; ++k) {
      double* col = sh + k * ldp;
      double best = -1.0;
      int prow = m;
      for (int r = k + tid; r < m; r += THREADS) { double v = fabs(col[r]); if (v > best) { best = v; prow = r; } }
      group_argmax(best, prow, 64);
      if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_row[tid >> 6] = prow; }
      __syncthreads();
      best = s_best[0]; prow = s_row[0];
#pragma unroll
      for (int j = 1; j < THREADS / 64; ++j) {
        const double ob = s_best[j];
        const int orow = s_row[j];
        if (ob > best || (ob == best && orow < prow)) { best = ob; prow = orow; }
      }
      if (prow >= m) prow = k;  // NaN column: keep the diagonal like the sequential scan
      const double diag = col[prow];
      const bool zero = diag == 0.0;
      if (zero) singular = true;
      if (tid == 0) s_piv[k] = jb + (zero ? k : prow);
      __syncthreads();
      if (!zero && prow != k && tid < w) { double tmp = sh[tid * ldp + k]; sh[tid * ldp + k] = sh[tid * ldp + prow]; sh[tid * ldp + prow] = tmp; }
      __syncthreads();
      if (!zero) {
        const double inv_diag = 1.0 / diag;
        for (int r = k + 1 + tid; r < m; r += THREADS) {
          const double l = col[r] * inv_diag;
          col[r] = l;
          for (int c = k + 1; c < w; ++c) sh[c * ldp + r] = (-sh[c * ldp + k]) * l + sh[c * ldp + r];
        }
      }
      __syncthreads();
    }
    mark(1);
    // ---- 3. panel and pivots back to global memory; L11 to its own LDS block
    for (int c = 0; c < w; ++c)
      for (int r = tid; r < m; r += THREADS) A[(size_t)(jb + c) * n + jb + r] = sh[c * ldp + r];
    if (tid < w) PIV[jb + tid] = s_piv[tid];
    for (int idx = tid; idx < NB * NB; idx += THREADS) {
      const int r = idx % NB, c = idx / NB;
      s_l11[c * (NB + 1) + r] = (r < w && c < w) ? sh[c * ldp + r] : 0.0;
    }
    __syncthreads();
    mark(2);
    // ---- 4. the panel's row interchanges on every other column (one thread per column)
    for (int c = tid; c < n; c += THREADS) {
      if (c >= jb && c < jb + w) continue;
      double* colg = A + (size_t)c * n;
      for (int k = 0; k < w; ++k) {
        const int p = s_piv[k];
        if (p != jb + k) { double tmp = colg[jb + k]; colg[jb + k] = colg[p]; colg[p] = tmp; }
      }
    }
    __syncthreads();
    mark(3);
    const int mc = n - jb - w;  // trailing columns
    if (mc <= 0) break;
    // ---- 5. U12 = L11^-1 A12, one column per thread; kept in LDS.  Trailing columns exist only behind a full panel: w == NB from here on.
    for (int cc = tid; cc < mc; cc += THREADS) {
      double* colg = A + (size_t)(jb + w + cc) * n + jb;
      double a[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) a[k] = colg[k];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int k = j + 1; k < NB; ++k)
          a[k] = (-a[j]) * s_l11[j * (NB + 1) + k] + a[k];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        colg[k] = a[k];
        sh[cc * LDU + k] = a[k];
      }
    }
    __syncthreads();
    mark(4);
    if constexpr (MFMA) {
      // ---- 6 (matrix cores). mc is a multiple of 16 here (n % 16 == 0, NB % 16 == 0)
      typedef double d4 __attribute__((ext_vector_type(4)));
      const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
      const int ntile = mc / 16;
      for (int tr = wave; tr < ntile; tr += THREADS / 64) {
        const int r0 = jb + w + tr * 16;
        double bl[NB / 4];  // L21^T operand of this row tile: L21[r0 + lr][4 kb + lg]
#pragma unroll
        for (int kb = 0; kb < NB / 4; ++kb) bl[kb] = A[(size_t)(jb + kb * 4 + lg) * n + r0 + lr];
        double* ctile = A + (size_t)(jb + w + lg) * n + r0 + lr;  // element (row r0 + lr, column jb + w + c0 + 4 reg + lg) at ctile[(c0 + 4 reg) * n]
        d4 acc, nxt;
#pragma unroll
        for (int q = 0; q < 4; ++q) nxt[q] = ctile[(size_t)(4 * q) * n];
        for (int tc = 0; tc < ntile; ++tc) {
          const int c0 = tc * 16;
          acc = nxt;
          if (tc + 1 < ntile) {
#pragma unroll
            for (int q = 0; q < 4; ++q) nxt[q] = ctile[(size_t)(c0 + 16 + 4 * q) * n];
          }
#pragma unroll
          for (int kb = 0; kb < NB / 4; ++kb) {
            const double au = -sh[(c0 + lr) * LDU + kb * 4 + lg];  // (-U12)^T operand: -U12[4 kb + lg][c0 + lr]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(au, bl[kb], acc, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) ctile[(size_t)(c0 + 4 * q) * n] = acc[q];
        }
      }
    } else
    // ---- 6. trailing update: thread = row, NB multipliers in registers, 4 columns in flight
    for (int r = jb + w + tid; r < n; r += THREADS) {
      double l[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) l[k] = A[(size_t)(jb + k) * n + r];
      double* row = A + (size_t)(jb + w) * n + r;
      // CW columns per group, the next group's loads issued before this group's arithmetic (one wave per SIMD: nothing else hides HBM latency)
      constexpr int CW = 8;
      double a[CW], nx[CW];
      int cc = 0;
      const int full = mc - mc % CW;
      if (full > 0) {
#pragma unroll
        for (int q = 0; q < CW; ++q) nx[q] = row[(size_t)q * n];
      }
      for (; cc < full; cc += CW) {
#pragma unroll
        for (int q = 0; q < CW; ++q) a[q] = nx[q];
        if (cc + CW < full) {
#pragma unroll
          for (int q = 0; q < CW; ++q) nx[q] = row[(size_t)(cc + CW + q) * n];
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
#pragma unroll
          for (int q = 0; q < CW; ++q) a[q] = (-sh[(cc + q) * LDU + k]) * l[k] + a[q];
          if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep later LDS reads from being hoisted (register pressure)
        }
#pragma unroll
        for (int q = 0; q < CW; ++q) row[(size_t)(cc + q) * n] = a[q];
      }
      for (; cc < mc; ++cc) {
        double a0 = row[(size_t)cc * n];
        const double* u0 = sh + cc * LDU;
#pragma unroll
        for (int k = 0; k < NB; ++k) a0 = (-u0[k]) * l[k] + a0;
        row[(size_t)cc * n] = a0;
      }
    }
    __syncthreads();
    mark(5);
  }
  if (singular && tid == 0) publish_singular(singular_word, 1ull, epoch);
}

// dynamic LDS bytes of k_lu_factor_blocked<NB> for order n
inline size_t blocked_lds_bytes(int64_t n, int nbk) {
  const size_t panel = (size_t)(n | 1) * nbk, ublock = (size_t)(n > nbk ? n - nbk : 0) * (nbk + 2);
  return sizeof(double) * (panel > ublock ? panel : ublock);
}

// Blocked triangular solves for n > 64, one workgroup per system, rhs in LDS, factors streamed once from HBM.  The unblocked kernel above needs two
// workgroup barriers per column (2n for L, 2n for U); here the B = 8 columns of a block are eliminated together: the 8 x 8 diagonal block by eight
// lanes of the first wavefront (shuffles), then every other row takes its eight updates v_r = (-v_k) * a_rk + v_r in the same k order as the
// column-by-column algorithm — bit-identical — with the eight loads of the row in flight together.  n/4 barriers instead of 4n.
constexpr int kSolveBlock = 8;
__global__ __launch_bounds__(kCoopThreads) void k_lu_solve_blocked(int n, int64_t nb, const double* __restrict__ f_aos, const int32_t* __restrict__ piv_aos,
                                                                  double* __restrict__ rhs, unsigned long long* rec, unsigned int seq) {
  extern __shared__ double v[];
  constexpr int B = kSolveBlock;
  const int64_t b = blockIdx.x;
  const double* A = f_aos + (size_t)b * n * n;
  const int tid = threadIdx.x;
  for (int r = tid; r < n; r += kCoopThreads) v[r] = rhs[(int64_t)r * nb + b];
  __syncthreads();
  if (tid == 0) {  // the recorded interchanges, in order (sequentially dependent: n cheap LDS operations by one lane)
    const int32_t* P = piv_aos + (size_t)b * n;
    for (int i = 0; i < n; ++i) { const int p = P[i]; if (p != i) { const double tmp = v[i]; v[i] = v[p]; v[p] = tmp; } }
  }
  __syncthreads();
  // ---- L y = P b (unit lower triangle)
  for (int kb = 0; kb < n; kb += B) {
    const int w = (n - kb) < B ? (n - kb) : B;
    if (tid < 64) {  // diagonal block: lane j owns row kb + j
      const int j = tid;
      double a[B];
#pragma unroll
      for (int i = 0; i < B; ++i) a[i] = (j < w && i < j) ? A[(size_t)(kb + i) * n + kb + j] : 0.0;
      double vj = j < w ? v[kb + j] : 0.0;
#pragma unroll
      for (int i = 0; i < B; ++i) {
        const double vi = __shfl(vj, i, 64);
        if (i < j && j < w) vj = (-vi) * a[i] + vj;
      }
      if (j < w) v[kb + j] = vj;
    }
    __syncthreads();
    for (int r = kb + w + tid; r < n; r += kCoopThreads) {
      double a[B];
#pragma unroll
      for (int i = 0; i < B; ++i) a[i] = i < w ? A[(size_t)(kb + i) * n + r] : 0.0;
      double vr = v[r];
#pragma unroll
      for (int i = 0; i < B; ++i) if (i < w) vr = (-v[kb + i]) * a[i] + vr;
      v[r] = vr;
    }
    __syncthreads();
  }
  // ---- U x = y
  bool ok = true;
  const int last = ((n - 1) / B) * B;
  for (int kb = last; kb >= 0; kb -= B) {
    const int w = (n - kb) < B ? (n - kb) : B;
    if (tid < 64) {  // diagonal block, columns kb+w-1 down to kb: lane j owns row kb + j
      const int j = tid;
      double a[B];
#pragma unroll
      for (int i = 0; i < B; ++i) a[i] = (j < w && i < w && i >= j) ? A[(size_t)(kb + i) * n + kb + j] : 1.0;
      double vj = j < w ? v[kb + j] : 0.0;
#pragma unroll
      for (int i = B - 1; i >= 0; --i) {
        if (i < w) {
          const double diag = __shfl(a[i], i, 64);  // U(kb+i, kb+i), held by lane i
          if (diag == 0.0) ok = false;
          const double coeff = __shfl(vj, i, 64) / diag;
          if (j == i) vj = coeff;
          else if (j < i) vj = (-coeff) * a[i] + vj;
        }
      }
      if (j < w) v[kb + j] = vj;
    }
    __syncthreads();
    for (int r = tid; r < kb; r += kCoopThreads) {
      double a[B];
#pragma unroll
      for (int i = 0; i < B; ++i) a[i] = i < w ? A[(size_t)(kb + i) * n + r] : 0.0;
      double vr = v[r];
#pragma unroll
      for (int i = B - 1; i >= 0; --i) if (i < w) vr = (-v[kb + i]) * a[i] + vr;
      v[r] = vr;
    }
    __syncthreads();
  }
  for (int r = tid; r < n; r += kCoopThreads) rhs[(int64_t)r * nb + b] = v[r];
  block_publish(0ull, 0ull, (tid == 0 && !ok) ? 1ull : 0ull, rec, seq);
}

// The same triangular solves with the factor panels PREFETCHED.  k_lu_solve_blocked is a chain of 2 n / B steps, each of which issues its loads, waits for HBM,
// computes and meets a barrier — twice (diagonal block, then the panel): at n = 962 x 256 systems 1.02 ms per solve for 1.9 GB of factors (1.9 TB/s), 45 % of
// the DFN model's kernel time (profiles/r03_dfn_kernel_stats_256.md).  Nothing a step loads depends on the right-hand side, so here every thread keeps a ring
// of D panels in registers: the loads of step s + D are issued as soon as the registers of step s are free, and D - 1 panels (~60 KB per workgroup) are
// in flight while a step computes.  Rows are owned by threads (row r = tid + q * kStreamThreads), the 8 x 8 diagonal block by eight lanes of the first
// wavefront; the interchanges are read into LDS and only the rows that really move are visited.  Every element receives the updates of the
// column-by-column algorithm in its order (v_r = (-v_k) * a_rk + v_r, k ascending for L, descending for U): bit-identical to k_lu_solve_blocked.
constexpr int kStreamThreads = 512;
template <int RPT, int D, int B = kSolveBlock, int T = kStreamThreads>
__global__ __launch_bounds__(T) void k_lu_solve_stream(int n, int64_t nb, const double* __restrict__ f_aos, const int32_t* __restrict__ piv_aos,
                                                                  double* __restrict__ rhs, unsigned long long* rec, unsigned int seq) {
  extern __shared__ double v[];  // n values, then n pivot rows (int), then the bit mask of the rows that move
  // B columns per block step: the fixed cost of a step (two barriers, the diagonal block's chain) is paid n / B times per sweep
  const int64_t b = blockIdx.x;
  const double* A = f_aos + (size_t)b * n * n;
  const int tid = threadIdx.x;
  int* P = reinterpret_cast<int*>(v + n);
  unsigned long long* moved = reinterpret_cast<unsigned long long*>(P + ((n + 1) & ~1));
  const int nwords = (n + 63) / 64;
  const int ns = (n + B - 1) / B;  // block steps of each sweep
  double ring[D][RPT][B];
  double dring[D][B];
  // panel of the L sweep, step s: columns kb .. kb + B - 1, rows below the block; diagonal block: row kb + j, columns left of it
  auto load_L = [&](int s, double (&buf)[RPT][B], double (&dg)[B]) __attribute__((always_inline)) {
    const int kb = s * B;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + q * T;
#pragma unroll
      for (int i = 0; i < B; ++i) buf[q][i] = (s < ns && r >= kb + B && r < n) ? A[(size_t)(kb + i) * n + r] : 0.0;
    }
    if (tid < 64) {
      const int j = tid;
#pragma unroll
      for (int i = 0; i < B; ++i) dg[i] = (s < ns && j < B && i < j && kb + j < n) ? A[(size_t)(kb + i) * n + kb + j] : 0.0;
    }
  };
  // panel of the U sweep, step s (counted from the last block upwards): columns kb .. kb + w - 1, rows above the block
  auto load_U = [&](int s, double (&buf)[RPT][B], double (&dg)[B]) __attribute__((always_inline)) {
    const int kb = (ns - 1 - s) * B;
    const int w = (n - kb) < B ? (n - kb) : B;
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int r = tid + q * T;
#pragma unroll
      for (int i = 0; i < B; ++i) buf[q][i] = (s < ns && r < kb && i < w) ? A[(size_t)(kb + i) * n + r] : 0.0;
    }
    if (tid < 64) {
      const int j = tid;
#pragma unroll
      for (int i = 0; i < B; ++i) dg[i] = (s < ns && j < w && i < w && i >= j) ? A[(size_t)(kb + i) * n + kb + j] : 1.0;
    }
  };
#pragma unroll
  for (int u = 0; u < D; ++u) load_L(u, ring[u], dring[u]);  // in flight while the right-hand side is permuted
  for (int r = tid; r < n; r += T) { v[r] = rhs[(int64_t)r * nb + b]; P[r] = piv_aos[(size_t)b * n + r]; }
  __syncthreads();
  for (int wd = tid >> 6; wd < nwords; wd += T / 64) {
    const int i = wd * 64 + (tid & 63);
    const unsigned long long m = __ballot(i < n && P[i] != i);
    if ((tid & 63) == 0) moved[wd] = m;
  }
  __syncthreads();
  if (tid == 0) {  // the recorded interchanges, in order, visiting only the rows that move
    for (int wd = 0; wd < nwords; ++wd) {
      unsigned long long m = moved[wd];
      while (m) {
        const int i = wd * 64 + __builtin_ctzll(m);
        m &= m - 1;
        const int p = P[i];
        const double tmp = v[i]; v[i] = v[p]; v[p] = tmp;
      }
    }
  }
  __syncthreads();
  // ---- L y = P b (unit lower triangle)
  for (int s0 = 0; s0 < ns; s0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int s = s0 + u;
      if (s < ns) {  // uniform
        const int kb = s * B;
        const int w = (n - kb) < B ? (n - kb) : B;
        if (tid < 64) {
          const int j = tid;
          double vj = j < w ? v[kb + j] : 0.0;
#pragma unroll
          for (int i = 0; i < B; ++i) {
            const double vi = __shfl(vj, i, 64);
            if (i < j && j < w) vj = (-vi) * dring[u][i] + vj;
          }
          if (j < w) v[kb + j] = vj;
        }
        __syncthreads();
        double vk[B];
#pragma unroll
        for (int i = 0; i < B; ++i) vk[i] = i < w ? v[kb + i] : 0.0;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          const int r = tid + q * T;
          if (r >= kb + w && r < n) {
            double vr = v[r];
#pragma unroll
            for (int i = 0; i < B; ++i) if (i < w) vr = (-vk[i]) * ring[u][q][i] + vr;
            v[r] = vr;
          }
        }
        __syncthreads();
        load_L(s + D, ring[u], dring[u]);
      }
    }
  }
  // ---- U x = y
#pragma unroll
  for (int u = 0; u < D; ++u) load_U(u, ring[u], dring[u]);
  bool ok = true;
  for (int s0 = 0; s0 < ns; s0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int s = s0 + u;
      if (s < ns) {
        const int kb = (ns - 1 - s) * B;
        const int w = (n - kb) < B ? (n - kb) : B;
        if (tid < 64) {
          const int j = tid;
          double vj = j < w ? v[kb + j] : 0.0;
#pragma unroll
          for (int i = B - 1; i >= 0; --i) {
            if (i < w) {
              const double diag = __shfl(dring[u][i], i, 64);  // U(kb+i, kb+i), held by lane i
              if (diag == 0.0) ok = false;
              const double coeff = __shfl(vj, i, 64) / diag;
              if (j == i) vj = coeff;
              else if (j < i) vj = (-coeff) * dring[u][i] + vj;
            }
          }
          if (j < w) v[kb + j] = vj;
        }
        __syncthreads();
        double vk[B];
#pragma unroll
        for (int i = 0; i < B; ++i) vk[i] = i < w ? v[kb + i] : 0.0;
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          const int r = tid + q * T;
          if (r < kb) {
            double vr = v[r];
#pragma unroll
            for (int i = B - 1; i >= 0; --i) if (i < w) vr = (-vk[i]) * ring[u][q][i] + vr;
            v[r] = vr;
          }
        }
        __syncthreads();
        load_U(s + D, ring[u], dring[u]);
      }
    }
  }
  for (int r = tid; r < n; r += T) rhs[(int64_t)r * nb + b] = v[r];
  block_publish(0ull, 0ull, (tid == 0 && !ok) ? 1ull : 0ull, rec, seq);
}
inline size_t stream_solve_lds_bytes(int64_t n) { return sizeof(double) * (size_t)n + sizeof(int) * (size_t)((n + 1) & ~1) + 8 * (size_t)((n + 63) / 64); }

}  // namespace dsh
