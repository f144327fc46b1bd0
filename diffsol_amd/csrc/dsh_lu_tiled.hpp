// Dense LU for 64 < n <= 1024 on the matrix cores, one 512-thread workgroup per system (gfx950).  The default for that range since round 3;
// DSH_LU_EXACT=1 keeps k_lu_factor_blocked (dsh_lu_coop.hpp), whose factors are bit-identical to the CPU path.
// Replaces, like every kernel of this library, the host loop over cusolverDnDgetrf of the reference (diffsol-la/src/linear_solver/cuda/lu.rs:59-125):
// same pivot rule (first largest magnitude in the column, rows in the order the interchanges so far have left them), same factor layout out.
//
// What bounded the blocked kernel (profiles/r02: 4-6 TFLOP/s): a panel factorisation with four workgroup barriers per pivot, row interchanges that
// touch one cache line per column (column-major storage), 16 round trips of the trailing matrix through HBM.  Here:
//   * the working copy W of a system is ROW-major (k_lu_stage_rowmajor makes it from the batch-fastest operand, pitch ldw = n rounded up to 64, padding
//     zeroed) and rows NEVER move: a row that has been chosen as a pivot is finished — its L part sits in W, its U part is computed once and written to the
//     factor storage F — and simply leaves the list of active rows.  No interchange costs memory traffic.  Every thread keeps, for the rows it owns,
//     the position the reference's interchanges would have put them at (`pos`), so that ties are broken and pivots are recorded exactly as the
//     sequential algorithm does (pivots[k] = position swapped with k at step k).
//   * panels of 64 columns (8 round trips at n = 512), factored as two sub-panels of 32 columns in REGISTERS, one thread per row (two rows for n > 512):
//     per pivot ONE workgroup barrier — every wavefront reduces its candidates with DPP, the lane that owns the wavefront's best row publishes that row
//     to LDS, and after the barrier all threads pick the same winner among the eight published rows and eliminate with it.
//   * U12 = L11^-1 A12 and the trailing update A22 -= L21 U12 run on v_mfma_f64_16x16x4_f64.  U12 is a blocked substitution with 16 x 16 blocks (the
//     inverses of the four unit-triangular diagonal blocks and the six blocks below them are the A operands; an accumulator register of one product is,
//     as it stands, the B operand of the next: d(l, r) = D[4 r + l/16][l%16] = b(l) of k-block r), kept in LDS for a chunk of 208 columns; the update
//     keeps the L21 operand of up to four row tiles per wavefront in registers and streams the C tiles of the active rows (gathered through the row
//     list) once per panel.
// Arithmetic differs from the exact kernels in fused multiply-adds and in the summation order inside a matrix-core instruction: tested to a tolerance
// (identical pivots, factors to 1e-11 of the largest entry; tests/test_gpu_lu_models.py), not bitwise.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "dsh_device.hpp"

namespace dsh {

constexpr int kTlThreads = 512;
constexpr int kTlWaves = kTlThreads / 64;
constexpr int kTlPW = 64;     // panel width
constexpr int kTlSW = 32;     // sub-panel width (columns of a row held in registers)
constexpr int kTlLDP = 208;   // pitch of the U12 chunk in LDS (doubles); LDP % 32 == 16 keeps the operand reads conflict-free
constexpr int kTlCH = 208;    // columns per chunk (13 tiles of 16)
constexpr int kTlL11P = 65;   // pitch of the staged L11
constexpr int kTlMaxN = 1024;
inline int tiled_ldw(int64_t n) { return (int)((n + 63) / 64 * 64); }
inline size_t tiled_lds_bytes() { return sizeof(double) * (size_t)(64 * kTlLDP + 64 * kTlL11P + 4 * 16 * 17); }

typedef double tl_d4 __attribute__((ext_vector_type(4)));
typedef double tl_d2 __attribute__((ext_vector_type(2)));
// The panel and the trailing phase are functions of their own (register allocation); their pointer arguments would be generic, and a FLAT store counts
// on the LDS counter as well as on the memory counter — every LDS-only barrier would wait for the scattered global stores.  Hence explicit address spaces.
typedef __attribute__((address_space(1))) double tl_gdouble;
typedef __attribute__((address_space(1))) tl_d2 tl_gd2;
typedef __attribute__((address_space(1))) char tl_gchar;

// batch-fastest operand a[(j*n + i)*nb + b] -> row-major working copies w[b][i][j], pitch ldw, columns n..ldw-1 zero.  One 32 x 32 (column x system)
// tile per workgroup and row: reads are coalesced along b, writes along j.
__global__ void k_lu_stage_rowmajor(int n, int ldw, int64_t nb, const double* __restrict__ a, double* __restrict__ w) {
  __shared__ double tile[32][33];
  const int i = blockIdx.z;
  const int64_t b0 = (int64_t)blockIdx.x * 32;
  const int j0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int j = j0 + k;
    const int64_t b = b0 + tx;
    tile[k][tx] = (j < n && b < nb) ? a[((int64_t)j * n + i) * nb + b] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int64_t b = b0 + k;
    const int j = j0 + tx;
    if (b < nb && j < ldw) w[(size_t)b * n * ldw + (size_t)i * ldw + j] = tile[tx][k];
  }
}

// trailing update of one group of RT row tiles over the column tiles tc0, tc0 + tcs, ... of the current chunk.  roffb: byte offsets of the lane's four
// rows of every tile (32-bit: the loads and stores take the system's base from scalar registers)
template <int RT>
__device__ __forceinline__ void tl_update_tiles(tl_gdouble* __restrict__ W, const double* __restrict__ u12s, const double (&aneg)[4][16], const unsigned (&roffb)[4][4],
                                                unsigned valid, int c_lo, int ntc, int tc0, int tcs, int lane) {
  const int q = lane >> 4, j = lane & 15;
  tl_gchar* const Wb = reinterpret_cast<tl_gchar*>(W);
  tl_d4 acc[RT], nxt[RT];
  const unsigned colb0 = (unsigned)(c_lo + j) * 8u;
  if (tc0 < ntc) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) nxt[t][r] = *reinterpret_cast<const tl_gdouble*>(Wb + (roffb[t][r] + colb0 + 128u * (unsigned)tc0));
  }
  for (int tc = tc0; tc < ntc; tc += tcs) {
    const unsigned colb = colb0 + 128u * (unsigned)tc;
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = nxt[t];
    if (tc + tcs < ntc) {
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) nxt[t][r] = *reinterpret_cast<const tl_gdouble*>(Wb + (roffb[t][r] + colb + 128u * (unsigned)tcs));
    }
    const double* ub = u12s + q * kTlLDP + 16 * tc + j;
    double bv[4], bn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bn[e] = ub[4 * e * kTlLDP];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {  // four k-blocks at a time, the next four U12 operands in flight behind them (and no more: registers)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = bn[e];
      if (kg < 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bn[e] = ub[4 * (4 * kg + 4 + e) * kTlLDP];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aneg[t][4 * kg + e], bv[e], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (valid & (1u << (4 * t + r))) *reinterpret_cast<tl_gdouble*>(Wb + (roffb[t][r] + colb)) = acc[t][r];
  }
}

// (value, position) arg-max over the wavefront, smallest position on ties, without branches: four DPP stages inside each row of 16 lanes, then the
// four row results through v_readlane.  Every lane returns the wavefront's result.
template <int CTRL>
__device__ __forceinline__ void tl_argmax_stage(double& v, int& p) {
  const double ov = __longlong_as_double((long long)dpp_move_u64<CTRL>((unsigned long long)__double_as_longlong(v)));
  const int op = __builtin_amdgcn_update_dpp(p, p, CTRL, 0xf, 0xf, false);
  const bool better = (ov > v) | ((ov == v) & (op < p));
  v = better ? ov : v;
  p = better ? op : p;
}
template <int CTRL>
__device__ __forceinline__ double tl_dpp_max(double v) {
  return __builtin_fmax(v, __longlong_as_double((long long)dpp_move_u64<CTRL>((unsigned long long)__double_as_longlong(v))));
}
// The common case first: the maximum alone (one v_max per stage), then a ballot of the lanes that hold it; only a tie (several lanes with the same
// magnitude) takes the reduction that carries the positions along.  No NaN reaches this (mapped to -1 before).
__device__ __forceinline__ void tl_wave_argmax(double& v, int& p) {
  double m = tl_dpp_max<kDppQuadXor1>(v);
  m = tl_dpp_max<kDppQuadXor2>(m);
  m = tl_dpp_max<kDppRowHalfMirror>(m);
  m = tl_dpp_max<kDppRowMirror>(m);
  const double m0 = __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(m), 0));
  const double m1 = __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(m), 16));
  const double m2 = __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(m), 32));
  const double m3 = __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(m), 48));
  const double wm = __builtin_fmax(__builtin_fmax(m0, m1), __builtin_fmax(m2, m3));
  const unsigned long long holders = __ballot(v == wm);
  if (__popcll(holders) == 1) {
    p = __builtin_amdgcn_readlane(p, __ffsll((long long)holders) - 1);
    v = wm;
    return;
  }
  tl_argmax_stage<kDppQuadXor1>(v, p);
  tl_argmax_stage<kDppQuadXor2>(v, p);
  tl_argmax_stage<kDppRowHalfMirror>(v, p);
  tl_argmax_stage<kDppRowMirror>(v, p);
  double bv = __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(v), 0));
  int bp = __builtin_amdgcn_readlane(p, 0);
#pragma unroll
  for (int r = 1; r < 4; ++r) {
    const double ov = __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(v), 16 * r));
    const int op = __builtin_amdgcn_readlane(p, 16 * r);
    const bool better = (ov > bv) | ((ov == bv) & (op < bp));
    bv = better ? ov : bv;
    bp = better ? op : bp;
  }
  v = bv;
  p = bp;
}

// The pivot steps of a sub-panel held in registers, as ONE rolled loop (a loop body per step, unrolled so that the register arrays get static indices, was
// 100 KB of code).  The array ROTATES instead: a[i][0] is always the pivot column, every step writes a[i][c-1] = a[i][c] - u[c] l and shifts a zero in at
// the right, so register indices are static while the step index is not.  What leaves the array goes to W at once: the multiplier of every active row (one
// 8-byte store per row and step), and the winner's row — its U entries from the pivot column on — copied from LDS by one wavefront.  Rows that are
// finished keep rotating garbage nobody reads.
// Per step two exchanges through LDS, each closed by a barrier that waits for LDS only (never for the global stores): (1) every wavefront reduces its
// candidates with DPP and publishes {|value|, position}; everybody picks the same winner; (2) the one thread that owns the winning row publishes the row
// (its 32 - k live columns), its reciprocal pivot and its row index.  (Publishing all eight wavefronts' candidate rows before a single barrier was
// measured: 2.4 us per step, the LDS port busy with 136 one-lane 16-byte writes.)
// s_slot[0][w] = header of wavefront w; s_slot[1][0] = {-, inverse pivot, row index} + the row at [4..36).
#ifdef TL_X_STEPPROF
__device__ unsigned long long tl_stepprof[5];
#endif
template <int R>
__device__ __forceinline__ void tl_subpanel_steps(double (&a)[R][kTlSW], bool (&act)[R], int (&pos)[R], bool& singular, double (*s_slot)[kTlWaves][36], int* s_prow,
                                                  int* s_ipiv, tl_gdouble* __restrict__ W, int ldw, int ws, int cb, int pbase, int tid, int wave, int lane) {
  double* const rowbuf = &s_slot[1][0][0];
#ifdef TL_X_STEPPROF
  unsigned long long tacc[5] = {0, 0, 0, 0, 0};
#define TL_T(ix) { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[ix] += now_ - t0_; t0_ = now_; }
#else
#define TL_T(ix)
#endif
#pragma nounroll
  for (int k = 0; k < ws; ++k) {
#ifdef TL_X_STEPPROF
    unsigned long long t0_ = __builtin_readcyclecounter();
#endif
    const int g = cb + k;
    const int nact = kTlSW - k;  // live columns of the rotating arrays
    double wv = -2.0;
    int wp = 0x7fffffff;
    double rinv[R];  // the step's division, for every row before anybody knows the winner: off the chain of dependent exchanges
#pragma unroll
    for (int i = 0; i < R; ++i) rinv[i] = 1.0 / a[i][0];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      double v = __builtin_fabs(a[i][0]);
      v = (v == v) ? v : -1.0;  // a NaN never beats a number; if nothing else is left the row at the diagonal position wins, like the sequential scan
      v = act[i] ? v : -2.0;
      const int pi = act[i] ? pos[i] : 0x7fffffff;
      const bool better = (v > wv) | ((v == wv) & (pi < wp));
      wv = better ? v : wv;
      wp = better ? pi : wp;
    }
    tl_wave_argmax(wv, wp);
    TL_T(0)
    if (lane == 0) { tl_d2 h; h[0] = wv; h[1] = __hiloint2double(0, wp); *reinterpret_cast<tl_d2*>(&s_slot[0][wave][0]) = h; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TL_T(1)
    // the same winner in every thread: lane l reads the header of wavefront l % 8 (ONE LDS round trip; eight reads in a row came out as four), three DPP
    // stages reduce every group of eight lanes
    int bpos;
    {
      const tl_d2 h = *reinterpret_cast<const tl_d2*>(&s_slot[0][lane & (kTlWaves - 1)][0]);
      double hv = h[0];
      int hp = __double2loint(h[1]);
      tl_argmax_stage<kDppQuadXor1>(hv, hp);
      tl_argmax_stage<kDppQuadXor2>(hv, hp);
      tl_argmax_stage<kDppRowHalfMirror>(hv, hp);
      bpos = __builtin_amdgcn_readfirstlane(hp);
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
      if (act[i] && pos[i] == bpos) {  // one thread of the workgroup
        rowbuf[1] = rinv[i];
        rowbuf[2] = __hiloint2double(0, tid + kTlThreads * i);
#pragma unroll
        for (int c = 0; c < kTlSW; c += 2)
          if (c < nact) { tl_d2 v; v[0] = a[i][c]; v[1] = a[i][c + 1]; *reinterpret_cast<tl_d2*>(rowbuf + 4 + c) = v; }
      }
    TL_T(2)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TL_T(3)
    const double inv = rowbuf[1];
    const int brow = __builtin_amdgcn_readfirstlane(__double2loint(rowbuf[2]));
    double u[kTlSW];
#pragma unroll
    for (int c = 0; c < kTlSW; c += 2) {
      if (c < nact) { const tl_d2 v = *reinterpret_cast<const tl_d2*>(rowbuf + 4 + c); u[c] = v[0]; u[c + 1] = v[1]; }
      else { u[c] = 0.0; u[c + 1] = 0.0; }
    }
    const double diag = u[0];
    const bool zero = diag == 0.0;
    if (zero) singular = true;
    if (tid == 0) { s_ipiv[pbase + k] = bpos; s_prow[pbase + k] = brow; }
    if (wave == (k & (kTlWaves - 1)) && lane < nact) W[(size_t)brow * ldw + g + lane] = rowbuf[4 + lane];  // the winner's row from its diagonal entry on
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const bool winner = act[i] & (pos[i] == bpos);
      const double l = zero ? a[i][0] : a[i][0] * inv;  // a zero pivot leaves its column (all zeros) as it is
      if (act[i] & !winner) W[(size_t)(tid + kTlThreads * i) * ldw + g] = l;
      if (act[i]) {
        if (winner) { act[i] = false; pos[i] = g; }
        else if (pos[i] == g) pos[i] = bpos;
      }
      const double le = zero ? 0.0 : l;
#pragma unroll
      for (int c = 1; c < kTlSW; ++c) a[i][c - 1] = __builtin_fma(-u[c], le, a[i][c]);
      a[i][kTlSW - 1] = 0.0;
    }
    // the row buffer is rewritten only behind the first barrier of the next step, the headers behind the second of this one: single buffers are enough
    TL_T(4)
  }
#ifdef TL_X_STEPPROF
  if (tid == 0 && blockIdx.x == 0) for (int e = 0; e < 5; ++e) atomicAdd(&tl_stepprof[e], tacc[e]);
#endif
}

// The panel of 64 columns at jb.  Not inlined: its registers (the rows' 32 columns, the pivot row) are allocated apart from the rest of the kernel —
// a spill reload inside the step loop would wait for the global stores of the previous steps (one counter for loads and stores: ~10 us per step).
// st: positions, activity flags of this thread's R rows, the singular flag (in / out).
template <int R>
__device__ __noinline__ void tl_panel(double* __restrict__ W_generic, int ldw, int n, int jb, double* dyn, double (*s_slot)[kTlWaves][36], int* s_prow, int* s_ipiv, int* st) {
  tl_gdouble* const W = (tl_gdouble*)W_generic;
  double* const u12s = dyn;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  bool act[R];
  int pos[R];
#pragma unroll
  for (int i = 0; i < R; ++i) { pos[i] = st[i]; act[i] = st[R + i] != 0; }
  bool singular = st[2 * R] != 0;
    double a[R][kTlSW];
#pragma nounroll
    for (int s = 0; s < 2; ++s) {
      const int cb = jb + kTlSW * s;
      const int ws = (n - cb) < kTlSW ? (n - cb) : kTlSW;
      if (ws <= 0) break;
      if (s == 0) {
#pragma unroll
        for (int i = 0; i < R; ++i)
          if (act[i]) {
            const tl_gd2* src = reinterpret_cast<const tl_gd2*>(W + (size_t)(tid + kTlThreads * i) * ldw + cb);
#pragma unroll
            for (int c = 0; c < kTlSW; c += 2) { const tl_d2 v = src[c >> 1]; a[i][c] = v[0]; a[i][c + 1] = v[1]; }
          }
      } else {
        // ---- columns cb..cb+31 take the 32 eliminations of the first sub-panel: U' = L11A^-1 (pivot rows' entries), then row -= L_row U'
        double* const Bp = u12s;
        double* const Us = u12s + 1024;
        double* const L11A = u12s + 2048;
        {
          const int k = tid >> 4, c2 = (tid & 15) * 2;  // 32 pivot rows x 16 pairs of columns
          const tl_gdouble* const prw = W + (size_t)s_prow[k] * ldw + jb;
          *reinterpret_cast<tl_d2*>(Bp + k * 32 + c2) = *reinterpret_cast<const tl_gd2*>(prw + kTlSW + c2);
          *reinterpret_cast<tl_d2*>(L11A + k * 32 + c2) = *reinterpret_cast<const tl_gd2*>(prw + c2);
        }
        __syncthreads();
        if (wave == 0 && lane < 32) {
          double x[32];
#pragma unroll
          for (int k = 0; k < 32; ++k) x[k] = Bp[k * 32 + lane];
#pragma unroll
          for (int i = 0; i < 31; ++i)
#pragma unroll
            for (int k = i + 1; k < 32; ++k) x[k] = __builtin_fma(-L11A[k * 32 + i], x[i], x[k]);
#pragma unroll
          for (int k = 0; k < 32; ++k) {
            Us[k * 32 + lane] = x[k];
            W[(size_t)s_prow[k] * ldw + cb + lane] = x[k];
          }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < R; ++i)
          if (act[i]) {
            double b[kTlSW], l[kTlSW];
            const tl_gd2* src = reinterpret_cast<const tl_gd2*>(W + (size_t)(tid + kTlThreads * i) * ldw + jb);
#pragma unroll
            for (int c = 0; c < kTlSW; c += 2) { const tl_d2 v = src[c >> 1]; l[c] = v[0]; l[c + 1] = v[1]; }
#pragma unroll
            for (int c = 0; c < kTlSW; c += 2) { const tl_d2 v = src[(kTlSW + c) >> 1]; b[c] = v[0]; b[c + 1] = v[1]; }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
#pragma unroll
              for (int c = 0; c < kTlSW; c += 2) {
                const tl_d2 u = *reinterpret_cast<const tl_d2*>(Us + k * 32 + c);
                b[c] = __builtin_fma(-l[k], u[0], b[c]);
                b[c + 1] = __builtin_fma(-l[k], u[1], b[c + 1]);
              }
            }
#pragma unroll
            for (int c = 0; c < kTlSW; ++c) a[i][c] = b[c];
          }
      }
      const int pbase = kTlSW * s;
      tl_subpanel_steps<R>(a, act, pos, singular, s_slot, s_prow, s_ipiv, W, ldw, ws, cb, pbase, tid, wave, lane);
      __syncthreads();
    }
#pragma unroll
  for (int i = 0; i < R; ++i) { st[i] = pos[i]; st[R + i] = act[i] ? 1 : 0; }
  st[2 * R] = singular ? 1 : 0;
}

// Everything behind a finished 64-column panel: U12 and the update of the active rows, a chunk of <= 208 trailing columns at a time.  A function of its own
// (not inlined) so that its registers — the L21 operand of four row tiles stays in them across the chunks — are allocated apart from the panel's.
__device__ __noinline__ void tl_trailing(double* __restrict__ W_generic, double* __restrict__ F_generic, double* dyn, const int* s_prow, const unsigned short* s_rowlist,
                                         unsigned long long* phase_clocks, int n, int ldw, int jb, int nct, int m2) {
  tl_gdouble* const W = (tl_gdouble*)W_generic;
  tl_gdouble* const F = (tl_gdouble*)F_generic;
  double* const u12s = dyn;
  const double* const l11 = dyn + 64 * kTlLDP;
  const double* const invd = l11 + 64 * kTlL11P;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool prof = phase_clocks != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long tprev = prof ? wall_clock64() : 0ull;
  auto mark = [&](int phase) {
    if (prof) { const unsigned long long now = wall_clock64(); phase_clocks[phase] += now - tprev; tprev = now; }
  };
    const int q = lane >> 4, j = lane & 15;
    // decomposition of the update: groups of RTw row tiles; with fewer groups than wavefronts the column tiles are split as well
    const int nrt = (m2 + 15) / 16;
    const int rtw = (nrt + kTlWaves - 1) / kTlWaves < 4 ? (nrt + kTlWaves - 1) / kTlWaves : 4;
    const int groups = (nrt + rtw - 1) / rtw;
    const int csplit = groups >= kTlWaves ? 1 : kTlWaves / groups;
    const bool fixed_group = groups <= kTlWaves;
    double aneg[4][16];
    unsigned roffb[4][4];
    unsigned valid = 0;
    int loaded_group = -1;
    for (int c_lo = jb + kTlPW; c_lo < nct; c_lo += kTlCH) {
      const int cw = (nct - c_lo) < kTlCH ? (nct - c_lo) : kTlCH;
      const int ntc = cw / 16;
      // ---- U12 of the chunk: blocked substitution on the matrix cores, one column tile per wavefront at a time
      {
        // the A operands (inverses of the diagonal blocks, negated blocks below them) come from LDS as they are needed: holding all 40 of them next
        // to the L21 operand of the update, which stays in registers across the chunks, does not fit in 256 registers
        const double* const dinv_l = invd + j * 17 + q;            // block b, k-block kb: + b * 272 + 4 kb
        const double* const l11_l = l11 + j * kTlL11P + q;         // block (rb, cbk), k-block kb: + 16 rb * pitch + 16 cbk + 4 kb
        for (int tc = wave; tc < ntc; tc += kTlWaves) {
          const int c0 = c_lo + 16 * tc;
          tl_d4 B[4], X[4];
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) B[rb][r] = W[(size_t)(s_prow[16 * rb + 4 * r + q] * ldw) + c0 + j];
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
            for (int cbk = 0; cbk < rb; ++cbk) {
              double lo[4];
#pragma unroll
              for (int kb = 0; kb < 4; ++kb) lo[kb] = -l11_l[16 * rb * kTlL11P + 16 * cbk + 4 * kb];
#pragma unroll
              for (int kb = 0; kb < 4; ++kb) B[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo[kb], X[cbk][kb], B[rb], 0, 0, 0);
            }
            double di[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) di[kb] = dinv_l[rb * 272 + 4 * kb];
            tl_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(di[kb], B[rb][kb], acc, 0, 0, 0);
            X[rb] = acc;
          }
          const bool incol = c0 + j < n;
#pragma unroll
          for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              u12s[(16 * rb + 4 * r + q) * kTlLDP + 16 * tc + j] = X[rb][r];
              if (incol) F[(size_t)(c0 + j) * n + jb + 16 * rb + 4 * r + q] = X[rb][r];
            }
        }
      }
      __syncthreads();
      mark(2);
      // ---- A22 -= L21 U12 for the chunk's columns
      for (int grp = fixed_group ? wave % groups : wave; grp < groups; grp += kTlWaves) {
        const int csub = fixed_group ? wave / groups : 0;
        if (csub >= csplit) break;
        if (grp != loaded_group) {
          loaded_group = grp;
          valid = 0;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int tile = grp * rtw + t;
            const bool tv = t < rtw && tile < nrt;
            const int arow = tv ? (int)s_rowlist[16 * tile + j] * ldw : 0;
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) aneg[t][kb] = tv ? -W[(size_t)arow + jb + 4 * kb + q] : 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              roffb[t][r] = tv ? (unsigned)s_rowlist[16 * tile + 4 * r + q] * (unsigned)ldw * 8u : 0u;
              if (tv && 16 * tile + 4 * r + q < m2) valid |= 1u << (4 * t + r);
            }
          }
        }
        const int ngt = (nrt - grp * rtw) < rtw ? (nrt - grp * rtw) : rtw;  // row tiles of this group
        switch (ngt) {
          case 1: tl_update_tiles<1>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
          case 2: tl_update_tiles<2>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
          case 3: tl_update_tiles<3>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
          default: tl_update_tiles<4>(W, u12s, aneg, roffb, valid, c_lo, ntc, csub, csplit, lane); break;
        }
        if (fixed_group) break;
      }
      __syncthreads();
      mark(3);
    }
  }

template <int R>
__global__ __launch_bounds__(kTlThreads) void k_lu_factor_tiled(int n, int ldw, double* __restrict__ w_all, double* __restrict__ f_all, int32_t* __restrict__ piv_all,
                                                                 unsigned long long* singular_word, unsigned int epoch, unsigned long long* phase_clocks) {
  const bool prof = phase_clocks != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long tprev = prof ? wall_clock64() : 0ull;
  auto mark = [&](int phase) {
    if (prof) { const unsigned long long now = wall_clock64(); phase_clocks[phase] += now - tprev; tprev = now; }
  };
  extern __shared__ double dyn[];
  double* const u12s = dyn;                      // [64][LDP]; during the panel: Bp [32][32], Us [32][32], L11A [32][32]
  double* const l11 = dyn + 64 * kTlLDP;         // [64][65]
  double* const invd = l11 + 64 * kTlL11P;       // [4][16][17]
  __shared__ double s_slot[2][kTlWaves][36];     // per wavefront: {value, (row, pos)} + the candidate row's 32 columns at [4..36)
  __shared__ int s_prow[kTlPW], s_ipiv[kTlPW];
  __shared__ unsigned short s_rowlist[kTlMaxN + 16];
  __shared__ int s_wcnt[R][kTlWaves];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double* const W = w_all + (size_t)blockIdx.x * n * ldw;
  double* const F = f_all + (size_t)blockIdx.x * n * n;
  int32_t* const PIV = piv_all + (size_t)blockIdx.x * n;
  bool act[R];
  int pos[R];
#pragma unroll
  for (int i = 0; i < R; ++i) { const int row = tid + kTlThreads * i; act[i] = row < n; pos[i] = row; }
  bool singular = false;
  const int nct = (n + 15) / 16 * 16;  // columns processed by the tile phases (the padding up to it stays isolated in its own columns)

  for (int jb = 0; jb < n; jb += kTlPW) {
    const int pw = (n - jb) < kTlPW ? (n - jb) : kTlPW;
    // =========================================================== panel: two sub-panels of 32 columns in registers
    {
      int st[2 * R + 1];  // the rows' bookkeeping travels through memory: the panel is a function of its own (registers)
#pragma unroll
      for (int i = 0; i < R; ++i) { st[i] = pos[i]; st[R + i] = act[i] ? 1 : 0; }
      st[2 * R] = singular ? 1 : 0;
      tl_panel<R>(W, ldw, n, jb, dyn, &s_slot[0], s_prow, s_ipiv, st);
#pragma unroll
      for (int i = 0; i < R; ++i) { pos[i] = st[i]; act[i] = st[R + i] != 0; }
      singular = st[2 * R] != 0;
    }
    mark(0);
    // =========================================================== the 64 finished rows -> F; list of the rows still active; L11 and its diagonal-block inverses
    int m2 = 0;
    {
      unsigned long long bal[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        bal[i] = __ballot(act[i]);
        if (lane == 0) s_wcnt[i][wave] = __popcll(bal[i]);
      }
      __syncthreads();
      int base = 0;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        int before = 0, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < kTlWaves; ++w2) { const int cnt = s_wcnt[i][w2]; total += cnt; if (w2 < wave) before += cnt; }
        if (act[i]) s_rowlist[base + before + __popcll(bal[i] & ((1ull << lane) - 1ull))] = (unsigned short)(tid + kTlThreads * i);
        base += total;
      }
      m2 = base;
    }
    const int mc = nct - jb - kTlPW;  // trailing columns (exist only behind a full panel)
    const bool trailing = pw == kTlPW && mc > 0 && m2 > 0;
    if (trailing) {
      for (int idx = tid; idx < 64 * 64; idx += kTlThreads) {
        const int k = idx >> 6, i = idx & 63;
        l11[k * kTlL11P + i] = i < k ? W[(size_t)s_prow[k] * ldw + jb + i] : (i == k ? 1.0 : 0.0);
      }
    }
    for (int c = tid; c < jb + pw; c += kTlThreads) {
      double* const dst = F + (size_t)c * n + jb;
      for (int k0 = 0; k0 < pw; k0 += 16) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (k0 + k < pw) ? W[(size_t)s_prow[k0 + k] * ldw + c] : 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k0 + k < pw) dst[k0 + k] = v[k];
      }
    }
    if (tid < pw) PIV[jb + tid] = s_ipiv[tid];
    __syncthreads();
    if (!trailing) { mark(1); continue; }
    if (tid < 16) s_rowlist[m2 + tid] = s_rowlist[m2 - 1];  // padding of the last row tile: a valid row, never stored
    if (wave == 1) {  // inverse of the four 16 x 16 unit lower triangular diagonal blocks: lane = (block, column)
      const int blk = lane >> 4, jc = lane & 15;
      double x[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = r == jc ? 1.0 : 0.0;
#pragma unroll
      for (int i = 0; i < 15; ++i)
#pragma unroll
        for (int r = i + 1; r < 16; ++r) x[r] = __builtin_fma(-l11[(16 * blk + r) * kTlL11P + 16 * blk + i], x[i], x[r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) invd[(blk * 16 + r) * 17 + jc] = x[r];
    }
    __syncthreads();
    mark(1);
    tl_trailing(W, F, dyn, s_prow, s_rowlist, phase_clocks, n, ldw, jb, nct, m2);
    if (prof) tprev = wall_clock64();
  }
  if (singular && tid == 0) publish_singular(singular_word, 1ull, epoch);
}

}  // namespace dsh
