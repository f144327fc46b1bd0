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
//   * panels of 64 columns (8 round trips at n = 512), factored as two sub-panels of 32 columns in REGISTERS with the columns dealt to the wavefronts
//     (tl_panel below): per pivot ONE workgroup barrier and one exchange through LDS — the column of multipliers.
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
#include <cstdlib>

#include "dsh_device.hpp"

namespace dsh {

constexpr int kTlMaxThreads = 512;
constexpr int kTlPW = 64;     // panel width
constexpr int kTlLC = 16;     // columns of multipliers kept in LDS at a time (a sub-panel is flushed to W in two halves)
constexpr int kTlL11P = 49;   // pitch of the staged L11 (rows 16..63, columns 0..47: the blocks below the diagonal blocks)
constexpr int kTlMaxN = 1024;
constexpr int kTlL21P = 66;   // pitch of the staged L21 chunk [64][kTlL21P] of the column-dealt trailing phase (operand reads free of bank conflicts)
// pitch of the row-major working copy.  DSH_TL_LDW_PAD (doubles, a multiple of 16) adds to it: experiment on memory-channel conflicts of a 4 KB pitch
inline int tiled_ldw(int64_t n) {
  static const int pad = [] { const char* e = getenv("DSH_TL_LDW_PAD"); return e ? atoi(e) & ~15 : 0; }();
  return (int)((n + 63) / 64 * 64) + pad;
}
typedef double tl_d4 __attribute__((ext_vector_type(4)));
typedef double tl_d2 __attribute__((ext_vector_type(2)));
// The panel and the trailing phase are functions of their own (register allocation); their pointer arguments would be generic, and a FLAT store counts
// on the LDS counter as well as on the memory counter — every LDS-only barrier would wait for the scattered global stores.  Hence explicit address spaces.
typedef __attribute__((address_space(1))) double tl_gdouble;
typedef __attribute__((address_space(1))) tl_d2 tl_gd2;
typedef __attribute__((address_space(1))) char tl_gchar;

// Arguments of a function that is not inlined arrive in vector registers, and the function cannot know that they are the same in every lane: these make
// them scalars again (in the two-workgroups-per-CU layout the ~25 registers they would occupy per lane are a fifth of the budget).
__device__ __forceinline__ int tl_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
template <class T> __device__ __forceinline__ T* tl_uni(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}

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
// rows of every tile (32-bit: the loads and stores take the system's base from scalar registers).  LDP: pitch of the U12 chunk in LDS.  PF: how many
// column tiles ahead the C tiles are loaded while one is multiplied.
template <int RT, int RTMAX, int LDP, int PF>
__device__ __forceinline__ void tl_update_tiles(tl_gdouble* __restrict__ W, const double* __restrict__ u12s, const double (&aneg)[RTMAX][16], const unsigned (&roffb)[RTMAX][4],
                                                unsigned valid, int c_lo, int ntc, int tc0, int tcs, int lane) {
  const int q = lane >> 4, j = lane & 15;
  tl_gchar* const Wb = reinterpret_cast<tl_gchar*>(W);
  tl_d4 acc[RT], nxt[PF >= 1 ? RT : 1], nx2[PF >= 2 ? RT : 1];
  const unsigned colb0 = (unsigned)(c_lo + j) * 8u;
  auto load_tiles = [&](tl_d4 (&dst)[RT], int tc) {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[t][r] = *reinterpret_cast<const tl_gdouble*>(Wb + (roffb[t][r] + colb0 + 128u * (unsigned)tc));
  };
  if constexpr (PF >= 1) { if (tc0 < ntc) load_tiles(nxt, tc0); }
  if constexpr (PF >= 2) { if (tc0 + tcs < ntc) load_tiles(nx2, tc0 + tcs); }
  for (int tc = tc0; tc < ntc; tc += tcs) {
    const unsigned colb = colb0 + 128u * (unsigned)tc;
    if constexpr (PF >= 2) {  // the C tiles of the next two column tiles are in flight while this one is multiplied
#pragma unroll
      for (int t = 0; t < RT; ++t) { acc[t] = nxt[t]; nxt[t] = nx2[t]; }
      if (tc + 2 * tcs < ntc) load_tiles(nx2, tc + 2 * tcs);
    } else if constexpr (PF == 1) {
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = nxt[t];
      if (tc + tcs < ntc) load_tiles(nxt, tc + tcs);
    } else {
      load_tiles(acc, tc);
    }
    const double* ub = u12s + q * LDP + 16 * tc + j;
    double bv[4], bn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bn[e] = ub[4 * e * LDP];
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {  // four k-blocks at a time, the next four U12 operands in flight behind them (and no more: registers)
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = bn[e];
      if (kg < 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bn[e] = ub[4 * (4 * kg + 4 + e) * LDP];
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

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The panel.  A sub-panel of 32 columns lives in registers with the COLUMNS dealt to the wavefronts — wavefront w holds columns w, w + 8, w + 16, w + 24
// of it, all rows (lane l: rows l, l + 64, ...; RS of them) — because then a pivot step needs ONE exchange: the wavefront that owns column k finds the
// pivot by itself (DPP reduction, no LDS), forms the multipliers of all rows and publishes them (a column of the LDS buffer Lbuf, which is also where the
// flush to W reads them from) together with {lane, slot} of the pivot row; behind ONE barrier every wavefront reads its rows' multipliers, takes the pivot
// row's entry of each of its own columns out of its own registers (v_readlane at the published lane; the slot selects the register through a uniform
// switch) and eliminates.  Measured before this form (one thread per row, the pivot row broadcast through LDS): 2.1-2.5 us per pivot step in three
// variants — two exchanges per step, the second one a 1 KB broadcast read per wavefront.
// Rows never move; where the interchanges of the reference would have put them is kept in LDS: pos[row] (rows with pos < the current step are finished)
// and its inverse rowat[pos]; only the owner of a step reads them, one lane updates them.
// Layout of the dynamic LDS during the panel (doubles), P = 64 RS rows: [0, 16 P) T / Lbuf — [16][P]: first the transposing stage (thread per row in,
// column per wavefront out, 16 columns at a time), then the multipliers of 16 steps, column k & 15 at (k & 15) P, flushed to W after step 15 and after
// step 31 — and before the stage of the second sub-panel the operands of its matrix-core update (L11A, the inverses of its diagonal blocks, U');
// [16 P, 16 P + 1056) Ubuf [32][33]: the pivot rows' entries of the sub-panel's own columns (U11), written by the column owners step by step.
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
__device__ __forceinline__ double tl_readlane_f64(double x, int lane) {
  return __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(x), lane));
}
// Lane of the wavefront's best candidate (largest v, smallest position p on ties).  The common case first: the maximum alone (one v_max per DPP stage),
// then a ballot of the lanes that hold it; only a tie takes the reduction that carries the positions along.  No NaN reaches this (mapped to -1 before).
__device__ __forceinline__ int tl_wave_argmax_lane(double v, int p) {
  double m = tl_dpp_max<kDppQuadXor1>(v);
  m = tl_dpp_max<kDppQuadXor2>(m);
  m = tl_dpp_max<kDppRowHalfMirror>(m);
  m = tl_dpp_max<kDppRowMirror>(m);
  const double wm = __builtin_fmax(__builtin_fmax(tl_readlane_f64(m, 0), tl_readlane_f64(m, 16)), __builtin_fmax(tl_readlane_f64(m, 32), tl_readlane_f64(m, 48)));
  unsigned long long holders = __ballot(v == wm);
  if (__popcll(holders) != 1) {  // several lanes hold the largest magnitude: the smallest position among them
    int q = (v == wm) ? p : 0x7fffffff;
    double dummy = 0.0;
    tl_argmax_stage<kDppQuadXor1>(dummy, q);
    tl_argmax_stage<kDppQuadXor2>(dummy, q);
    tl_argmax_stage<kDppRowHalfMirror>(dummy, q);
    tl_argmax_stage<kDppRowMirror>(dummy, q);
    int bq = __builtin_amdgcn_readlane(q, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) { const int oq = __builtin_amdgcn_readlane(q, 16 * r); bq = oq < bq ? oq : bq; }
    holders = __ballot((v == wm) & (p == bq));
  }
  return __ffsll((long long)holders) - 1;
}

template <int RS> struct tl_colvec;
template <> struct tl_colvec<8> { typedef double type __attribute__((ext_vector_type(8))); };
template <> struct tl_colvec<16> { typedef double type __attribute__((ext_vector_type(16))); };
// a register column of the panel: RS rows of one lane.  A VECTOR, so that a wavefront-uniform run-time slot becomes relative register addressing
// (v_movrels) instead of a switch over the slots — whose joins cost ~60 register moves per step.
template <int RS> using tl_col = typename tl_colvec<RS>::type;
// entry `ss` (the same in every lane) of a register column: relative register addressing
template <int RS>
__device__ __forceinline__ double tl_slot(const tl_col<RS>& c, int ss) { return c[ss]; }

// start-up stagger (experiment, profiles/r04_lu_bench.md): workgroup b waits (b & 3) * tl_stagger_ticks ticks of the 100 MHz clock before it starts, so that the CUs do not
// go through their panel / update phases in step (all in the update at once = HBM contention); 0 = off
static __device__ int tl_stagger_ticks = 0;
constexpr int kTlLaP = 49;                // pitch of L11A in LDS
constexpr int kTlUs = 2400, kTlUsP = 48;  // offset (doubles) of Us in the dynamic LDS (behind L11A [48][49]) and its pitch (conflict-free operand reads)

#ifdef TL_X_STEPPROF
__device__ unsigned long long tl_stepprof[8];
#endif
}  // namespace dsh

// one workgroup of eight wavefronts per CU (n <= 1024)
#define DSH_TL_LAYOUT8 0
namespace dsh { namespace tl_one {
#include "dsh_lu_tiled_impl.hpp"
} }
#undef DSH_TL_LAYOUT8
#undef DSH_TL_COLS
#undef DSH_TL_FUSED_STAGE
// two workgroups of four wavefronts per CU, 16-column sub-panels (n <= 512)
#define DSH_TL_LAYOUT8 1
namespace dsh { namespace tl_two {
#include "dsh_lu_tiled_impl.hpp"
} }
#undef DSH_TL_LAYOUT8
#undef DSH_TL_COLS
#undef DSH_TL_FUSED_STAGE

namespace dsh {
// Which layout factors a system of n rows.  Round 4: two workgroups per CU paid only while the trailing update was the smaller part of the work (n <= 416) — their 80 KB of
// LDS held a U12 chunk of 64 columns.  Round 5: the two-workgroup layout runs the column-dealt trailing phase (no U12 in LDS), stages its later sub-panels without the round
// trip through W and wins up to its limit of 512 rows (4096 systems, profiles/r05_lu_bench.md: n = 320 8.0 against 10.7 ms, n = 512 20.6 against 26.3).
// DSH_LU_TILED_LAYOUT=1|2 forces one.
inline int tiled_layout(int64_t n) {
  if (n > 512) return 1;
  static const int forced = [] { const char* e = getenv("DSH_LU_TILED_LAYOUT"); return e ? atoi(e) : 0; }();
  if (forced == 1 || forced == 2) return forced;
  return 2;
}
inline int tiled_threads(int64_t n) { return n > 512 ? tl_one::tiled_threads(n) : (tiled_layout(n) == 2 ? tl_two::tiled_threads(n) : tl_one::tiled_threads(n)); }
inline size_t tiled_lds_bytes(int64_t n) { return n > 512 ? tl_one::tiled_lds_bytes(n) : (tiled_layout(n) == 2 ? tl_two::tiled_lds_bytes(n) : tl_one::tiled_lds_bytes(n)); }
}  // namespace dsh
