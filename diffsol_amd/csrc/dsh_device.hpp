// Device-side helpers shared by every kernel of libdiffsol_hip.so (gfx950): bit-pattern maxima, DPP wavefront reductions, per-workgroup result
// records.  Device code only — this header is also what the run-time-compiled (hiprtc) model modules include, so nothing host-side lives here.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// loops over the n state components: fully unrolled (the state of the register-resident models lives in registers); the run-time-compiled banded form,
// whose state is in per-lane memory anyway, may keep them rolled (DSH_NOUNROLL_N, dsh_jit.hip)
#ifdef DSH_NOUNROLL_N
#define DSH_UNROLL_N _Pragma("nounroll")
#else
#define DSH_UNROLL_N _Pragma("unroll")
#endif

namespace dsh {

// Result records: kRecWords u64 per workgroup (layout in dsh_internal.hpp); kRecRegions reducing launches may be in flight before a region is reused.
constexpr int kRecWords = 4;
constexpr int kRecRegions = 8;


__device__ __forceinline__ unsigned long long d2u(double x) { return (unsigned long long)__double_as_longlong(x); }

// NaN-propagating maximum in the bit-pattern domain (inputs are squares / counts: never negative zero or negative)
// ---- An IEEE FP64 division split into its denominator half and its numerator half.
// hipcc expands x / y for gfx950 into v_div_scale(y), v_rcp, four FMAs that refine the reciprocal, v_div_scale(x), a multiply, an FMA, v_div_fmas and
// v_div_fixup: ~11 dependent instructions.  When neither operand needs scaling (v_div_scale returns its operand, VCC is clear, v_div_fmas is a plain FMA
// and v_div_fixup passes its first operand through — true whenever both magnitudes are within [2^-300, 2^300]) the first six depend on y alone.
// div_refined_rcp(y) is that half — computed once per denominator, off any dependent chain — and div_by_refined(x, y, r) the other: three dependent
// instructions that return the bits of x / y (scripts/ubench/div_split.hip: 2.6e11 pairs incl. extreme mantissas and exact / halfway quotients, no
// mismatch).  Two ways to know that a quotient made this way is the quotient:
//   * div_split_ok(x, y): both operands in range; a numerator of +0 also is (the three instructions give the right signed zero), -0 is not (they
//     return +0 for a positive denominator);
//   * after the fact, with no instruction on the chain: div_den_ok(y) (|y| within [2^-49, 2^49]) and div_quot_ok(q) (|q| within [2^-250, 2^250])
//     together imply |x| within [2^-300, 2^300] — a numerator outside of it (zero, denormal, huge, Inf, NaN) cannot produce a quotient inside.
// Whoever finds the test false divides the ordinary way.
constexpr double kDivSplitLo = 0x1p-300, kDivSplitHi = 0x1p300, kDivDenLo = 0x1p-49, kDivDenHi = 0x1p49, kDivQuotLo = 0x1p-250, kDivQuotHi = 0x1p250;
__device__ __forceinline__ double div_refined_rcp(double y) {
  double r = __builtin_amdgcn_rcp(y);
  double e = __builtin_fma(-y, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-y, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
__device__ __forceinline__ double div_by_refined(double x, double y, double r) {
  const double q = x * r;
  const double e = __builtin_fma(-y, q, x);
  return __builtin_fma(e, r, q);
}
// bitwise, not short-circuit, operators: compares and scalar mask logic, no branches
__device__ __forceinline__ bool div_split_ok(double x, double y) {
  const double a = __builtin_fabs(x), d = __builtin_fabs(y);
  return (((a >= kDivSplitLo) & (a <= kDivSplitHi)) | __builtin_amdgcn_class(x, 1 << 6)) & (d >= kDivSplitLo) & (d <= kDivSplitHi);  // class bit 6: +0
}
__device__ __forceinline__ bool div_den_ok(double y) { const double d = __builtin_fabs(y); return (d >= kDivDenLo) & (d <= kDivDenHi); }
__device__ __forceinline__ bool div_quot_ok(double q) { const double a = __builtin_fabs(q); return (a >= kDivQuotLo) & (a <= kDivQuotHi); }

__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

// Wavefront reductions without the LDS crossbar: four DPP butterfly stages inside each row of 16 lanes (quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror, row_mirror — after every stage the lanes of a 2/4/8/16-group hold the same value, so the mirrors act as xor 4 / xor 8), then the
// four row results are read with v_readlane and combined as scalars.  ~35 VALU instructions instead of six dependent ds_bpermute round trips.
// All 64 lanes must be active (every caller reduces over whole wavefronts).
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_move_u64(unsigned long long v) {
  int lo = (int)(unsigned int)v, hi = (int)(unsigned int)(v >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int lane) {
  const int lo = __builtin_amdgcn_readlane((int)(unsigned int)v, lane), hi = __builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), lane);
  return ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo;
}
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E, kDppRowHalfMirror = 0x141, kDppRowMirror = 0x140;

// maximum of a 32-bit unsigned value over the wavefront: four DPP stages inside each row of 16 lanes (v_max_u32 takes the DPP operand directly), the four
// row results combined as scalars
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, kDppQuadXor1, 0xf, 0xf, false));
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, kDppQuadXor2, 0xf, 0xf, false));
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, kDppRowHalfMirror, 0xf, 0xf, false));
  v = max(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, kDppRowMirror, 0xf, 0xf, false));
  const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned int c = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
  return max(max(a, b), max(c, d));
}
// maximum of a 64-bit pattern, high words first: the largest high word over the wavefront, then the largest low word among the lanes that hold it — the
// same value as a 64-bit compare-and-select reduction, in about half the vector instructions (a 64-bit stage is two DPP moves, a compare and two selects)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  const unsigned int hi = (unsigned int)(v >> 32), lo = (unsigned int)v;
  const unsigned int mh = wave_max_u32(hi);
  const unsigned int ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | (unsigned long long)ml;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
  v += dpp_move_u64<kDppQuadXor1>(v);
  v += dpp_move_u64<kDppQuadXor2>(v);
  v += dpp_move_u64<kDppRowHalfMirror>(v);
  v += dpp_move_u64<kDppRowMirror>(v);
  return (readlane_u64(v, 0) + readlane_u64(v, 16)) + (readlane_u64(v, 32) + readlane_u64(v, 48));
}

// (|value|, row) arg-max with smallest row on ties, reduced over the `tps` consecutive lanes of a system (tps is a power of two <= 64
// and the group is aligned inside a wave, so xor-shuffles below tps stay inside the group)
// The four stages inside a row of 16 lanes are DPP moves (no LDS crossbar, see wave_max_u64); wider groups finish with shuffles (32) or with four
// v_readlane per value (64).  All lanes of the wavefront must be active.
__device__ __forceinline__ void argmax_take(double& best, int& row, double ob, int orow) {
  if (ob > best || (ob == best && orow < row)) { best = ob; row = orow; }
}
template <int CTRL>
__device__ __forceinline__ void argmax_dpp_stage(double& best, int& row) {
  const double ob = __longlong_as_double((long long)dpp_move_u64<CTRL>((unsigned long long)__double_as_longlong(best)));
  const int orow = __builtin_amdgcn_update_dpp(row, row, CTRL, 0xf, 0xf, false);
  argmax_take(best, row, ob, orow);
}
__device__ __forceinline__ void group_argmax(double& best, int& row, int tps) {
  if (tps >= 16) {
    argmax_dpp_stage<kDppQuadXor1>(best, row);
    argmax_dpp_stage<kDppQuadXor2>(best, row);
    argmax_dpp_stage<kDppRowHalfMirror>(best, row);
    argmax_dpp_stage<kDppRowMirror>(best, row);
    if (tps == 32) {
      argmax_take(best, row, __shfl_xor(best, 16, 64), __shfl_xor(row, 16, 64));
    } else if (tps == 64) {
      double b0 = __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(best), 0));
      int r0 = __builtin_amdgcn_readlane(row, 0);
#pragma unroll
      for (int q = 1; q < 4; ++q)
        argmax_take(b0, r0, __longlong_as_double((long long)readlane_u64((unsigned long long)__double_as_longlong(best), 16 * q)), __builtin_amdgcn_readlane(row, 16 * q));
      best = b0; row = r0;
    }
    return;
  }
  for (int off = tps >> 1; off > 0; off >>= 1) argmax_take(best, row, __shfl_xor(best, off, 64), __shfl_xor(row, off, 64));
}

__device__ __forceinline__ void publish_singular(unsigned long long* word, unsigned long long count, unsigned int epoch) {
  unsigned long long old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) {
    const unsigned long long base = (old >> 32) == epoch ? (old & 0xffffffffull) : 0ull;
    const unsigned long long desired = ((unsigned long long)epoch << 32) | (base + count);
    const unsigned long long prev = atomicCAS(word, old, desired);
    if (prev == old) break;
    old = prev;
  }
}

// Block-level reduction of two maxima and one count, then ONE record (two 16-byte stores by thread 0) to host-mapped memory.
// Works for any block size that is a multiple of 64 up to 1024.  Every thread of the block must call it.
__device__ __forceinline__ void block_publish(unsigned long long m0, unsigned long long m1, unsigned long long cnt, unsigned long long* rec,
                                              unsigned int seq) {
  m0 = wave_max_u64(m0);
  m1 = wave_max_u64(m1);
  cnt = wave_sum_u64(cnt);
  const int lane = threadIdx.x & 63;
  const int nwaves = blockDim.x >> 6;
  if (nwaves > 1) {
    __shared__ unsigned long long sh[3][16];
    const int w = threadIdx.x >> 6;
    if (lane == 0) { sh[0][w] = m0; sh[1][w] = m1; sh[2][w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 1; k < nwaves; ++k) { m0 = umax64(m0, sh[0][k]); m1 = umax64(m1, sh[1][k]); cnt += sh[2][k]; }
    }
  }
  if (threadIdx.x == 0) {
    const unsigned long long tag = ((unsigned long long)seq << 32) | (cnt & 0xffffffffull);
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2* r = reinterpret_cast<u64x2*>(rec + (size_t)blockIdx.x * kRecWords);
    u64x2 g0 = {m0, tag}, g1 = {m1, tag};
    r[0] = g0;  // one global_store_dwordx4 each: {payload, tag} granules
    r[1] = g1;
  }
  if (nwaves > 1) __syncthreads();  // the LDS staging array may be reused by another reduction of the same launch
}


}  // namespace dsh
