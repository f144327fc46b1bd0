// Internal definitions shared by the HIP translation units of libdiffsol_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>

#include "../../include/diffsol_hip.h"

namespace dsh {

void set_error(const std::string& msg);

#define DSH_HIP_CHECK(expr)                                                                              \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) {                                                                              \
      ::dsh::set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                       std::to_string(__LINE__) + ")");                                                  \
      return DSH_E_HIP;                                                                                  \
    }                                                                                                    \
  } while (0)

#define DSH_REQUIRE(cond, msg)                         \
  do {                                                 \
    if (!(cond)) {                                     \
      ::dsh::set_error(std::string(__func__) + ": " + (msg)); \
      return DSH_E_INVALID;                            \
    }                                                  \
  } while (0)

// operand nbatch must be 1 (broadcast) or the context batch size
// incompatible batch sizes: the reference panics (Context::check_compatible, context/mod.rs:49-62); here a distinct error code
#define DSH_CHECK_NB(nb_op, nb)                                                              \
  do {                                                                                       \
    if (!((nb_op) == 1 || (nb_op) == (nb))) {                                                \
      ::dsh::set_error(std::string(__func__) + ": operand nbatch must be 1 or nbatch");     \
      return DSH_E_BATCH_MISMATCH;                                                           \
    }                                                                                        \
  } while (0)

// Reduction records.  Every reducing launch writes ONE 32-byte record per workgroup straight into pinned, device-mapped host memory:
//   word0 = max of f64 bit patterns (non-negative values and NaN order correctly as unsigned integers: NaN > +inf > finite)
//   word1 = (sequence << 32) | count        word2 = second max        word3 = (sequence << 32) | count
// i.e. two self-validating 16-byte granules {payload, tag}, each written by a single 16-byte store.  No atomics, no device-side
// zeroing, no D2H copy kernel: the host reduces the (few hundred) records itself, either after a stream synchronise or — in polling
// mode — as soon as every record carries the launch's sequence tag (the granules travel as single PCIe writes, so a tag that has
// arrived implies its payload has).
constexpr int kRecWords = 4;
// Records of the last kRecRegions reducing launches stay readable (launch seq uses region seq % kRecRegions), so a caller may keep
// a few reducing launches in flight and collect their results later (speculative Newton pipelining).
constexpr int kRecRegions = 8;

}  // namespace dsh

struct dsh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  int block = 64;   // threads per workgroup for one-lane-per-system kernels
  int num_cu = 256;
  unsigned long long* rec_host = nullptr;  // pinned + mapped host memory, rec_capacity * kRecWords
  unsigned long long* rec_dev = nullptr;   // device view of rec_host
  int64_t rec_capacity = 0;                // records (= max workgroups of a reducing launch)
  unsigned int seq = 0;                    // sequence tag of the last reducing launch
  bool poll = true;                        // true: spin on the record tags; false: hipStreamSynchronize then read
  unsigned long long res_m0 = 0, res_m1 = 0, res_cnt = 0;  // host-side reduction of the last fetched launch
  int32_t* i32_scratch = nullptr;          // device scratch for root finding results etc.
  int64_t i32_scratch_len = 0;
  double* f64_scratch = nullptr;
  int64_t f64_scratch_len = 0;
  // Stream-ordered allocation cache: dsh_free parks blocks here (keyed by size) and dsh_malloc reuses them.  All users of a context
  // issue work on its one in-order stream, so handing a parked block to a new owner is safe without synchronising: every kernel
  // of the old owner was enqueued before any kernel of the new one.  (hipMalloc/hipFree cost 50-200 us each and hipFree synchronises.)
  std::multimap<size_t, void*>* pool = nullptr;
  std::map<void*, size_t>* live = nullptr;
  size_t pool_bytes = 0;
  // optional HIP-event timing of the dominant (fused Newton iteration) kernel on this context's stream
  bool timing = false;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool ev_pending = false;
  double timed_ms = 0.0;
  double timed_clock_ms = 0.0;       // same launches measured with the in-kernel 100 MHz device clock (max block end - min block start)
  double bracket_overhead_ms = 0.0;  // elapsed time of an empty event bracket (calibrated when timing is enabled)
  int64_t timed_launches = 0;
};

struct dsh_lu {
  dsh_ctx* ctx = nullptr;
  int64_t n = 0, nbatch = 0;
  // n <= 8 (register kernels): batch-fastest, factors (j*n+i)*nbatch + b, pivots k*nbatch + b.
  // n  > 8 (cooperative kernels): system-major, factors b*n*n + j*n + i, pivots b*n + k.   pivots[k] = row swapped with row k at step k.
  double* factors = nullptr;
  int32_t* pivots = nullptr;
  bool system_major = false;
  // device word: (epoch << 32) | number of systems with a zero pivot found by the factorisation launch of that epoch.  A launch of a
  // newer epoch replaces an older word (CAS loop, only executed by waves that actually found a singular system), so no reset
  // launch is needed between factorisations.
  unsigned long long* singular = nullptr;
  unsigned int singular_epoch = 0;
  bool factored = false;
};

namespace dsh {

// Start a reducing launch of `nblocks` workgroups: makes sure the record buffer is large enough and returns the device pointer the
// kernel writes to plus the sequence tag it must stamp.
int begin_records(dsh_ctx* ctx, int64_t nblocks, unsigned long long** rec_dev, unsigned int* seq);
// Wait for the launch's records and reduce them on the host into ctx->res_m0 / res_m1 / res_cnt.
int fetch_records(dsh_ctx* ctx, int64_t nblocks, unsigned int seq, int64_t first_record = 0);
int ensure_i32_scratch(dsh_ctx* ctx, int64_t len);
int ensure_f64_scratch(dsh_ctx* ctx, int64_t len);

inline double bits_to_double(unsigned long long b) {
  double d;
  __builtin_memcpy(&d, &b, sizeof(d));
  return d;
}

inline dim3 grid_for(int64_t work, int block) { return dim3((unsigned)((work + block - 1) / block)); }

// ---------------------------------------------------------------- device helpers
#if defined(__HIPCC__)

__device__ __forceinline__ unsigned long long d2u(double x) { return (unsigned long long)__double_as_longlong(x); }

// NaN-propagating maximum in the bit-pattern domain (inputs are squares / counts: never negative zero or negative)
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

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = umax64(v, dpp_move_u64<kDppQuadXor1>(v));
  v = umax64(v, dpp_move_u64<kDppQuadXor2>(v));
  v = umax64(v, dpp_move_u64<kDppRowHalfMirror>(v));
  v = umax64(v, dpp_move_u64<kDppRowMirror>(v));
  return umax64(umax64(readlane_u64(v, 0), readlane_u64(v, 16)), umax64(readlane_u64(v, 32), readlane_u64(v, 48)));
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

#endif  // __HIPCC__

}  // namespace dsh
