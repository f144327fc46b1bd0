// Internal definitions shared by the HIP translation units of libdiffsol_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
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
#define DSH_CHECK_NB(nb_op, nb) DSH_REQUIRE((nb_op) == 1 || (nb_op) == (nb), "operand nbatch must be 1 or nbatch")

// Reduction slots: each reducing launch gets 4 zero-initialised 64-bit words
//   [0] max of f64 bit patterns (non-negative values and NaN order correctly as unsigned integers: NaN > +inf > finite)
//   [1] second max   [2] counter   [3] flags
// taken from a ring in device memory that is re-zeroed (stream-ordered memset) each time it wraps; results come back to
// the host through one 32-byte D2H copy into pinned memory + a stream synchronise.
constexpr int kSlotWords = 4;
constexpr int kRingEntries = 512;

}  // namespace dsh

struct dsh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  int block = 64;   // threads per workgroup for one-lane-per-system kernels
  int num_cu = 256;
  unsigned long long* ring = nullptr;      // device, kRingEntries * kSlotWords
  unsigned long long* mailbox = nullptr;   // pinned host, kSlotWords (+ spare)
  int ring_cursor = 0;
  int32_t* i32_scratch = nullptr;          // device scratch for root finding results etc.
  int64_t i32_scratch_len = 0;
  double* f64_scratch = nullptr;
  int64_t f64_scratch_len = 0;
  // optional HIP-event timing of the dominant (fused Newton iteration) kernel on this context's stream
  bool timing = false;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool ev_pending = false;
  double timed_ms = 0.0;
  int64_t timed_launches = 0;
};

struct dsh_lu {
  dsh_ctx* ctx = nullptr;
  int64_t n = 0, nbatch = 0;
  double* factors = nullptr;   // (j*n+i)*nbatch + b
  int32_t* pivots = nullptr;   // k*nbatch + b : row swapped with row k at elimination step k
  unsigned long long* singular = nullptr;  // device counter of systems with a zero pivot in the last factorisation
  bool factored = false;
};

namespace dsh {

// next zeroed slot group (device pointer); enqueues a memset when the ring wraps
int take_slots(dsh_ctx* ctx, unsigned long long** out);
// copy a slot group to the host mailbox and wait; afterwards ctx->mailbox[0..4) is valid
int fetch_slots(dsh_ctx* ctx, const unsigned long long* slots);
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

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(v, off, 64);
    v = umax64(v, o);
  }
  return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// publish a per-wave maximum: skip the atomic when the slot already holds a value >= ours (device-scope relaxed load first;
// the slot only ever grows, so a stale read can only cause a redundant atomic, never a lost update)
__device__ __forceinline__ void publish_max(unsigned long long* slot, unsigned long long v) {
  if (v == 0ull) return;
  unsigned long long cur = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (v > cur) atomicMax(slot, v);
}

// Block-level reduction of up to 3 quantities (two maxima + one count) and publication to the slot group.
// Works for any block size that is a multiple of 64 up to 1024.
__device__ __forceinline__ void block_publish(unsigned long long m0, unsigned long long m1, unsigned long long cnt, unsigned long long* slots,
                                              bool use_m1, bool use_cnt) {
  m0 = wave_max_u64(m0);
  if (use_m1) m1 = wave_max_u64(m1);
  if (use_cnt) cnt = wave_sum_u64(cnt);
  const int lane = threadIdx.x & 63;
  const int nwaves = blockDim.x >> 6;
  if (nwaves == 1) {
    if (lane == 0) {
      publish_max(slots + 0, m0);
      if (use_m1) publish_max(slots + 1, m1);
      if (use_cnt && cnt) atomicAdd(slots + 2, cnt);
    }
    return;
  }
  __shared__ unsigned long long sh[3][16];
  const int w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = m0; sh[1][w] = m1; sh[2][w] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long a = sh[0][0], b = sh[1][0], c = sh[2][0];
    for (int k = 1; k < nwaves; ++k) { a = umax64(a, sh[0][k]); b = umax64(b, sh[1][k]); c += sh[2][k]; }
    publish_max(slots + 0, a);
    if (use_m1) publish_max(slots + 1, b);
    if (use_cnt && c) atomicAdd(slots + 2, c);
  }
}

#endif  // __HIPCC__

}  // namespace dsh
