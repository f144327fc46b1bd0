// Internal definitions shared by the HIP translation units of libdiffsol_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include <string>

#include "../../include/diffsol_hip.h"
#include "dsh_device.hpp"

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
// Records of the last kRecRegions reducing launches stay readable (launch seq uses region seq % kRecRegions), so a caller may keep
// a few reducing launches in flight and collect their results later (speculative Newton pipelining).

}  // namespace dsh

struct dsh_ctx {
  // Every extern "C" entry point that touches a context holds this for the duration of the call (DSH_ENTER below): the record ring, the scratch buffers and the
  // allocation cache need no further locks, and two host threads that use clones of one context (safe Rust can do that: HipContext / HipVec are Clone + Send) are
  // serialised call by call on the context's one in-order stream instead of racing.  Recursive: entry points call each other.
  std::recursive_mutex mu;
  std::thread::id last_thread;  // the thread whose HIP device binding is known to be this context's device (the current device is per-thread state)
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
  size_t pool_limit = (size_t)64 << 30;  // parked bytes above which dsh_free returns blocks to the runtime: a quarter of the device's memory (set by dsh_ctx_create)
  // optional HIP-event timing of the dominant (fused Newton iteration) kernel on this context's stream
  bool timing = false;
  int timing_target = 0;  // which launches the brackets go around (DSH_TIMING_*): 0 = the device-resident integrators / the fused Newton launch, 1 = dsh_lu_solve, 2 = dsh_lu_factor
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool ev_pending = false;
  double timed_ms = 0.0;
  double timed_clock_ms = 0.0;       // same launches measured with the in-kernel 100 MHz device clock (max block end - min block start)
  double bracket_overhead_ms = 0.0;  // elapsed time of an empty event bracket (calibrated when timing is enabled)
  int64_t timed_launches = 0;
  int solve_mode = 0;  // DSH_SOLVE_EXACT (default: every solve in the reference's order of operations) | DSH_SOLVE_REORDERED (opt-in: dsh_lu_band_affine.hpp where it applies)
  // constants + save points of the last device-resident solve, kept on the device between solves (dsh_adaptive.hip); freed by dsh_ctx_destroy
  unsigned char* const_cache_dev = nullptr;
  std::vector<unsigned char>* const_cache_host = nullptr;
};

struct dsh_lu {
  dsh_ctx* ctx = nullptr;
  int64_t n = 0, nbatch = 0;
  // n <= 8 (register kernels): batch-fastest, factors (j*n+i)*nbatch + b, pivots k*nbatch + b.
  // n  > 8 (cooperative kernels): system-major, factors b*n*n + j*n + i, pivots b*n + k.   pivots[k] = row swapped with row k at step k.
  double* factors = nullptr;
  int32_t* pivots = nullptr;
  double* work = nullptr;  // row-major working copies of the systems for the matrix-core kernel (dsh_lu_tiled.hpp), allocated by its first use
  bool system_major = false;
  // device word: (epoch << 32) | number of systems with a zero pivot found by the factorisation launch of that epoch.  A launch of a
  // newer epoch replaces an older word (CAS loop, only executed by waves that actually found a singular system), so no reset
  // launch is needed between factorisations.
  unsigned long long* singular = nullptr;
  unsigned int singular_epoch = 0;
  bool factored = false;
  // Banded matrices in dense containers (dsh_lu_band.hpp): structure 0 = probe the bandwidth of every operand and use the banded kernels when
  // max(kl, ku) <= 4 (results are bit-identical to the dense kernels'), 1 = always dense.  band_k = K of the current factorisation (0: dense
  // factors).  Banded factors live in `factors` too: U(r, r+d) at (d*n + r)*nbatch + b (d <= 2K), multipliers at ((2K+1+r-1)*n + j)*nbatch + b,
  // pivots batch-fastest.
  int structure = 0;
  int band_k = 0;
  // general band (dsh_lu_gband.hpp, max(kl, ku) in 5 .. 64): >= 0 when the current factors are in its system-major layout ((2 kl + ku + 1) n doubles per system, pivots
  // [b][k]); band_k is 0 then
  int gb_kl = -1, gb_ku = -1;
  double* gb_work = nullptr;  // staged system-major copy of the operand's band (k_gband_stage), gb_work_len doubles, grown on demand
  int64_t gb_work_len = 0;
  int packed_k = 0;  // > 0: a handle made by dsh_lu_create_banded — `factors` holds (3 packed_k + 1) n doubles per system and takes banded factorisations with K <= packed_k only
  int* band_probe = nullptr;
};
// allocates the factor / pivot storage of an LU handle on first use (dsh_lu.hip): every entry point that WRITES factors calls it
extern "C" __attribute__((visibility("hidden"))) int lu_ensure_storage(dsh_lu* lu);


namespace dsh {

// Scope guard of an entry point: takes the context's lock and, when the calling thread is not the one that used the context last, binds this thread's current HIP
// device to the context's (what dsh_ctx_bind_thread did on request; a context re-created at the same address on another thread is covered too).
struct ctx_guard {
  dsh_ctx* c;
  explicit ctx_guard(const dsh_ctx* cc) : c(const_cast<dsh_ctx*>(cc)) {
    if (!c) return;
    c->mu.lock();
    const std::thread::id me = std::this_thread::get_id();
    if (c->last_thread != me) { (void)hipSetDevice(c->device); c->last_thread = me; }
  }
  ~ctx_guard() { if (c) c->mu.unlock(); }
  ctx_guard(const ctx_guard&) = delete;
  ctx_guard& operator=(const ctx_guard&) = delete;
};
#define DSH_ENTER(ctxptr) ::dsh::ctx_guard _dsh_ctx_guard(ctxptr)

// Start a reducing launch of `nblocks` workgroups: makes sure the record buffer is large enough and returns the device pointer the
// kernel writes to plus the sequence tag it must stamp.
int begin_records(dsh_ctx* ctx, int64_t nblocks, unsigned long long** rec_dev, unsigned int* seq);
// Wait for the launch's records and reduce them on the host into ctx->res_m0 / res_m1 / res_cnt.
int fetch_records(dsh_ctx* ctx, int64_t nblocks, unsigned int seq, int64_t first_record = 0);
// the pieces of the staged SDIRK Newton iteration of run-time-sized models (dsh_fused.hip): each enqueues only
bool model_has_staged_newton(int model, int64_t size);
bool model_dyn_sdirk_residual(dsh_ctx* ctx, int model, int64_t size, int64_t nb, double t, double c, double h, const double* phi, const double* k, const double* p, double* out);
int lu_solve_launch(const dsh_lu* lu, double* rhs, unsigned int* gx, unsigned int* seq);
int lu_solve_norm_launch(const dsh_lu* lu, double* rhs, const double* xin, double* xout, const double* y, int64_t ynb, const double* atol, int64_t anb, double rtol,
                         unsigned int* gx, unsigned int* seq, bool* fused);
int vec_sub_squared_norm_launch(dsh_ctx* ctx, int64_t n, int64_t nb, const double* delta, const double* xin, double* xout, const double* y, int64_t ynb,
                                const double* atol, int64_t anb, double rtol, unsigned int* gx, unsigned int* seq);
int ensure_i32_scratch(dsh_ctx* ctx, int64_t len);
int ensure_f64_scratch(dsh_ctx* ctx, int64_t len);

inline double bits_to_double(unsigned long long b) {
  double d;
  __builtin_memcpy(&d, &b, sizeof(d));
  return d;
}

// HIP-event bracket around ONE launch on the context's stream (dsh_ctx_set_timing): record before / after the launch, collect after the stream
// has been synchronised.  Used by the device-resident integrators, whose single launch per ensemble solve is the whole timed region.
inline bool timing_on(const dsh_ctx* ctx, int target = 0) { return ctx->timing && ctx->timing_target == target; }
inline hipError_t timing_begin(dsh_ctx* ctx, int target = 0) { return timing_on(ctx, target) ? hipEventRecord(ctx->ev_start, ctx->stream) : hipSuccess; }
inline hipError_t timing_end(dsh_ctx* ctx, int target = 0) { return timing_on(ctx, target) ? hipEventRecord(ctx->ev_stop, ctx->stream) : hipSuccess; }
inline hipError_t timing_collect(dsh_ctx* ctx, int target = 0) {
  if (!timing_on(ctx, target)) return hipSuccess;
  hipError_t e = hipEventSynchronize(ctx->ev_stop);
  if (e != hipSuccess) return e;
  float ms = 0.f;
  e = hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop);
  if (e != hipSuccess) return e;
  ctx->timed_ms += (double)ms;
  ctx->timed_launches += 1;
  return hipSuccess;
}
// bracket around everything `f` enqueues (one kernel for the solves; staging + factor kernel for the tiled dense factorisation); timed calls are synchronous
template <class F>
inline int timed_call(dsh_ctx* ctx, int target, F&& f) {
  if (!timing_on(ctx, target)) return f();
  if (hipEventRecord(ctx->ev_start, ctx->stream) != hipSuccess) { set_error("timing: hipEventRecord failed"); return DSH_E_HIP; }
  const int rc = f();
  if (rc != DSH_OK) return rc;
  if (timing_end(ctx, target) != hipSuccess || timing_collect(ctx, target) != hipSuccess) { set_error("timing: event collection failed"); return DSH_E_HIP; }
  return rc;
}

inline dim3 grid_for(int64_t work, int block) { return dim3((unsigned)((work + block - 1) / block)); }

}  // namespace dsh
