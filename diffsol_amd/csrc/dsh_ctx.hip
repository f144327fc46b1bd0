// Context, device memory and layout conversion for libdiffsol_hip.so (gfx950).
// Replaces CudaContext (diffsol-la/src/context/cuda.rs:41-144) and the cudarc driver calls listed in SURVEY §2a.
#include "dsh_internal.hpp"

#include <algorithm>
#include <cstdlib>

#include <cstring>
#include <mutex>

namespace dsh {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

int begin_records(dsh_ctx* ctx, int64_t nblocks, unsigned long long** rec_dev, unsigned int* seq) {
  if (nblocks > ctx->rec_capacity) {
    int64_t cap = ctx->rec_capacity > 0 ? ctx->rec_capacity : 1024;
    while (cap < nblocks) cap *= 2;
    // Launches already enqueued may still owe their records (the staged Newton iteration enqueues the solve, then the norm, and redeems both afterwards): after
    // the synchronize they are all in the old ring, so the ring grows by COPYING every region to the same region of the new one — a later fetch_records of an
    // older sequence number finds its records where it expects them (same region, same record index).
    DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    unsigned long long* grown = nullptr;
    DSH_HIP_CHECK(hipHostMalloc((void**)&grown, sizeof(unsigned long long) * kRecWords * cap * kRecRegions, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(grown, 0, sizeof(unsigned long long) * kRecWords * cap * kRecRegions);
    if (ctx->rec_host) {
      for (int64_t r = 0; r < (int64_t)kRecRegions; ++r)
        std::memcpy(grown + (size_t)r * cap * kRecWords, ctx->rec_host + (size_t)r * ctx->rec_capacity * kRecWords, sizeof(unsigned long long) * kRecWords * ctx->rec_capacity);
      DSH_HIP_CHECK(hipHostFree(ctx->rec_host));
    }
    ctx->rec_host = grown;
    DSH_HIP_CHECK(hipHostGetDevicePointer((void**)&ctx->rec_dev, ctx->rec_host, 0));
    ctx->rec_capacity = cap;
  }
  ctx->seq += 1;
  if (ctx->seq == 0) ctx->seq = 1;  // tag 0 is "never written"
  *rec_dev = ctx->rec_dev + (size_t)(ctx->seq % kRecRegions) * ctx->rec_capacity * kRecWords;
  *seq = ctx->seq;
  return DSH_OK;
}

int fetch_records(dsh_ctx* ctx, int64_t nblocks, unsigned int seq, int64_t first_record) {
  volatile unsigned long long* rec = ctx->rec_host + ((size_t)(seq % kRecRegions) * ctx->rec_capacity + (size_t)first_record) * kRecWords;
  if (!ctx->poll) DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  unsigned long long m0 = 0, m1 = 0, cnt = 0;
  const unsigned long long want = (unsigned long long)seq;
  long spins = 0;
  for (int64_t b = 0; b < nblocks; ++b) {
    volatile unsigned long long* r = rec + (size_t)b * kRecWords;
    unsigned long long t0, t1;
    while (((t0 = r[1]) >> 32) != want || ((t1 = r[3]) >> 32) != want) {
      // a launch that failed never tags its records: fall back to the runtime to surface the error instead of spinning forever
      if (++spins > 2000000) {
        hipError_t e = hipStreamQuery(ctx->stream);
        if (e != hipSuccess && e != hipErrorNotReady) { DSH_HIP_CHECK(e); }
        if (e == hipSuccess && ((r[1] >> 32) != want || (r[3] >> 32) != want)) {
          set_error("reduction records were not written by the kernel (stream idle)");
          return DSH_E_HIP;
        }
        spins = 0;
      }
      __builtin_ia32_pause();
    }
    unsigned long long a = r[0], c = r[2];
    if (a > m0) m0 = a;
    if (c > m1) m1 = c;
    cnt += (t0 & 0xffffffffull);
  }
  ctx->res_m0 = m0; ctx->res_m1 = m1; ctx->res_cnt = cnt;
  return DSH_OK;
}

int ensure_i32_scratch(dsh_ctx* ctx, int64_t len) {
  if (ctx->i32_scratch_len >= len) return DSH_OK;
  if (ctx->i32_scratch) { DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream)); DSH_HIP_CHECK(hipFree(ctx->i32_scratch)); }
  DSH_HIP_CHECK(hipMalloc((void**)&ctx->i32_scratch, sizeof(int32_t) * len));
  ctx->i32_scratch_len = len;
  return DSH_OK;
}
int ensure_f64_scratch(dsh_ctx* ctx, int64_t len) {
  if (ctx->f64_scratch_len >= len) return DSH_OK;
  if (ctx->f64_scratch) { DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream)); DSH_HIP_CHECK(hipFree(ctx->f64_scratch)); }
  DSH_HIP_CHECK(hipMalloc((void**)&ctx->f64_scratch, sizeof(double) * len));
  ctx->f64_scratch_len = len;
  return DSH_OK;
}
}  // namespace dsh

using namespace dsh;

// [b][i] (host order) <-> [i][b] (device order) through a padded LDS tile so both sides stay coalesced
__global__ void k_transpose(const double* __restrict__ src, double* __restrict__ dst, int64_t rows, int64_t cols, int64_t row_block0) {
  // src is rows x cols row-major; dst is cols x rows row-major
  __shared__ double tile[32][33];
  int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (row_block0 + (int64_t)blockIdx.y) * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int k = ty; k < 32; k += 8) {
    int64_t r = r0 + k, c = c0 + tx;
    if (r < rows && c < cols) tile[k][tx] = src[r * cols + c];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    int64_t c = c0 + k, r = r0 + tx;
    if (r < rows && c < cols) dst[c * rows + r] = tile[tx][k];
  }
}

static int launch_transpose(dsh_ctx* ctx, const double* src, double* dst, int64_t rows, int64_t cols) {
  if (rows == 0 || cols == 0) return DSH_OK;
  // grid.y holds at most 65535 row tiles: ensembles beyond 2.09M members (rows = nbatch on upload) go in several launches
  const int64_t row_tiles = (rows + 31) / 32, col_tiles = (cols + 31) / 32;
  DSH_REQUIRE(col_tiles <= 2147483647LL, "transpose: too many columns");
  for (int64_t rb = 0; rb < row_tiles; rb += 65535) {
    dim3 grid((unsigned)col_tiles, (unsigned)std::min<int64_t>(65535, row_tiles - rb));
    hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, ctx->stream, src, dst, rows, cols, rb);
  }
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

namespace {
template <class T>
__global__ void k_permute_members(int64_t rows, int64_t nb, const T* __restrict__ src, const int32_t* __restrict__ idx, T* __restrict__ dst) {
  const int64_t total = rows * nb;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / nb, b = e % nb;
    dst[e] = src[r * nb + idx[b]];
  }
}
}  // namespace

extern "C" {

const char* dsh_last_error(void) { return g_err.c_str(); }
int dsh_version(void) { return 1; }

int dsh_ctx_create(int device, void* stream, dsh_ctx** out) {
  DSH_REQUIRE(out != nullptr, "out is null");
  int count = 0;
  DSH_HIP_CHECK(hipGetDeviceCount(&count));
  DSH_REQUIRE(device >= 0 && device < count, "no such HIP device (a gfx950 GPU is required; there is no CPU fallback)");
  {
    // One process per GPU: launches do not switch devices, and run-time-compiled modules / function attributes are loaded once per
    // process.  A second context on ANOTHER device in the same process would launch on the wrong device: refuse it instead.
    static std::mutex mu;
    static int first_device = -1;
    std::lock_guard<std::mutex> lk(mu);
    if (first_device < 0) first_device = device;
    if (device != first_device) {
      set_error("dsh_ctx_create: this process already uses HIP device " + std::to_string(first_device) + "; the backend runs one process per GPU (use one rank per device)");
      return DSH_E_UNSUPPORTED;
    }
  }
  DSH_HIP_CHECK(hipSetDevice(device));
  dsh_ctx* ctx = new dsh_ctx();
  ctx->device = device;
  ctx->last_thread = std::this_thread::get_id();
  if (stream) { ctx->stream = (hipStream_t)stream; ctx->owns_stream = false; }
  else { DSH_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)); ctx->owns_stream = true; }
  hipDeviceProp_t prop;
  DSH_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  ctx->num_cu = prop.multiProcessorCount;
  ctx->pool_limit = prop.totalGlobalMem / 4;
  ctx->pool = new std::multimap<size_t, void*>();
  ctx->live = new std::map<void*, size_t>();
  {
    // "poll" or "sync".  Default: poll for a single process (a spinning host core buys ~10 us per reduction), sync when the process is one of several
    // ranks (WORLD_SIZE > 1: one process per GPU — eight spinning cores on a node help nobody)
    const char* env = std::getenv("DSH_SYNC_MODE");
    const char* ws = std::getenv("WORLD_SIZE");
    const bool multi = ws && std::atoi(ws) > 1;
    ctx->poll = env ? std::string(env) != "sync" : !multi;
  }
  *out = ctx;
  return DSH_OK;
}

void dsh_ctx_destroy(dsh_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->pool) { for (auto& kv : *ctx->pool) (void)hipFree(kv.second); delete ctx->pool; }
  if (ctx->live) { for (auto& kv : *ctx->live) (void)hipFree(kv.first); delete ctx->live; }
  if (ctx->rec_host) (void)hipHostFree(ctx->rec_host);
  if (ctx->i32_scratch) (void)hipFree(ctx->i32_scratch);
  if (ctx->f64_scratch) (void)hipFree(ctx->f64_scratch);
  if (ctx->const_cache_dev) (void)hipFree(ctx->const_cache_dev);
  delete ctx->const_cache_host;
  if (ctx->ev_start) { (void)hipEventDestroy(ctx->ev_start); (void)hipEventDestroy(ctx->ev_stop); }
  if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// A context (and everything created from it) may move between host threads.  Every entry point takes the context's lock (DSH_ENTER) and re-binds the calling
// thread's current HIP device when the thread changed, so concurrent callers are serialised and this call is only needed by code that issues its own HIP calls
// on the context's stream from a new thread.
int dsh_ctx_bind_thread(dsh_ctx* ctx) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(ctx != nullptr, "null context");
  DSH_HIP_CHECK(hipSetDevice(ctx->device));
  return DSH_OK;
}

int dsh_ctx_sync(dsh_ctx* ctx) {
  DSH_ENTER(ctx);
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return DSH_OK;
}
void* dsh_ctx_stream(dsh_ctx* ctx) { return (void*)ctx->stream; }
int dsh_ctx_device(dsh_ctx* ctx) { return ctx->device; }
int dsh_ctx_set_block(dsh_ctx* ctx, int threads) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(threads >= 64 && threads <= 1024 && (threads & (threads - 1)) == 0, "block must be a power of two in [64,1024]");
  ctx->block = threads;
  return DSH_OK;
}

int dsh_ctx_set_timing(dsh_ctx* ctx, int enable) {
  DSH_ENTER(ctx);
  if (enable && !ctx->ev_start) {
    DSH_HIP_CHECK(hipEventCreate(&ctx->ev_start));
    DSH_HIP_CHECK(hipEventCreate(&ctx->ev_stop));
  }
  ctx->timing = enable != 0;  // note: event timing needs completed events, so timed launches synchronise the stream
  ctx->ev_pending = false;
  ctx->timed_ms = 0.0;
  ctx->timed_clock_ms = 0.0;
  ctx->timed_launches = 0;
  if (enable) {
    // calibrate the cost of the bracket itself: elapsed time between two back-to-back event records with nothing in between
    DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double acc = 0.0;
    const int reps = 200;
    for (int i = 0; i < reps; ++i) {
      DSH_HIP_CHECK(hipEventRecord(ctx->ev_start, ctx->stream));
      DSH_HIP_CHECK(hipEventRecord(ctx->ev_stop, ctx->stream));
      DSH_HIP_CHECK(hipEventSynchronize(ctx->ev_stop));
      float ms = 0.f;
      DSH_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
      acc += (double)ms;
    }
    ctx->bracket_overhead_ms = acc / reps;
  }
  return DSH_OK;
}
int dsh_ctx_set_solve_mode(dsh_ctx* ctx, int mode) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(ctx != nullptr && (mode == DSH_SOLVE_EXACT || mode == DSH_SOLVE_REORDERED), "dsh_ctx_set_solve_mode: unknown mode");
  ctx->solve_mode = mode;
  return DSH_OK;
}
int dsh_ctx_get_solve_mode(const dsh_ctx* ctx) { return ctx ? ctx->solve_mode : -1; }
int dsh_ctx_set_timing_target(dsh_ctx* ctx, int target) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(target >= DSH_TIMING_RESIDENT && target <= DSH_TIMING_LU_FACTOR, "dsh_ctx_set_timing_target: unknown target");
  ctx->timing_target = target;
  ctx->timed_ms = 0.0;
  ctx->timed_clock_ms = 0.0;
  ctx->timed_launches = 0;
  return DSH_OK;
}
int dsh_ctx_get_timing_overhead(dsh_ctx* ctx, double* empty_bracket_ms, double* device_clock_total_ms) {
  DSH_ENTER(ctx);
  if (empty_bracket_ms) *empty_bracket_ms = ctx->bracket_overhead_ms;
  if (device_clock_total_ms) *device_clock_total_ms = ctx->timed_clock_ms;
  return DSH_OK;
}
int dsh_ctx_set_poll(dsh_ctx* ctx, int poll) {
  DSH_ENTER(ctx);
  ctx->poll = poll != 0;
  return DSH_OK;
}
int dsh_ctx_get_timing(dsh_ctx* ctx, int64_t* launches, double* total_ms) {
  DSH_ENTER(ctx);
  if (launches) *launches = ctx->timed_launches;
  if (total_ms) *total_ms = ctx->timed_ms;
  return DSH_OK;
}

int dsh_malloc(dsh_ctx* ctx, int64_t nbytes, int zero, void** out) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(nbytes >= 0 && out, "bad arguments");
  const size_t want = nbytes > 0 ? (size_t)nbytes : 8;
  void* p = nullptr;
  auto it = ctx->pool->find(want);
  if (it != ctx->pool->end()) {
    p = it->second;
    ctx->pool->erase(it);
    ctx->pool_bytes -= want;
  } else {
    DSH_HIP_CHECK(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess && !ctx->pool->empty()) {
      // out of memory with blocks still parked (a sweep over ensemble sizes leaves dead sizes behind — the cache only reuses exact matches): give every parked
      // block back to the runtime and try once more.  The parked blocks may still be read by work in flight, hence the synchronise.
      (void)hipGetLastError();
      DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      for (auto& kv : *ctx->pool) (void)hipFree(kv.second);
      ctx->pool->clear();
      ctx->pool_bytes = 0;
      e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      set_error("dsh_malloc: hipMalloc of " + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
      return DSH_E_HIP;
    }
  }
  (*ctx->live)[p] = want;
  if (zero && nbytes > 0) DSH_HIP_CHECK(hipMemsetAsync(p, 0, (size_t)nbytes, ctx->stream));
  *out = p;
  return DSH_OK;
}
int dsh_free(dsh_ctx* ctx, void* p) {
  DSH_ENTER(ctx);
  if (!p) return DSH_OK;
  auto it = ctx->live->find(p);
  if (it == ctx->live->end()) {  // not ours (or already freed): fall back to the runtime
    DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    DSH_HIP_CHECK(hipFree(p));
    return DSH_OK;
  }
  const size_t sz = it->second;
  ctx->live->erase(it);
  if (ctx->pool_bytes + sz > ctx->pool_limit) {  // a quarter of the device's memory (72 GB of the 288 GB of an MI355X) may stay parked per context
    DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    DSH_HIP_CHECK(hipFree(p));
    return DSH_OK;
  }
  ctx->pool->emplace(sz, p);
  ctx->pool_bytes += sz;
  return DSH_OK;
}
int dsh_memset_zero(dsh_ctx* ctx, void* p, int64_t nbytes) {
  DSH_ENTER(ctx);
  if (nbytes > 0) DSH_HIP_CHECK(hipMemsetAsync(p, 0, (size_t)nbytes, ctx->stream));
  return DSH_OK;
}
int dsh_h2d(dsh_ctx* ctx, void* dst, const void* src, int64_t nbytes) {
  DSH_ENTER(ctx);
  if (nbytes <= 0) return DSH_OK;
  DSH_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream));
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return DSH_OK;
}
int dsh_d2h(dsh_ctx* ctx, void* dst, const void* src, int64_t nbytes) {
  DSH_ENTER(ctx);
  if (nbytes <= 0) return DSH_OK;
  DSH_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream));
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return DSH_OK;
}
int dsh_d2d(dsh_ctx* ctx, void* dst, const void* src, int64_t nbytes) {
  DSH_ENTER(ctx);
  if (nbytes <= 0) return DSH_OK;
  DSH_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, ctx->stream));
  return DSH_OK;
}

int dsh_vec_upload(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* host, double* dev) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(n >= 0 && nbatch >= 1, "bad shape");
  if (n == 0) return DSH_OK;
  if (nbatch == 1 || n == 1) return dsh_h2d(ctx, dev, host, sizeof(double) * n * nbatch);
  DSH_REQUIRE(ensure_f64_scratch(ctx, n * nbatch) == DSH_OK, "scratch allocation failed");
  DSH_HIP_CHECK(hipMemcpyAsync(ctx->f64_scratch, host, sizeof(double) * n * nbatch, hipMemcpyHostToDevice, ctx->stream));
  int rc = launch_transpose(ctx, ctx->f64_scratch, dev, nbatch, n);  // [b][i] -> [i][b]
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return DSH_OK;
}
int dsh_vec_download(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* dev, double* host) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(n >= 0 && nbatch >= 1, "bad shape");
  if (n == 0) return DSH_OK;
  if (nbatch == 1 || n == 1) return dsh_d2h(ctx, host, dev, sizeof(double) * n * nbatch);
  DSH_REQUIRE(ensure_f64_scratch(ctx, n * nbatch) == DSH_OK, "scratch allocation failed");
  int rc = launch_transpose(ctx, dev, ctx->f64_scratch, n, nbatch);  // [i][b] -> [b][i]
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipMemcpyAsync(host, ctx->f64_scratch, sizeof(double) * n * nbatch, hipMemcpyDeviceToHost, ctx->stream));
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return DSH_OK;
}

int dsh_vec_get_index(dsh_ctx* ctx, int64_t nbatch, const double* v, int64_t i, int64_t b, double* out) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(b >= 0 && b < nbatch && i >= 0, "index out of range");
  return dsh_d2h(ctx, out, v + i * nbatch + b, sizeof(double));
}
int dsh_vec_set_index(dsh_ctx* ctx, int64_t nbatch, double* v, int64_t i, int64_t b, double value) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(b >= 0 && b < nbatch && i >= 0, "index out of range");
  return dsh_h2d(ctx, v + i * nbatch + b, &value, sizeof(double));
}
// One member of a batched vector as a contiguous vector with nbatch = 1 and back (Vector::get_batch / get_batch_mut, vector/mod.rs:227-231): with the
// batch-fastest layout member b is the stride-nbatch slice v[i * nbatch + b]; a stream-ordered strided copy, no kernel.
int dsh_vec_extract_batch(dsh_ctx* ctx, int64_t n, int64_t nbatch, const double* v, int64_t b, double* dst) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(ctx != nullptr && b >= 0 && b < nbatch && n >= 0, "batch index out of range");
  if (n == 0) return DSH_OK;
  DSH_HIP_CHECK(hipMemcpy2DAsync(dst, sizeof(double), v + b, sizeof(double) * (size_t)nbatch, sizeof(double), (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  return DSH_OK;
}
// Members of a batched array re-ordered: dst[r * nbatch + b] = src[r * nbatch + idx[b]] for every row r (rows of elem_bytes = 4 or 8).  The per-member
// device-resident solves run the ensemble sorted by parameters (members with similar parameters take similar paths: a wavefront of neighbours diverges
// less) and hand results back in the caller's order with the inverse permutation.
int dsh_permute_members(dsh_ctx* ctx, int64_t rows, int64_t nbatch, int elem_bytes, const void* src, const int32_t* idx_dev, void* dst) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(ctx != nullptr && rows >= 0 && nbatch >= 1 && (elem_bytes == 4 || elem_bytes == 8) && src != dst, "dsh_permute_members: bad argument");
  if (rows == 0) return DSH_OK;
  const int64_t total = rows * nbatch;
  const dim3 g((unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20)), blk(256);
  if (elem_bytes == 8) hipLaunchKernelGGL((k_permute_members<unsigned long long>), g, blk, 0, ctx->stream, rows, nbatch, (const unsigned long long*)src, idx_dev, (unsigned long long*)dst);
  else hipLaunchKernelGGL((k_permute_members<unsigned int>), g, blk, 0, ctx->stream, rows, nbatch, (const unsigned int*)src, idx_dev, (unsigned int*)dst);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_vec_insert_batch(dsh_ctx* ctx, int64_t n, int64_t nbatch, double* v, int64_t b, const double* src) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(ctx != nullptr && b >= 0 && b < nbatch && n >= 0, "batch index out of range");
  if (n == 0) return DSH_OK;
  DSH_HIP_CHECK(hipMemcpy2DAsync(v + b, sizeof(double) * (size_t)nbatch, src, sizeof(double), sizeof(double), (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  return DSH_OK;
}

}  // extern "C"
