// Multi-GPU trajectory collection at the C ABI: one process per GPU, every rank integrates the contiguous shard
// [lo, hi) = dsh_dist_shard_bounds(n_total, rank, world) of the ensemble with NO collective inside the integration, and the solve_dense output is gathered
// along the batch axis once — one ncclAllGather over RCCL / xGMI.  The reference has no distributed layer; this is what BASELINE.json's
// configs[3] ("sharded over 8 MI355X with RCCL gather") asks of the backend, offered below the Rust shim / diffsol_c_hip.h so that callers shard without Python.
//
// librccl is bound at RUN time (dlopen): a process that already carries an RCCL (PyTorch ships its own) keeps exactly one copy — the loaded one is reused —
// and a single-GPU user of libdiffsol_hip.so never needs the library at all.  Only the types come from <rccl/rccl.h>.
//
// Layout: the device buffers are batch-fastest, [lead][nb_local] (lead = save points x states).  Shards differ in size by at most one member: every rank sends
// [lead][m], m = ceil(n_total / world) (k_pack pads; even shards send the solver's buffer itself), the all-gather delivers [world][lead][m], and k_unpack writes
// [lead][n_total] dropping the padding.  The collective and both copies run on the communicator's OWN stream behind an event of the solver's stream, so the next
// solve of this rank (on the solver's stream) overlaps the transfer; dsh_gather_wait joins.
#include "dsh_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

using namespace dsh;

namespace {
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // an RCCL that is already mapped (e.g. PyTorch's) first, then the system one
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) { api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (api.handle) break; }
    if (!api.handle) {
      if (const char* e = std::getenv("DSH_RCCL_LIB")) api.handle = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
      for (const char* nm : names) { if (api.handle) break; api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); }
    }
    if (!api.handle) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
  });
  return api;
}
int rccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return DSH_OK;
  set_error(std::string(what) + " failed: " + rccl().GetErrorString(r));
  return DSH_E_HIP;
}

// local [lead][nl] -> send [lead][m] (columns nl..m-1 zero)
__global__ void k_pack(const double* __restrict__ local, double* __restrict__ send, int64_t lead, int64_t nl, int64_t m) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t l = blockIdx.y;
  if (k < m && l < lead) send[l * m + k] = k < nl ? local[l * nl + k] : 0.0;
}
// recv [world][lead][m] -> out [lead][n_total]: member g of rank r's shard sits at column lo(r) + g
__global__ void k_unpack(const double* __restrict__ recv, double* __restrict__ out, int64_t lead, int64_t n_total, int world, int64_t m) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // column of `out`
  const int64_t l = blockIdx.y;
  if (col >= n_total || l >= lead) return;
  const int64_t base = n_total / world, rem = n_total % world;
  // shards 0..rem-1 hold base + 1 members, the others base (dsh_dist_shard_bounds)
  const int64_t split = rem * (base + 1);
  int64_t r, g;
  if (col < split) { r = col / (base + 1); g = col - r * (base + 1); }
  else { r = base > 0 ? rem + (col - split) / base : 0; g = base > 0 ? (col - split) % base : 0; }
  out[l * n_total + col] = recv[((int64_t)r * lead + l) * m + g];
}
}  // namespace

struct dsh_dist {
  dsh_ctx* ctx = nullptr;   // used while gathering (the solver's stream); never touched by dsh_dist_destroy — the context may be gone by then
  int device = 0;           // the context's device, cached at init
  int rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;   // the collective's own stream
  hipEvent_t ready = nullptr;     // "the solver's stream has produced the shard"
  hipEvent_t done = nullptr;      // "the gathered trajectories are in `out`"
  double *send = nullptr, *recv = nullptr;
  size_t send_len = 0, recv_len = 0;
  bool pending = false;
};

extern "C" {

int dsh_dist_shard_bounds(int64_t n_total, int rank, int world, int64_t* lo, int64_t* hi) {
  if (n_total < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) { set_error("dsh_dist_shard_bounds: bad arguments"); return DSH_E_INVALID; }
  const int64_t base = n_total / world, rem = n_total % world;
  *lo = rank * base + std::min<int64_t>(rank, rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
  return DSH_OK;
}

int dsh_dist_unique_id(unsigned char* id128) {
  if (!id128) { set_error("dsh_dist_unique_id: null argument"); return DSH_E_INVALID; }
  if (!rccl().ok) { set_error("dsh_dist: librccl could not be loaded (DSH_RCCL_LIB names it explicitly)"); return DSH_E_UNSUPPORTED; }
  ncclUniqueId id;
  int rc = rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  if (rc != DSH_OK) return rc;
  static_assert(sizeof(id) == DSH_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id128, &id, sizeof(id));
  return DSH_OK;
}

int dsh_dist_init(dsh_ctx* ctx, int rank, int world, const unsigned char* id128, dsh_dist** out) {
  DSH_ENTER(ctx);
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world || !id128) { set_error("dsh_dist_init: bad arguments"); return DSH_E_INVALID; }
  if (!rccl().ok) { set_error("dsh_dist: librccl could not be loaded (DSH_RCCL_LIB names it explicitly)"); return DSH_E_UNSUPPORTED; }
  DSH_HIP_CHECK(hipSetDevice(ctx->device));
  dsh_dist* d = new dsh_dist();
  d->ctx = ctx; d->device = ctx->device; d->rank = rank; d->world = world;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  int rc = rccl_check(rccl().CommInitRank(&d->comm, world, id, rank), "ncclCommInitRank");
  if (rc != DSH_OK) { delete d; return rc; }
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&d->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&d->done, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    set_error("dsh_dist_init: could not create the communication stream");
    dsh_dist_destroy(d);
    return DSH_E_HIP;
  }
  *out = d;
  return DSH_OK;
}

void dsh_dist_destroy(dsh_dist* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->comm) (void)rccl().CommDestroy(d->comm);
  (void)hipFree(d->send);
  (void)hipFree(d->recv);
  if (d->ready) (void)hipEventDestroy(d->ready);
  if (d->done) (void)hipEventDestroy(d->done);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

int dsh_dist_rank(const dsh_dist* d) { return d ? d->rank : -1; }
int dsh_dist_world(const dsh_dist* d) { return d ? d->world : -1; }

// the two copies on their own (also what the GPU tier drives with synthetic buffers for world sizes the box does not have)
int dsh_dist_pack_shard(dsh_ctx* ctx, void* stream, const double* local, int64_t lead, int64_t nb_local, int64_t m, double* send) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(ctx && (local || nb_local == 0) && send && lead >= 0 && nb_local >= 0 && m >= nb_local, "dsh_dist_pack_shard: bad arguments");  // an empty shard (more ranks than members) has no buffer
  if (lead == 0 || m == 0) return DSH_OK;
  DSH_REQUIRE(lead <= 65535, "dsh_dist_pack_shard: more than 65535 rows (save points x states) per call");
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((m + 255) / 256), (unsigned)lead), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream, local, send, lead, nb_local, m);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_dist_unpack_gathered(dsh_ctx* ctx, void* stream, const double* recv, int64_t lead, int64_t n_total, int world, double* out) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(ctx && recv && out && lead >= 0 && n_total >= 0 && world >= 1, "dsh_dist_unpack_gathered: bad arguments");
  if (lead == 0 || n_total == 0) return DSH_OK;
  DSH_REQUIRE(lead <= 65535, "dsh_dist_unpack_gathered: more than 65535 rows (save points x states) per call");
  const int64_t m = (n_total + world - 1) / world;
  hipLaunchKernelGGL(k_unpack, dim3((unsigned)((n_total + 255) / 256), (unsigned)lead), dim3(256), 0, stream ? (hipStream_t)stream : ctx->stream, recv, out, lead, n_total, world, m);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

// All-gather of this rank's [lead][hi - lo] along the batch axis into out [lead][n_total] (every rank gets all of it), issued behind everything the solver's
// stream holds at the time of the call and run on the communicator's stream: returns at once.  `local` and `out` must stay untouched until dsh_gather_wait.
int dsh_gather_batch_axis_async(dsh_dist* d, const double* local, int64_t lead, int64_t n_total, double* out) {
  DSH_ENTER(d ? d->ctx : nullptr);
  if (!d || !out || lead < 0 || n_total < 0) { set_error("dsh_gather_batch_axis: bad arguments"); return DSH_E_INVALID; }
  if (d->pending) { set_error("dsh_gather_batch_axis_async: the previous gather of this communicator has not been waited for (dsh_gather_wait)"); return DSH_E_INVALID; }
  int64_t lo = 0, hi = 0;
  int rc = dsh_dist_shard_bounds(n_total, d->rank, d->world, &lo, &hi);
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipSetDevice(d->ctx->device));
  const int64_t m = (n_total + d->world - 1) / d->world, nl = hi - lo;
  if (!local && nl > 0) { set_error("dsh_gather_batch_axis: null shard buffer"); return DSH_E_INVALID; }
  const size_t send_len = (size_t)(lead * m), recv_len = send_len * (size_t)d->world;
  DSH_HIP_CHECK(hipEventRecord(d->ready, d->ctx->stream));
  DSH_HIP_CHECK(hipStreamWaitEvent(d->stream, d->ready, 0));
  if (send_len == 0) { DSH_HIP_CHECK(hipEventRecord(d->done, d->stream)); d->pending = true; return DSH_OK; }
  if (recv_len > d->recv_len) {
    DSH_HIP_CHECK(hipStreamSynchronize(d->stream));
    (void)hipFree(d->recv); d->recv = nullptr; d->recv_len = 0;
    DSH_HIP_CHECK(hipMalloc((void**)&d->recv, sizeof(double) * recv_len));
    d->recv_len = recv_len;
  }
  const double* src = local;
  if (nl != m) {  // the shorter shards pad
    if (send_len > d->send_len) {
      DSH_HIP_CHECK(hipStreamSynchronize(d->stream));
      (void)hipFree(d->send); d->send = nullptr; d->send_len = 0;
      DSH_HIP_CHECK(hipMalloc((void**)&d->send, sizeof(double) * send_len));
      d->send_len = send_len;
    }
    for (int64_t l0 = 0; l0 < lead; l0 += 65535) {
      const int64_t ll = std::min<int64_t>(65535, lead - l0);
      rc = dsh_dist_pack_shard(d->ctx, d->stream, local + l0 * nl, ll, nl, m, d->send + l0 * m);
      if (rc != DSH_OK) { (void)hipStreamSynchronize(d->stream); return rc; }  // nothing of a failed gather stays queued behind the caller's back
    }
    src = d->send;
  }
  rc = rccl_check(rccl().AllGather(src, d->recv, send_len, ncclFloat64, d->comm, d->stream), "ncclAllGather");
  if (rc != DSH_OK) { (void)hipStreamSynchronize(d->stream); return rc; }
  // recv is [world][lead][m]: unpack in row blocks (the grid's y extent)
  for (int64_t l0 = 0; l0 < lead; l0 += 65535) {
    const int64_t ll = std::min<int64_t>(65535, lead - l0);
    hipLaunchKernelGGL(k_unpack, dim3((unsigned)((n_total + 255) / 256), (unsigned)ll), dim3(256), 0, d->stream, d->recv + l0 * m, out + l0 * n_total, lead, n_total, d->world, m);
  }
  if (hipGetLastError() != hipSuccess || hipEventRecord(d->done, d->stream) != hipSuccess) {
    (void)hipStreamSynchronize(d->stream);
    set_error("dsh_gather_batch_axis_async: the unpack launch / completion event failed");
    return DSH_E_HIP;
  }
  d->pending = true;
  return DSH_OK;
}
// host waits until the gathered trajectories are in `out` (and `local` may be overwritten by the next solve)
int dsh_gather_wait(dsh_dist* d) {
  // no context lock here: the wait touches the communicator's own event only, must not hold up another thread's launches on the solver's context while it blocks
  // (the overlap of a gather with the next solve is the point), and may be called after the context is gone
  if (!d) { set_error("dsh_gather_wait: null communicator"); return DSH_E_INVALID; }
  if (!d->pending) return DSH_OK;
  DSH_HIP_CHECK(hipEventSynchronize(d->done));
  d->pending = false;
  return DSH_OK;
}
int dsh_gather_batch_axis(dsh_dist* d, const double* local, int64_t lead, int64_t n_total, double* out) {
  const int rc = dsh_gather_batch_axis_async(d, local, lead, n_total, out);
  return rc != DSH_OK ? rc : dsh_gather_wait(d);
}

}  // extern "C"
