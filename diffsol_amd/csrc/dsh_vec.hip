// Vector kernels of libdiffsol_hip.so (gfx950): the HIP counterparts of diffsol-la/src/cuda_kernels/vec_*.cu.
//
// Layout is batch-fastest ([i][b]), so every op whose operands all carry the full batch is a flat, fully coalesced
// sweep over n*nbatch doubles (the reference launches grid (ceil(n/B), nbatch) with n = 3 live threads per block —
// SURVEY §2a).  A broadcast operand (nbatch 1) is indexed by the state index idx / nbatch.
// Arithmetic is written exactly as in the CPU reference path (no FMA contraction: built with -ffp-contract=off) so the
// results are bit-identical to the oracle.
#include <cstdlib>
#include "dsh_internal.hpp"

using namespace dsh;

namespace {

constexpr int kEwBlock = 256;
constexpr int kEwMaxBlocks = 4096;

inline dim3 ew_grid(int64_t total) {
  int64_t blocks = (total + kEwBlock - 1) / kEwBlock;
  if (blocks > kEwMaxBlocks) blocks = kEwMaxBlocks;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

// operand index: full operands use the flat index, broadcast operands the state index
template <bool BC>
__device__ __forceinline__ int64_t opidx(int64_t idx, int64_t nb) {
  return BC ? idx / nb : idx;
}

// ---- ternary: ret = f(lhs, rhs)
template <class F, bool BL, bool BR>
__global__ void k_ternary(int64_t total, int64_t nb, const double* __restrict__ lhs, const double* __restrict__ rhs, double* __restrict__ ret, F f) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    ret[idx] = f(lhs[opidx<BL>(idx, nb)], rhs[opidx<BR>(idx, nb)]);
}
// ---- binary in place: lhs = f(lhs, rhs)
template <class F, bool BR>
__global__ void k_binary(int64_t total, int64_t nb, double* __restrict__ lhs, const double* __restrict__ rhs, F f) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    lhs[idx] = f(lhs[idx], rhs[opidx<BR>(idx, nb)]);
}
// ---- unary in place / out of place with scalar
template <class F>
__global__ void k_unary(int64_t total, const double* __restrict__ src, double* __restrict__ dst, F f) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) dst[idx] = f(src[idx]);
}
template <class F>
__global__ void k_generate(int64_t total, double* __restrict__ dst, F f) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) dst[idx] = f(idx);
}

struct FAdd { __device__ double operator()(double a, double b) const { return a + b; } };
struct FSub { __device__ double operator()(double a, double b) const { return a - b; } };
struct FMul { __device__ double operator()(double a, double b) const { return a * b; } };
struct FDiv { __device__ double operator()(double a, double b) const { return a / b; } };
struct FScale { double s; __device__ double operator()(double a) const { return a * s; } };
struct FConst { double v; __device__ double operator()(int64_t) const { return v; } };
// y = alpha*x + beta*y ; beta == 0 never reads y (nalgebra axcpy / the oracle's axpy)
struct FAxpy { double alpha, beta; __device__ double operator()(double y, double x) const { return alpha * x + beta * y; } };
struct FAxpy0 { double alpha; __device__ double operator()(double, double x) const { return alpha * x; } };

template <class F>
int launch_ternary(dsh_ctx* ctx, int64_t n, int64_t nb, const double* lhs, int64_t lnb, const double* rhs, int64_t rnb, double* ret, F f) {
  int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  bool bl = lnb == 1 && nb != 1, br = rnb == 1 && nb != 1;
  dim3 g = ew_grid(total), b(kEwBlock);
  if (!bl && !br) hipLaunchKernelGGL((k_ternary<F, false, false>), g, b, 0, ctx->stream, total, nb, lhs, rhs, ret, f);
  else if (bl && !br) hipLaunchKernelGGL((k_ternary<F, true, false>), g, b, 0, ctx->stream, total, nb, lhs, rhs, ret, f);
  else if (!bl && br) hipLaunchKernelGGL((k_ternary<F, false, true>), g, b, 0, ctx->stream, total, nb, lhs, rhs, ret, f);
  else hipLaunchKernelGGL((k_ternary<F, true, true>), g, b, 0, ctx->stream, total, nb, lhs, rhs, ret, f);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
template <class F>
int launch_binary(dsh_ctx* ctx, int64_t n, int64_t nb, double* lhs, const double* rhs, int64_t rnb, F f) {
  int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  bool br = rnb == 1 && nb != 1;
  dim3 g = ew_grid(total), b(kEwBlock);
  if (!br) hipLaunchKernelGGL((k_binary<F, false>), g, b, 0, ctx->stream, total, nb, lhs, rhs, f);
  else hipLaunchKernelGGL((k_binary<F, true>), g, b, 0, ctx->stream, total, nb, lhs, rhs, f);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

// ---- per-batch alpha axpy: y_b = alpha[b]*x_b + beta*y_b
template <bool BX>
__global__ void k_batched_axpy(int64_t total, int64_t nb, const double* __restrict__ alpha, const double* __restrict__ x, double beta, double* __restrict__ y) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = idx % nb;
    double xv = x[opidx<BX>(idx, nb)];
    y[idx] = beta == 0.0 ? alpha[b] * xv : alpha[b] * xv + beta * y[idx];
  }
}

// ---- index kernels (one thread per (k, b))
__global__ void k_gather(int64_t nidx, int64_t nb, const double* __restrict__ src, const int32_t* __restrict__ idx, double* __restrict__ dst) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nidx * nb; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = t / nb, b = t % nb;
    dst[k * nb + b] = src[(int64_t)idx[k] * nb + b];
  }
}
__global__ void k_scatter(int64_t nidx, int64_t nb, const double* __restrict__ src, const int32_t* __restrict__ idx, double* __restrict__ dst) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nidx * nb; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = t / nb, b = t % nb;
    dst[(int64_t)idx[k] * nb + b] = src[k * nb + b];
  }
}
__global__ void k_copy_from_indices(int64_t nidx, int64_t nb, const double* __restrict__ src, const int32_t* __restrict__ idx, double* __restrict__ dst) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nidx * nb; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = t / nb, b = t % nb;
    int64_t o = (int64_t)idx[k] * nb + b;
    dst[o] = src[o];
  }
}
__global__ void k_assign_at_indices(int64_t nidx, int64_t nb, const int32_t* __restrict__ idx, double value, double* __restrict__ dst) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nidx * nb; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = t / nb, b = t % nb;
    dst[(int64_t)idx[k] * nb + b] = value;
  }
}

// ---- reductions: one lane per system, sequential over the n states (same summation order as the CPU path), then
// wave shuffle max -> conditional atomicMax into the slot group.  Loads are coalesced: lane b reads p[i*nb + b].
template <bool BY, bool BA>
__global__ __launch_bounds__(256) void k_squared_norm(int64_t n, int64_t nb, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ atol,
                               double rtol, unsigned long long* rec, unsigned int seq, double* __restrict__ per_batch) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bits = 0ull;
  if (b < nb) {
    // one lane per system, the sum in index order (the reference's sequential accumulation).  For large n only nb lanes are busy, so the loop is
    // latency-bound: 16 independent loads / divisions are kept in flight, only the additions are a dependent chain.
    double acc = 0.0;
    int64_t i = 0;
    if (n >= 32) {
      // software-pipelined: the operands of chunk c+1 are requested before the divisions of chunk c are issued, so the memory latency of one
      // chunk hides behind the arithmetic of the previous one
      double xa[16], ya[16], aa[16];
      auto fetch = [&](int64_t base, double (&xs)[16], double (&ys)[16], double (&as)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          xs[q] = x[(base + q) * nb + b];
          ys[q] = BY ? y[base + q] : y[(base + q) * nb + b];
          as[q] = BA ? atol[base + q] : atol[(base + q) * nb + b];
        }
      };
      fetch(0, xa, ya, aa);
      for (; i + 16 <= n; i += 16) {
        double xn[16], yn[16], an[16];
        const bool more = i + 32 <= n;
        if (more) fetch(i + 16, xn, yn, an);
        double term[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) term[q] = xa[q] / (fabs(ya[q]) * rtol + aa[q]);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += term[q] * term[q];
        if (more) {
#pragma unroll
          for (int q = 0; q < 16; ++q) { xa[q] = xn[q]; ya[q] = yn[q]; aa[q] = an[q]; }
        }
      }
    }
    for (; i < n; ++i) {
      double yi = BY ? y[i] : y[i * nb + b];
      double ai = BA ? atol[i] : atol[i * nb + b];
      double term = x[i * nb + b] / (fabs(yi) * rtol + ai);
      acc += term * term;
    }
    double nrm = acc / (double)n;
    if (per_batch) per_batch[b] = nrm;
    bits = d2u(nrm);
  }
  block_publish(bits, 0ull, 0ull, rec, seq);
}

// The same norm for SMALL ensembles of LONG vectors (config 3: n = 512, 4096 members): one lane per member is 64 wavefronts walking 512 dependent
// loads-divisions-additions each (60 us, 0.56 TB/s).  Here a wavefront owns 8 members; its lanes are 8 row groups x 8 members, so the loads and the
// divisions of 32 components per member run in parallel on all 64 lanes, and only the additions — in index order, as Vector::squared_norm sums —
// are a chain, on the 8 lanes of row group 0, fed through LDS.  Same terms, same order of additions: the same bits.
// SUB: the lanes that fetch the terms also apply the Newton update the norm belongs to, xout = xin - x (x = the Newton step delta; xout may be xin) —
// NoLineSearch::take_optimal_step's `xn -= delta` and Convergence::norm(delta) in one pass over delta (line_search.rs:43-72)
template <bool BY, bool BA, bool SUB = false>
__global__ __launch_bounds__(64) void k_squared_norm_wide(int64_t n, int64_t nb, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ atol,
                                                          double rtol, unsigned long long* rec, unsigned int seq, double* __restrict__ per_batch,
                                                          const double* xin = nullptr, double* xout = nullptr) {
  constexpr int S = 8, G = 8, QL = 4, CH = G * QL, LD = CH + 8;  // LD: row stride of the staging array (bank spread)
  __shared__ double sT[S][LD];
  const int lane = threadIdx.x, s = lane % S, g = lane / S;
  const int64_t b0 = (int64_t)blockIdx.x * S + s;
  const bool valid = b0 < nb;
  const int64_t b = valid ? b0 : nb - 1;
  double acc = 0.0;
  double xa[QL], ya[QL], aa[QL], ia[SUB ? QL : 1];
  auto fetch = [&](int64_t base, double (&xs)[QL], double (&ys)[QL], double (&as)[QL], double (&is)[SUB ? QL : 1]) {
#pragma unroll
    for (int q = 0; q < QL; ++q) {
      const int64_t r = min(base + q * G + g, n - 1);  // clamped: components past the end are not added
      xs[q] = x[r * nb + b];
      if constexpr (SUB) is[q] = xin[r * nb + b];  // the update is stored when the chunk is consumed: a store here would wait for the loads just issued
      ys[q] = BY ? y[r] : y[r * nb + b];
      as[q] = BA ? atol[r] : atol[r * nb + b];
    }
  };
  fetch(0, xa, ya, aa, ia);
  for (int64_t i0 = 0; i0 < n; i0 += CH) {
    double xn[QL], yn[QL], an[QL], in[SUB ? QL : 1];
    const bool more = i0 + CH < n;
    if (more) fetch(i0 + CH, xn, yn, an, in);
    if constexpr (SUB) {
#pragma unroll
      for (int q = 0; q < QL; ++q) { const int64_t r = i0 + q * G + g; if (valid && r < n) xout[r * nb + b] = ia[q] - xa[q]; }
    }
#pragma unroll
    for (int q = 0; q < QL; ++q) {
      const double term = xa[q] / (fabs(ya[q]) * rtol + aa[q]);
      sT[s][q * G + g] = term * term;
    }
    __builtin_amdgcn_wave_barrier();
    if (g == 0) {
      const int cnt = (int)min((int64_t)CH, n - i0);
      if (cnt == CH) {
#pragma unroll
        for (int t = 0; t < CH; ++t) acc += sT[s][t];
      } else {
        for (int t = 0; t < cnt; ++t) acc += sT[s][t];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (more) {
#pragma unroll
      for (int q = 0; q < QL; ++q) { xa[q] = xn[q]; ya[q] = yn[q]; aa[q] = an[q]; if constexpr (SUB) ia[q] = in[q]; }
    }
  }
  unsigned long long bits = 0ull;
  if (g == 0 && valid) {
    const double nrm = acc / (double)n;
    if (per_batch) per_batch[b0] = nrm;
    bits = d2u(nrm);
  }
  block_publish(bits, 0ull, 0ull, rec, seq);
}

// Third form of the same norm, for the same ensembles: one WORKGROUP of four wavefronts per 8 members.  k_squared_norm_wide is one wavefront, one instruction
// stream: its loads, its 64 divisions per lane and its additions stand in line (26 us at n = 512 x 4096).  Here the 256 lanes are 32 row groups x 8 members, each
// lane has 16 rows of a 512-row block in flight at once and divides 16 times; the terms go to LDS in [row][member] order and 8 lanes of wavefront 0 add them up in
// index order — same terms, same order of additions: the same bits.  SUB as in k_squared_norm_wide.
constexpr int kNormTeamThreads = 256, kNormTeamRows = 512;
template <bool BY, bool BA, bool SUB = false>
__global__ __launch_bounds__(kNormTeamThreads) void k_squared_norm_team(int64_t n, int64_t nb, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ atol,
                                                                        double rtol, unsigned long long* rec, unsigned int seq, double* __restrict__ per_batch,
                                                                        const double* xin = nullptr, double* xout = nullptr) {
  constexpr int S = 8, G = kNormTeamThreads / S, RB = kNormTeamRows, QL = RB / G;
  __shared__ double sT[RB * S];  // term of row t, member s at t * S + s
  const int tid = threadIdx.x, s = tid % S, g = tid / S;
  const int64_t b0 = (int64_t)blockIdx.x * S + s;
  const bool valid = b0 < nb;
  const int64_t b = valid ? b0 : nb - 1;
  double acc = 0.0;
  for (int64_t r0 = 0; r0 < n; r0 += RB) {
    double xa[QL], ya[QL], aa[QL], ia[SUB ? QL : 1];
#pragma unroll
    for (int q = 0; q < QL; ++q) {
      const int64_t r = min(r0 + q * G + g, n - 1);  // clamped: components past the end are not added
      xa[q] = x[r * nb + b];
      if constexpr (SUB) ia[q] = xin[r * nb + b];
      ya[q] = BY ? y[r] : y[r * nb + b];
      aa[q] = BA ? atol[r] : atol[r * nb + b];
    }
#pragma unroll
    for (int q = 0; q < QL; ++q) {
      if constexpr (SUB) { const int64_t r = r0 + q * G + g; if (valid && r < n) xout[r * nb + b] = ia[q] - xa[q]; }
      const double term = xa[q] / (fabs(ya[q]) * rtol + aa[q]);
      sT[(q * G + g) * S + s] = term * term;
    }
    __syncthreads();
    if (tid < S) {
      const int cnt = (int)min((int64_t)RB, n - r0);
      int t = 0;
      for (; t + 16 <= cnt; t += 16) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = sT[(t + k) * S + s];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
      }
      for (; t < cnt; ++t) acc += sT[t * S + s];
    }
    __syncthreads();
  }
  unsigned long long bits = 0ull;
  if (tid < S && valid) {
    const double nrm = acc / (double)n;
    if (per_batch) per_batch[b0] = nrm;
    bits = d2u(nrm);
  }
  block_publish(bits, 0ull, 0ull, rec, seq);
}

// which of the two small-ensemble forms (DSH_NORM_TEAM=0 / 1 forces one).  Measured at n = 512 x 4096 (profiles/r04_c3_kernel_stats.md): with the Newton update
// in the same pass (SUB: four vectors) 23.7 us against 26.1; the norm alone 15.4 against 14.1 — both forms are one memory pass followed by what is left of the
// 512-addition chain, so the workgroup form is taken for the fused update only, from 256 rows on
static bool norm_team(int64_t n, bool sub) {
  static const int env = [] { const char* e = std::getenv("DSH_NORM_TEAM"); return e && *e ? std::atoi(e) : -1; }();
  return env >= 0 ? env != 0 : (sub && n >= 256);
}

__global__ void k_norm(int64_t n, int64_t nb, const double* __restrict__ x, int k, unsigned long long* rec, unsigned int seq) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bits = 0ull;
  if (b < nb) {
    double acc = 0.0;
    if (k == 2) { for (int64_t i = 0; i < n; ++i) { double v = x[i * nb + b]; acc += v * v; } acc = sqrt(acc); }
    else if (k == 1) { for (int64_t i = 0; i < n; ++i) acc += fabs(x[i * nb + b]); }
    else { for (int64_t i = 0; i < n; ++i) acc += pow(fabs(x[i * nb + b]), (double)k); acc = pow(acc, 1.0 / (double)k); }
    bits = d2u(acc);
  }
  block_publish(bits, 0ull, 0ull, rec, seq);
}

// per-system root finding triple; results to three device arrays of nbatch entries
__global__ void k_root_finding(int64_t n, int64_t nb, const double* __restrict__ g0, const double* __restrict__ g1, int32_t* __restrict__ found,
                               int32_t* __restrict__ midx, double* __restrict__ frac, unsigned long long* rec, unsigned int seq) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mism = 0ull;
  int f = 0, mi = -1;
  double mx = 0.0;
  if (b < nb) {
    for (int64_t i = 0; i < n; ++i) {
      double v0 = g0[i * nb + b], v1 = g1[i * nb + b];
      if (v1 == 0.0) f = 1;
      if (v0 * v1 < 0.0) { double val = fabs(v1 / (v1 - v0)); if (val > mx) { mx = val; mi = (int)i; } }
    }
    found[b] = f; midx[b] = mi; frac[b] = mx;
  }
  // compare with batch member 0 (lane 0 of block 0 recomputes it cheaply: n is tiny for root functions)
  int f0 = 0, mi0 = -1;
  {
    double mx0 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
      double v0 = g0[i * nb], v1 = g1[i * nb];
      if (v1 == 0.0) f0 = 1;
      if (v0 * v1 < 0.0) { double val = fabs(v1 / (v1 - v0)); if (val > mx0) { mx0 = val; mi0 = (int)i; } }
    }
  }
  if (b < nb && (f != f0 || mi != mi0)) mism = 1ull;
  block_publish(0ull, 0ull, mism, rec, seq);
}

}  // namespace

extern "C" {

int dsh_vec_add(dsh_ctx* ctx, int64_t n, int64_t nb, const double* lhs, int64_t lnb, const double* rhs, int64_t rnb, double* ret) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(lnb, nb); DSH_CHECK_NB(rnb, nb);
  return launch_ternary(ctx, n, nb, lhs, lnb, rhs, rnb, ret, FAdd{});
}
int dsh_vec_sub(dsh_ctx* ctx, int64_t n, int64_t nb, const double* lhs, int64_t lnb, const double* rhs, int64_t rnb, double* ret) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(lnb, nb); DSH_CHECK_NB(rnb, nb);
  return launch_ternary(ctx, n, nb, lhs, lnb, rhs, rnb, ret, FSub{});
}
int dsh_vec_add_assign(dsh_ctx* ctx, int64_t n, int64_t nb, double* lhs, const double* rhs, int64_t rnb) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(rnb, nb);
  return launch_binary(ctx, n, nb, lhs, rhs, rnb, FAdd{});
}
int dsh_vec_sub_assign(dsh_ctx* ctx, int64_t n, int64_t nb, double* lhs, const double* rhs, int64_t rnb) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(rnb, nb);
  return launch_binary(ctx, n, nb, lhs, rhs, rnb, FSub{});
}
int dsh_vec_mul_assign(dsh_ctx* ctx, int64_t n, int64_t nb, double* lhs, const double* rhs, int64_t rnb) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(rnb, nb);
  return launch_binary(ctx, n, nb, lhs, rhs, rnb, FMul{});
}
int dsh_vec_div_assign(dsh_ctx* ctx, int64_t n, int64_t nb, double* lhs, const double* rhs, int64_t rnb) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(rnb, nb);
  return launch_binary(ctx, n, nb, lhs, rhs, rnb, FDiv{});
}
int dsh_vec_mul_assign_scalar(dsh_ctx* ctx, int64_t n, int64_t nb, double* v, double s) {
  DSH_ENTER(ctx);
  int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  hipLaunchKernelGGL((k_unary<FScale>), ew_grid(total), dim3(kEwBlock), 0, ctx->stream, total, (const double*)v, v, FScale{s});
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_vec_mul_scalar(dsh_ctx* ctx, int64_t n, int64_t nb, const double* v, double s, double* res) {
  DSH_ENTER(ctx);
  int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  hipLaunchKernelGGL((k_unary<FScale>), ew_grid(total), dim3(kEwBlock), 0, ctx->stream, total, v, res, FScale{s});
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_vec_axpy(dsh_ctx* ctx, int64_t n, int64_t nb, double alpha, const double* x, int64_t xnb, double beta, double* y) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(xnb, nb);
  if (beta == 0.0) return launch_binary(ctx, n, nb, y, x, xnb, FAxpy0{alpha});
  return launch_binary(ctx, n, nb, y, x, xnb, FAxpy{alpha, beta});
}
// out = alpha*x + beta*y0 (`out.copy_from(y0); out.axpy(alpha, x, beta)` in one pass); copy_x_to (optional) additionally receives x
__global__ void k_axpby_to(int64_t total, double alpha, const double* __restrict__ x, double beta, const double* __restrict__ y0, double* __restrict__ out,
                           double* __restrict__ copy_x_to) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const double xv = x[idx];
    out[idx] = alpha * xv + beta * y0[idx];
    if (copy_x_to) copy_x_to[idx] = xv;
  }
}
int dsh_vec_axpby_to(dsh_ctx* ctx, int64_t n, int64_t nb, double alpha, const double* x, double beta, const double* y0, double* out, double* copy_x_to) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(x && y0 && out, "null argument");
  DSH_REQUIRE(out != x && out != copy_x_to, "dsh_vec_axpby_to: out must not alias x or the copy");
  const int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  hipLaunchKernelGGL(k_axpby_to, ew_grid(total), dim3(kEwBlock), 0, ctx->stream, total, alpha, x, beta, y0, out, copy_x_to);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_vec_batched_axpy(dsh_ctx* ctx, int64_t n, int64_t nb, const double* alpha_host, const double* x, int64_t xnb, double beta, double* y) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(xnb, nb);
  int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  int rc = ensure_f64_scratch(ctx, nb);
  if (rc != DSH_OK) return rc;
  DSH_HIP_CHECK(hipMemcpyAsync(ctx->f64_scratch, alpha_host, sizeof(double) * nb, hipMemcpyHostToDevice, ctx->stream));
  if (xnb == 1 && nb != 1) hipLaunchKernelGGL((k_batched_axpy<true>), ew_grid(total), dim3(kEwBlock), 0, ctx->stream, total, nb, (const double*)ctx->f64_scratch, x, beta, y);
  else hipLaunchKernelGGL((k_batched_axpy<false>), ew_grid(total), dim3(kEwBlock), 0, ctx->stream, total, nb, (const double*)ctx->f64_scratch, x, beta, y);
  DSH_HIP_CHECK(hipGetLastError());
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // alpha_host / scratch may be reused by the caller
  return DSH_OK;
}
int dsh_vec_copy(dsh_ctx* ctx, int64_t n, int64_t nb, const double* src, int64_t snb, double* dst) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(snb, nb);
  if (snb == nb) return dsh_d2d(ctx, dst, src, sizeof(double) * n * nb);
  return launch_binary(ctx, n, nb, dst, src, snb, FAxpy0{1.0});
}
int dsh_vec_fill(dsh_ctx* ctx, int64_t n, int64_t nb, double* v, double value) {
  DSH_ENTER(ctx);
  int64_t total = n * nb;
  if (total == 0) return DSH_OK;
  hipLaunchKernelGGL((k_generate<FConst>), ew_grid(total), dim3(kEwBlock), 0, ctx->stream, total, v, FConst{value});
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_vec_set_index_all(dsh_ctx* ctx, int64_t nb, double* v, int64_t i, double value) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(i >= 0, "index out of range");
  return dsh_vec_fill(ctx, 1, nb, v + i * nb, value);
}

int dsh_vec_gather(dsh_ctx* ctx, int64_t n_src, int64_t nb, const double* src, const int32_t* idx, int64_t nidx, double* dst) {
  DSH_ENTER(ctx);
  (void)n_src;
  if (nidx * nb == 0) return DSH_OK;
  hipLaunchKernelGGL(k_gather, ew_grid(nidx * nb), dim3(kEwBlock), 0, ctx->stream, nidx, nb, src, idx, dst);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_vec_scatter(dsh_ctx* ctx, int64_t n_dst, int64_t nb, const double* src, const int32_t* idx, int64_t nidx, double* dst) {
  DSH_ENTER(ctx);
  (void)n_dst;
  if (nidx * nb == 0) return DSH_OK;
  hipLaunchKernelGGL(k_scatter, ew_grid(nidx * nb), dim3(kEwBlock), 0, ctx->stream, nidx, nb, src, idx, dst);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_vec_copy_from_indices(dsh_ctx* ctx, int64_t n, int64_t nb, const double* src, const int32_t* idx, int64_t nidx, double* dst) {
  DSH_ENTER(ctx);
  (void)n;
  if (nidx * nb == 0) return DSH_OK;
  hipLaunchKernelGGL(k_copy_from_indices, ew_grid(nidx * nb), dim3(kEwBlock), 0, ctx->stream, nidx, nb, src, idx, dst);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_vec_assign_at_indices(dsh_ctx* ctx, int64_t n, int64_t nb, const int32_t* idx, int64_t nidx, double value, double* dst) {
  DSH_ENTER(ctx);
  (void)n;
  if (nidx * nb == 0) return DSH_OK;
  hipLaunchKernelGGL(k_assign_at_indices, ew_grid(nidx * nb), dim3(kEwBlock), 0, ctx->stream, nidx, nb, idx, value, dst);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

int dsh_vec_norm(dsh_ctx* ctx, int64_t n, int64_t nb, const double* x, int k, double* out_max) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(k >= 1 && out_max, "bad arguments");
  if (n == 0) { *out_max = 0.0; return DSH_OK; }
  unsigned long long* rec; unsigned int seq;
  dim3 g = grid_for(nb, ctx->block);
  int rc = begin_records(ctx, g.x, &rec, &seq);
  if (rc != DSH_OK) return rc;
  hipLaunchKernelGGL(k_norm, g, dim3(ctx->block), 0, ctx->stream, n, nb, x, k, rec, seq);
  DSH_HIP_CHECK(hipGetLastError());
  rc = fetch_records(ctx, g.x, seq);
  if (rc != DSH_OK) return rc;
  *out_max = bits_to_double(ctx->res_m0);
  return DSH_OK;
}

}  // extern "C"
namespace dsh {
// xout = xin - delta and the squared norm of delta in one launch where the wide kernel applies (long vectors, few members), two launches otherwise; the
// caller redeems the norm's records (fetch_records(ctx, *gx, *seq): res_m0).  Used by the staged SDIRK Newton iteration (dsh_fused.hip).
int vec_sub_squared_norm_launch(dsh_ctx* ctx, int64_t n, int64_t nb, const double* delta, const double* xin, double* xout, const double* y, int64_t ynb,
                                const double* atol, int64_t anb, double rtol, unsigned int* gx, unsigned int* seq_out) {
  DSH_CHECK_NB(ynb, nb); DSH_CHECK_NB(anb, nb);
  unsigned long long* rec; unsigned int seq;
  const int threads = ctx->block < 256 ? ctx->block : 256;
  static const int wide_env = [] { const char* e = std::getenv("DSH_NORM_WIDE"); return e && *e ? std::atoi(e) : -1; }();
  const bool wide = wide_env >= 0 ? wide_env != 0 : (n >= 128 && nb <= 16384);
  const bool by = ynb == 1 && nb != 1, ba = anb == 1 && nb != 1;
  if (!wide) {
    int rc = launch_ternary(ctx, n, nb, xin, nb, delta, nb, xout, FSub{});
    if (rc != DSH_OK) return rc;
  }
  dim3 g = wide ? grid_for(nb, 8) : grid_for(nb, threads), b(threads);
  int rc = begin_records(ctx, g.x, &rec, &seq);
  if (rc != DSH_OK) return rc;
  double* none = nullptr;
  if (wide) {
    if (norm_team(n, true)) {
      const dim3 tb(kNormTeamThreads);
      if (!by && !ba) hipLaunchKernelGGL((k_squared_norm_team<false, false, true>), g, tb, 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none, xin, xout);
      else if (by && !ba) hipLaunchKernelGGL((k_squared_norm_team<true, false, true>), g, tb, 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none, xin, xout);
      else if (!by && ba) hipLaunchKernelGGL((k_squared_norm_team<false, true, true>), g, tb, 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none, xin, xout);
      else hipLaunchKernelGGL((k_squared_norm_team<true, true, true>), g, tb, 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none, xin, xout);
    } else
    if (!by && !ba) hipLaunchKernelGGL((k_squared_norm_wide<false, false, true>), g, dim3(64), 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none, xin, xout);
    else if (by && !ba) hipLaunchKernelGGL((k_squared_norm_wide<true, false, true>), g, dim3(64), 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none, xin, xout);
    else if (!by && ba) hipLaunchKernelGGL((k_squared_norm_wide<false, true, true>), g, dim3(64), 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none, xin, xout);
    else hipLaunchKernelGGL((k_squared_norm_wide<true, true, true>), g, dim3(64), 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none, xin, xout);
  } else
  if (!by && !ba) hipLaunchKernelGGL((k_squared_norm<false, false>), g, b, 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none);
  else if (by && !ba) hipLaunchKernelGGL((k_squared_norm<true, false>), g, b, 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none);
  else if (!by && ba) hipLaunchKernelGGL((k_squared_norm<false, true>), g, b, 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none);
  else hipLaunchKernelGGL((k_squared_norm<true, true>), g, b, 0, ctx->stream, n, nb, delta, y, atol, rtol, rec, seq, none);
  DSH_HIP_CHECK(hipGetLastError());
  *gx = g.x; *seq_out = seq;
  return DSH_OK;
}
}  // namespace dsh
extern "C" {

int dsh_vec_squared_norm(dsh_ctx* ctx, int64_t n, int64_t nb, const double* x, const double* y, int64_t ynb, const double* atol, int64_t anb,
                         double rtol, double* out_max, double* per_batch_dev) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(ynb, nb); DSH_CHECK_NB(anb, nb);
  DSH_REQUIRE(out_max != nullptr, "out_max is null");
  if (n == 0) { *out_max = 0.0; return DSH_OK; }  // vector/cuda.rs:1365-1367
  unsigned long long* rec; unsigned int seq;
  const int threads = ctx->block < 256 ? ctx->block : 256;  // the kernel is compiled for at most 256 threads (register budget of its pipelined loop)
  // long vectors, few members: 8 members per wavefront, terms on all lanes (k_squared_norm_wide: same bits).  DSH_NORM_WIDE=0 / 1 forces either.
  static const int wide_env = [] { const char* e = std::getenv("DSH_NORM_WIDE"); return e && *e ? std::atoi(e) : -1; }();
  const bool wide = wide_env >= 0 ? wide_env != 0 : (n >= 128 && nb <= 16384);
  dim3 g = wide ? grid_for(nb, 8) : grid_for(nb, threads), b(threads);
  int rc = begin_records(ctx, g.x, &rec, &seq);
  if (rc != DSH_OK) return rc;
  bool by = ynb == 1 && nb != 1, ba = anb == 1 && nb != 1;
  if (wide) {
    if (norm_team(n, false)) {
      const dim3 tb(kNormTeamThreads);
      if (!by && !ba) hipLaunchKernelGGL((k_squared_norm_team<false, false>), g, tb, 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
      else if (by && !ba) hipLaunchKernelGGL((k_squared_norm_team<true, false>), g, tb, 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
      else if (!by && ba) hipLaunchKernelGGL((k_squared_norm_team<false, true>), g, tb, 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
      else hipLaunchKernelGGL((k_squared_norm_team<true, true>), g, tb, 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
    } else
    if (!by && !ba) hipLaunchKernelGGL((k_squared_norm_wide<false, false>), g, dim3(64), 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
    else if (by && !ba) hipLaunchKernelGGL((k_squared_norm_wide<true, false>), g, dim3(64), 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
    else if (!by && ba) hipLaunchKernelGGL((k_squared_norm_wide<false, true>), g, dim3(64), 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
    else hipLaunchKernelGGL((k_squared_norm_wide<true, true>), g, dim3(64), 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
  } else
  if (!by && !ba) hipLaunchKernelGGL((k_squared_norm<false, false>), g, b, 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
  else if (by && !ba) hipLaunchKernelGGL((k_squared_norm<true, false>), g, b, 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
  else if (!by && ba) hipLaunchKernelGGL((k_squared_norm<false, true>), g, b, 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
  else hipLaunchKernelGGL((k_squared_norm<true, true>), g, b, 0, ctx->stream, n, nb, x, y, atol, rtol, rec, seq, per_batch_dev);
  DSH_HIP_CHECK(hipGetLastError());
  rc = fetch_records(ctx, g.x, seq);
  if (rc != DSH_OK) return rc;
  *out_max = bits_to_double(ctx->res_m0);
  return DSH_OK;
}

int dsh_vec_root_finding(dsh_ctx* ctx, int64_t n, int64_t nb, const double* g0, const double* g1, int* found, double* frac, int* idx) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(found && frac && idx, "null output");
  if (n == 0) { *found = 0; *frac = 0.0; *idx = -1; return DSH_OK; }
  int rc = ensure_i32_scratch(ctx, 2 * nb);
  if (rc != DSH_OK) return rc;
  rc = ensure_f64_scratch(ctx, nb);
  if (rc != DSH_OK) return rc;
  unsigned long long* rec; unsigned int seq;
  dim3 g = grid_for(nb, ctx->block);
  rc = begin_records(ctx, g.x, &rec, &seq);
  if (rc != DSH_OK) return rc;
  hipLaunchKernelGGL(k_root_finding, g, dim3(ctx->block), 0, ctx->stream, n, nb, g0, g1, ctx->i32_scratch, ctx->i32_scratch + nb, ctx->f64_scratch, rec, seq);
  DSH_HIP_CHECK(hipGetLastError());
  int32_t h_found = 0, h_idx = -1;
  double h_frac = 0.0;
  DSH_HIP_CHECK(hipMemcpyAsync(&h_found, ctx->i32_scratch, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  DSH_HIP_CHECK(hipMemcpyAsync(&h_idx, ctx->i32_scratch + nb, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  DSH_HIP_CHECK(hipMemcpyAsync(&h_frac, ctx->f64_scratch, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  rc = fetch_records(ctx, g.x, seq);
  if (rc != DSH_OK) return rc;
  *found = h_found; *idx = h_idx; *frac = h_frac;
  if (ctx->res_cnt != 0ull) {
    set_error("dsh_vec_root_finding: root finding results differ across batches (" + std::to_string((long long)ctx->res_cnt) + " of " +
              std::to_string((long long)nb) + " batch members disagree with member 0)");
    return DSH_E_BATCH_MISMATCH;
  }
  return DSH_OK;
}

}  // extern "C"
