// Dense-matrix kernels of libdiffsol_hip.so (gfx950): HIP counterparts of diffsol-la/src/cuda_kernels/mat_*.cu and of the
// cuBLAS gemv / gemmStridedBatched call sites (diffsol-la/src/matrix/cuda.rs:80-96, :757-822).
// Batched matrices are column-major per system, batch-fastest across systems: A_b(i,j) at p[(j*nrows+i)*nb + b].
// The matrices on the hot path are tiny per system (n x 8 difference array, (k+1)^2 R*U, n x s stage matrix), so "GEMM" here is
// a per-lane dot product streamed from HBM — bandwidth-bound, deliberately not an MFMA kernel.
#include "dsh_internal.hpp"

using namespace dsh;

namespace {
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 4096;
inline dim3 ew_grid(int64_t total) {
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > kMaxBlocks) blocks = kMaxBlocks;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks);
}

template <bool BV>
__global__ void k_from_diagonal(int64_t n, int64_t nb, const double* __restrict__ v, double* __restrict__ mat) {
  int64_t total = n * n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t e = idx / nb, b = idx % nb;
    int64_t i = e % n, j = e / n;
    mat[idx] = (i == j) ? (BV ? v[i] : v[i * nb + b]) : 0.0;
  }
}
__global__ void k_get_diagonal(int64_t n, int64_t nb, const double* __restrict__ mat, double* __restrict__ v) {
  int64_t total = n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = idx / nb, b = idx % nb;
    v[idx] = mat[(i * n + i) * nb + b];
  }
}
// self = y*beta + x  (dense_nalgebra_serial.rs:325-329 order: copy y, scale by beta, add x)
template <bool BX, bool BY>
__global__ void k_scale_add_assign(int64_t total, int64_t nb, double* __restrict__ self, const double* __restrict__ x, double beta,
                                   const double* __restrict__ y) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    double xv = x[BX ? idx / nb : idx], yv = y[BY ? idx / nb : idx];
    self[idx] = yv * beta + xv;
  }
}
// the same on the band |i - j| <= (kl below, ku above) of n x n matrices only: thread (d, i, b) handles entry (i, i + d - kl) — the entries outside the
// band are not touched (callers guarantee they are structural zeros of all three operands)
template <bool BX, bool BY>
__global__ void k_scale_add_assign_banded(int64_t n, int64_t nb, int kl, int ku, double* __restrict__ self, const double* __restrict__ x, double beta,
                                          const double* __restrict__ y) {
  const int64_t w = kl + ku + 1, total = w * n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = idx % nb, r = idx / nb, i = r % n, j = i + r / n - kl;
    if (j < 0 || j >= n) continue;
    const int64_t e = j * n + i;
    const double xv = x[BX ? e : e * nb + b], yv = y[BY ? e : e * nb + b];
    self[e * nb + b] = yv * beta + xv;
  }
}
// ---- band containers: entry (i, j), -kl <= j - i <= ku, at ((j - i + kl) * n + i) * nb + b; the corners that fall outside the matrix are kept at zero
template <bool BV>
__global__ void k_band_from_diagonal(int64_t n, int64_t nb, int kl, int ku, const double* __restrict__ v, double* __restrict__ band) {
  const int64_t total = (int64_t)(kl + ku + 1) * n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = idx % nb, r = idx / nb, i = r % n, pl = r / n;
    band[idx] = pl == kl ? (BV ? v[i] : v[i * nb + b]) : 0.0;
  }
}
// y = alpha * A x + beta * y over the band, columns ascending: Matrix::gemv's order (the first term carries beta * y, dense_nalgebra_serial gemv) restricted to
// the entries the container holds — the skipped terms of a dense gemv are alpha * 0 * x_j + acc
__global__ void k_band_gemv(int64_t n, int64_t nb, int kl, int ku, double alpha, const double* __restrict__ band, const double* __restrict__ x, int64_t xnb, double beta,
                            double* __restrict__ y) {
  const int64_t total = n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx / nb, b = idx % nb;
    auto A = [&](int64_t j) { const int64_t d = j - i; return (d >= -kl && d <= ku) ? band[((d + kl) * n + i) * nb + b] : 0.0; };
    auto X = [&](int64_t j) { return xnb == 1 && nb != 1 ? x[j] : x[j * nb + b]; };
    double acc = beta == 0.0 ? alpha * A(0) * X(0) : alpha * A(0) * X(0) + beta * y[idx];
    const int64_t lo = i - kl > 1 ? i - kl : 1, hi = i + ku < n - 1 ? i + ku : n - 1;
    for (int64_t j = lo; j <= hi; ++j) acc = alpha * A(j) * X(j) + acc;
    y[idx] = acc;
  }
}
__global__ void k_set_data_with_indices(int64_t nidx, int64_t nb, double* __restrict__ self, const int32_t* __restrict__ dst_idx,
                                        const int32_t* __restrict__ src_idx, const double* __restrict__ data) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nidx * nb; t += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = t / nb, b = t % nb;
    self[(int64_t)dst_idx[k] * nb + b] = data[(int64_t)src_idx[k] * nb + b];
  }
}
// column i += alpha*column j  (value = self[k,i] + alpha*self[k,j])
__global__ void k_column_axpy(int64_t total, double* __restrict__ ci, const double* __restrict__ cj, double alpha) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x)
    ci[idx] = ci[idx] + alpha * cj[idx];
}

// y = alpha*A*x + beta*y with nalgebra's accumulation order: first column carries beta (beta==0 never reads y),
// remaining columns accumulate  acc = alpha*A[i,j]*x[j] + acc.  One thread per (row, system).
// y0 != nullptr: y = alpha*A*x + beta*y0 — `y.copy_from(y0); A.gemv(alpha, x, beta, y)` in one pass (the same first term alpha*A(0)*X(0) + beta*y0_i)
template <bool BA, bool BX>
__global__ void k_gemv(int64_t nrows, int64_t ncols, int64_t nb, double alpha, const double* __restrict__ a, const double* __restrict__ x, double beta,
                       double* y, const double* y0 = nullptr) {
  int64_t total = nrows * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = idx / nb, b = idx % nb;
    const double yin = beta == 0.0 ? 0.0 : (y0 ? y0[idx] : y[idx]);
    if (ncols == 0) { y[idx] = beta == 0.0 ? 0.0 : yin * beta; continue; }
    auto A = [&](int64_t j) { return BA ? a[j * nrows + i] : a[(j * nrows + i) * nb + b]; };
    auto X = [&](int64_t j) { return BX ? x[j] : x[j * nb + b]; };
    double acc = beta == 0.0 ? alpha * A(0) * X(0) : alpha * A(0) * X(0) + beta * yin;
    for (int64_t j = 1; j < ncols; ++j) acc = alpha * A(j) * X(j) + acc;
    y[idx] = acc;
  }
}
// C = alpha*A*B + beta*C, per column of C the gemv order above.  One thread per (row of C, column of C, system).
template <bool BA, bool BB>
__global__ void k_gemm(int64_t m, int64_t n, int64_t k, int64_t nb, double alpha, const double* __restrict__ a, const double* __restrict__ bm, double beta,
                       double* __restrict__ c) {
  int64_t total = m * n * nb;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t e = idx / nb, b = idx % nb;
    int64_t i = e % m, j = e / m;
    if (k == 0) { c[idx] = beta == 0.0 ? 0.0 : c[idx] * beta; continue; }
    auto A = [&](int64_t l) { return BA ? a[l * m + i] : a[(l * m + i) * nb + b]; };
    auto B = [&](int64_t l) { return BB ? bm[j * k + l] : bm[(j * k + l) * nb + b]; };
    double acc = beta == 0.0 ? alpha * A(0) * B(0) : alpha * A(0) * B(0) + beta * c[idx];
    for (int64_t l = 1; l < k; ++l) acc = alpha * A(l) * B(l) + acc;
    c[idx] = acc;
  }
}
}  // namespace

extern "C" {

int dsh_mat_from_diagonal(dsh_ctx* ctx, int64_t n, int64_t nb, const double* v, int64_t vnb, double* mat) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(vnb, nb);
  int64_t total = n * n * nb;
  if (total == 0) return DSH_OK;
  if (vnb == 1 && nb != 1) hipLaunchKernelGGL((k_from_diagonal<true>), ew_grid(total), dim3(kBlock), 0, ctx->stream, n, nb, v, mat);
  else hipLaunchKernelGGL((k_from_diagonal<false>), ew_grid(total), dim3(kBlock), 0, ctx->stream, n, nb, v, mat);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_mat_band_from_diagonal(dsh_ctx* ctx, int64_t n, int64_t nb, int kl, int ku, const double* v, int64_t vnb, double* band) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(vnb, nb);
  DSH_REQUIRE(kl >= 0 && ku >= 0 && band != nullptr, "bad arguments");
  const int64_t total = (int64_t)(kl + ku + 1) * n * nb;
  if (total == 0) return DSH_OK;
  if (vnb == 1 && nb != 1) hipLaunchKernelGGL((k_band_from_diagonal<true>), ew_grid(total), dim3(kBlock), 0, ctx->stream, n, nb, kl, ku, v, band);
  else hipLaunchKernelGGL((k_band_from_diagonal<false>), ew_grid(total), dim3(kBlock), 0, ctx->stream, n, nb, kl, ku, v, band);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_mat_band_gemv(dsh_ctx* ctx, int64_t n, int64_t nb, int kl, int ku, double alpha, const double* band, const double* x, int64_t xnb, double beta, double* y) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(xnb, nb);
  DSH_REQUIRE(kl >= 0 && ku >= 0 && band && x && y, "bad arguments");
  if (n * nb == 0) return DSH_OK;
  hipLaunchKernelGGL(k_band_gemv, ew_grid(n * nb), dim3(kBlock), 0, ctx->stream, n, nb, kl, ku, alpha, band, x, xnb, beta, y);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_mat_get_diagonal(dsh_ctx* ctx, int64_t n, int64_t nb, const double* mat, double* v) {
  DSH_ENTER(ctx);
  if (n * nb == 0) return DSH_OK;
  hipLaunchKernelGGL(k_get_diagonal, ew_grid(n * nb), dim3(kBlock), 0, ctx->stream, n, nb, mat, v);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_mat_set_column(dsh_ctx* ctx, int64_t nrows, int64_t ncols, int64_t nb, double* mat, int64_t j, const double* v, int64_t vnb) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(j >= 0 && j < ncols, "column index out of bounds");
  return dsh_vec_copy(ctx, nrows, nb, v, vnb, mat + j * nrows * nb);
}
int dsh_mat_scale_add_assign(dsh_ctx* ctx, int64_t nelem, int64_t nb, double* self, const double* x, int64_t xnb, double beta, const double* y,
                             int64_t ynb) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(xnb, nb); DSH_CHECK_NB(ynb, nb);
  int64_t total = nelem * nb;
  if (total == 0) return DSH_OK;
  bool bx = xnb == 1 && nb != 1, by = ynb == 1 && nb != 1;
  dim3 g = ew_grid(total), b(kBlock);
  if (!bx && !by) hipLaunchKernelGGL((k_scale_add_assign<false, false>), g, b, 0, ctx->stream, total, nb, self, x, beta, y);
  else if (bx && !by) hipLaunchKernelGGL((k_scale_add_assign<true, false>), g, b, 0, ctx->stream, total, nb, self, x, beta, y);
  else if (!bx && by) hipLaunchKernelGGL((k_scale_add_assign<false, true>), g, b, 0, ctx->stream, total, nb, self, x, beta, y);
  else hipLaunchKernelGGL((k_scale_add_assign<true, true>), g, b, 0, ctx->stream, total, nb, self, x, beta, y);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_mat_scale_add_assign_banded(dsh_ctx* ctx, int64_t n, int64_t nb, int kl, int ku, double* self, const double* x, int64_t xnb, double beta, const double* y,
                                    int64_t ynb) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(xnb, nb); DSH_CHECK_NB(ynb, nb);
  DSH_REQUIRE(kl >= 0 && ku >= 0 && kl < n && ku < n, "bandwidths out of range");
  const int64_t total = (int64_t)(kl + ku + 1) * n * nb;
  if (total == 0) return DSH_OK;
  bool bx = xnb == 1 && nb != 1, by = ynb == 1 && nb != 1;
  dim3 g = ew_grid(total), b(kBlock);
  if (!bx && !by) hipLaunchKernelGGL((k_scale_add_assign_banded<false, false>), g, b, 0, ctx->stream, n, nb, kl, ku, self, x, beta, y);
  else if (bx && !by) hipLaunchKernelGGL((k_scale_add_assign_banded<true, false>), g, b, 0, ctx->stream, n, nb, kl, ku, self, x, beta, y);
  else if (!bx && by) hipLaunchKernelGGL((k_scale_add_assign_banded<false, true>), g, b, 0, ctx->stream, n, nb, kl, ku, self, x, beta, y);
  else hipLaunchKernelGGL((k_scale_add_assign_banded<true, true>), g, b, 0, ctx->stream, n, nb, kl, ku, self, x, beta, y);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_mat_set_data_with_indices(dsh_ctx* ctx, int64_t nelem_self, int64_t nelem_data, int64_t nb, double* self, const int32_t* dst_idx,
                                  const int32_t* src_idx, int64_t nidx, const double* data) {
  DSH_ENTER(ctx);
  (void)nelem_self; (void)nelem_data;
  if (nidx * nb == 0) return DSH_OK;
  hipLaunchKernelGGL(k_set_data_with_indices, ew_grid(nidx * nb), dim3(kBlock), 0, ctx->stream, nidx, nb, self, dst_idx, src_idx, data);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_mat_column_axpy(dsh_ctx* ctx, int64_t nrows, int64_t nb, double* mat, double alpha, int64_t j, int64_t i) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(i != j, "column index cannot be the same");
  DSH_REQUIRE(i >= 0 && j >= 0, "column index out of bounds");
  int64_t total = nrows * nb;
  if (total == 0) return DSH_OK;
  hipLaunchKernelGGL(k_column_axpy, ew_grid(total), dim3(kBlock), 0, ctx->stream, total, mat + i * total, (const double*)(mat + j * total), alpha);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_mat_gemv_from(dsh_ctx* ctx, int64_t nrows, int64_t ncols, int64_t nb, double alpha, const double* a, int64_t anb, const double* x, int64_t xnb,
                      double beta, const double* y0, double* y) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(anb, nb); DSH_CHECK_NB(xnb, nb);
  int64_t total = nrows * nb;
  if (total == 0) return DSH_OK;
  bool ba = anb == 1 && nb != 1, bx = xnb == 1 && nb != 1;
  dim3 g = ew_grid(total), b(kBlock);
  if (!ba && !bx) hipLaunchKernelGGL((k_gemv<false, false>), g, b, 0, ctx->stream, nrows, ncols, nb, alpha, a, x, beta, y, y0);
  else if (ba && !bx) hipLaunchKernelGGL((k_gemv<true, false>), g, b, 0, ctx->stream, nrows, ncols, nb, alpha, a, x, beta, y, y0);
  else if (!ba && bx) hipLaunchKernelGGL((k_gemv<false, true>), g, b, 0, ctx->stream, nrows, ncols, nb, alpha, a, x, beta, y, y0);
  else hipLaunchKernelGGL((k_gemv<true, true>), g, b, 0, ctx->stream, nrows, ncols, nb, alpha, a, x, beta, y, y0);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_mat_gemv(dsh_ctx* ctx, int64_t nrows, int64_t ncols, int64_t nb, double alpha, const double* a, int64_t anb, const double* x, int64_t xnb,
                 double beta, double* y) {
  DSH_ENTER(ctx);
  return dsh_mat_gemv_from(ctx, nrows, ncols, nb, alpha, a, anb, x, xnb, beta, nullptr, y);
}
int dsh_mat_gemm(dsh_ctx* ctx, int64_t m, int64_t n, int64_t k, int64_t nb, double alpha, const double* a, int64_t anb, const double* bm, int64_t bnb,
                 double beta, double* c) {
  DSH_ENTER(ctx);
  DSH_CHECK_NB(anb, nb); DSH_CHECK_NB(bnb, nb);
  int64_t total = m * n * nb;
  if (total == 0) return DSH_OK;
  bool ba = anb == 1 && nb != 1, bb = bnb == 1 && nb != 1;
  dim3 g = ew_grid(total), b(kBlock);
  if (!ba && !bb) hipLaunchKernelGGL((k_gemm<false, false>), g, b, 0, ctx->stream, m, n, k, nb, alpha, a, bm, beta, c);
  else if (ba && !bb) hipLaunchKernelGGL((k_gemm<true, false>), g, b, 0, ctx->stream, m, n, k, nb, alpha, a, bm, beta, c);
  else if (!ba && bb) hipLaunchKernelGGL((k_gemm<false, true>), g, b, 0, ctx->stream, m, n, k, nb, alpha, a, bm, beta, c);
  else hipLaunchKernelGGL((k_gemm<true, true>), g, b, 0, ctx->stream, m, n, k, nb, alpha, a, bm, beta, c);
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

}  // extern "C"
