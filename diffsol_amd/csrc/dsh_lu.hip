// Batched dense LU of libdiffsol_hip.so (gfx950): replaces CudaLU (diffsol-la/src/linear_solver/cuda/lu.rs:59-191), whose
// set_linearisation / solve_in_place loop over the batch on the HOST calling cusolverDnDgetrf / Dgetrs once per system.
// Here every system of the ensemble is factored / solved by ONE launch: one lane per system, factors in registers for n <= 8
// (18 flop and 132 algorithmic bytes per n=3 solve: purely HBM-bound, so the job of the kernel is to keep every access coalesced),
// one wavefront per system for n <= 64 (dsh_lu_wave.hpp), one workgroup per system with a blocked factorisation beyond (dsh_lu_coop.hpp).
#include <algorithm>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <vector>

#include "dsh_internal.hpp"
#include "dsh_lu_dev.hpp"
#include "dsh_lu_coop.hpp"
#include "dsh_lu_wave.hpp"
#include "dsh_lu_band.hpp"
#include "dsh_lu_gband.hpp"
#include "dsh_lu_band_team.hpp"
#include "dsh_lu_band_affine.hpp"
#include "dsh_lu_tiled.hpp"

using namespace dsh;

namespace {

template <int N>
__global__ void k_lu_factor_reg(int64_t nb, const double* __restrict__ a, double* __restrict__ factors, int32_t* __restrict__ piv,
                                unsigned long long* singular_count, unsigned int epoch) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long sing = 0ull;
  if (b < nb) {
    double A[N * N];
    int P[N];
    load_mat<N>(a, nb, b, A);
    bool s = false;
    lu_factor_reg<N>(A, P, s);
    store_mat<N>(factors, nb, b, A);
    store_piv<N>(piv, nb, b, P);
    sing = s ? 1ull : 0ull;
  }
  sing = wave_sum_u64(sing);
  if ((threadIdx.x & 63) == 0 && sing) publish_singular(singular_count, sing, epoch);
}

template <int N>
__global__ void k_lu_solve_reg(int64_t nb, const double* __restrict__ factors, const int32_t* __restrict__ piv, double* __restrict__ rhs,
                               unsigned long long* rec, unsigned int seq) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad = 0ull;
  if (b < nb) {
    double A[N * N], v[N];
    int P[N];
    load_mat<N>(factors, nb, b, A);
    load_piv<N>(piv, nb, b, P);
    load_vec<N>(rhs, nb, b, v);
    bool ok = lu_solve_reg<N>(A, P, v);
    store_vec<N>(rhs, nb, b, v);
    bad = ok ? 0ull : 1ull;
  }
  block_publish(0ull, 0ull, bad, rec, seq);
}

// several right-hand sides per system with the factors loaded once (forward-sensitivity columns: Bdf::sensitivity_solve, bdf.rs:934-989, solves one
// system per parameter with the SAME LU): column r of system b at rhs[(r*n + i)*nb + b], i.e. an n x nrhs matrix in the library's matrix layout
template <int N>
__global__ void k_lu_solve_multi_reg(int64_t nb, int64_t nrhs, const double* __restrict__ factors, const int32_t* __restrict__ piv, double* __restrict__ rhs,
                                     unsigned long long* rec, unsigned int seq) {
  int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad = 0ull;
  if (b < nb) {
    double A[N * N];
    int P[N];
    load_mat<N>(factors, nb, b, A);
    load_piv<N>(piv, nb, b, P);
    for (int64_t r = 0; r < nrhs; ++r) {
      double v[N];
      load_vec<N>(rhs + r * N * nb, nb, b, v);
      if (!lu_solve_reg<N>(A, P, v)) bad = 1ull;
      store_vec<N>(rhs + r * N * nb, nb, b, v);
    }
  }
  block_publish(0ull, 0ull, bad, rec, seq);
}

}  // namespace

namespace {
constexpr int64_t kGbMaxN = 1024;  // the solve keeps the right-hand side in registers: 16 per lane
// factor every system's band (kl, ku) with the one-wavefront-per-system kernel; the operand is a dense container or (packed) a band container of bandwidths (pkl, pku)
int gband_factor_launch(dsh_lu* lu, const double* a, int kl, int ku, bool packed, int pkl, int pku) {
  dsh_ctx* ctx = lu->ctx;
  const int64_t n = lu->n, nb = lu->nbatch;
  const int Wb = kl + ku + 1;
  const int64_t need = n * Wb * nb;
  lu->factored = false;  // until the launches below are queued: a failure must not leave the handle claiming factors it does not have
  lu->band_k = 0;
  lu->gb_kl = lu->gb_ku = -1;
  if (lu->gb_work_len < need) {
    if (lu->gb_work) (void)dsh_free(ctx, lu->gb_work);
    lu->gb_work = nullptr; lu->gb_work_len = 0;
    if (dsh_malloc(ctx, (int64_t)sizeof(double) * need, 0, (void**)&lu->gb_work) != DSH_OK) {
      lu->gb_work = nullptr;
      set_error("dsh_lu: out of device memory for the staged band of the general banded factorisation (" + std::to_string((long long)(need >> 17)) + " MiB)");
      return DSH_E_HIP;
    }
    lu->gb_work_len = need;
  }
  const dim3 sg((unsigned)((nb + 31) / 32), (unsigned)((n * Wb + 31) / 32));
  if (packed) hipLaunchKernelGGL((k_gband_stage<true>), sg, dim3(256), 0, ctx->stream, (int)n, nb, kl, ku, a, pkl, pku, lu->gb_work);
  else hipLaunchKernelGGL((k_gband_stage<false>), sg, dim3(256), 0, ctx->stream, (int)n, nb, kl, ku, a, pkl, pku, lu->gb_work);
  const int waves = gband_factor_waves(kl, ku);
  const size_t lds = gband_window_bytes(kl, ku) * (size_t)waves;
  const unsigned grid = (unsigned)((nb + waves - 1) / waves);
  const int cpl = (Wb + 63) / 64;
#define DSH_GB_FACTOR(CPLV)                                                                                                                                       \
  do {                                                                                                                                                            \
    if (lds > (size_t)64 * 1024) {                                                                                                                                \
      static signed char attr_dev[64] = {0};                                                                                                                      \
      signed char& at = attr_dev[ctx->device & 63];                                                                                                               \
      if (at == 0) { at = hipFuncSetAttribute((const void*)k_lu_gband_factor<CPLV>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess ? 1 : -1; (void)hipGetLastError(); } \
      if (at < 0) { set_error("dsh_lu: the device refuses 72 KB of LDS for the general banded factorisation"); return DSH_E_HIP; }                               \
    }                                                                                                                                                             \
    hipLaunchKernelGGL((k_lu_gband_factor<CPLV>), dim3(grid), dim3(64 * waves), lds, ctx->stream, (int)n, nb, kl, ku, (const double*)lu->gb_work, lu->factors,    \
                       lu->pivots, lu->singular, lu->singular_epoch);                                                                                             \
  } while (0)
  if (cpl == 1) DSH_GB_FACTOR(1); else if (cpl == 2) DSH_GB_FACTOR(2); else DSH_GB_FACTOR(3);
#undef DSH_GB_FACTOR
  DSH_HIP_CHECK(hipGetLastError());
  lu->factored = true;  // (the caller has advanced lu->singular_epoch)
  lu->gb_kl = kl; lu->gb_ku = ku;
  return DSH_OK;
}
}  // namespace

extern "C" {

int dsh_lu_create(dsh_ctx* ctx, int64_t n, int64_t nbatch, dsh_lu** out) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(n >= 0 && nbatch >= 1 && out, "bad arguments");
  dsh_lu* lu = new dsh_lu();
  lu->ctx = ctx; lu->n = n; lu->nbatch = nbatch;
  lu->system_major = n > 8;
  // the factor storage (n^2 doubles per system) is allocated by the first factorisation: a solver whose ensemble is integrated by the device-resident
  // kernels never factors through this handle, and at n = 512 x 32 768 members the storage alone would be 69 GB
  DSH_HIP_CHECK(hipMalloc((void**)&lu->singular, sizeof(unsigned long long)));
  DSH_HIP_CHECK(hipMalloc((void**)&lu->band_probe, 2 * sizeof(int)));
  const char* structure_env = getenv("DSH_LU_STRUCTURE");  // read per handle: a process may make solvers of both kinds (tests/test_gpu_configs.py does)
  lu->structure = structure_env && std::string(structure_env) == "dense" ? 1 : 0;
  DSH_HIP_CHECK(hipMemsetAsync(lu->singular, 0, sizeof(unsigned long long), ctx->stream));
  *out = lu;
  return DSH_OK;
}
__attribute__((visibility("hidden"))) int lu_ensure_storage(dsh_lu* lu) {
  if (lu->factors && lu->pivots) return DSH_OK;
  const int64_t n = lu->n, nbatch = lu->nbatch;
  const int64_t per_system = lu->packed_k > 0 ? (int64_t)(3 * lu->packed_k + 1) * n : n * n;  // banded handle: U on 2K+1 diagonals, K multipliers per column
  // factor / pivot / working storage comes from the context's stream-ordered cache (dsh_malloc): a solver that is reset (a fresh .bdf() on the same problem) gets the
  // blocks of the handle it replaces back without a hipFree / hipMalloc round trip (8.6 GB each for config 3 in dense mode: seconds of driver time)
  if (!lu->factors && dsh_malloc(lu->ctx, (int64_t)sizeof(double) * (per_system * nbatch > 0 ? per_system * nbatch : 1), 0, (void**)&lu->factors) != DSH_OK) {
    lu->factors = nullptr;
    return DSH_E_HIP;
  }
  if (dsh_malloc(lu->ctx, (int64_t)sizeof(int32_t) * (n * nbatch > 0 ? n * nbatch : 1), 0, (void**)&lu->pivots) != DSH_OK) {
    (void)dsh_free(lu->ctx, lu->factors);  // all or nothing: a later call must not find half of the storage and report success
    lu->factors = nullptr; lu->pivots = nullptr;
    set_error("lu_ensure_storage: out of device memory for the pivots");
    return DSH_E_HIP;
  }
  return DSH_OK;
}
void dsh_lu_destroy(dsh_lu* lu) {
  if (!lu) return;
  (void)hipStreamSynchronize(lu->ctx->stream);
  (void)dsh_free(lu->ctx, lu->factors);
  (void)dsh_free(lu->ctx, lu->pivots);
  (void)dsh_free(lu->ctx, lu->work);
  (void)dsh_free(lu->ctx, lu->gb_work);
  (void)hipFree(lu->singular);
  (void)hipFree(lu->band_probe);
  delete lu;
}
double* dsh_lu_factors(dsh_lu* lu) { return lu_ensure_storage(lu) == DSH_OK ? lu->factors : nullptr; }
int32_t* dsh_lu_pivots(dsh_lu* lu) { return lu_ensure_storage(lu) == DSH_OK ? lu->pivots : nullptr; }
int dsh_lu_system_major(const dsh_lu* lu) { return lu->system_major ? 1 : 0; }
int dsh_lu_set_structure(dsh_lu* lu, int structure) {
  DSH_ENTER(lu ? lu->ctx : nullptr);
  DSH_REQUIRE(lu != nullptr && (structure == DSH_LU_STRUCTURE_AUTO || structure == DSH_LU_STRUCTURE_DENSE), "bad arguments");
  lu->structure = structure;
  return DSH_OK;
}
int dsh_lu_band_width(const dsh_lu* lu) { return lu->gb_kl >= 0 ? std::max(lu->gb_kl, lu->gb_ku) : lu->band_k; }

int dsh_lu_download(dsh_lu* lu, double* factors_host, int32_t* pivots_host) {
  DSH_ENTER(lu ? lu->ctx : nullptr);
  dsh_ctx* ctx = lu->ctx;
  const int64_t n = lu->n, nb = lu->nbatch;
  if (n == 0) return DSH_OK;
  { const int rc = lu_ensure_storage(lu); if (rc != DSH_OK) return rc; }
  if (lu->band_k > 0 || lu->gb_kl >= 0) { set_error("dsh_lu_download: the current factors are banded (dsh_lu_band_width); factor with DSH_LU_STRUCTURE_DENSE to download dense factors"); return DSH_E_UNSUPPORTED; }
  if (lu->system_major) {  // already [b][col][row] / [b][k]
    if (factors_host) { int rc = dsh_d2h(ctx, factors_host, lu->factors, sizeof(double) * n * n * nb); if (rc != DSH_OK) return rc; }
    if (pivots_host) { int rc = dsh_d2h(ctx, pivots_host, lu->pivots, sizeof(int32_t) * n * nb); if (rc != DSH_OK) return rc; }
    return DSH_OK;
  }
  if (factors_host) { int rc = dsh_vec_download(ctx, n * n, nb, lu->factors, factors_host); if (rc != DSH_OK) return rc; }
  if (pivots_host) {
    std::vector<int32_t> tmp((size_t)(n * nb));
    int rc = dsh_d2h(ctx, tmp.data(), lu->pivots, sizeof(int32_t) * n * nb);
    if (rc != DSH_OK) return rc;
    for (int64_t b = 0; b < nb; ++b) for (int64_t k = 0; k < n; ++k) pivots_host[b * n + k] = tmp[(size_t)(k * nb + b)];
  }
  return DSH_OK;
}

static int lu_factor_core(dsh_lu* lu, const double* a, int declared_k, int declared_kl, int declared_ku);
static int lu_factor_impl(dsh_lu* lu, const double* a, int declared_k, int declared_kl = -1, int declared_ku = -1) {
  return timed_call(lu->ctx, DSH_TIMING_LU_FACTOR, [&] { return lu_factor_core(lu, a, declared_k, declared_kl, declared_ku); });
}

int dsh_lu_factor(dsh_lu* lu, const double* a) { DSH_ENTER(lu ? lu->ctx : nullptr); return lu_factor_impl(lu, a, -1); }
// An LU handle for banded systems only: (3k + 1) n doubles of factor storage per system instead of n^2 (config 3: 13 KB instead of 2 MB; heat1d n = 512 x
// 65 536 members: 0.9 GB instead of 137 GB).  Takes dsh_lu_factor_packed with max(kl, ku) <= k; the solve is dsh_lu_solve as for any handle.
int dsh_lu_create_banded(dsh_ctx* ctx, int64_t n, int64_t nbatch, int k, dsh_lu** out) {
  DSH_ENTER(ctx);
  DSH_REQUIRE(k >= 1 && k <= kGbMaxK, "dsh_lu_create_banded: k must be in 1..64");
  DSH_REQUIRE(n >= 16, "dsh_lu_create_banded: n must be at least 16 (smaller systems use the dense kernels)");
  int rc = dsh_lu_create(ctx, n, nbatch, out);
  if (rc != DSH_OK) return rc;
  (*out)->packed_k = k;
  (*out)->structure = DSH_LU_STRUCTURE_AUTO;
  return DSH_OK;
}
// Factor a band container (dsh_mat_band_*: entry (i, j) at ((j - i + kl) n + i) nbatch + b): the eliminations of dsh_lu_factor_banded on the same entries
int dsh_lu_factor_packed(dsh_lu* lu, const double* band, int kl, int ku) {
  DSH_ENTER(lu ? lu->ctx : nullptr);
  DSH_REQUIRE(lu != nullptr && band != nullptr && kl >= 0 && ku >= 0, "bad arguments");
  const int k = std::max(1, std::max(kl, ku));
  DSH_REQUIRE(k <= kGbMaxK, "dsh_lu_factor_packed: bandwidths up to 64");
  DSH_REQUIRE(lu->n >= 16, "dsh_lu_factor_packed: n must be at least 16");
  if (lu->packed_k > 0 && k > lu->packed_k) { set_error("dsh_lu_factor_packed: the operand is wider than the handle's factor storage"); return DSH_E_INVALID; }
  dsh_ctx* ctx = lu->ctx;
  const int64_t n = lu->n, nb = lu->nbatch;
  if (k > 4) {  // general bandwidth: one wavefront per system (dsh_lu_gband.hpp)
    DSH_REQUIRE(lu->packed_k > 0 || (int64_t)(2 * kl + ku + 1) <= n, "dsh_lu_factor_packed: the band's factors do not fit a dense handle's storage");
    DSH_REQUIRE(n <= kGbMaxN, "dsh_lu_factor_packed: general bandwidths (k > 4) up to n = 1024");
    { const int rc = lu_ensure_storage(lu); if (rc != DSH_OK) return rc; }
    lu->singular_epoch += 1;
    return gband_factor_launch(lu, band, kl, ku, true, kl, ku);
  }
  { const int rc = lu_ensure_storage(lu); if (rc != DSH_OK) return rc; }
  lu->singular_epoch += 1;
  lu->factored = true;
  lu->band_k = k;
  lu->gb_kl = lu->gb_ku = -1;
  const dim3 bg = grid_for(nb, 64), bblk(64);
  switch (k) {
    case 1: hipLaunchKernelGGL((k_lu_band_factor<1, true>), bg, bblk, 0, ctx->stream, n, nb, band, lu->factors, lu->pivots, lu->singular, lu->singular_epoch, kl, ku); break;
    case 2: hipLaunchKernelGGL((k_lu_band_factor<2, true>), bg, bblk, 0, ctx->stream, n, nb, band, lu->factors, lu->pivots, lu->singular, lu->singular_epoch, kl, ku); break;
    case 3: hipLaunchKernelGGL((k_lu_band_factor<3, true>), bg, bblk, 0, ctx->stream, n, nb, band, lu->factors, lu->pivots, lu->singular, lu->singular_epoch, kl, ku); break;
    default: hipLaunchKernelGGL((k_lu_band_factor<4, true>), bg, bblk, 0, ctx->stream, n, nb, band, lu->factors, lu->pivots, lu->singular, lu->singular_epoch, kl, ku); break;
  }
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}
int dsh_lu_factor_banded(dsh_lu* lu, const double* a, int kl, int ku) {
  DSH_ENTER(lu ? lu->ctx : nullptr);
  DSH_REQUIRE(kl >= 0 && ku >= 0, "bandwidths must be non-negative");
  return lu_factor_impl(lu, a, std::max(1, std::max(kl, ku)), kl, ku);
}

static int lu_factor_core(dsh_lu* lu, const double* a, int declared_k, int declared_kl, int declared_ku) {
  if (lu->packed_k > 0) { set_error("this LU handle was made by dsh_lu_create_banded: it takes band containers (dsh_lu_factor_packed), not dense operands"); return DSH_E_UNSUPPORTED; }
  dsh_ctx* ctx = lu->ctx;
  const int64_t n = lu->n, nb = lu->nbatch;
  { const int rc = lu_ensure_storage(lu); if (rc != DSH_OK) return rc; }
  lu->singular_epoch += 1;
  lu->factored = true;
  if (n == 0) return DSH_OK;
  dim3 g = grid_for(nb, ctx->block), blk(ctx->block);
#define DSH_LU_FACTOR_CASE(N) \
  case N: hipLaunchKernelGGL((k_lu_factor_reg<N>), g, blk, 0, ctx->stream, nb, a, lu->factors, lu->pivots, lu->singular, lu->singular_epoch); break;
  switch (n) {
    DSH_LU_FACTOR_CASE(1) DSH_LU_FACTOR_CASE(2) DSH_LU_FACTOR_CASE(3) DSH_LU_FACTOR_CASE(4)
    DSH_LU_FACTOR_CASE(5) DSH_LU_FACTOR_CASE(6) DSH_LU_FACTOR_CASE(7) DSH_LU_FACTOR_CASE(8)
    default: {
      lu->band_k = 0;
      lu->gb_kl = lu->gb_ku = -1;
      static const bool check_declared = [] { const char* e = getenv("DSH_CHECK_BAND"); return e && atoi(e) != 0; }();
      if (lu->structure == DSH_LU_STRUCTURE_AUTO && n >= 16) {  // dense container of a narrow band?
        int k = declared_k, gkl = declared_kl, gku = declared_ku;  // the general banded kernels take the two bandwidths separately
        static const bool gband_on = [] { const char* e = getenv("DSH_LU_GBAND"); return !e || atoi(e) != 0; }();  // DSH_LU_GBAND=0: the dense kernels for every band wider than 4
        // a declared band wider than 4 is taken at its word by the general banded kernels (a declaration is an upper bound; DSH_CHECK_BAND verifies it)
        const bool declared_general = gband_on && k > 4 && k <= kGbMaxK && gkl >= 0 && gku >= 0 && n <= kGbMaxN && (int64_t)(2 * gkl + gku + 1) * 2 <= n;
        if (k < 0 || (k > 4 && !declared_general) || check_declared) {  // not declared by the caller (or declared too wide — a declaration is an upper bound): one read of the operand decides
          DSH_HIP_CHECK(hipMemsetAsync(lu->band_probe, 0, 2 * sizeof(int), ctx->stream));
          int64_t pblocks = (n * n * nb + 255) / 256;
          if (pblocks > 8192) pblocks = 8192;
          hipLaunchKernelGGL(k_band_probe, dim3((unsigned)pblocks), dim3(256), 0, ctx->stream, n, nb, a, lu->band_probe);
          int h[2] = {0, 0};
          DSH_HIP_CHECK(hipMemcpyAsync(h, lu->band_probe, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
          DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
          const int probed = std::max(1, std::max(h[0], h[1]));
          if (k >= 0 && probed > k) { set_error("dsh_lu_factor_banded: the operand has entries outside the declared band (DSH_CHECK_BAND)"); return DSH_E_INVALID; }
          if (k < 0 || (k > 4 && !declared_general)) { k = probed; gkl = h[0]; gku = h[1]; }
        }
        if (gband_on && k > 4 && k <= kGbMaxK && gkl >= 0 && gku >= 0 && n <= kGbMaxN && (int64_t)(2 * gkl + gku + 1) * 2 <= n) {
          const int rc = gband_factor_launch(lu, a, gkl, gku, false, 0, 0);
          if (rc != DSH_OK) return rc;
          break;
        }
        if (k <= 4) {
          lu->band_k = k;
          const dim3 bg = grid_for(nb, 64), bblk(64);
          switch (k) {
            case 1: hipLaunchKernelGGL((k_lu_band_factor<1>), bg, bblk, 0, ctx->stream, n, nb, a, lu->factors, lu->pivots, lu->singular, lu->singular_epoch); break;
            case 2: hipLaunchKernelGGL((k_lu_band_factor<2>), bg, bblk, 0, ctx->stream, n, nb, a, lu->factors, lu->pivots, lu->singular, lu->singular_epoch); break;
            case 3: hipLaunchKernelGGL((k_lu_band_factor<3>), bg, bblk, 0, ctx->stream, n, nb, a, lu->factors, lu->pivots, lu->singular, lu->singular_epoch); break;
            default: hipLaunchKernelGGL((k_lu_band_factor<4>), bg, bblk, 0, ctx->stream, n, nb, a, lu->factors, lu->pivots, lu->singular, lu->singular_epoch); break;
          }
          break;
        }
      }
      // Default for 256 <= n <= 1024: the matrix-core kernel on a row-major working copy (dsh_lu_tiled.hpp): same pivots, factors within rounding of the
      // exact kernels' (fused multiply-adds, the matrix cores' summation order).  DSH_LU_EXACT=1 keeps the kernels below, bit-identical to the CPU path;
      // DSH_LU_TILED_MIN moves the lower end (measured crossover with the exact blocked kernel: 13.3 vs 16.8 ms at 256 x 4096, 33.4 vs 19.0 ms at 320 x 4096).  Both read per call so that tests can compare.
      {
        const char* ex = getenv("DSH_LU_EXACT");
        const char* tm = getenv("DSH_LU_TILED_MIN");
        const int64_t tiled_min = tm && *tm ? atoll(tm) : 256;  // round 6: 256 (5.3 against 13.3 ms at 256 x 4096 for the exact blocked kernel, DESIGN 16.11 lead 1)
        if (!(ex && ex[0] == '1') && n >= std::max<int64_t>(tiled_min, 65) && n <= kTlMaxN) {
          const int ldw = tiled_ldw(n);
          if (!lu->work) {
            if (dsh_malloc(ctx, (int64_t)sizeof(double) * n * ldw * nb, 0, (void**)&lu->work) != DSH_OK) { (void)hipGetLastError(); lu->work = nullptr; }
          }
          if (lu->work) {  // no room for the working copy: the in-place exact kernels below
            static bool tl_attr_dev[64] = {false};
            bool& tl_attr = tl_attr_dev[ctx->device & 63];
            if (!tl_attr) {
              DSH_HIP_CHECK(hipFuncSetAttribute((const void*)tl_one::k_lu_factor_tiled<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl_one::tiled_lds_bytes(512)));
              DSH_HIP_CHECK(hipFuncSetAttribute((const void*)tl_one::k_lu_factor_tiled<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl_one::tiled_lds_bytes(1024)));
              DSH_HIP_CHECK(hipFuncSetAttribute((const void*)tl_two::k_lu_factor_tiled<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl_two::tiled_lds_bytes(512)));
              tl_attr = true;
            }
            static const bool tl_prof = [] { const char* e = getenv("DSH_LU_PHASE_PROFILE"); return e && atoi(e) != 0; }();
            unsigned long long* clk = nullptr;
            if (tl_prof) { DSH_HIP_CHECK(hipMalloc(&clk, 8 * sizeof(unsigned long long))); DSH_HIP_CHECK(hipMemset(clk, 0, 8 * sizeof(unsigned long long))); }
            const dim3 sg((unsigned)((nb + 31) / 32), (unsigned)((ldw + 31) / 32), (unsigned)n);
            hipLaunchKernelGGL(k_lu_stage_rowmajor, sg, dim3(256), 0, ctx->stream, (int)n, ldw, nb, a, lu->work);
            if (n > 512)
              hipLaunchKernelGGL((tl_one::k_lu_factor_tiled<16>), dim3((unsigned)nb), dim3(tiled_threads(n)), tiled_lds_bytes(n), ctx->stream, (int)n, ldw, lu->work, lu->factors,
                                 lu->pivots, lu->singular, lu->singular_epoch, clk);
            else if (tiled_layout(n) == 2)  // two workgroups of four wavefronts per CU
              hipLaunchKernelGGL((tl_two::k_lu_factor_tiled<8>), dim3((unsigned)nb), dim3(tiled_threads(n)), tiled_lds_bytes(n), ctx->stream, (int)n, ldw, lu->work, lu->factors,
                                 lu->pivots, lu->singular, lu->singular_epoch, clk);
            else
              hipLaunchKernelGGL((tl_one::k_lu_factor_tiled<8>), dim3((unsigned)nb), dim3(tiled_threads(n)), tiled_lds_bytes(n), ctx->stream, (int)n, ldw, lu->work, lu->factors,
                                 lu->pivots, lu->singular, lu->singular_epoch, clk);
            if (tl_prof) {
              unsigned long long h[8];
              DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
              DSH_HIP_CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
              DSH_HIP_CHECK(hipFree(clk));
              fprintf(stderr, "[dsh_lu tiled phase us, workgroup 0] panel %.1f  finish %.1f  u12 %.1f  update %.1f\n", h[0] / 100.0, h[1] / 100.0, h[2] / 100.0, h[3] / 100.0);
            }
            break;
          }
        }
      }
      // system-major copy of the operand, then factor in place
      DSH_REQUIRE((n * n + 31) / 32 <= 65535, "dsh_lu_factor: n > 1448 exceeds the launch geometry of the operand transpose (grid.y)");
      dim3 tg((unsigned)((nb + 31) / 32), (unsigned)((n * n + 31) / 32));
      hipLaunchKernelGGL(k_soa_to_aos, tg, dim3(256), 0, ctx->stream, n * n, nb, a, lu->factors);
      if (n <= 64) {  // one system per (part of a) wavefront, rows in registers
#define DSH_LU_WAVE(NP, GW)                                                                                                               \
  hipLaunchKernelGGL((k_lu_factor_wave<NP, GW>), dim3((unsigned)((nb + kWaveLuThreads / GW - 1) / (kWaveLuThreads / GW))), dim3(kWaveLuThreads), 0, \
                     ctx->stream, (int)n, nb, lu->factors, lu->pivots, lu->singular, lu->singular_epoch)
        if (n <= 16) DSH_LU_WAVE(16, 16);
        else if (n <= 32) DSH_LU_WAVE(32, 32);
        else if (n <= 48) DSH_LU_WAVE(48, 64);
        else DSH_LU_WAVE(64, 64);
#undef DSH_LU_WAVE
      } else {  // one workgroup per system
        const size_t budget = 140 * 1024;
        static bool attr_set_dev[64] = {false};  // hipFuncSetAttribute applies per device
        bool& attr_set = attr_set_dev[ctx->device & 63];
        static const int wg_threads = [] { const char* e = getenv("DSH_LU_BLOCKED_THREADS"); return e ? atoi(e) : 0; }();  // tuning knob: 256 | 512
        // measured (profiles/r01_lu_bench.md): 512 threads win for n around 512 (58 vs 71 ms at 512 x 4096), lose below 384 and at 1024
        const int threads = wg_threads == 256 || wg_threads == 512 ? wg_threads : (n >= 384 && n < 900 ? 512 : 256);
        if (!attr_set) {
          DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_lu_factor_blocked<32, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 141 * 1024));
          DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_lu_factor_blocked<32, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 141 * 1024));
          DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_lu_factor_blocked<16, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 141 * 1024));
          DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_lu_factor_blocked<16, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 141 * 1024));
          DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_lu_factor_blocked<8, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 141 * 1024));
          DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_lu_factor_blocked<8, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 141 * 1024));
          DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_lu_factor_blocked<32, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 141 * 1024));
          DSH_HIP_CHECK(hipFuncSetAttribute((const void*)k_lu_factor_blocked<32, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 141 * 1024));
          attr_set = true;
        }
        static const bool phase_profile = [] { const char* e = getenv("DSH_LU_PHASE_PROFILE"); return e && atoi(e) != 0; }();
        unsigned long long* phase_clocks = nullptr;
        if (phase_profile) { DSH_HIP_CHECK(hipMalloc(&phase_clocks, 8 * sizeof(unsigned long long))); DSH_HIP_CHECK(hipMemset(phase_clocks, 0, 8 * sizeof(unsigned long long))); }
#define DSH_LU_BLOCKED(NBK)                                                                                                                             \
  do {                                                                                                                                                  \
    if (threads == 512)                                                                                                                                 \
      hipLaunchKernelGGL((k_lu_factor_blocked<NBK, 512>), dim3((unsigned)nb), dim3(512), blocked_lds_bytes(n, NBK), ctx->stream, (int)n, nb, lu->factors, \
                         lu->pivots, lu->singular, lu->singular_epoch, phase_clocks);                                                                   \
    else                                                                                                                                                \
      hipLaunchKernelGGL((k_lu_factor_blocked<NBK, 256>), dim3((unsigned)nb), dim3(256), blocked_lds_bytes(n, NBK), ctx->stream, (int)n, nb, lu->factors, \
                         lu->pivots, lu->singular, lu->singular_epoch, phase_clocks);                                                                   \
  } while (0)
        // DSH_LU_BLOCKED_NB = 16 | 8: narrower panels than would fit (less LDS per workgroup, more workgroups per CU); same bits for every width
        static const int nb_cap = [] { const char* e = getenv("DSH_LU_BLOCKED_NB"); const int v = e ? atoi(e) : 32; return v == 8 || v == 16 ? v : 32; }();
        // opt-in: trailing update on the FP64 matrix cores (not bit-identical: tested to a tolerance).  Read per call so that tests can compare both.
        const bool mfma = [] { const char* e = getenv("DSH_LU_MFMA"); return e && e[0] == '1'; }() && n % 16 == 0 && blocked_lds_bytes(n, 32) <= budget;
        if (mfma) {
          if (threads == 512)
            hipLaunchKernelGGL((k_lu_factor_blocked<32, 512, true>), dim3((unsigned)nb), dim3(512), blocked_lds_bytes(n, 32), ctx->stream, (int)n, nb, lu->factors, lu->pivots,
                               lu->singular, lu->singular_epoch, phase_clocks);
          else
            hipLaunchKernelGGL((k_lu_factor_blocked<32, 256, true>), dim3((unsigned)nb), dim3(256), blocked_lds_bytes(n, 32), ctx->stream, (int)n, nb, lu->factors, lu->pivots,
                               lu->singular, lu->singular_epoch, phase_clocks);
        } else
        if (nb_cap >= 32 && blocked_lds_bytes(n, 32) <= budget) DSH_LU_BLOCKED(32);
        else if (nb_cap >= 16 && blocked_lds_bytes(n, 16) <= budget) DSH_LU_BLOCKED(16);
        else if (blocked_lds_bytes(n, 8) <= budget) DSH_LU_BLOCKED(8);
        else
          hipLaunchKernelGGL(k_lu_factor_global_coop, dim3((unsigned)nb), dim3(kCoopThreads), 0, ctx->stream, (int)n, nb, lu->factors, lu->pivots, lu->singular,
                             lu->singular_epoch);
#undef DSH_LU_BLOCKED
        if (phase_profile) {  // diagnostic only: synchronises
          unsigned long long h[8];
          DSH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
          DSH_HIP_CHECK(hipMemcpy(h, phase_clocks, sizeof(h), hipMemcpyDeviceToHost));
          DSH_HIP_CHECK(hipFree(phase_clocks));
          fprintf(stderr, "[dsh_lu phase us, workgroup 0] load %.1f  panel %.1f  writeback %.1f  swaps %.1f  u12 %.1f  update %.1f\n", h[0] / 100.0, h[1] / 100.0,
                  h[2] / 100.0, h[3] / 100.0, h[4] / 100.0, h[5] / 100.0);
        }
      }
    }
  }
#undef DSH_LU_FACTOR_CASE
  DSH_HIP_CHECK(hipGetLastError());
  return DSH_OK;
}

static int lu_solve_launch_core(const dsh_lu* lu, double* rhs, bool wait, unsigned int* gx_out, unsigned int* seq_out);
static int lu_solve_launch_impl(const dsh_lu* lu, double* rhs, bool wait, unsigned int* gx_out, unsigned int* seq_out) {
  return timed_call(lu->ctx, DSH_TIMING_LU_SOLVE, [&] { return lu_solve_launch_core(lu, rhs, wait, gx_out, seq_out); });
}
int dsh_lu_solve(const dsh_lu* lu, double* rhs) { DSH_ENTER(lu ? lu->ctx : nullptr); return lu_solve_launch_impl(lu, rhs, true, nullptr, nullptr); }
}  // extern "C"
namespace dsh {
// the solve enqueued only: the zero-pivot count arrives in the launch's records (fetch_records(ctx, *gx, *seq): res_cnt) — for callers that redeem it
// together with a later reduction (the staged SDIRK Newton iteration, dsh_fused.hip)
int lu_solve_launch(const dsh_lu* lu, double* rhs, unsigned int* gx, unsigned int* seq) { return lu_solve_launch_impl(lu, rhs, false, gx, seq); }
}  // namespace dsh
namespace dsh {
// The banded solve with its caller's next pass fused in (k_lu_band_solve_team<.., EPI>): x <- A^-1 x, the squared norm of x against y / atol / rtol in the launch's
// record (res_m0, next to the zero-pivot count res_cnt) and, when xin is given, xout = xin - x.  *fused = false (nothing enqueued) where the workgroup form of the
// banded solve does not apply — the caller then runs the separate launches.  Same bits either way.  DSH_LU_SOLVE_EPI=0 disables.
int lu_solve_norm_launch(const dsh_lu* lu, double* rhs, const double* xin, double* xout, const double* y, int64_t ynb, const double* atol, int64_t anb, double rtol,
                         unsigned int* gx, unsigned int* seq_out, bool* fused) {
  *fused = false;
  dsh_ctx* ctx = lu->ctx;
  if (!lu->factored) { set_error("dsh_lu_solve: LU not initialised"); return DSH_E_NOT_SETUP; }
  const int64_t n = lu->n, nb = lu->nbatch;
  static const int epi_env = [] { const char* e = std::getenv("DSH_LU_SOLVE_EPI"); return e && *e ? std::atoi(e) : 1; }();
  static const int wide_env = [] { const char* e = std::getenv("DSH_LU_BAND_WIDE"); return e && *e ? std::atoi(e) : -1; }();
  const bool wide = (wide_env >= 0 ? wide_env != 0 : (n >= 128 && nb <= 16384)) && n * nb < (1ll << 28);
  if (!epi_env || !(n > 8 && lu->band_k >= 1 && lu->band_k <= 2) || !wide || wide_env == 2 || nb > 4096) return DSH_OK;
  if (ctx->solve_mode == DSH_SOLVE_REORDERED && lu->band_k == 1 && n >= 32 && n <= 1024) return DSH_OK;  // the opt-in reordered solve keeps its own kernel
  if (!((ynb == 1 || ynb == nb) && (anb == 1 || anb == nb))) return DSH_OK;
  const size_t dyn = lu->band_k == 1 ? band_team_epi_lds_bytes<1, 16>(n) : band_team_epi_lds_bytes<2, 16>(n);
  if (dyn > (size_t)96 * 1024) return DSH_OK;  // the squares of the whole vector live in LDS: n <= 768 at 16 systems per workgroup
  {  // more than 64 KB of dynamic LDS needs the attribute, which applies per DEVICE (a process-wide flag left the second GPU of a process without it);
     // a device that refuses it takes the two launches (*fused stays false) instead of failing the launch
    static signed char attr_dev[64][2] = {};  // 0 not tried, 1 set, -1 refused
    signed char& at = attr_dev[ctx->device & 63][lu->band_k - 1];
    if (at == 0) {
      const hipError_t e = lu->band_k == 1 ? hipFuncSetAttribute((const void*)k_lu_band_solve_team<1, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)
                                           : hipFuncSetAttribute((const void*)k_lu_band_solve_team<2, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
      (void)hipGetLastError();
      at = e == hipSuccess ? 1 : -1;
    }
    if (at < 0 && dyn > (size_t)64 * 1024) return DSH_OK;
  }
  band_epi_args ea;
  ea.xin = xin; ea.xout = xout; ea.y = y; ea.atol = atol; ea.rtol = rtol; ea.by = (ynb == 1 && nb != 1) ? 1 : 0; ea.ba = (anb == 1 && nb != 1) ? 1 : 0;
  return timed_call(ctx, DSH_TIMING_LU_SOLVE, [&]() -> int {
    unsigned long long* rec; unsigned int seq;
    const dim3 g = grid_for(nb, 16);
    int rc = begin_records(ctx, g.x, &rec, &seq);
    if (rc != DSH_OK) return rc;
#define DSH_TEAM_EPI(KK)                                                                                                                                          \
  do {                                                                                                                                                            \
    hipLaunchKernelGGL((k_lu_band_solve_team<KK, 16, true>), g, dim3(kTeamThreads), dyn, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq, ea); \
  } while (0)
    if (lu->band_k == 1) DSH_TEAM_EPI(1); else DSH_TEAM_EPI(2);
#undef DSH_TEAM_EPI
    DSH_HIP_CHECK(hipGetLastError());
    *gx = g.x; *seq_out = seq; *fused = true;
    return DSH_OK;
  });
}
}  // namespace dsh
extern "C" {
static int lu_solve_launch_core(const dsh_lu* lu, double* rhs, bool wait, unsigned int* gx_out, unsigned int* seq_out) {
  dsh_ctx* ctx = lu->ctx;
  if (!lu->factored) { set_error("dsh_lu_solve: LU not initialised"); return DSH_E_NOT_SETUP; }
  const int64_t n = lu->n, nb = lu->nbatch;
  if (n == 0) return DSH_OK;
  unsigned long long* rec; unsigned int seq;
  dim3 g = grid_for(nb, ctx->block), blk(ctx->block);
  if (n > 8 && lu->gb_kl >= 0) {  // general banded factors: one wavefront per system, the right-hand side in registers
    g = dim3((unsigned)((nb + kGbSolveWaves - 1) / kGbSolveWaves));
    int rc = begin_records(ctx, g.x, &rec, &seq);
    if (rc != DSH_OK) return rc;
#define DSH_GB_SOLVE(MBV, W3V) hipLaunchKernelGGL((k_lu_gband_solve<MBV, W3V>), g, dim3(64 * kGbSolveWaves), gband_solve_lds_bytes(W3V), ctx->stream, (int)n, nb, lu->gb_kl, lu->gb_ku, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq)
    if (lu->gb_kl + lu->gb_ku > 64) { if (n <= 128) DSH_GB_SOLVE(2, true); else if (n <= 256) DSH_GB_SOLVE(4, true); else if (n <= 512) DSH_GB_SOLVE(8, true); else DSH_GB_SOLVE(16, true); }
    else { if (n <= 128) DSH_GB_SOLVE(2, false); else if (n <= 256) DSH_GB_SOLVE(4, false); else if (n <= 512) DSH_GB_SOLVE(8, false); else DSH_GB_SOLVE(16, false); }
#undef DSH_GB_SOLVE
    DSH_HIP_CHECK(hipGetLastError());
    if (!wait) { *gx_out = g.x; *seq_out = seq; return DSH_OK; }
    rc = fetch_records(ctx, g.x, seq);
    if (rc != DSH_OK) return rc;
    if (ctx->res_cnt != 0ull) {
      set_error("dsh_lu_solve: zero pivot in " + std::to_string((long long)ctx->res_cnt) + " system(s) (LuSolveFailed)");
      return DSH_E_SINGULAR;
    }
    return DSH_OK;
  }
  if (n > 8 && lu->band_k > 0) {  // banded factors: one lane per system
    // small ensembles, long chains: one workgroup per 16-64 systems, a chain wavefront fed through LDS by four loader wavefronts (k_lu_band_solve_team: same
    // bits, the memory traffic off the chain wavefront's instruction stream); DSH_LU_BAND_WIDE=2 selects the round-2 kernel instead (8 systems per wavefront,
    // k_lu_band_solve_wide); large ensembles fill the machine with one lane per system.  DSH_LU_BAND_WIDE=0 / 1 / 2 forces one of them.
    static const int wide_env = [] { const char* e = std::getenv("DSH_LU_BAND_WIDE"); return e && *e ? std::atoi(e) : -1; }();
    const bool wide = (wide_env >= 0 ? wide_env != 0 : (n >= 128 && nb <= 16384)) && n * nb < (1ll << 28);  // 32-bit byte offsets; n = 42 x 32 768: 43 us one lane per system, 171 us wide
    // opt-in (dsh_ctx_set_solve_mode): chunked affine maps, another order of the same sums — tridiagonal systems, n <= 1024
    if (ctx->solve_mode == DSH_SOLVE_REORDERED && lu->band_k == 1 && n >= 32 && n <= 1024 && nb <= 16384) {
      g = grid_for(nb, kAffSys);
      int rc = begin_records(ctx, g.x, &rec, &seq);
      if (rc != DSH_OK) return rc;
#define DSH_AFF(CLV) hipLaunchKernelGGL((k_lu_band_solve_affine<CLV>), g, dim3(kAffThreads), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq)
      if (n <= 128) DSH_AFF(8); else if (n <= 256) DSH_AFF(16); else if (n <= 512) DSH_AFF(32); else DSH_AFF(64);
#undef DSH_AFF
      DSH_HIP_CHECK(hipGetLastError());
      if (!wait) { *gx_out = g.x; *seq_out = seq; return DSH_OK; }
      rc = fetch_records(ctx, g.x, seq);
      if (rc != DSH_OK) return rc;
      if (ctx->res_cnt != 0ull) {
        set_error("dsh_lu_solve: zero pivot in " + std::to_string((long long)ctx->res_cnt) + " system(s) (LuSolveFailed)");
        return DSH_E_SINGULAR;
      }
      return DSH_OK;
    }
    const bool team = wide && wide_env != 2;
    const int sys = nb <= 4096 ? 16 : (nb <= 8192 ? 32 : 64);
    g = team ? grid_for(nb, sys) : (wide ? grid_for(nb, 8) : grid_for(nb, 64));
    int rc = begin_records(ctx, g.x, &rec, &seq);
    if (rc != DSH_OK) return rc;
    if (team) {
#define DSH_TEAM_CASE(KK, SS) hipLaunchKernelGGL((k_lu_band_solve_team<KK, SS>), g, dim3(kTeamThreads), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq)
#define DSH_TEAM_K(KK) do { if (sys == 16) DSH_TEAM_CASE(KK, 16); else if (sys == 32) DSH_TEAM_CASE(KK, 32); else DSH_TEAM_CASE(KK, 64); } while (0)
      switch (lu->band_k) {
        case 1: DSH_TEAM_K(1); break;
        case 2: DSH_TEAM_K(2); break;
        case 3: DSH_TEAM_K(3); break;
        default: DSH_TEAM_K(4); break;
      }
#undef DSH_TEAM_K
#undef DSH_TEAM_CASE
    } else if (wide) {
      switch (lu->band_k) {
        case 1: hipLaunchKernelGGL((k_lu_band_solve_wide<1, 8>), g, dim3(64), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
        case 2: hipLaunchKernelGGL((k_lu_band_solve_wide<2, 8>), g, dim3(64), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
        case 3: hipLaunchKernelGGL((k_lu_band_solve_wide<3, 8>), g, dim3(64), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
        default: hipLaunchKernelGGL((k_lu_band_solve_wide<4, 8>), g, dim3(64), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
      }
    } else
    switch (lu->band_k) {
      case 1: hipLaunchKernelGGL((k_lu_band_solve<1>), g, dim3(64), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
      case 2: hipLaunchKernelGGL((k_lu_band_solve<2>), g, dim3(64), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
      case 3: hipLaunchKernelGGL((k_lu_band_solve<3>), g, dim3(64), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
      default: hipLaunchKernelGGL((k_lu_band_solve<4>), g, dim3(64), 0, ctx->stream, n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
    }
    DSH_HIP_CHECK(hipGetLastError());
    if (!wait) { *gx_out = g.x; *seq_out = seq; return DSH_OK; }
    rc = fetch_records(ctx, g.x, seq);
    if (rc != DSH_OK) return rc;
    if (ctx->res_cnt != 0ull) {
      set_error("dsh_lu_solve: zero pivot in " + std::to_string((long long)ctx->res_cnt) + " system(s) (LuSolveFailed)");
      return DSH_E_SINGULAR;
    }
    return DSH_OK;
  }
  if (n > 64) g = dim3((unsigned)nb);  // one workgroup per system
  else if (n > 8) {
    const int per_block = kWaveLuThreads / (n <= 16 ? 16 : n <= 32 ? 32 : 64);
    g = dim3((unsigned)((nb + per_block - 1) / per_block));
  }
  int rc = begin_records(ctx, g.x, &rec, &seq);
  if (rc != DSH_OK) return rc;
#define DSH_LU_SOLVE_CASE(N) \
  case N: hipLaunchKernelGGL((k_lu_solve_reg<N>), g, blk, 0, ctx->stream, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
  switch (n) {
    DSH_LU_SOLVE_CASE(1) DSH_LU_SOLVE_CASE(2) DSH_LU_SOLVE_CASE(3) DSH_LU_SOLVE_CASE(4)
    DSH_LU_SOLVE_CASE(5) DSH_LU_SOLVE_CASE(6) DSH_LU_SOLVE_CASE(7) DSH_LU_SOLVE_CASE(8)
    default: {
      if (n <= 64) {
#define DSH_LU_WAVE(NP, GW)                                                                                                                  \
  hipLaunchKernelGGL((k_lu_solve_wave<NP, GW>), g, dim3(kWaveLuThreads), 0, ctx->stream, (int)n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, \
                     rhs, rec, seq)
        if (n <= 16) DSH_LU_WAVE(16, 16);
        else if (n <= 32) DSH_LU_WAVE(32, 32);
        else if (n <= 48) DSH_LU_WAVE(48, 64);
        else DSH_LU_WAVE(64, 64);
#undef DSH_LU_WAVE
      } else {
        static const bool blocked_solve = [] { const char* e = getenv("DSH_LU_BLOCKED_SOLVE"); return !e || atoi(e) != 0; }();
        // factor panels prefetched through a register ring (same bits) when there is at most one system per compute unit — with several workgroups per
        // CU the plain kernel's occupancy hides the latency better (512 x 4096: 1.8 ms against 4.4 ms; 962 x 256: 1.06 ms against 0.67 ms,
        // gpurun_out/r03_solve_check.txt).  DSH_LU_STREAM_SOLVE=0 keeps k_lu_solve_blocked, =2/4/6 forces the ring depth for any ensemble size.
        const int stream_env = [] { const char* e = getenv("DSH_LU_STREAM_SOLVE"); return e ? atoi(e) : -1; }();  // read per call: the GPU tier runs both kernels in one process
        const int stream_depth = stream_env >= 0 ? stream_env : (nb <= (int64_t)ctx->num_cu ? 2 : 0);
        if (blocked_solve && stream_depth > 0 && n <= 2 * kStreamThreads) {
          const size_t lds = stream_solve_lds_bytes(n);
#define DSH_STREAM(RPT, D, BB) hipLaunchKernelGGL((k_lu_solve_stream<RPT, D, BB>), g, dim3(kStreamThreads), lds, ctx->stream, (int)n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq)
          static const int stream_b = [] { const char* e = getenv("DSH_LU_STREAM_B"); return e ? atoi(e) : 8; }();
#ifdef DSH_EXPERIMENTS
          static const int stream_t = [] { const char* e = getenv("DSH_LU_STREAM_THREADS"); return e ? atoi(e) : 512; }();
          if (stream_t >= 1024 && n <= 1024) {  // 16 wavefronts, one row per thread: more loads in flight per CU (measured slower: profiles/r03_dense_solve.txt)
#define DSH_STREAM_T(D) hipLaunchKernelGGL((k_lu_solve_stream<1, D, 8, 1024>), g, dim3(1024), lds, ctx->stream, (int)n, nb, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq)
            if (stream_depth >= 4) DSH_STREAM_T(4); else if (stream_depth >= 3) DSH_STREAM_T(3); else DSH_STREAM_T(2);
#undef DSH_STREAM_T
          } else
#endif
          if (n <= kStreamThreads) {
            if (stream_b >= 32) DSH_STREAM(1, 2, 32); else if (stream_b >= 16) { if (stream_depth >= 4) DSH_STREAM(1, 4, 16); else DSH_STREAM(1, 2, 16); }
            else if (stream_depth >= 6) DSH_STREAM(1, 6, 8); else if (stream_depth >= 4) DSH_STREAM(1, 4, 8); else DSH_STREAM(1, 2, 8);
          } else {
            if (stream_b >= 32) DSH_STREAM(2, 2, 32); else if (stream_b >= 16) { if (stream_depth >= 3) DSH_STREAM(2, 3, 16); else DSH_STREAM(2, 2, 16); }
            else if (stream_depth >= 6) DSH_STREAM(2, 6, 8); else if (stream_depth >= 4) DSH_STREAM(2, 4, 8); else DSH_STREAM(2, 2, 8);
          }
#undef DSH_STREAM
        } else if (blocked_solve)
          hipLaunchKernelGGL(k_lu_solve_blocked, g, dim3(kCoopThreads), sizeof(double) * n, ctx->stream, (int)n, nb, (const double*)lu->factors,
                             (const int32_t*)lu->pivots, rhs, rec, seq);
        else
          hipLaunchKernelGGL(k_lu_solve_global_coop, g, dim3(kCoopThreads), sizeof(double) * n, ctx->stream, (int)n, nb, (const double*)lu->factors,
                             (const int32_t*)lu->pivots, rhs, rec, seq);
      }
    }
  }
#undef DSH_LU_SOLVE_CASE
  DSH_HIP_CHECK(hipGetLastError());
  // LinearSolver::solve_in_place returns Result<(), LaError>: the zero-pivot flag has to come back (blocking, like the reference's getrs loop)
  if (!wait) { *gx_out = g.x; *seq_out = seq; return DSH_OK; }
  rc = fetch_records(ctx, g.x, seq);
  if (rc != DSH_OK) return rc;
  if (ctx->res_cnt != 0ull) {
    set_error("dsh_lu_solve: zero pivot in " + std::to_string((long long)ctx->res_cnt) + " system(s) (LuSolveFailed)");
    return DSH_E_SINGULAR;
  }
  return DSH_OK;
}

int dsh_lu_solve_multi(const dsh_lu* lu, double* rhs, int64_t nrhs) {
  DSH_ENTER(lu ? lu->ctx : nullptr);
  DSH_REQUIRE(lu != nullptr && nrhs >= 0, "bad arguments");
  if (!lu->factored) { set_error("dsh_lu_solve_multi: LU not initialised"); return DSH_E_NOT_SETUP; }
  const int64_t n = lu->n, nb = lu->nbatch;
  if (n == 0 || nrhs == 0) return DSH_OK;
  if (n > 8) {  // wavefront / workgroup / banded kernels: one solve launch per column (each reads the factors once anyway)
    for (int64_t r = 0; r < nrhs; ++r) {
      int rc = dsh_lu_solve(lu, rhs + r * n * nb);
      if (rc != DSH_OK) return rc;
    }
    return DSH_OK;
  }
  dsh_ctx* ctx = lu->ctx;
  unsigned long long* rec; unsigned int seq;
  dim3 g = grid_for(nb, ctx->block), blk(ctx->block);
  int rc = begin_records(ctx, g.x, &rec, &seq);
  if (rc != DSH_OK) return rc;
#define DSH_LU_MULTI_CASE(N) \
  case N: hipLaunchKernelGGL((k_lu_solve_multi_reg<N>), g, blk, 0, ctx->stream, nb, nrhs, (const double*)lu->factors, (const int32_t*)lu->pivots, rhs, rec, seq); break;
  switch (n) {
    DSH_LU_MULTI_CASE(1) DSH_LU_MULTI_CASE(2) DSH_LU_MULTI_CASE(3) DSH_LU_MULTI_CASE(4)
    DSH_LU_MULTI_CASE(5) DSH_LU_MULTI_CASE(6) DSH_LU_MULTI_CASE(7) DSH_LU_MULTI_CASE(8)
  }
#undef DSH_LU_MULTI_CASE
  DSH_HIP_CHECK(hipGetLastError());
  rc = fetch_records(ctx, g.x, seq);
  if (rc != DSH_OK) return rc;
  if (ctx->res_cnt != 0ull) {
    set_error("dsh_lu_solve_multi: zero pivot in " + std::to_string((long long)ctx->res_cnt) + " system(s) (LuSolveFailed)");
    return DSH_E_SINGULAR;
  }
  return DSH_OK;
}

int dsh_lu_info(const dsh_lu* lu, int64_t* n_singular) {
  DSH_ENTER(lu ? lu->ctx : nullptr);
  unsigned long long h = 0;
  DSH_HIP_CHECK(hipMemcpyAsync(&h, lu->singular, sizeof(h), hipMemcpyDeviceToHost, lu->ctx->stream));
  DSH_HIP_CHECK(hipStreamSynchronize(lu->ctx->stream));
  *n_singular = (unsigned int)(h >> 32) == lu->singular_epoch ? (int64_t)(h & 0xffffffffull) : 0;
  return DSH_OK;
}

}  // extern "C"
