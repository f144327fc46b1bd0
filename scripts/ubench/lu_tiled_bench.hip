// Stand-alone check + timing of the matrix-core dense LU (diffsol_amd/csrc/dsh_lu_tiled.hpp) without the library around it:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/lu_tiled_bench.hip -o scripts/ubench/_build/lu_tiled_bench
//   lu_tiled_bench n nbatch reps kind        kind: dense (random, every step pivots) | dd (diagonally dominant, no interchange) | sing (one zero column)
// Prints pivot mismatches and the largest factor deviation (relative to the largest entry of the factors) against a host LU with the same pivot rule for the
// distinct systems of the batch, the kernel time by HIP events, TFLOP/s at 2/3 n^3 flop per system, and the phase profile of workgroup 0.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../diffsol_amd/csrc/dsh_lu_tiled.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

static void host_lu(int n, std::vector<double>& A /* column-major */, std::vector<int>& piv) {
  piv.assign(n, 0);
  for (int k = 0; k < n; ++k) {
    double best = -1.0; int p = k;
    for (int r = k; r < n; ++r) { const double v = std::fabs(A[(size_t)k * n + r]); if (v > best) { best = v; p = r; } }
    const double diag = A[(size_t)k * n + p];
    if (diag == 0.0) { piv[k] = k; continue; }
    piv[k] = p;
    if (p != k) for (int c = 0; c < n; ++c) std::swap(A[(size_t)c * n + k], A[(size_t)c * n + p]);
    const double inv = 1.0 / diag;
    for (int r = k + 1; r < n; ++r) A[(size_t)k * n + r] *= inv;
    for (int c = k + 1; c < n; ++c) {
      const double u = A[(size_t)c * n + k];
      for (int r = k + 1; r < n; ++r) A[(size_t)c * n + r] = (-u) * A[(size_t)k * n + r] + A[(size_t)c * n + r];
    }
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  const int64_t nb = argc > 2 ? atoll(argv[2]) : 256;
  const int reps = argc > 3 ? atoi(argv[3]) : 3;
  const char* kind = argc > 4 ? argv[4] : "dense";
  const int distinct = 4;
  { const char* e = getenv("TL_STAGGER_US"); const int ticks = e ? atoi(e) * 100 : 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(dsh::tl_stagger_ticks), &ticks, sizeof ticks)); }
  const int ldw = dsh::tiled_ldw(n);
  std::mt19937_64 rng(1234 + n);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<std::vector<double>> mats(distinct, std::vector<double>((size_t)n * n));
  for (int d = 0; d < distinct; ++d) {
    for (auto& v : mats[d]) v = nd(rng);
    if (!strcmp(kind, "dd")) for (int i = 0; i < n; ++i) mats[d][(size_t)i * n + i] += 4.0 * n;
    if (!strcmp(kind, "sing") && d == 1) for (int r = 0; r < n; ++r) mats[d][(size_t)(n / 3) * n + r] = 0.0;
  }
  std::vector<double> soa((size_t)n * n * nb);
  for (int64_t b = 0; b < nb; ++b) {
    const auto& m = mats[b % distinct];
    for (size_t e = 0; e < (size_t)n * n; ++e) soa[e * nb + b] = m[e];
  }
  double *d_a, *d_w, *d_f; int32_t* d_p; unsigned long long *d_sing, *d_clk;
  CK(hipMalloc(&d_a, sizeof(double) * soa.size()));
  CK(hipMalloc(&d_w, sizeof(double) * (size_t)n * ldw * nb));
  CK(hipMalloc(&d_f, sizeof(double) * (size_t)n * n * nb));
  CK(hipMalloc(&d_p, sizeof(int32_t) * (size_t)n * nb));
  CK(hipMalloc(&d_sing, 8)); CK(hipMalloc(&d_clk, 128));
  CK(hipMemset(d_sing, 0, 8)); CK(hipMemset(d_clk, 0, 128));
  CK(hipMemset(d_f, 0xff, sizeof(double) * (size_t)n * n * nb));
  CK(hipMemcpy(d_a, soa.data(), sizeof(double) * soa.size(), hipMemcpyHostToDevice));
  const size_t lds = dsh::tiled_lds_bytes(n);
  CK(hipFuncSetAttribute((const void*)dsh::tl_one::k_lu_factor_tiled<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsh::tl_one::tiled_lds_bytes(512)));
  CK(hipFuncSetAttribute((const void*)dsh::tl_one::k_lu_factor_tiled<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsh::tl_one::tiled_lds_bytes(1024)));
  CK(hipFuncSetAttribute((const void*)dsh::tl_two::k_lu_factor_tiled<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dsh::tl_two::tiled_lds_bytes(512)));
  { int occ = 0;
    const void* kf = n > 512 ? (const void*)dsh::tl_one::k_lu_factor_tiled<16> : (dsh::tiled_layout(n) == 2 ? (const void*)dsh::tl_two::k_lu_factor_tiled<8> : (const void*)dsh::tl_one::k_lu_factor_tiled<8>);
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kf, dsh::tiled_threads(n), lds));
    printf("layout %d: %d threads, %zu B of dynamic LDS, %d workgroups per CU\n", dsh::tiled_layout(n), dsh::tiled_threads(n), lds, occ); }
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  float best_stage = 1e30f, best_factor = 1e30f;
  const dim3 sg((unsigned)((nb + 31) / 32), (unsigned)((ldw + 31) / 32), (unsigned)n);
  for (int rep = 0; rep < reps + 1; ++rep) {
    unsigned long long* clk = rep == reps ? d_clk : nullptr;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(dsh::k_lu_stage_rowmajor, sg, dim3(256), 0, 0, n, ldw, nb, (const double*)d_a, d_w);
    CK(hipEventRecord(e1));
    if (n > 512) hipLaunchKernelGGL((dsh::tl_one::k_lu_factor_tiled<16>), dim3((unsigned)nb), dim3(dsh::tiled_threads(n)), lds, 0, n, ldw, d_w, d_f, d_p, d_sing, 1u, clk);
    else if (dsh::tiled_layout(n) == 2) hipLaunchKernelGGL((dsh::tl_two::k_lu_factor_tiled<8>), dim3((unsigned)nb), dim3(dsh::tiled_threads(n)), lds, 0, n, ldw, d_w, d_f, d_p, d_sing, 1u, clk);
    else hipLaunchKernelGGL((dsh::tl_one::k_lu_factor_tiled<8>), dim3((unsigned)nb), dim3(dsh::tiled_threads(n)), lds, 0, n, ldw, d_w, d_f, d_p, d_sing, 1u, clk);
    CK(hipEventRecord(e2));
    CK(hipGetLastError());
    CK(hipEventSynchronize(e2));
    float ms_s, ms_f;
    CK(hipEventElapsedTime(&ms_s, e0, e1)); CK(hipEventElapsedTime(&ms_f, e1, e2));
    if (rep > 0 && rep < reps) { best_stage = std::min(best_stage, ms_s); best_factor = std::min(best_factor, ms_f); }
    if (reps == 1) { best_stage = ms_s; best_factor = ms_f; }
  }
  // ---- check the distinct systems at both ends of the batch
  std::vector<double> f((size_t)n * n);
  std::vector<int32_t> p(n);
  int pivbad = 0; double worst = 0.0;
  std::vector<int64_t> which;
  for (int d = 0; d < distinct && d < nb; ++d) { which.push_back(d); if (nb - 1 - d >= distinct) which.push_back(nb - 1 - d); }
  for (int64_t b : which) {
    CK(hipMemcpy(f.data(), d_f + (size_t)b * n * n, sizeof(double) * f.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(p.data(), d_p + (size_t)b * n, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    std::vector<double> ref = mats[b % distinct];
    std::vector<int> rp;
    host_lu(n, ref, rp);
    double big = 0.0;
    for (double v : ref) big = std::max(big, std::fabs(v));
    int bad = 0;
    for (int k = 0; k < n; ++k) if (rp[k] != p[k]) ++bad;
    double dev = 0.0;
    for (size_t e = 0; e < f.size(); ++e) { const double dd = std::fabs(f[e] - ref[e]); if (!(dd <= dev)) dev = dd; }
    if (bad) printf("  system %lld: %d pivot mismatches\n", (long long)b, bad);
    pivbad += bad;
    if (!(dev / big <= worst)) worst = dev / big;
  }
  unsigned long long sing = 0, clk[16];
  CK(hipMemcpy(&sing, d_sing, 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(clk, d_clk, 128, hipMemcpyDeviceToHost));
  const double flop = 2.0 / 3.0 * n * (double)n * n * nb;
  printf("n=%d nb=%lld %s: stage %.3f ms  factor %.3f ms  %.2f TFLOP/s (kernel)  %.2f TFLOP/s (with staging)   pivots wrong %d  max dev %.3e  singular %llu\n", n,
         (long long)nb, kind, best_stage, best_factor, flop / best_factor / 1e9, flop / (best_factor + best_stage) / 1e9, pivbad, worst,
         (unsigned long long)(sing & 0xffffffffull));
  printf("  phases of workgroup 0 (us): panel %.1f  finish+lists %.1f  u12 %.1f  update %.1f\n", clk[0] / 100.0, clk[1] / 100.0, clk[2] / 100.0, clk[3] / 100.0);
  printf("  inside the panel (us): U' solve %.1f  stage %.1f  pivot steps %.1f  flush %.1f\n", clk[4] / 100.0, clk[5] / 100.0, clk[6] / 100.0, clk[7] / 100.0);
  printf("  inside stage (us): loads+B-update %.1f  T writes %.1f  CO reads %.1f\n", clk[8] / 100.0, clk[9] / 100.0, clk[10] / 100.0);
#ifdef TL_X_STEPPROF
  { unsigned long long sp[8]; CK(hipMemcpyFromSymbol(sp, HIP_SYMBOL(dsh::tl_stepprof), sizeof sp));
    printf("  step profile (shader cycles over %d launches, thread 0 of workgroup 0): reads %llu  pivot-row entries %llu  update/search %llu  barrier %llu; searches by thread 0: %llu cycles in %llu calls\n", reps + 1, sp[0], sp[1], sp[2], sp[3], sp[4], sp[5]); }
#endif
  return (pivbad == 0 && worst < 1e-10) ? 0 : 1;
}
