// The register-resident LU of the workgroup-per-member integrators (csrc/dsh_team_reg_lu.hpp: 2 x 2 wavefronts, a 64 x 64 block per wavefront, a block row per lane)
// against the LDS-resident form it replaces (team_lu_factor<2> / team_lu_solve<2>, dsh_team_member_kernel.hpp): same bits (factors, permutation, solutions), time per
// factorisation and per solve with one system per workgroup.
//   scripts/ubench/build_team_reg_lu.sh && scripts/ubench/_build/team_reg_lu_bench <n> <systems> <dense|dd|sing|ties> [nsolve]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../diffsol_amd/csrc/dsh_internal.hpp"
#include "../../diffsol_amd/csrc/dsh_resident.hpp"
#include "../../diffsol_amd/csrc/dsh_wave_member_kernel.hpp"
#include "../../diffsol_amd/csrc/dsh_team_member_kernel.hpp"
#include "../../diffsol_amd/csrc/dsh_team_reg_lu.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace dsh;
#ifndef TRG_NL
#define TRG_NL 128
#endif
#ifndef TRG_WPE
#define TRG_WPE 2
#endif

// M: [sys][c * n + r]; B: [sys][s][n]; X likewise; F: factors [sys][c * n + r] at positions; PM: [sys][n]
__global__ __launch_bounds__(128) void k_ref(int n, int nfac, int nsol, const double* __restrict__ M, const double* __restrict__ B, double* __restrict__ X, double* __restrict__ F,
                                             int* __restrict__ PM, int* __restrict__ SG) {
  extern __shared__ double lds[];
  constexpr int P = team_pitch_w(2);
  double* xch = lds;
  double* cand = lds + 128;
  int* perm = reinterpret_cast<int*>(lds + 132);
  double* A = lds + 132 + 64;
  const int ln = threadIdx.x;
  const bool rowlive = ln < n;
  const size_t s = blockIdx.x;
  bool singular = false;
  for (int f = 0; f < nfac; ++f) {
    __syncthreads();
    if (rowlive) for (int j = 0; j < n; ++j) A[j * P + ln] = M[s * n * n + (size_t)j * n + ln];
    team_lu_factor<2>(A, P, n, ln, rowlive, cand, perm, singular);
  }
  if (rowlive) { for (int j = 0; j < n; ++j) F[s * n * n + (size_t)j * n + ln] = A[j * P + ln]; PM[s * n + ln] = perm[ln]; }
  if (ln == 0) SG[s] = singular;
  for (int q = 0; q < nsol; ++q) {
    double v = rowlive ? B[(s * nsol + q) * n + ln] : 0.0;
    team_lu_solve<2>(A, P, n, ln, rowlive, perm, xch, singular, v);
    if (rowlive) X[(s * nsol + q) * n + ln] = v;
  }
}

__global__ __launch_bounds__(trg_threads(TRG_NL), TRG_WPE) void k_reg(int n, int nfac, int nsol, const double* __restrict__ M, const double* __restrict__ B, double* __restrict__ X,
                                                     double* __restrict__ F, int* __restrict__ PM, int* __restrict__ SG) {
  extern __shared__ double w[];
  constexpr int RBN = trg_rbn(TRG_NL);
  const int tid = threadIdx.x, row = tid & (64 * RBN - 1), h = tid >> (5 + RBN);
  const bool rowlive = row < n;
  const size_t s = blockIdx.x;
  double a[64], dself = 1.0, rself = 1.0;
  bool singular = false;
  for (int f = 0; f < nfac; ++f) {
#pragma unroll
    for (int j = 0; j < 64; ++j) a[j] = (rowlive && trg_gcol(h, j) < n) ? M[s * n * n + (size_t)trg_gcol(h, j) * n + row] : 0.0;  // the elimination's column layout
#ifndef TRG_NO_FACTOR
    team_reg_lu_factor<TRG_NL>(a, n, tid, w, singular, dself, rself);
#endif
  }
  if (rowlive) {
#pragma unroll
    for (int j = 0; j < 64; ++j) if (64 * h + j < n) F[s * n * n + (size_t)(64 * h + j) * n + row] = a[j];
    if (h == 0) PM[s * n + row] = reinterpret_cast<int*>(w + kTrgOffPerm)[row];
  }
  if (tid == 0) SG[s] = singular;
  for (int q = 0; q < nsol; ++q) {
    double v = rowlive ? B[(s * nsol + q) * n + row] : 0.0;
#ifndef TRG_NO_SOLVE
    team_reg_lu_solve<TRG_NL>(a, n, tid, w, singular, dself, rself, v);
#endif
    if (rowlive && h == 0) X[(s * nsol + q) * n + row] = v;
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 120;
  const int nb = argc > 2 ? atoi(argv[2]) : 256;
  const std::string kind = argc > 3 ? argv[3] : "dense";
  const int nsol = argc > 4 ? atoi(argv[4]) : 4;
  if (n < 9 || n > TRG_NL) { printf("8 < n <= %d\n", TRG_NL); return 2; }
  std::vector<double> M((size_t)nb * n * n), B((size_t)nb * nsol * n);
  uint64_t z = 987654321ull + n;
  auto rnd = [&] { z = z * 6364136223846793005ull + 1442695040888963407ull; return (double)(z >> 11) / 9007199254740992.0 - 0.5; };
  for (int s = 0; s < nb; ++s)
    for (int c = 0; c < n; ++c)
      for (int r = 0; r < n; ++r) {
        double v = rnd();
        if (kind == "ties") v = (double)((int)(v * 6.0));  // small integers: many equal magnitudes in the pivot search, exact zeros
        if (kind == "dd" && r == c) v += (double)n;
        M[(size_t)s * n * n + (size_t)c * n + r] = v;
      }
  if (kind == "sing")
    for (int s = 0; s < nb; s += 3)
      for (int r = 0; r < n; ++r) M[(size_t)s * n * n + (size_t)(s % n) * n + r] = 0.0;  // a zero column
  for (auto& v : B) v = rnd();
  double *dM, *dB, *dX[2], *dF[2]; int *dP[2], *dS[2];
  CK(hipMalloc(&dM, M.size() * 8)); CK(hipMalloc(&dB, B.size() * 8));
  for (int q = 0; q < 2; ++q) { CK(hipMalloc(&dX[q], B.size() * 8)); CK(hipMalloc(&dF[q], M.size() * 8)); CK(hipMalloc(&dP[q], (size_t)nb * n * 4)); CK(hipMalloc(&dS[q], (size_t)nb * 4));
    CK(hipMemset(dX[q], 0, B.size() * 8)); CK(hipMemset(dF[q], 0, M.size() * 8)); }
  CK(hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice));
  const size_t ref_lds = (size_t)(132 + 64 + (size_t)n * team_pitch_w(2)) * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ref), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ref_lds));
  const size_t reg_lds = (size_t)trg_lds_doubles(TRG_NL) * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_reg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)reg_lds));
  { int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_reg, trg_threads(TRG_NL), reg_lds)); printf("registers form: %zu bytes of LDS, %d workgroups per CU\n", reg_lds, occ); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int which, int nfac, int ns) -> float {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_ref, dim3(nb), dim3(128), ref_lds, 0, n, nfac, ns, dM, dB, dX[0], dF[0], dP[0], dS[0]);
      else hipLaunchKernelGGL(k_reg, dim3(nb), dim3(trg_threads(TRG_NL)), reg_lds, 0, n, nfac, ns, dM, dB, dX[1], dF[1], dP[1], dS[1]);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
  };
  for (int which = 0; which < 2; ++which) {
    const float t1 = run(which, 1, 0), t5 = run(which, 5, 0), ts = run(which, 1, nsol);
    CK(hipDeviceSynchronize());
    printf("%s n=%d systems=%d %s: factor %.1f us per launch of one factorisation each ((5x - 1x)/4 = %.1f us), %d solves + 1 factorisation %.1f us -> %.1f us per solve\n",
           which ? (trg_rbn(TRG_NL) == 2 ? "registers (4 wavefronts)" : "registers (2 wavefronts)") : "LDS       (2 wavefronts)", n, nb, kind.c_str(), t1 * 1e3, (t5 - t1) / 4 * 1e3, nsol, ts * 1e3, (ts - t1) / nsol * 1e3);
  }
#ifdef DSH_TRG_PROF
  { unsigned long long pr[4][8]; CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(dsh::g_trg), sizeof pr));
    const double steps = 35.0 * n;  // factorisations of workgroup 0 in the launches above x pivots
    for (int q = 0; q < 4; ++q) printf("  cycles per pivot step, wavefront %d (rb %d, h %d): publish + search %.0f | LDS writes done %.0f | barrier %.0f | candidates, interchange, multiplier %.0f | update %.0f\n", q, q & 1, q >> 1,
                                       pr[q][0] / steps, pr[q][2] / steps, pr[q][1] / steps, pr[q][3] / steps, pr[q][4] / steps);
    unsigned long long ps[4][8]; CK(hipMemcpyFromSymbol(ps, HIP_SYMBOL(dsh::g_trs), sizeof ps));
    const double solves = 5.0 * nsol;
    for (int q = 0; q < 4; ++q) printf("  cycles per solve, wavefront %d (rb %d, h %d): gather P b %.0f | fwd block 0 %.0f | rows 64.. take y[0..63] %.0f | fwd + back block 1 %.0f | rows 0..63 take x[64..] %.0f | back block 0 %.0f | barriers and waits %.0f\n", q, q & 1, q >> 1,
                                       ps[q][0] / solves, ps[q][1] / solves, ps[q][3] / solves, ps[q][4] / solves, ps[q][5] / solves, ps[q][6] / solves, ps[q][2] / solves); }
#endif
  std::vector<double> X0(B.size()), X1(B.size()), F0(M.size()), F1(M.size());
  std::vector<int> P0((size_t)nb * n), P1((size_t)nb * n), S0(nb), S1(nb);
  CK(hipMemcpy(X0.data(), dX[0], B.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(X1.data(), dX[1], B.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(F0.data(), dF[0], M.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(F1.data(), dF[1], M.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(P0.data(), dP[0], P0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(P1.data(), dP[1], P1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(S0.data(), dS[0], S0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(S1.data(), dS[1], S1.size() * 4, hipMemcpyDeviceToHost));
  int nsing = 0, nswaps = 0;
  for (int s = 0; s < nb; ++s) nsing += S0[s];
  for (size_t i = 0; i < P0.size(); ++i) nswaps += P0[i] != (int)(i % n);
  const bool okF = memcmp(F0.data(), F1.data(), M.size() * 8) == 0, okP = P0 == P1, okS = S0 == S1, okX = memcmp(X0.data(), X1.data(), B.size() * 8) == 0;
  printf("  same bits: factors %s, permutation %s, singular flags %s (%d singular), solutions %s   (rows off their place: %d of %zu; x[0] = %.17g)\n", okF ? "yes" : "NO",
         okP ? "yes" : "NO", okS ? "yes" : "NO", nsing, okX ? "yes" : "NO", nswaps, P0.size(), X0[0]);
  if (!okF) { for (size_t i = 0; i < M.size(); ++i) if (memcmp(&F0[i], &F1[i], 8)) { printf("  first factor mismatch: system %zu column %zu row %zu: %.17g vs %.17g\n", i / ((size_t)n * n), (i / n) % n, i % n, F0[i], F1[i]); break; } }
  return (okF && okP && okS && okX) ? 0 : 1;
}
