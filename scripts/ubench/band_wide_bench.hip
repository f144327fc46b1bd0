// Times k_lu_band_solve_wide<1,8> / k_lu_band_solve<1> alone on synthetic tridiagonal factors (no pivoting, heat1d-like), n = 512:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include band_wide_bench.hip -o _build/band_wide_bench && _build/band_wide_bench [nb]
// and checks that both kernels return the same bits.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../diffsol_amd/csrc/dsh_lu_band.hpp"
#include "../../diffsol_amd/csrc/dsh_lu_band_team.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int64_t n = 512, nb = argc > 1 ? atoll(argv[1]) : 4096;
  constexpr int K = 1, C = 3, ROWS = 3 * K + 1;
  std::vector<double> fac((size_t)ROWS * n * nb), b((size_t)n * nb);
  std::vector<int32_t> piv((size_t)n * nb);
  uint64_t z = 12345;
  auto rnd = [&] { z = z * 6364136223846793005ull + 1442695040888963407ull; return (double)(z >> 11) / 9007199254740992.0; };
  for (int64_t j = 0; j < n; ++j)
    for (int64_t s = 0; s < nb; ++s) {
      fac[(0 * n + j) * nb + s] = 3.7 + 0.3 * rnd();    // U diagonal
      fac[(1 * n + j) * nb + s] = -1.0 + 0.1 * rnd();   // U(r, r+1)
      fac[(2 * n + j) * nb + s] = 0.0;                  // U(r, r+2): fill-in of interchanges, none here
      fac[((C + 0) * n + j) * nb + s] = -0.27 + 0.02 * rnd();  // multiplier
      piv[j * nb + s] = (int32_t)j;
      b[j * nb + s] = rnd() - 0.5;
    }
  double *dfac, *drhs, *drhs2, *drhs3; int32_t* dpiv; unsigned long long* rec;
  CK(hipMalloc(&dfac, fac.size() * 8)); CK(hipMalloc(&drhs, b.size() * 8)); CK(hipMalloc(&drhs2, b.size() * 8)); CK(hipMalloc(&drhs3, b.size() * 8)); CK(hipMalloc(&dpiv, piv.size() * 4));
  CK(hipMalloc(&rec, (size_t)(nb / 8 + 64) * dsh::kRecWords * 8));
  CK(hipMemcpy(dfac, fac.data(), fac.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dpiv, piv.data(), piv.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  for (int which = 0; which < 3; ++which) {
    double* r = which == 2 ? drhs3 : (which ? drhs2 : drhs);
    float best = 1e30f;
    for (int rep = 0; rep < reps; ++rep) {
      CK(hipMemcpy(r, b.data(), b.size() * 8, hipMemcpyHostToDevice));
      CK(hipEventRecord(e0));
      if (which == 0) hipLaunchKernelGGL((dsh::k_lu_band_solve_wide<K, 8>), dim3((nb + 7) / 8), dim3(64), 0, 0, n, nb, dfac, dpiv, r, rec, 1u);
      else if (which == 2) { if (nb <= 4096) hipLaunchKernelGGL((dsh::k_lu_band_solve_team<K, 16>), dim3((nb + 15) / 16), dim3(dsh::kTeamThreads), 0, 0, n, nb, dfac, dpiv, r, rec, 1u);
        else if (nb <= 8192) hipLaunchKernelGGL((dsh::k_lu_band_solve_team<K, 32>), dim3((nb + 31) / 32), dim3(dsh::kTeamThreads), 0, 0, n, nb, dfac, dpiv, r, rec, 1u);
        else hipLaunchKernelGGL((dsh::k_lu_band_solve_team<K, 64>), dim3((nb + 63) / 64), dim3(dsh::kTeamThreads), 0, 0, n, nb, dfac, dpiv, r, rec, 1u); }
      else hipLaunchKernelGGL((dsh::k_lu_band_solve<K>), dim3((nb + 63) / 64), dim3(64), 0, 0, n, nb, dfac, dpiv, r, rec, 1u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%s n=%lld nb=%lld: %.1f us (best of %d, HIP events)\n", which == 2 ? "team (16-64 systems / workgroup, chain + loader wavefronts)" : which ? "one lane per system" : "wide (8 systems / wavefront)", (long long)n, (long long)nb, best * 1e3, reps);
  }
  std::vector<double> x0(b.size()), x1(b.size()), x2(b.size());
  CK(hipMemcpy(x2.data(), drhs3, b.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(x0.data(), drhs, b.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x1.data(), drhs2, b.size() * 8, hipMemcpyDeviceToHost));
  printf("same bits: %s   x[0]=%.17g\n", memcmp(x0.data(), x1.data(), b.size() * 8) == 0 ? "yes" : "NO", x0[0]);
#ifdef DSH_TEAM_PROF
  { unsigned long long pr[8]; CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(dsh::g_team_prof), sizeof pr));
    printf("team prof (cycles, block 0): chain fwd busy %llu of %llu, bwd busy %llu of %llu; loader fwd busy %llu of %llu, bwd busy %llu of %llu\n", pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7]); }
  { unsigned long long st[2][8]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(dsh::g_team_stamp), sizeof st));
    for (int k = 0; k < 2; ++k) printf("team timeline %s block (us since its entry; entry of last vs first %+.2f): forward chain starts %.2f, ends %.2f, backward chain starts %.2f, ends %.2f\n", k ? "last" : "first",
      ((double)st[1][0] - (double)st[0][0]) * 0.01, (st[k][1] - st[k][0]) * 0.01, (st[k][2] - st[k][0]) * 0.01, (st[k][3] - st[k][0]) * 0.01, (st[k][4] - st[k][0]) * 0.01); }
#endif
  printf("team same bits: %s\n", memcmp(x2.data(), x1.data(), b.size() * 8) == 0 ? "yes" : "NO");
  return 0;
}
