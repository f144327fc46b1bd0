// Operand layout of v_mfma_f64_16x16x4_f64 on gfx950, found by experiment: D = A (16x4) * B (4x16) + C with one double of A and of B per lane and four
// doubles of C / D per lane.  Prints, for every (lane, register) of D, the (row, column) whose value it holds, under the hypothesis a(l) = A[l % 16][l / 16],
// b(l) = B[l / 16][l % 16]; exits 1 if the hypothesis does not explain every register.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/mfma_f64_layout.hip -o scripts/ubench/_build/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k(const double* a, const double* b, const double* c, double* d) {
  const int l = threadIdx.x;
  double4_t acc = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

int main() {
  double ha[64], hb[64], hc[256], hd[256];
  srand(7);
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0 + (rand() % 1000) / 128.0; hb[l] = 1.0 + (rand() % 1000) / 64.0; }
  for (int e = 0; e < 256; ++e) hc[e] = (rand() % 1000) * 1000.0;
  double *da, *db, *dc, *dd;
  hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, sizeof hc); hipMalloc(&dd, sizeof hd);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice); hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
  hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
  double AB[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += ha[kk * 16 + i] * hb[kk * 16 + j]; AB[i][j] = s; }
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const double v = hd[l * 4 + r] - hc[l * 4 + r];
      int fi = -1, fj = -1, n = 0;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(AB[i][j] - v) < 1e-9 * fabs(v)) { fi = i; fj = j; ++n; }
      if (n != 1) { ++bad; printf("lane %2d reg %d: %d matches (value %g)\n", l, r, n, v); }
      else if (l < 20 || l % 16 == 0) printf("lane %2d reg %d -> D[%2d][%2d]   (i == 4*(l/16)+r: %d, j == l%%16: %d)\n", l, r, fi, fj, fi == 4 * (l / 16) + r, fj == l % 16);
      if (n == 1 && !(fi == 4 * (l / 16) + r && fj == l % 16)) ++bad;
    }
  printf(bad ? "hypothesis D(l, r) = D[4*(l/16)+r][l%%16] FAILED for %d registers\n" : "layout: a(l) = A[l%%16][l/16], b(l) = B[l/16][l%%16], d(l, r) = D[4*(l/16)+r][l%%16]  (%d mismatches)\n", bad);
  return bad ? 1 : 0;
}
