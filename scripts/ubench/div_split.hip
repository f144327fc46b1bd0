// Is an IEEE FP64 division, as hipcc expands it for gfx950 (v_div_scale x2, v_rcp, 4 FMAs, mul, fma, v_div_fmas, v_div_fixup), reproduced bit for bit
// by "refined reciprocal of the denominator, then mul / fma / fma on the numerator" whenever no operand scaling is needed?  The denominator half can then
// be computed off a dependent chain (dsh_lu_band.hpp, backward sweep).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off div_split.hip -o div_split
#include <hip/hip_runtime.h>
#include "../../diffsol_amd/csrc/dsh_device.hpp"
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ inline uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
// mode 0: random mantissas, exponents uniform in [-E, E];  1: mantissas near all-ones / all-zeros;  2: x = y * small integer ratio (exact or halfway quotients);
// 3: raw random bit patterns with zeros / infinities / NaNs / denormals mixed in (the range guard must send them to the ordinary division)
__global__ void k(uint64_t seed, int E, int mode, int test, unsigned long long* nbad, double* ex, unsigned long long* nfast) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long bad = 0, fast = 0;
  for (int it = 0; it < 4096; ++it) {
    uint64_t a = mix(seed + i * 8192 + 2 * it), b = mix(seed + i * 8192 + 2 * it + 1);
    uint64_t mx = a & 0xfffffffffffffull, my = b & 0xfffffffffffffull;
    if (mode == 1) { int sh = (a >> 56) % 52; mx = (a & 1) ? (0xfffffffffffffull >> sh << sh) & 0xfffffffffffffull : (mx >> sh); sh = (b >> 56) % 52; my = (b & 1) ? 0xfffffffffffffull - (my >> sh) : (my >> sh); }
    int ex_ = (int)((a >> 52) % (2 * E + 1)) - E, ey = (int)((b >> 52) % (2 * E + 1)) - E;
    uint64_t bx = ((uint64_t)(1023 + ex_) << 52) | mx | ((a >> 63) << 63), by = ((uint64_t)(1023 + ey) << 52) | my | ((b >> 63) << 63);
    double x = __longlong_as_double((long long)bx), y = __longlong_as_double((long long)by);
    if (mode == 2) { double m = (double)(1 + (a >> 40) % 4097); x = y * m; if (a & 2) x = __longlong_as_double(__double_as_longlong(x) + (long long)((a >> 8) % 3) - 1); }
    if (mode == 3) {
      x = __longlong_as_double((long long)a); y = __longlong_as_double((long long)b);
      const double sp[8] = {0.0, -0.0, __builtin_inf(), -__builtin_inf(), __builtin_nan(""), 4.9e-324, -2.2e-308, 1.0};
      if ((a & 0x30) == 0) x = sp[(a >> 8) & 7];
      if ((b & 0x30) == 0) y = sp[(b >> 8) & 7];
    }
    const double ref = x / y;
    const double r = dsh::div_refined_rcp(y);
    const double sp = dsh::div_by_refined(x, y, r);
    // test 0: operands in range (div_split_ok);  test 1: denominator in its narrower range and the QUOTIENT in range (div_den_ok, div_quot_ok)
    const bool ok = test == 0 ? dsh::div_split_ok(x, y) : (dsh::div_den_ok(y) & dsh::div_quot_ok(sp));
    fast += ok;
    const double got = ok ? sp : ref;
    if (__double_as_longlong(ref) != __double_as_longlong(got)) { if (!bad && atomicAdd(nbad, 0ull) == 0) { ex[0] = x; ex[1] = y; ex[2] = ref; ex[3] = got; } ++bad; }
  }
  if (bad) atomicAdd(nbad, bad);
  atomicAdd(nfast, fast);
}

int main() {
  unsigned long long *nbad, *nfast; double* ex;
  (void)hipMalloc(&nbad, 8); (void)hipMalloc(&nfast, 8); (void)hipMalloc(&ex, 32);
  for (int test = 0; test < 2; ++test)
  for (int mode = 0; mode < 4; ++mode)
    for (int E : {8, 60, 300, 380, 1000}) {
      (void)hipMemset(nbad, 0, 8); (void)hipMemset(nfast, 0, 8);
      for (int rep = 0; rep < 4; ++rep) k<<<4096, 256>>>(0x1234567ull * (rep + 1) + mode, E, mode, test, nbad, ex, nfast);
      unsigned long long h, hf; double hx[4];
      (void)hipMemcpy(&h, nbad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hf, nfast, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hx, ex, 32, hipMemcpyDeviceToHost);
      printf("test %d mode %d  |exponent| <= %3d : %llu mismatches of %.3g, %.1f %% through the split form", test, mode, E, h, 4.0 * 4096 * 256 * 4096, 100.0 * hf / (4.0 * 4096 * 256 * 4096));
      if (h) printf("   e.g. x=%a y=%a  x/y=%a  split=%a", hx[0], hx[1], hx[2], hx[3]);
      printf("\n");
    }
  return 0;
}
