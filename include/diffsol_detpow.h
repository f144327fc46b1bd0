/* diffsol_detpow.h — a DETERMINISTIC pow(x, y) for positive x, built only from IEEE-754 basic operations (+ - * /, floor, bit manipulation) in a fixed
 * order: compiled without FMA contraction it returns the same bits on the host (gcc) and on the device (hipcc, gfx950).
 *
 * Why: the device-resident integrators (dsh_bdf_solve_adaptive, dsh_sdirk_solve_resident, dsh_bdf_solve_wave_member) take every step-size / convergence
 * decision on the GPU, where pow() is ocml's, while the CPU restatement of the reference uses libm's — equal to within an ulp, not bitwise.  With
 * dsh_adaptive_options.deterministic_pow = 1 the kernels call this function instead, and the oracle can be switched to it too (orc_set_det_pow): then the
 * two must agree BIT FOR BIT, which verifies every line of the device-side control logic against the restatement.  It is a verification vehicle, not a
 * better pow: accuracy is a few ulp (|y ln x| <~ 50), special cases are handled only as far as the integrators need them (0, inf, NaN, x = 1, y = 0).
 */
#ifndef DIFFSOL_DETPOW_H
#define DIFFSOL_DETPOW_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define DSH_DETPOW_FN __host__ __device__ inline
#else
#define DSH_DETPOW_FN inline
#endif

DSH_DETPOW_FN uint64_t dsh_detpow_bits(double x) { uint64_t u; memcpy(&u, &x, sizeof(u)); return u; }
DSH_DETPOW_FN double dsh_detpow_from_bits(uint64_t u) { double x; memcpy(&x, &u, sizeof(x)); return x; }

/* 2^k for -1022 <= k <= 1023 */
DSH_DETPOW_FN double dsh_detpow_exp2i(int k) { return dsh_detpow_from_bits((uint64_t)(k + 1023) << 52); }

/* nearest integer to z / ln 2 (ties irrelevant), via floor: deterministic on both sides */
DSH_DETPOW_FN double dsh_detpow_floor_half(double z) {
  const double t = z * 1.4426950408889634 + 0.5;
  double f = (double)(long long)t; /* truncation toward zero */
  if (f > t) f -= 1.0;              /* -> floor */
  return f;
}

DSH_DETPOW_FN double dsh_det_pow(double x, double y) {
  if (y == 0.0 || x == 1.0) return 1.0;
  if (x != x || y != y) return x + y; /* NaN */
  const double inf = dsh_detpow_from_bits(0x7ff0000000000000ull);
  if (x < 0.0) return dsh_detpow_from_bits(0x7ff8000000000000ull);
  if (x == 0.0) return y > 0.0 ? 0.0 : inf;
  if (x == inf) return y > 0.0 ? inf : 0.0;
  if (y == inf) return x > 1.0 ? inf : 0.0;
  if (y == -inf) return x > 1.0 ? 0.0 : inf;
  /* x = m 2^e, m in [sqrt(1/2), sqrt(2)) */
  uint64_t u = dsh_detpow_bits(x);
  int e = (int)((u >> 52) & 0x7ff);
  if (e == 0) { /* subnormal: scale up by 2^54 (exact) */
    x = x * 18014398509481984.0;
    u = dsh_detpow_bits(x);
    e = (int)((u >> 52) & 0x7ff) - 54;
  }
  e -= 1022; /* x = m 2^e with m in [0.5, 1) */
  double m = dsh_detpow_from_bits((u & 0x000fffffffffffffull) | 0x3fe0000000000000ull);
  if (m < 0.70710678118654757) { m = m * 2.0; e -= 1; }
  /* ln m = 2 s (1 + s^2/3 + s^4/5 + ...), s = (m-1)/(m+1), |s| <= 0.1716 */
  const double s = (m - 1.0) / (m + 1.0);
  const double s2 = s * s;
  double p = 1.0 / 25.0;
  p = p * s2 + 1.0 / 23.0;
  p = p * s2 + 1.0 / 21.0;
  p = p * s2 + 1.0 / 19.0;
  p = p * s2 + 1.0 / 17.0;
  p = p * s2 + 1.0 / 15.0;
  p = p * s2 + 1.0 / 13.0;
  p = p * s2 + 1.0 / 11.0;
  p = p * s2 + 1.0 / 9.0;
  p = p * s2 + 1.0 / 7.0;
  p = p * s2 + 1.0 / 5.0;
  p = p * s2 + 1.0 / 3.0;
  /* ln m = 2s + c, c = 2s (s^2 p) small; ln x = e ln2_hi (exact: ln2_hi has 21 trailing zero bits) + [2s + (c + e ln2_lo)] as a hi/lo pair */
  const double two_s = 2.0 * s;
  const double c = two_s * (s2 * p);
  const double ln2_hi = 0.69314718036912382, ln2_lo = 1.9082149292705877e-10;
  const double ed = (double)e;
  const double a1 = ed * ln2_hi;
  /* two_sum(a1, two_s) */
  double l_hi = a1 + two_s;
  double bb = l_hi - a1;
  double l_lo = (a1 - (l_hi - bb)) + (two_s - bb);
  l_lo = l_lo + (c + ed * ln2_lo);
  /* the division behind s carries a rounding error of its own: recover it, s_err = ((m-1) - s (m+1)) / (m+1) with Dekker's exact product */
  {
    const double den = m + 1.0, num = m - 1.0; /* num is exact; den may round by at most half an ulp, which is far below what matters here */
    const double sp = 134217729.0 * s, s_h = sp - (sp - s), s_l = s - s_h;
    const double dp = 134217729.0 * den, d_h = dp - (dp - den), d_l = den - d_h;
    const double prod = s * den;
    const double perr = ((s_h * d_h - prod) + s_h * d_l + s_l * d_h) + s_l * d_l; /* s*den = prod + perr exactly */
    const double s_err = ((num - prod) - perr) / den;
    l_lo = l_lo + 2.0 * s_err;
  }
  /* z = y (l_hi + l_lo) as a hi/lo pair (Dekker product for y l_hi) */
  const double yp = 134217729.0 * y, y_h = yp - (yp - y), y_l = y - y_h;
  const double lp = 134217729.0 * l_hi, lh_h = lp - (lp - l_hi), lh_l = l_hi - lh_h;
  const double z_hi = y * l_hi;
  const double z_lo = (((y_h * lh_h - z_hi) + y_h * lh_l + y_l * lh_h) + y_l * lh_l) + y * l_lo;
  const double z = z_hi;
  if (z > 709.0) return inf;
  if (z < -745.0) return 0.0;
  /* exp z = 2^k exp r, r = z - k ln2, |r| <= 0.35 */
  const double kf = dsh_detpow_floor_half(z);
  const int k = (int)kf;
  const double r = ((z_hi - kf * ln2_hi) - kf * ln2_lo) + z_lo;
  double q = 1.0 / 6227020800.0; /* 1/13! */
  q = q * r + 1.0 / 479001600.0;
  q = q * r + 1.0 / 39916800.0;
  q = q * r + 1.0 / 3628800.0;
  q = q * r + 1.0 / 362880.0;
  q = q * r + 1.0 / 40320.0;
  q = q * r + 1.0 / 5040.0;
  q = q * r + 1.0 / 720.0;
  q = q * r + 1.0 / 120.0;
  q = q * r + 1.0 / 24.0;
  q = q * r + 1.0 / 6.0;
  q = q * r + 0.5;
  const double er = 1.0 + (r + (r * r) * q);
  /* scale in two exact steps so that k may leave the normal exponent range of one factor */
  const int k1 = k / 2, k2 = k - k1;
  return (er * dsh_detpow_exp2i(k1)) * dsh_detpow_exp2i(k2);
}

#endif /* DIFFSOL_DETPOW_H */
