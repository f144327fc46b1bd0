// Run-time-sized models of the registry (one thread per state component): shared by the element-wise kernels of dsh_models.hip and the
// wavefront-per-member integrator (dsh_wave_member.hip).  Expressions follow the reference closures literally (cited per model).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/diffsol_hip.h"
#include "../../include/diffsol_detpow.h"

namespace dsh {

__host__ __device__ inline int64_t dyn_isqrt(int64_t v) { int64_t r = (int64_t)(sqrt((double)v) + 0.5); while (r * r > v) --r; while ((r + 1) * (r + 1) <= v) ++r; return r; }

// value of component i of f(x) (or of J(x) v when `v` is given) for the dynamic models; X(i)/V(i) read system b
template <class XF, class VF, class PF>
__device__ __forceinline__ double dyn_component(int model, int64_t n, double t, int64_t i, XF X, VF V, PF P, bool jac) {
  switch (model) {
    case DSH_MODEL_DYDT_Y2:  // test_models/dydt_y2.rs:9-19
      return jac ? V(i) * X(i) * 2.0 : X(i) * X(i);
    case DSH_MODEL_GAUSSIAN_DECAY:  // test_models/gaussian_decay.rs:12-23
      return (jac ? V(i) : X(i)) * P(i) * (-t);
    case DSH_MODEL_HEAT1D: {  // test_models/heat1d.rs:16-52 : D*(A u)/h^2, A = tridiag(1,-2,1), h = 1/(n+1)
      double h = 1.0 / (double)(n + 1);
      auto U = [&](int64_t k) { return jac ? V(k) : X(k); };
      double left = i > 0 ? U(i - 1) : 0.0, right = i + 1 < n ? U(i + 1) : 0.0;
      double heat = left + (-2.0) * U(i) + right;
      return P(0) * heat / (h * h);
    }
    case DSH_MODEL_SPM: {  // book/src/primer/src/spm.ds F_i: (I/3600, |I|/3600, A_neg x_neg + flux_neg e_last, A_pos x_pos + flux_pos e_last)
      const int64_t m = (n - 2) / 2;
      if (i < 2) return jac ? 0.0 : (i == 0 ? 0.0002777777777777778 * P(0) : 0.0002777777777777778 * fabs(P(0)));
      const bool pos = i >= 2 + m;
      const int64_t base = pos ? 2 + m : 2, k = i - base;
      auto U = [&](int64_t q) { return jac ? V(base + q) : X(base + q); };
      // spherical finite-volume Laplacian on m uniform shells times D/R^2 (constant6_ij / constant7_ij of spm.ds)
      const double s = pos ? 1.0e-3 : 0.39e-3, dr = 1.0 / (double)m, i0 = (double)k, i1 = (double)(k + 1);
      const double vol = i1 * i1 * i1 - i0 * i0 * i0;
      const double lower = 3.0 * i0 * i0 / vol / (dr * dr) * s, upper = k + 1 < m ? 3.0 * i1 * i1 / vol / (dr * dr) * s : 0.0;
      double acc = (-(lower + upper)) * U(k);
      if (k > 0) acc += lower * U(k - 1);
      if (k + 1 < m) acc += upper * U(k + 1);
      if (!jac && k == m - 1) acc += (pos ? 4.106800547504748e-12 * 243644455.17866704 : 3.2835305549534856e-12 * -520607810.21082705) * P(0);
      return acc;
    }
    case DSH_MODEL_HEAT2D: {  // test_models/heat2d.rs:105-123 (rhs), :125-149 (jac_mul): 5-point differences on an m x m grid, boundary rows res_i = u_i
      const int64_t m = dyn_isqrt(n), jy = i / m, ix = i % m;
      auto U = [&](int64_t k) { return jac ? V(k) : X(k); };
      if (jy == 0 || jy == m - 1 || ix == 0 || ix == m - 1) return U(i);  // y.copy_from(x): the boundary equations
      const double mm = (double)m, four = 4.0;
      const double dx = 1.0 / (mm - 1.0);
      const double coeff = P(0) / (dx * dx);  // P(0) = 1: the reference's 1 / (dx dx)
      return coeff * (U(i - 1) + U(i + 1) + U(i - m) + U(i + m) - four * U(i));
    }
    case DSH_MODEL_FOODWEB: {  // test_models/foodweb.rs:419-494 (rhs), :502-582 (jac_mul): species interleaved, Neumann boundaries by mirror points
      const int64_t nx = dyn_isqrt(n / 2), nsmx = 2 * nx;
      const int64_t is = i & 1, loc = i - is, jx = (loc / 2) % nx, jy = (loc / 2) / nx;
      const double dx = 1.0 / ((double)nx - 1.0), dy = 1.0 / ((double)nx - 1.0);
      const double yy = (double)jy * dy, xx = (double)jx * dx;
      const int64_t idyu = jy != nx - 1 ? nsmx : -nsmx, idyl = jy != 0 ? nsmx : -nsmx, idxu = jx != nx - 1 ? 2 : -2, idxl = jx != 0 ? 2 : -2;
      const int64_t locxu = loc + idxu, locxl = loc - idxl, locyu = loc + idyu, locyl = loc - idyl;
      const double AA = 1.0, EE = 10000.0, GG = 0.5e-6, BB = 1.0, DPREY = 1.0, DPRED = 0.05;
      const double a0 = is == 0 ? -AA : EE, a1 = is == 0 ? -GG : -AA;  // acoef[is][0], acoef[is][1]
      const double bco = is == 0 ? BB : -BB;
      const double cox = (is == 0 ? DPREY : DPRED) / (dx * dx), coy = (is == 0 ? DPREY : DPRED) / (dy * dy);
      double dp = 0.0, ddp = 0.0;
      dp += a0 * X(loc); dp += a1 * X(loc + 1);
      if (jac) { ddp += a0 * V(loc); ddp += a1 * V(loc + 1); }
      const double fac = 1.0 + P(0) * xx * yy + P(1) * dsh_det_sin(4.0 * 3.14159265358979323846 * xx) * dsh_det_sin(4.0 * 3.14159265358979323846 * yy);
      const double rate = jac ? X(i) * ddp + V(i) * (bco * fac + dp) : X(i) * (bco * fac + dp);
      auto U = [&](int64_t k) { return jac ? V(k) : X(k); };
      const double dcyli = U(i) - U(locyl + is), dcyui = U(locyu + is) - U(i);
      const double dcxli = U(i) - U(locxl + is), dcxui = U(locxu + is) - U(i);
      return coy * (dcyui - dcyli) + cox * (dcxui - dcxli) + rate;
    }
    case DSH_MODEL_ROBERTSON_ODE: {  // test_models/robertson_ode.rs:71-90
      int64_t g = (i / 3) * 3, r = i % 3;
      if (!jac) {
        if (r == 0) return -P(0) * X(g) + P(1) * X(g + 1) * X(g + 2);
        if (r == 1) return P(0) * X(g) - P(1) * X(g + 1) * X(g + 2) - P(2) * X(g + 1) * X(g + 1);
        return P(2) * X(g + 1) * X(g + 1);
      }
      if (r == 0) return -P(0) * V(g) + P(1) * V(g + 1) * X(g + 2) + P(1) * X(g + 1) * V(g + 2);
      if (r == 1) return P(0) * V(g) - P(1) * V(g + 1) * X(g + 2) - P(1) * X(g + 1) * V(g + 2) - 2.0 * P(2) * X(g + 1) * V(g + 1);
      return 2.0 * P(2) * X(g + 1) * V(g + 1);
    }
  }
  return 0.0;
}


// initial value of component i (Init::call_inplace of the registry models)
__device__ __forceinline__ double dyn_init_value(int model, int64_t n, int64_t i) {
  switch (model) {
    case DSH_MODEL_DYDT_Y2: return -200.0;
    case DSH_MODEL_GAUSSIAN_DECAY: return 1.0;
    case DSH_MODEL_HEAT1D: { double h = 1.0 / (double)(n + 1); double xx = (double)(i + 1) * h; return xx < 0.5 ? 2.0 * xx : 2.0 * (1.0 - xx); }
    case DSH_MODEL_ROBERTSON_ODE: return (i % 3 == 0) ? 1.0 : 0.0;
    case DSH_MODEL_SPM: return i < 2 ? 0.0 : (i < 2 + (n - 2) / 2 ? 0.8000000000000016 : 0.6000000000000001);  // spm.ds u_i
    case DSH_MODEL_HEAT2D: {  // heat2d.rs:151-183
      const int64_t m = dyn_isqrt(n), jy = i / m, ix = i % m;
      if (jy == 0 || jy == m - 1 || ix == 0 || ix == m - 1) return 0.0;
      const double mm = (double)m, one = 1.0, sixteen = 16.0;
      const double dx = one / (mm - one);
      const double yfact = dx * (double)jy, xfact = dx * (double)ix;
      return sixteen * xfact * (one - xfact) * yfact * (one - yfact);
    }
    case DSH_MODEL_FOODWEB: {  // foodweb.rs:342-366: prey from a polynomial, predators flat 1e5 (corrected by the consistent initialisation)
      if (i & 1) return 1.0e5;
      const int64_t nx = dyn_isqrt(n / 2), jx = (i / 2) % nx, jy = (i / 2) / nx;
      const double dx = 1.0 / ((double)nx - 1.0), dy = 1.0 / ((double)nx - 1.0);
      const double yy = (double)jy * dy, xx = (double)jx * dx;
      double xyfactor = 16.0 * xx * (1.0 - xx) * yy * (1.0 - yy);
      xyfactor = xyfactor * xyfactor;
      return 10.0 + 1.0 * xyfactor;
    }
  }
  return 0.0;
}

// diagonal of the mass matrix of the run-time-sized registry models that have one (heat2d.rs:185-205: boundary rows algebraic; foodweb.rs:639-655: predators algebraic)
__device__ __forceinline__ double dyn_mass_diag(int model, int64_t n, int64_t i) {
  switch (model) {
    case DSH_MODEL_HEAT2D: { const int64_t m = dyn_isqrt(n), jy = i / m, ix = i % m; return (jy == 0 || jy == m - 1 || ix == 0 || ix == m - 1) ? 0.0 : 1.0; }
    case DSH_MODEL_FOODWEB: return (i & 1) ? 0.0 : 1.0;
  }
  return 1.0;
}

// ---- terminal voltage of the single-particle model (spm.ds varying2..5, out_i) and its two stop conditions
__device__ __forceinline__ double spm_clamp(double v, double lo, double hi) { return v < hi ? (v > lo ? v : lo) : hi; }
__device__ inline double spm_ocp_pos(double s) {
  return 2.16216 + 0.07645 * dsh_det_tanh(30.834 - 57.858397200000006 * s) + 2.1581 * dsh_det_tanh(52.294 - 53.412228 * s) - 0.14169 * dsh_det_tanh(11.0923 - 21.0852666 * s) +
         0.2051 * dsh_det_tanh(1.4684 - 5.829105600000001 * s) + 0.2531 * dsh_det_tanh(4.291641337386018 - 8.069908814589667 * s) - 0.02167 * dsh_det_tanh(-87.5 + 177.0 * s) +
         1e-06 * (1.0 / s + 1.0 / (-1.0 + s));
}
__device__ inline double spm_ocp_neg(double s) {
  return 0.194 + 1.5 * dsh_det_exp(-120.0 * s) + 0.0351 * dsh_det_tanh(-3.44578313253012 + 12.048192771084336 * s) - 0.0045 * dsh_det_tanh(-7.1344537815126055 + 8.403361344537815 * s) -
         0.035 * dsh_det_tanh(-18.466 + 20.0 * s) - 0.0147 * dsh_det_tanh(-14.705882352941176 + 29.41176470588235 * s) - 0.102 * dsh_det_tanh(-1.3661971830985917 + 7.042253521126761 * s) -
         0.022 * dsh_det_tanh(-54.8780487804878 + 60.975609756097555 * s) - 0.011 * dsh_det_tanh(-5.486725663716814 + 44.24778761061947 * s) +
         0.0155 * dsh_det_tanh(-3.6206896551724133 + 34.48275862068965 * s) + 1e-06 * (1.0 / s + 1.0 / (-1.0 + s));
}
__device__ inline double spm_voltage(double neg_in, double neg_out, double pos_in, double pos_out, double current) {
  const double sp = -0.4999999999999983 * pos_in + 1.4999999999999982 * pos_out, sn = -0.4999999999999983 * neg_in + 1.4999999999999984 * neg_out;
  const double cp = spm_clamp(-25608.96286546366 * pos_in + 76826.88859639116 * pos_out, 0.000512179257309275, 51217.92521874824);
  const double cn = spm_clamp(-12491.630996921805 * neg_in + 37474.892990765504 * neg_out, 0.000249832619938437, 24983.261744011077);
  const double stp = spm_clamp(sp, 1e-10, 0.9999999999), stn = spm_clamp(sn, 1e-10, 0.9999999999);
  const double eta_p = 0.05138515824298745 * dsh_det_asinh((-2.3508116177110145 * current) / (2.0 * ((1.8973665961010275e-05 * sqrt(cp)) * sqrt(51217.9257309275 - cp))));
  const double eta_n = 0.05138515824298745 * dsh_det_asinh((1.9590096814258458 * current) / (2.0 * ((0.0006324555320336759 * sqrt(cn)) * sqrt(24983.2619938437 - cn))));
  return (eta_p + spm_ocp_pos(stp)) - (eta_n + spm_ocp_neg(stn));
}

// root functions of one system (at most 2); returns their number
template <class XF, class PF>
__device__ __forceinline__ int dyn_root_values(int model, int64_t n, double t, XF X, PF P, double (&g)[2]) {
  if (model == DSH_MODEL_SPM) {
    const int64_t m = (n - 2) / 2;
    const double v = spm_voltage(X(2 + m - 2), X(2 + m - 1), X(2 + 2 * m - 2), X(2 + 2 * m - 1), P(0));
    g[0] = -3.105 + v;
    g[1] = 4.1 - v;
    return 2;
  }
  return 0;
}

}  // namespace dsh
