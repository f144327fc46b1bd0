// Run-time-sized models of the registry (one thread per state component): shared by the element-wise kernels of dsh_models.hip and the
// wavefront-per-member integrator (dsh_wave_member.hip).  Expressions follow the reference closures literally (cited per model).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/diffsol_hip.h"
#include "../../include/diffsol_detpow.h"

namespace dsh {

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
  }
  return 0.0;
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
