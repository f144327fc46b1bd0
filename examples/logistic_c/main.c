/* A C program against include/diffsol_c_hip.h — the reference's C API (crates/diffsol-c) served by the HIP backend.
 *
 *   gcc -O2 -I include examples/logistic_c/main.c -L diffsol_amd/lib -ldiffsol_hip_host -ldiffsol_hip -Wl,-rpath,$PWD/diffsol_amd/lib -lm -o logistic_c
 *   ./logistic_c [nmembers]
 *
 * Logistic growth y' = r y (1 - y/k) written in DiffSL, compiled for the GPU at run time, integrated (a) for one parameter set exactly as with the
 * reference API and (b) for an ensemble of growth rates in one call; every value is checked against the closed-form solution. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "diffsol_c_hip.h"

static const char* kModel =
    "in = [r, k]\n"
    "r { 1 } k { 1 }\n"
    "u_i { y = 0.1 }\n"
    "F_i { r * y * (1 - y / k) }\n";

static double exact(double r, double k, double t) { return k / (1.0 + (k / 0.1 - 1.0) * exp(-r * t)); }

#define CHECK(call)                                                                                                        \
  do {                                                                                                                     \
    if ((call) != DIFFSOL_OK) {                                                                                            \
      fprintf(stderr, "%s failed: %s (%s:%u)\n", #call, diffsol_last_error_message(), diffsol_last_error_file(), diffsol_last_error_line()); \
      return 1;                                                                                                            \
    }                                                                                                                      \
  } while (0)

int main(int argc, char** argv) {
  const size_t nb = argc > 1 ? (size_t)atol(argv[1]) : 1000;
  OdeWrapper* ode = diffsol_ode_new_jit(kModel, DIFFSOL_JIT_HIPRTC, DIFFSOL_MATRIX_HIP_DENSE, DIFFSOL_LINEAR_SOLVER_DEFAULT, DIFFSOL_ODE_SOLVER_BDF);
  if (!ode) { fprintf(stderr, "diffsol_ode_new_jit: %s\n", diffsol_last_error_message()); return 1; }
  CHECK(diffsol_ode_set_rtol(ode, 1e-8));
  CHECK(diffsol_ode_set_atol(ode, 1e-10));

  /* (a) one parameter set: solve to t = 4, every internal step comes back (ys: nstates x ncols, column-major) */
  const double p1[2] = {1.5, 2.0};
  SolutionWrapper* sol = NULL;
  CHECK(diffsol_ode_solve(ode, p1, 2, 4.0, &sol));
  HostArray *ys = NULL, *ts = NULL;
  CHECK(diffsol_solution_wrapper_get_ys(sol, &ys));
  CHECK(diffsol_solution_wrapper_get_ts(sol, &ts));
  const size_t ncols = diffsol_host_array_dim(ts, 0);
  const double* y = (const double*)diffsol_host_array_ptr(ys);
  const double* t = (const double*)diffsol_host_array_ptr(ts);
  double worst = 0.0;
  for (size_t c = 0; c < ncols; ++c) worst = fmax(worst, fabs(y[c] - exact(1.5, 2.0, t[c])) / exact(1.5, 2.0, t[c]));
  printf("single solve: %zu steps to t = %.1f, y = %.10f (exact %.10f), max relative error %.2e\n", ncols - 1, t[ncols - 1], y[ncols - 1], exact(1.5, 2.0, 4.0), worst);
  if (diffsol_host_array_ndim(ys) != 2 || worst > 1e-6) return 2;
  diffsol_host_array_free(ys); diffsol_host_array_free(ts); diffsol_solution_wrapper_free(sol);

  /* (b) an ensemble: nb growth rates, dense output at three times, every member integrated on the device with its own step sizes */
  double* p = (double*)malloc(sizeof(double) * 2 * nb);
  for (size_t b = 0; b < nb; ++b) { p[2 * b] = 0.5 + 2.0 * (double)b / (double)nb; p[2 * b + 1] = 2.0; }
  const double t_eval[3] = {0.5, 2.0, 4.0};
  CHECK(diffsol_ode_set_ensemble_mode(ode, DIFFSOL_ENSEMBLE_PER_MEMBER));
  CHECK(diffsol_ode_solve_dense(ode, p, 2 * nb, t_eval, 3, &sol));
  CHECK(diffsol_solution_wrapper_get_ys(sol, &ys));
  y = (const double*)diffsol_host_array_ptr(ys);
  const size_t s_col = diffsol_host_array_stride(ys, 1) / sizeof(double), s_b = nb > 1 ? diffsol_host_array_stride(ys, 2) / sizeof(double) : 0;
  worst = 0.0;
  for (size_t b = 0; b < nb; ++b)
    for (size_t c = 0; c < 3; ++c) {
      const double ref = exact(p[2 * b], 2.0, t_eval[c]);
      worst = fmax(worst, fabs(y[c * s_col + b * s_b] - ref) / ref);
    }
  printf("ensemble of %zu members (one launch): max relative error at t = 0.5, 2, 4: %.2e\n", nb, worst);
  diffsol_host_array_free(ys); diffsol_solution_wrapper_free(sol); free(p);
  diffsol_ode_free(ode);
  return worst < 1e-6 ? 0 : 2;
}
