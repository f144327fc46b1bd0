// The FAST arithmetic variant of the device-resident TR-BDF2 / ESDIRK34 (k_sdirk_resident<.., FAST = true>; dsh_adaptive_options::deterministic_pow == 2) — the
// counterpart of dsh_adaptive_fast.hip for BASELINE config 5's integrator.
//
// This translation unit is compiled with -ffp-contract=fast -freciprocal-math -fapprox-func (csrc/Makefile): the same kernel source as the exact variant with
// multiply-adds fused and divisions by reciprocal + refinement instead of the IEEE sequence.  Its results are NOT bit-comparable with the oracle; the tests hold it
// to identical per-member counters and 1e-9 relative on states and event times at BASELINE config 5's full size.  It is what dshs_solve_dense launches by default
// for static models (dshs_set_resident_arithmetic); the parity tier pins the exact kernel.
#include "dsh_internal.hpp"
#include "dsh_resident.hpp"
#include "dsh_sdirk_kernel.hpp"

namespace dsh {

bool sdirk_fast_launch(int method, int model, int64_t size, bool ba, bool wave, dim3 grid, hipStream_t stream, int64_t nb, const double* p, const double* atol,
                       const SdirkConsts* consts, const double* t_eval, double* y_out, int32_t* stats, int32_t* status, double* t_root, int32_t* root_idx, int32_t* ncols,
                       unsigned long long* totals) {
  const dim3 blk(64);
  bool launched = false;
  dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    if constexpr (Mdl::N <= 4) {
#define DSH_FAST_LAUNCH(BA, WAVE, S) \
  hipLaunchKernelGGL((k_sdirk_resident<Mdl, BA, WAVE, S, false, true>), grid, blk, 0, stream, nb, p, atol, consts, t_eval, y_out, stats, status, t_root, root_idx, ncols, totals)
#define DSH_FAST_LAUNCH_S(BA, WAVE) do { if (method == 1) DSH_FAST_LAUNCH(BA, WAVE, 3); else DSH_FAST_LAUNCH(BA, WAVE, 4); } while (0)
      if (ba) { if (wave) DSH_FAST_LAUNCH_S(true, true); else DSH_FAST_LAUNCH_S(true, false); }
      else { if (wave) DSH_FAST_LAUNCH_S(false, true); else DSH_FAST_LAUNCH_S(false, false); }
#undef DSH_FAST_LAUNCH_S
#undef DSH_FAST_LAUNCH
      launched = true;
    }
  });
  return launched;
}

}  // namespace dsh
