// The opt-in FAST arithmetic variant of the device-resident BDF (k_bdf_adaptive<.., FAST = true>; dsh_adaptive_options::deterministic_pow == 2).
//
// This translation unit alone is compiled with -ffp-contract=fast -freciprocal-math -fapprox-func (csrc/Makefile): the same kernel source as the exact
// variant, with multiply-adds fused, divisions by reciprocal + refinement instead of the IEEE sequence, ocml's pow, and the Newton norm's weights as
// reciprocals.  Its results are NOT bit-comparable with the oracle (every other kernel of the library is); north_star asks for 1e-6 relative on the states,
// which the tests hold it to at tight tolerances.  It never produces bench.py's `value`: the bench reports it under an extra key.
#include "dsh_internal.hpp"
#include "dsh_resident.hpp"
#include "dsh_adaptive_kernel.hpp"

namespace dsh {

bool adaptive_fast_launch(int model, int64_t size, bool ba, bool wave, dim3 grid, hipStream_t stream, int64_t nb, const double* p, const double* atol,
                          const AdaptiveConsts* consts, const double* t_eval, double* y_out, int32_t* stats, int32_t* status, double* t_root, int32_t* root_idx,
                          int32_t* ncols, unsigned long long* totals) {
  const dim3 blk(64);
  bool launched = false;
  dispatch_static_model(model, size, [&](auto mdl) {
    using Mdl = decltype(mdl);
    if constexpr (Mdl::N <= 4) {
#define DSH_FAST_LAUNCH(BA, WAVE) \
  hipLaunchKernelGGL((k_bdf_adaptive<Mdl, BA, WAVE, false, false, true>), grid, blk, 0, stream, nb, p, atol, consts, t_eval, y_out, stats, status, t_root, root_idx, ncols, totals)
      if (wave) { if (ba) DSH_FAST_LAUNCH(true, true); else DSH_FAST_LAUNCH(false, true); }
      else { if (ba) DSH_FAST_LAUNCH(true, false); else DSH_FAST_LAUNCH(false, false); }
#undef DSH_FAST_LAUNCH
      launched = true;
    }
  });
  return launched;
}

}  // namespace dsh
