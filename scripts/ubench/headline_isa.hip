// Only the headline kernel — k_bdf_adaptive<RobertsonOde<1>, broadcast atol, wavefront lock-step> — for the ISA account of profiles/r03_isa_account.md:
//   scripts/isa_account.sh   (compiles this with line tables and histograms the instructions by phase)
#include "../../diffsol_amd/csrc/dsh_adaptive_kernel.hpp"
namespace dsh {
template __global__ void k_bdf_adaptive<RobertsonOde1, true, true>(int64_t, const double*, const double*, const AdaptiveConsts*, const double*, double*, int32_t*, int32_t*,
                                                                       double*, int32_t*, int32_t*, unsigned long long*);
}
