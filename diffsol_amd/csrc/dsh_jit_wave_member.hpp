// hiprtc translation unit tail for the wavefront-per-member integrator of a run-time-compiled, run-time-sized model: included after the generated
// jit_* functions (diffsol_amd/host/diffsl.hpp, Target::HipDynamic), it routes the kernel's model hooks to them.
#pragma once
#define DSH_JIT_DYNAMIC 1
#include "dsh_wave_member_kernel.hpp"
