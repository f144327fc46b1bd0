// hiprtc translation unit tail for the wavefront-per-member SDIRK integrators of a run-time-compiled, run-time-sized model (see dsh_jit_wave_member.hpp)
#pragma once
#define DSH_JIT_DYNAMIC 1
#include "dsh_sdirk_wave_member_kernel.hpp"
