// hiprtc translation unit tail for the workgroup-per-member BDF of a run-time-compiled, run-time-sized model with 64 < n <= 140 (dsh_team_member_kernel.hpp):
// included after the generated jit_* functions (diffsol_amd/host/diffsl.hpp, Target::HipDynamic), it routes the kernel's model hooks to them.
#pragma once
#define DSH_JIT_DYNAMIC 1
#include "dsh_team_member_kernel.hpp"
