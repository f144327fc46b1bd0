// Run-time-compiled models (hiprtc): registry and launch helper shared by the launch code of every kernel family.
#pragma once
#include <string>
#include <vector>

#include "dsh_internal.hpp"

namespace dsh {

struct JitInfo {
  int form = 0;  // DSH_JIT_FORM_*
  int64_t n = 0, np = 0, nroots = 0, nout = 0;
  int has_mass = 0;
  int has_reset = 0; // the source carries a reset operator (marker DSH_JIT_HAS_RESET)
  int64_t jac_nnz = 0; // dynamic form: structural nonzeros of f_y listed in the source (marker DSH_JIT_JAC_NNZ): the dense Jacobian is assembled from them alone
  int has_sens = 0;  // the source carries sens_mul / init_sens_mul (marker DSH_JIT_HAS_SENS written by the front end)
  int jac_kl = -1, jac_ku = -1, mass_kl = -1, mass_ku = -1;
  int twin = -1;  // the same model in the banded lane-per-member form (dsh_model_set_twin): used for device-resident per-member solves  // structural bandwidths declared with dsh_model_set_band (-1: dense / unknown)
};
inline bool is_jit_model(int model) { return model >= DSH_MODEL_JIT_BASE; }
// nullptr (+ error set) if `model` is not a live run-time-compiled model
const JitInfo* jit_info(int model);
// The kernel `name` (a C name, or a C++ name expression such as "dsh::k_jac_factor<dsh::JitModel>") of the module built from
// <model source> + #include "<header>"; `group` lists every name expression compiled into that module (one hiprtc compile per (model, group_key)).
int jit_get_function(int model, const char* header, const std::string& group_key, const std::vector<std::string>& group, const std::string& name,
                     hipFunction_t* f);

const std::vector<std::string>& jit_static_op_names();

template <class... Args>
int jit_launch(dsh_ctx* ctx, int model, const char* header, const std::string& group_key, const std::vector<std::string>& group, const std::string& name,
               dim3 grid, dim3 block, unsigned shmem, Args... args) {
  hipFunction_t f = nullptr;
  int rc = jit_get_function(model, header, group_key, group, name, &f);
  if (rc != DSH_OK) return rc;
  void* argv[] = {(void*)&args...};
  if (shmem > 64u * 1024u) { (void)hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); (void)hipGetLastError(); }  // up to the 160 KB of a CU
  DSH_HIP_CHECK(hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem, ctx->stream, argv, nullptr));
  return DSH_OK;
}

}  // namespace dsh
