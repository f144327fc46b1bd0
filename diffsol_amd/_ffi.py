"""ctypes declarations for the two C ABIs.  The declared symbol lists are the single source the tests check against
include/*.h (every symbol the headers declare must be exported by the built libraries)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_DIR = os.environ.get("DSH_LIB_DIR") or os.path.join(_HERE, "lib")  # DSH_LIB_DIR: another build of both libraries (occupancy experiments, scripts/occupancy_experiment.sh)

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
i64 = C.c_int64
dbl = C.c_double
vp = C.c_void_p
cint = C.c_int


class DiffsolHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


def lib_paths():
    return os.path.join(_LIB_DIR, "libdiffsol_hip.so"), os.path.join(_LIB_DIR, "libdiffsol_hip_host.so")


# name -> (restype, argtypes); mirrors include/diffsol_hip.h
DEVICE_ABI = {
    "dsh_last_error": (C.c_char_p, []),
    "dsh_version": (cint, []),
    "dsh_ctx_create": (cint, [cint, vp, C.POINTER(vp)]),
    "dsh_ctx_destroy": (None, [vp]),
    "dsh_ctx_sync": (cint, [vp]),
    "dsh_ctx_bind_thread": (cint, [vp]),
    "dsh_ctx_stream": (vp, [vp]),
    "dsh_ctx_device": (cint, [vp]),
    "dsh_ctx_set_block": (cint, [vp, cint]),
    "dsh_experiments_enabled": (cint, []),
    "dsh_dist_shard_bounds": (cint, [i64, cint, cint, c_i64p, c_i64p]),
    "dsh_dist_unique_id": (cint, [C.c_char_p]),
    "dsh_dist_init": (cint, [vp, cint, cint, C.c_char_p, C.POINTER(vp)]),
    "dsh_dist_destroy": (None, [vp]),
    "dsh_dist_rank": (cint, [vp]),
    "dsh_dist_world": (cint, [vp]),
    "dsh_gather_batch_axis": (cint, [vp, vp, i64, i64, vp]),
    "dsh_gather_batch_axis_async": (cint, [vp, vp, i64, i64, vp]),
    "dsh_gather_wait": (cint, [vp]),
    "dsh_dist_pack_shard": (cint, [vp, vp, vp, i64, i64, i64, vp]),
    "dsh_dist_unpack_gathered": (cint, [vp, vp, vp, i64, i64, cint, vp]),
    "dsh_ctx_set_timing": (cint, [vp, cint]),
    "dsh_ctx_set_timing_target": (cint, [vp, cint]),
    "dsh_ctx_set_solve_mode": (cint, [vp, cint]),
    "dsh_ctx_get_solve_mode": (cint, [vp]),
    "dsh_ctx_set_poll": (cint, [vp, cint]),
    "dsh_ctx_get_timing": (cint, [vp, c_i64p, c_dp]),
    "dsh_ctx_get_timing_overhead": (cint, [vp, c_dp, c_dp]),
    "dsh_malloc": (cint, [vp, i64, cint, C.POINTER(vp)]),
    "dsh_free": (cint, [vp, vp]),
    "dsh_memset_zero": (cint, [vp, vp, i64]),
    "dsh_h2d": (cint, [vp, vp, vp, i64]),
    "dsh_d2h": (cint, [vp, vp, vp, i64]),
    "dsh_d2d": (cint, [vp, vp, vp, i64]),
    "dsh_vec_upload": (cint, [vp, i64, i64, c_dp, vp]),
    "dsh_vec_download": (cint, [vp, i64, i64, vp, c_dp]),
    "dsh_vec_get_index": (cint, [vp, i64, vp, i64, i64, c_dp]),
    "dsh_vec_set_index": (cint, [vp, i64, vp, i64, i64, dbl]),
    "dsh_vec_extract_batch": (cint, [vp, i64, i64, vp, i64, vp]),
    "dsh_permute_members": (cint, [vp, i64, i64, cint, vp, vp, vp]),
    "dsh_vec_insert_batch": (cint, [vp, i64, i64, vp, i64, vp]),
    "dsh_vec_set_index_all": (cint, [vp, i64, vp, i64, dbl]),
    "dsh_vec_add": (cint, [vp, i64, i64, vp, i64, vp, i64, vp]),
    "dsh_vec_sub": (cint, [vp, i64, i64, vp, i64, vp, i64, vp]),
    "dsh_vec_add_assign": (cint, [vp, i64, i64, vp, vp, i64]),
    "dsh_vec_sub_assign": (cint, [vp, i64, i64, vp, vp, i64]),
    "dsh_vec_mul_assign": (cint, [vp, i64, i64, vp, vp, i64]),
    "dsh_vec_div_assign": (cint, [vp, i64, i64, vp, vp, i64]),
    "dsh_vec_mul_assign_scalar": (cint, [vp, i64, i64, vp, dbl]),
    "dsh_vec_mul_scalar": (cint, [vp, i64, i64, vp, dbl, vp]),
    "dsh_vec_axpy": (cint, [vp, i64, i64, dbl, vp, i64, dbl, vp]),
    "dsh_vec_axpby_to": (cint, [vp, i64, i64, dbl, vp, dbl, vp, vp, vp]),
    "dsh_vec_batched_axpy": (cint, [vp, i64, i64, c_dp, vp, i64, dbl, vp]),
    "dsh_vec_copy": (cint, [vp, i64, i64, vp, i64, vp]),
    "dsh_vec_fill": (cint, [vp, i64, i64, vp, dbl]),
    "dsh_vec_gather": (cint, [vp, i64, i64, vp, vp, i64, vp]),
    "dsh_vec_scatter": (cint, [vp, i64, i64, vp, vp, i64, vp]),
    "dsh_vec_copy_from_indices": (cint, [vp, i64, i64, vp, vp, i64, vp]),
    "dsh_vec_assign_at_indices": (cint, [vp, i64, i64, vp, i64, dbl, vp]),
    "dsh_vec_norm": (cint, [vp, i64, i64, vp, cint, c_dp]),
    "dsh_vec_squared_norm": (cint, [vp, i64, i64, vp, vp, i64, vp, i64, dbl, c_dp, vp]),
    "dsh_vec_root_finding": (cint, [vp, i64, i64, vp, vp, c_ip, c_dp, c_ip]),
    "dsh_mat_from_diagonal": (cint, [vp, i64, i64, vp, i64, vp]),
    "dsh_mat_band_from_diagonal": (cint, [vp, i64, i64, cint, cint, vp, i64, vp]),
    "dsh_mat_band_gemv": (cint, [vp, i64, i64, cint, cint, dbl, vp, vp, i64, dbl, vp]),
    "dsh_mat_get_diagonal": (cint, [vp, i64, i64, vp, vp]),
    "dsh_mat_set_column": (cint, [vp, i64, i64, i64, vp, i64, vp, i64]),
    "dsh_mat_scale_add_assign": (cint, [vp, i64, i64, vp, vp, i64, dbl, vp, i64]),
    "dsh_mat_set_data_with_indices": (cint, [vp, i64, i64, i64, vp, vp, vp, i64, vp]),
    "dsh_mat_column_axpy": (cint, [vp, i64, i64, vp, dbl, i64, i64]),
    "dsh_mat_gemv": (cint, [vp, i64, i64, i64, dbl, vp, i64, vp, i64, dbl, vp]),
    "dsh_mat_gemv_from": (cint, [vp, i64, i64, i64, dbl, vp, i64, vp, i64, dbl, vp, vp]),
    "dsh_mat_gemm": (cint, [vp, i64, i64, i64, i64, dbl, vp, i64, vp, i64, dbl, vp]),
    "dsh_lu_create": (cint, [vp, i64, i64, C.POINTER(vp)]),
    "dsh_lu_destroy": (None, [vp]),
    "dsh_lu_factor": (cint, [vp, vp]),
    "dsh_lu_solve": (cint, [vp, vp]),
    "dsh_lu_solve_squared_norm": (cint, [vp, vp, vp, i64, vp, i64, dbl, vp]),
    "dsh_lu_info": (cint, [vp, c_i64p]),
    "dsh_lu_factors": (vp, [vp]),
    "dsh_lu_pivots": (vp, [vp]),
    "dsh_lu_system_major": (cint, [vp]),
    "dsh_lu_download": (cint, [vp, c_dp, c_i32p]),
    "dsh_model_info": (cint, [cint, i64, c_i64p, c_i64p, c_ip, c_i64p]),
    "dsh_model_rhs": (cint, [vp, cint, i64, i64, dbl, vp, vp, vp]),
    "dsh_model_jac_mul": (cint, [vp, cint, i64, i64, dbl, vp, vp, vp, vp]),
    "dsh_model_jacobian": (cint, [vp, cint, i64, i64, dbl, vp, vp, vp]),
    "dsh_model_has_band_jacobian": (cint, [cint, i64]),
    "dsh_model_jacobian_band": (cint, [vp, cint, i64, i64, dbl, vp, vp, cint, cint, vp]),
    "dsh_model_jacobian_band_packed": (cint, [vp, cint, i64, i64, dbl, vp, vp, cint, cint, vp]),
    "dsh_model_mass_gemv": (cint, [vp, cint, i64, i64, dbl, vp, vp, dbl, vp]),
    "dsh_model_mass_matrix": (cint, [vp, cint, i64, i64, dbl, vp, vp]),
    "dsh_model_init": (cint, [vp, cint, i64, i64, dbl, vp, vp]),
    "dsh_lu_solve_multi": (cint, [vp, vp, i64]),
    "dsh_lu_set_structure": (cint, [vp, cint]),
    "dsh_lu_band_width": (cint, [vp]),
    "dsh_model_root": (cint, [vp, cint, i64, i64, dbl, vp, vp, vp]),
    "dsh_model_has_reset": (cint, [cint, i64]),
    "dsh_model_reset": (cint, [vp, cint, i64, i64, dbl, vp, vp, vp]),
    "dsh_model_has_sens": (cint, [cint, i64]),
    "dsh_model_rhs_sens": (cint, [vp, cint, i64, i64, dbl, vp, vp, vp]),
    "dsh_model_init_sens": (cint, [vp, cint, i64, i64, dbl, vp, vp]),
    "dsh_model_out": (cint, [vp, cint, i64, i64, dbl, vp, vp, vp]),
    "dsh_model_compile": (cint, [C.c_char_p, cint, i64, i64, i64, i64, cint, c_ip]),
    "dsh_model_release": (cint, [cint]),
    "dsh_model_precompile": (cint, [cint, cint]),
    "dsh_jit_compile_count": (C.c_int64, []),
    "dsh_jit_replay": (cint, [C.c_char_p, cint, cint, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dsh_model_set_member_twin_source": (cint, [cint, C.c_char_p, i64, i64, i64, i64]),
    "dsh_model_member_twin": (cint, [cint]),
    "dsh_model_set_twin": (cint, [cint, cint]),
    "dsh_model_twin": (cint, [cint]),
    "dsh_model_lane_twin": (cint, [cint, i64]),
    "dsh_model_set_band": (cint, [cint, cint, cint, cint, cint]),
    "dsh_model_band": (cint, [cint, i64, c_ip, c_ip, c_ip, c_ip]),
    "dsh_lu_factor_banded": (cint, [vp, vp, cint, cint]),
    "dsh_lu_create_banded": (cint, [vp, i64, i64, cint, C.POINTER(vp)]),
    "dsh_lu_factor_packed": (cint, [vp, vp, cint, cint]),
    "dsh_mat_scale_add_assign_banded": (cint, [vp, i64, i64, cint, cint, vp, vp, i64, dbl, vp, i64]),
    "dsh_bdf_newton_iter": (cint, [vp, cint, i64, i64, dbl, dbl, vp, vp, vp, vp, vp, vp, vp, vp, i64, dbl, c_dp]),
    "dsh_bdf_newton_iter_async": (cint, [vp, cint, i64, i64, dbl, dbl, cint, vp, vp, vp, vp, vp, vp, vp, vp, i64, dbl, c_i64p]),
    "dsh_reduction_wait": (cint, [vp, i64, c_dp]),
    "dsh_sdirk_begin_attempt": (cint, [vp, i64, i64, dbl, dbl, vp, vp, vp, vp, vp]),
    "dsh_sdirk_next_stage": (cint, [vp, i64, i64, cint, dbl, vp, vp, vp, vp, vp, c_dp, dbl, dbl]),
    "dsh_sdirk_finish_error": (cint, [vp, i64, i64, cint, dbl, vp, vp, vp, vp, c_dp, vp]),
    "dsh_sdirk_newton_iter": (cint, [vp, cint, i64, i64, dbl, dbl, dbl, vp, vp, vp, vp, vp, vp, vp, i64, dbl, c_dp]),
    "dsh_sdirk_newton_iter_async": (cint, [vp, cint, i64, i64, dbl, dbl, dbl, cint, vp, vp, vp, vp, vp, vp, vp, i64, dbl, c_i64p]),
    "dsh_jac_factor": (cint, [vp, cint, i64, i64, dbl, dbl, vp, vp, cint, vp, vp, vp]),
    "dsh_model_has_fused": (cint, [cint, i64]),
    "dsh_model_has_staged_newton": (cint, [cint, i64]),
    "dsh_model_has_adaptive": (cint, [cint, i64]),
    "dsh_adaptive_default_options": (None, [vp]),
    "dsh_model_has_wave_member": (cint, [cint, i64]),
    "dsh_bdf_solve_wave_member": (cint, [vp, cint, i64, i64, vp, vp, i64, dbl, dbl, dbl, vp, c_dp, i64, vp, vp, vp, vp, vp, vp, c_i64p]),
    "dsh_model_has_wave_member_sdirk": (cint, [cint, i64]),
    "dsh_sdirk_solve_wave_member": (cint, [vp, cint, i64, cint, i64, vp, vp, i64, dbl, dbl, dbl, vp, c_dp, i64, vp, vp, vp, vp, vp, vp, c_i64p]),
    "dsh_model_has_resident": (cint, [cint, cint, i64]),
    "dsh_sdirk_solve_resident": (cint, [vp, cint, cint, i64, i64, vp, vp, i64, dbl, dbl, dbl, vp, c_dp, i64, vp, vp, vp, vp, vp, vp, c_i64p]),
    "dsh_model_has_adaptive_sens": (cint, [cint, i64]),
    "dsh_model_has_adaptive_reset": (cint, [cint, i64]),
    "dsh_sdirk_solve_resident_sens": (cint, [vp, cint, cint, i64, i64, vp, vp, i64, dbl, dbl, dbl, vp, c_dp, i64, dbl, c_dp, i64, vp, vp, vp, vp, c_i64p]),
    "dsh_model_has_wave_member_sens": (cint, [cint, i64]),
    "dsh_model_has_wave_member_reset": (cint, [cint, i64]),
    "dsh_sdirk_solve_wave_member_sens": (cint, [vp, cint, i64, cint, i64, vp, vp, i64, dbl, dbl, dbl, vp, c_dp, i64, dbl, c_dp, i64, vp, vp, vp, vp, c_i64p]),
    "dsh_bdf_solve_wave_member_sens": (cint, [vp, cint, i64, i64, vp, vp, i64, dbl, dbl, dbl, vp, c_dp, i64, dbl, c_dp, i64, vp, vp, vp, vp, c_i64p]),
    "dsh_bdf_solve_adaptive_sens": (cint, [vp, cint, i64, i64, vp, vp, i64, dbl, dbl, dbl, vp, c_dp, i64, dbl, c_dp, i64, vp, vp, vp, vp, c_i64p]),
    "dsh_model_has_adaptive_steps": (cint, [cint, i64]),
    "dsh_bdf_solve_adaptive_steps": (cint, [vp, cint, i64, i64, vp, vp, i64, dbl, dbl, dbl, vp, dbl, i64, vp, vp, vp, vp, vp, vp, vp, c_i64p]),
    "dsh_sdirk_solve_resident_steps": (cint, [vp, cint, cint, i64, i64, vp, vp, i64, dbl, dbl, dbl, vp, dbl, i64, vp, vp, vp, vp, vp, vp, vp, c_i64p]),
    "dsh_bdf_solve_wave_member_steps": (cint, [vp, cint, i64, i64, vp, vp, i64, dbl, dbl, dbl, vp, dbl, i64, vp, vp, vp, vp, vp, vp, vp, c_i64p]),
    "dsh_sdirk_solve_wave_member_steps": (cint, [vp, cint, i64, cint, i64, vp, vp, i64, dbl, dbl, dbl, vp, dbl, i64, vp, vp, vp, vp, vp, vp, vp, c_i64p]),
    "dsh_bdf_solve_adaptive": (cint, [vp, cint, i64, i64, vp, vp, i64, dbl, dbl, dbl, vp, c_dp, i64, vp, vp, vp, vp, vp, vp, c_i64p]),
    "dsh_bdf_prepare_step": (cint, [vp, i64, i64, cint, vp, vp, c_dp, c_dp, dbl, vp, vp]),
    "dsh_bdf_accept_newton_async": (cint, [vp, cint, i64, i64, cint, dbl, vp, vp, vp, vp, vp, vp, i64, dbl, c_dp, dbl, vp, dbl, dbl, cint, vp, vp, vp, c_i64p, c_i64p]),
    "dsh_bdf_accept_step_async": (cint, [vp, i64, i64, cint, dbl, vp, vp, vp, vp, vp, vp, i64, dbl, c_dp, dbl, vp, c_i64p]),
    "dsh_bdf_accept_step": (cint, [vp, i64, i64, cint, dbl, vp, vp, vp, vp, vp, vp, i64, dbl, c_dp, dbl, vp, cint, c_dp]),
}


class DshsOptions(C.Structure):
    _fields_ = [
        ("max_nonlinear_solver_iterations", cint), ("max_error_test_failures", cint), ("max_nonlinear_solver_failures", cint),
        ("nonlinear_solver_tolerance", dbl), ("min_timestep", dbl), ("update_jacobian_after_steps", cint),
        ("update_rhs_jacobian_after_steps", cint), ("threshold_to_update_jacobian", dbl), ("threshold_to_update_rhs_jacobian", dbl),
        ("ic_use_linesearch", cint), ("use_fused_kernels", cint), ("block_threads", cint),
        ("ic_max_linesearch_iterations", cint), ("ic_max_linear_solver_setups", cint), ("ic_max_newton_iterations", cint),
        ("ic_step_reduction_factor", dbl), ("ic_armijo_constant", dbl),
    ]


# mirrors include/diffsol_hip_solver.h
HOST_ABI = {
    "dshs_last_error": (C.c_char_p, []),
    "dshs_default_options": (None, [C.POINTER(DshsOptions)]),
    "dshs_create": (cint, [cint, vp, cint, i64, i64, c_dp, i64, dbl, c_dp, i64, dbl, dbl, cint, C.POINTER(DshsOptions), C.POINTER(vp)]),
    "dshs_create_sens": (cint, [cint, vp, cint, i64, i64, c_dp, i64, dbl, c_dp, i64, dbl, dbl, cint, C.POINTER(DshsOptions), cint, dbl, c_dp, i64, C.POINTER(vp)]),
    "dshs_destroy": (None, [vp]),
    "dshs_nparams": (i64, [vp]),
    "dshs_interpolate_sens": (cint, [vp, dbl, c_dp]),
    "dshs_reset": (cint, [vp]),
    "dshs_context": (vp, [vp]),
    "dshs_set_linear_solve_mode": (cint, [vp, cint]),
    "dshs_set_kernel_timing": (cint, [vp, cint]),
    "dshs_set_kernel_timing_target": (cint, [vp, cint]),
    "dshs_get_kernel_timing": (cint, [vp, c_i64p, c_dp]),
    "dshs_get_kernel_timing_overhead": (cint, [vp, c_dp, c_dp]),
    "dshs_nstates": (i64, [vp]),
    "dshs_nbatch": (i64, [vp]),
    "dshs_is_fused": (cint, [vp]),
    "dshs_set_ensemble_mode": (cint, [vp, cint]),
    "dshs_set_deterministic_pow": (cint, [cint]),
    "dshs_set_resident_arithmetic": (cint, [cint]),
    "dshs_get_resident_arithmetic": (cint, []),
    "dshs_get_ensemble_mode": (cint, [vp, c_ip, c_ip]),
    "dshs_last_solve_info": (cint, [vp, c_ip, c_i64p]),
    "dshs_step": (cint, [vp, c_ip]),
    "dshs_set_stop_time": (cint, [vp, dbl]),
    "dshs_interpolate": (cint, [vp, dbl, c_dp]),
    "dshs_get_state": (cint, [vp, c_dp, c_dp, c_ip, c_dp, c_dp]),
    "dshs_root_info": (cint, [vp, c_dp, c_ip]),
    "dshs_bdf_get_diff": (cint, [vp, c_dp]),
    "dshs_stats": (cint, [vp, c_i64p]),
    "dshs_solve_to_points": (cint, [vp, c_dp, i64, c_dp]),
    "dshs_solve": (cint, [vp, dbl, cint, c_dp, c_i64p, c_ip]),
    "dshs_trajectory": (cint, [vp, c_dp, c_dp]),
    "dshs_solve_dense": (cint, [vp, c_dp, i64, c_dp, vp, c_ip]),
    "dshs_diffsl_generate": (cint, [C.c_char_p, cint, C.POINTER(vp), c_i64p, c_dp, i64]),
    "dshs_diffsl_generate_indexed": (cint, [C.c_char_p, cint, cint, C.POINTER(vp), c_i64p, c_dp, i64]),
    "dshs_free_string": (None, [vp]),
    "dshs_diffsl_set_model_index": (cint, [cint]),
    "dshs_solve_dense_adaptive_sens": (cint, [vp, c_dp, i64, cint, cint, c_dp, c_dp, c_i32p, c_i32p, c_i64p]),
    "dshs_solve_dense_adaptive": (cint, [vp, c_dp, i64, cint, cint, c_dp, vp, c_i32p, c_i32p, c_dp, c_i32p, c_i32p, c_i64p]),
    "dshs_solve_adaptive": (cint, [vp, dbl, i64, cint, cint, c_dp, c_dp, c_i32p, c_i32p, c_i32p, c_dp, c_i32p, c_i64p]),
}

_dev = None
_host = None


def _bind(lib, abi):
    for name, (res, args) in abi.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


def load_device_lib():
    """Load libdiffsol_hip.so (does not touch a GPU)."""
    global _dev
    if _dev is None:
        path = lib_paths()[0]
        if not os.path.exists(path):
            raise DiffsolHipError(-1, f"{path} not found: build it with `python -m diffsol_amd.build` (there is no CPU fallback)")
        _dev = _bind(C.CDLL(path, mode=C.RTLD_GLOBAL), DEVICE_ABI)
    return _dev


def load_host_lib():
    """Load libdiffsol_hip_host.so (does not touch a GPU)."""
    global _host
    if _host is None:
        load_device_lib()
        path = lib_paths()[1]
        if not os.path.exists(path):
            raise DiffsolHipError(-1, f"{path} not found: build it with `python -m diffsol_amd.build` (there is no CPU fallback)")
        _host = _bind(C.CDLL(path), HOST_ABI)
    return _host


def check(rc, lib=None, host=False):
    if rc < 0:
        L = load_host_lib() if host else load_device_lib()
        msg = (L.dshs_last_error() if host else L.dsh_last_error()) or b""
        raise DiffsolHipError(rc, msg.decode(errors="replace"))
    return rc
