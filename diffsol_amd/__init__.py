"""diffsol_amd — MI355X (gfx950) native batched implicit ODE/DAE integration behind diffsol's trait surface.

Python is only the test / benchmark harness: it binds the two C ABIs with ctypes

    include/diffsol_hip.h         -> diffsol_amd/lib/libdiffsol_hip.so       (HIP device backend: Context/Vector/Matrix/LU/models/fused)
    include/diffsol_hip_solver.h  -> diffsol_amd/lib/libdiffsol_hip_host.so  (host-side integrators: OdeBuilder/Bdf/Sdirk)

There is NO CPU fallback: importing works anywhere (so the C ABI can be inspected), but every compute entry point needs a HIP
device and raises `DiffsolHipError` loudly if the native libraries are missing or no GPU is present.
"""
from . import _ffi
from ._ffi import DiffsolHipError, lib_paths, load_device_lib, load_host_lib
from .la import HipContext, HipIndex, HipLU, HipMat, HipMatView, HipVec
from .solver import (ENSEMBLE_AUTO, ENSEMBLE_LOCKSTEP, ENSEMBLE_PER_MEMBER, ENSEMBLE_WAVEFRONT, METHOD_BDF, METHOD_ESDIRK34, METHOD_TR_BDF2, MODELS,
                     STAT_NAMES, ARITH_EXACT, ARITH_FAST, OdeBuilder, Solver, get_resident_arithmetic, set_deterministic_pow, set_resident_arithmetic)

__all__ = [
    "DiffsolHipError", "lib_paths", "load_device_lib", "load_host_lib", "HipContext", "HipVec", "HipMat", "HipMatView", "HipIndex", "HipLU", "OdeBuilder", "Solver",
    "METHOD_BDF", "METHOD_TR_BDF2", "METHOD_ESDIRK34", "MODELS", "STAT_NAMES", "ENSEMBLE_AUTO", "ENSEMBLE_LOCKSTEP", "ENSEMBLE_PER_MEMBER",
    "ENSEMBLE_WAVEFRONT", "set_deterministic_pow", "set_resident_arithmetic", "get_resident_arithmetic", "ARITH_EXACT", "ARITH_FAST",
]
