"""Host-side launch cost and launch->result round-trip latency of the C ABI on this box."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import diffsol_amd
from diffsol_amd import _ffi
L = _ffi.load_device_lib()
ctx = diffsol_amd.HipContext(0, nbatch=1)
v = diffsol_amd.HipVec.zeros(64, ctx)
big = diffsol_amd.HipContext(0, nbatch=1).clone_with_nbatch(100000)
N = 20000
# warm
for _ in range(100): L.dsh_vec_fill(ctx._h, 64, 1, v.ptr, 1.0)
ctx.sync()
t0 = time.perf_counter()
for _ in range(N): L.dsh_vec_fill(ctx._h, 64, 1, v.ptr, 1.0)
t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
print(f"async launch (python ctypes loop): {1e6*(t1-t0)/N:.2f} us per call issued, {1e6*(t2-t0)/N:.2f} us per call incl. drain")
out = C.c_double()
for mode in (1, 0):
    L.dsh_ctx_set_poll(ctx._h, mode)
    for _ in range(100): L.dsh_vec_norm(ctx._h, 64, 1, v.ptr, 2, C.byref(out))
    t0 = time.perf_counter()
    for _ in range(N): L.dsh_vec_norm(ctx._h, 64, 1, v.ptr, 2, C.byref(out))
    t1 = time.perf_counter()
    print(f"launch+result round trip, {'poll' if mode else 'sync'} mode: {1e6*(t1-t0)/N:.2f} us")
# python overhead of a trivial ctypes call
t0 = time.perf_counter()
for _ in range(N): L.dsh_version()
print(f"ctypes call overhead: {1e6*(time.perf_counter()-t0)/N:.2f} us")
