#!/usr/bin/env python
"""Banded solve (tridiagonal, n = 512 x 4096) with and without the fused norm epilogue: HIP-event time per dsh_lu_solve / dsh_lu_solve_squared_norm launch."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffsol_amd as H

n, nb, reps = 512, 4096, 100
rng = np.random.default_rng(1)
c = H.HipContext(nbatch=nb)
L = c._L
band = np.zeros((nb, n, n))
i = np.arange(n)
band[:, i, i] = 2.0 + rng.random((nb, n))
band[:, i[:-1], i[:-1] + 1] = -rng.random((nb, n - 1))
band[:, i[1:], i[1:] - 1] = -rng.random((nb, n - 1))
lu = H.HipLU(c, n)
lu.factor(H.HipMat.from_array(band, c))
del band
x = H.HipVec.from_vec(rng.standard_normal((nb, n)), c)
y = H.HipVec.from_vec(rng.standard_normal((nb, n)), c)
a = H.HipVec.from_vec(np.full((1, n), 1e-6), c.clone_with_nbatch(1))
out = C.c_double()


def timed(f):
    f()
    L.dsh_ctx_set_timing(c._h, 1); L.dsh_ctx_set_timing_target(c._h, 1)
    for _ in range(reps):
        f()
    nl, ms = C.c_int64(), C.c_double()
    L.dsh_ctx_get_timing(c._h, C.byref(nl), C.byref(ms))
    L.dsh_ctx_set_timing(c._h, 0)
    return 1e3 * ms.value / max(nl.value, 1), nl.value


t1, k1 = timed(lambda: L.dsh_lu_solve(lu._h, x.ptr))
t2, k2 = timed(lambda: L.dsh_lu_solve_squared_norm(lu._h, x.ptr, y.ptr, nb, a.ptr, 1, 1e-6, C.byref(out)))
print(f"DSH_LU_SOLVE_EPI={os.environ.get('DSH_LU_SOLVE_EPI')} DSH_TEAM_EPI_X={os.environ.get('DSH_TEAM_EPI_X')}: dsh_lu_solve {t1:.2f} us ({k1}); solve launch of dsh_lu_solve_squared_norm {t2:.2f} us ({k2})")
# the Newton-update form of the epilogue (xout = xin - delta): the staged SDIRK Newton iteration of heat1d, the solve launch bracketed alone (timing target 1)
k = H.HipVec.from_vec(rng.standard_normal((nb, n)) * 1e-3, c)
phi = H.HipVec.from_vec(rng.standard_normal((nb, n)), c)
p = H.HipVec.from_vec(rng.uniform(0.5, 2.0, (nb, 1)), c)
o3 = (C.c_double * 3)()
MODEL_HEAT1D = 7
t3, k3 = timed(lambda: L.dsh_sdirk_newton_iter(c._h, MODEL_HEAT1D, n, nb, 0.0, 1e-3, 2e-4, k.ptr, k.ptr, phi.ptr, p.ptr, lu._h, y.ptr, a.ptr, 1, 1e-6, o3))
print(f"  solve launch of dsh_sdirk_newton_iter (update + norm epilogue) {t3:.2f} us ({k3})")
