#!/usr/bin/env python
"""Exact vs reordered banded solve at BASELINE config 3's shape, and config 3 end to end in both modes:  python scripts/band_solve_modes.py [n=512] [nb=4096]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from diffsol_amd import _ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
L = _ffi.load_device_lib()
ctx = H.HipContext(0, nbatch=nb)
a1 = np.diag(np.full(n, 4.0)) + np.diag(np.full(n - 1, -1.0), 1) + np.diag(np.full(n - 1, -1.0), -1)
A = H.HipMat.from_array(np.broadcast_to(a1, (nb, n, n)).copy(), ctx)
b = H.HipVec.from_vec(np.random.default_rng(0).standard_normal((nb, n)), ctx)
lu = H.HipLU(ctx, n)
lu.factor(A)
out = {"n": n, "systems": nb, "algorithmic_bytes": nb * 52 * n}
for mode, name in ((0, "exact"), (1, "reordered")):
    _ffi.check(L.dsh_ctx_set_solve_mode(ctx._h, mode))
    for _ in range(5):
        lu.solve_in_place(b)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        lu.solve_in_place(b)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 200
    out[name] = {"us_per_solve_incl_host": dt * 1e6, "algorithmic_GBs": nb * 52 * n / dt / 1e9}
_ffi.check(L.dsh_ctx_set_solve_mode(ctx._h, 0))
if n == 512:
    D = np.random.default_rng(12345).uniform(0.5, 2.0, (nb, 1))
    for mode, name in ((0, "c3_exact"), (1, "c3_reordered")):
        s = H.Solver("heat1d", D, nbatch=nb, model_size=n, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
        s.set_linear_solve_mode(mode)
        s.solve_to_points([0.5])
        w = []
        for _ in range(3):
            s.reset()
            t0 = time.perf_counter(); y, _ = s.solve_to_points([0.5]); w.append(time.perf_counter() - t0)
        out[name] = {"wall_s": min(w), "steps": s.stats()["number_of_steps"]}
        if mode == 0:
            y0 = y
        else:
            out[name]["max_abs_diff_vs_exact"] = float(np.abs(y - y0).max())
print(json.dumps(out))
