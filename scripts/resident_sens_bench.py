#!/usr/bin/env python
"""Forward sensitivities of the C2 Robertson ensemble (3 parameters): the device-resident integrators (one launch: states + dy/dp at the save points) against the
host-driven lock-step path (trait operations, sensitivities read with interpolate_sens at the end).   python scripts/resident_sens_bench.py [nb]   (GPU only)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from bench import robertson_params, T_EVAL, RTOL, ATOL

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
p = robertson_params(nb)
rows = []
for method, mname in ((H.METHOD_BDF, "bdf"), (H.METHOD_TR_BDF2, "tr_bdf2"), (H.METHOD_ESDIRK34, "esdirk34")):
    for sens_tol in (None, (1e-4, [1e-6])):
        kw = dict(sens_rtol=sens_tol[0], sens_atol=sens_tol[1]) if sens_tol else {}
        s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, method=method, rtol=RTOL, atol=ATOL, sens=True, **kw)
        for group in (64, 1):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                y, sens, tot = s.solve_dense_adaptive_sens(T_EVAL, group=group)
                best = min(best, time.perf_counter() - t0)
            rows.append(dict(method=mname, sens_error_control=bool(sens_tol), group=group, members=nb, wall_ms_incl_download=best * 1e3, steps=tot["number_of_steps"],
                             newton=tot["number_of_nonlinear_solver_iterations"], failed=tot["failed_members"], finite=bool(np.isfinite(sens).all())))
            print(json.dumps(rows[-1]), flush=True)
# the same ensemble without sensitivities, for the cost of carrying them (BDF, group 64)
s0 = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); y0, tot0 = s0.solve_dense_adaptive(T_EVAL, group=64); best = min(best, time.perf_counter() - t0)
print(json.dumps(dict(method="bdf", sens=False, group=64, members=nb, wall_ms_incl_download=best * 1e3, steps=tot0["number_of_steps"])), flush=True)
# host-driven lock-step with sensitivities on a smaller ensemble (one (t, h, order) sequence for all members)
nh = min(nb, 16384)
sh = H.Solver("robertson_ode", p[:nh], nbatch=nh, model_size=1, rtol=RTOL, atol=ATOL, sens=True, ensemble_mode=H.ENSEMBLE_LOCKSTEP)
t0 = time.perf_counter(); sh.solve(T_EVAL[-1]); sv = sh.interpolate_sens(); th = time.perf_counter() - t0
print(json.dumps(dict(path="host-driven lock-step with sensitivities", members=nh, wall_ms=th * 1e3, steps=sh.stats()["number_of_steps"])), flush=True)
