#!/usr/bin/env python
"""Forward sensitivities in the banded lane-per-member BDF at ensemble size: the single-particle battery model from DiffSL (n = 42, no stop
conditions) with d(state)/d(current) integrated alongside, one launch for the whole ensemble.
    python scripts/spm_sens_once.py [nb=32768] [repeats=3]  -> one JSON line (time with and without sensitivities, error control on / off)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffsol_amd as H
from diffsol_amd import diffsl
import diffsl_models as D

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cur = np.random.default_rng(12345).uniform(0.6, 1.4, (nb, 1))
t_eval = np.linspace(360.0, 3600.0, 10)
m = diffsl.DiffslModel(D.spm(20, no_stops=True))
out = {"members": nb, "n": m.n}
s0 = H.Solver(m, cur, nbatch=nb, rtol=1e-6, atol=[1e-6])
s0.solve_dense_adaptive(t_eval, want_host=False)
w = []
for _ in range(reps):
    t0 = time.perf_counter(); _, tot = s0.solve_dense_adaptive(t_eval, want_host=False); w.append(time.perf_counter() - t0)
out["states_only"] = {"wall_s": min(w), "steps": tot["number_of_steps"], "newton": tot["number_of_nonlinear_solver_iterations"]}
for name, kw in (("sens_no_error_control", {}), ("sens_error_control", dict(sens_rtol=1e-6, sens_atol=[1e-6]))):
    s = H.Solver(m, cur, nbatch=nb, rtol=1e-6, atol=[1e-6], sens=True, **kw)
    t0 = time.perf_counter(); y, sens, tot = s.solve_dense_adaptive_sens(t_eval); first = time.perf_counter() - t0
    w = []
    for _ in range(reps):
        t0 = time.perf_counter(); y, sens, tot = s.solve_dense_adaptive_sens(t_eval); w.append(time.perf_counter() - t0)
    # dq/dI = t / 3600 exactly (q' = I / 3600): a property the front end did not produce
    err = float(np.abs(sens[0, :, :, 0] - (t_eval / 3600.0)[:, None]).max())
    out[name] = {"wall_s_incl_download": min(w), "first_call_s": first, "steps": tot["number_of_steps"], "newton": tot["number_of_nonlinear_solver_iterations"],
                 "failed": tot["failed_members"], "max_abs_error_dq_dI_vs_t_over_3600": err, "finite": bool(np.isfinite(sens).all())}
print(json.dumps(out))
