#!/usr/bin/env python
"""BASELINE config 2 through the pure 1:1 trait composition (fused = False, lock-step): wall clock, for rocprofv3 --kernel-trace --stats"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffsol_amd as H
from diffsol_amd.solver import ENSEMBLE_LOCKSTEP
from bench import robertson_params, T_EVAL, RTOL, ATOL
nb = 100000
s = H.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL, fused=False, ensemble_mode=ENSEMBLE_LOCKSTEP)
s.solve_dense(T_EVAL, want_host=False)
w = []
for _ in range(3):
    s.reset(); t0 = time.perf_counter(); s.solve_dense(T_EVAL, want_host=False); w.append(time.perf_counter() - t0)
print("trait path: ms per solve", [round(1e3 * x, 2) for x in w], s.stats())
