#!/usr/bin/env python
"""Workgroup-per-member BDF (64 < n <= 140) timed on the reference's benchmark family: robertson_ode x ngroups (n = 3 ngroups), 4096 members, tol 1e-4 and 1e-8.
    python scripts/team_member_once.py [ngroups ...]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DSH_RESIDENT_LANE"] = "0"  # the dense per-member route (the block-diagonal family also has the banded lane form)
import diffsol_amd as H
from bench import robertson_params

T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]
nb = 4096
for groups in [int(a) for a in sys.argv[1:]] or [40, 30, 22]:
    n = 3 * groups
    for tol in (1e-4, 1e-8):
        s = H.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=groups, rtol=tol, atol=[tol] * n)
        y, tot = s.solve_dense_adaptive(T_EVAL, group=1)
        t0 = time.perf_counter(); y, tot = s.solve_dense_adaptive(T_EVAL, group=1); dt = time.perf_counter() - t0
        print(f"n = {n} tol {tol:g}: {dt:.4f} s per {nb} members ({1e6 * dt / nb:.1f} us per member), steps/member {tot['number_of_steps'] / nb:.0f}, setups/member "
              f"{tot['number_of_linear_solver_setups'] / nb:.1f}, failed {tot['failed_members']}, sha {hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest()[:12]}", flush=True)
        del s
