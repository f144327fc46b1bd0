#!/usr/bin/env python
"""BASELINE config 3 (heat1d n = 512 x 4096, TR-BDF2, host-driven lock-step) once: wall clock of the warmed solve (min of 3) and the HIP-event totals of the
dsh_lu_solve launches.   python scripts/c3_once.py [banded|dense] [nb]      (A/B: DSH_LU_SOLVE_EPI=0|1)"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffsol_amd as H
from bench import heat_params

route = sys.argv[1] if len(sys.argv) > 1 else "banded"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
if route == "dense":
    os.environ["DSH_LU_STRUCTURE"] = "dense"
D = heat_params(nb)
s = H.Solver("heat1d", D, nbatch=nb, model_size=512, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
y, _ = s.solve_to_points([0.5])
walls = []
for _ in range(3):
    s.reset()
    t0 = time.perf_counter(); y2, _ = s.solve_to_points([0.5]); walls.append(time.perf_counter() - t0)
assert np.array_equal(y, y2)
st = s.stats()
s.reset(); s.set_kernel_timing(True); s.set_kernel_timing_target(1)
s.solve_to_points([0.5])
ns, ms = s.kernel_timing()
print(f"c3 {route} nb={nb} DSH_LU_SOLVE_EPI={os.environ.get('DSH_LU_SOLVE_EPI')}: wall min {1e3 * min(walls):.2f} ms {[round(1e3 * w, 1) for w in walls]}; steps {st['number_of_steps']}, "
      f"newton {st['number_of_nonlinear_solver_iterations']}; {ns} solve launches, avg {1e3 * ms / max(ns, 1):.2f} us; output sha {hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest()[:16]}")
