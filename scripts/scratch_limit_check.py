#!/usr/bin/env python
"""C4 (k_bdf_lane_banded: ~9.6 KB of per-lane scratch) timed with HIP events and wall clock, for the HSA_SCRATCH_SINGLE_LIMIT experiment (HIP-event time
70.7 ms vs rocprof kernel time 64.6 ms).  Run once per environment:   python scripts/scratch_limit_check.py [c4_ode|c4_dae] [nb]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (device memory for the output)
import diffsol_amd as H
from bench import spm_params

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4_ode"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
t_eval = np.linspace(360.0, 3600.0, 10)
if cfg == "c4_dae":
    from diffsol_amd import diffsl
    import diffsl_models as DM
    s = H.Solver(diffsl.DiffslModel(DM.spm_dae(20)), spm_params(nb), nbatch=nb, rtol=1e-6, atol=[1e-6])
else:
    s = H.Solver("spm", spm_params(nb), nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6])
out = torch.empty((len(t_eval), s.n, nb), dtype=torch.float64, device="cuda:0")
s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=out.data_ptr())
walls = []
for _ in range(5):
    t0 = time.perf_counter(); s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=out.data_ptr()); walls.append(time.perf_counter() - t0)
s.set_kernel_timing(True); s.set_kernel_timing_target(0)
for _ in range(3):
    s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=out.data_ptr())
nl, ms = s.kernel_timing()
print(f"{cfg} nb={nb} HSA_SCRATCH_SINGLE_LIMIT={os.environ.get('HSA_SCRATCH_SINGLE_LIMIT')} HSA_SCRATCH_SINGLE_LIMIT_ASYNC={os.environ.get('HSA_SCRATCH_SINGLE_LIMIT_ASYNC')}: "
      f"wall min {1e3 * min(walls):.2f} ms (all {[round(1e3 * w, 2) for w in walls]}), HIP-event avg {ms / max(nl, 1):.2f} ms over {nl} launches")
