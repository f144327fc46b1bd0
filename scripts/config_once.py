#!/usr/bin/env python
"""One BASELINE config's device-resident solve reduced to K launches (for rocprofv3 passes; the workloads are bench.py's):
    python scripts/config_once.py c4_ode|c4_dae|c5_per_member|c5_group64 [nb] [K]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffsol_amd as H
from bench import rlc_params, spm_params

cfg = sys.argv[1]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if cfg.startswith("c4"):
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    t_eval = np.linspace(360.0, 3600.0, 10)
    if cfg == "c4_dae":
        from diffsol_amd import diffsl
        import diffsl_models as DM
        s = H.Solver(diffsl.DiffslModel(DM.spm_dae(20)), spm_params(nb), nbatch=nb, rtol=1e-6, atol=[1e-6])
    else:
        s = H.Solver("spm", spm_params(nb), nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6])
    group = 1
else:
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    group = 1 if cfg == "c5_per_member" else 64
    t_eval = np.linspace(0.1, 1.0, 10)
    s = H.Solver("rlc", rlc_params(nb, 0.03 if group == 1 else 1e3), nbatch=nb, model_size=1, rtol=1e-6, atol=[1e-6], method=H.METHOD_ESDIRK34)
import time
for _ in range(K):
    t0 = time.perf_counter()
    # the library's default arithmetic (fast builds where they exist) unless DSH_RESIDENT_ARITH=exact: what bench.py's rows and their counters are taken with
    _, tot = s.solve_dense_adaptive(t_eval, want_host=False, group=group, deterministic_pow=2 if H.get_resident_arithmetic() == H.ARITH_FAST else 1)
    print(cfg, nb, "wall ms %.2f" % (1e3 * (time.perf_counter() - t0)), tot)
