#!/usr/bin/env python
"""One host-driven BDF solve of a 4096-member heat2d / foodweb ensemble (for rocprofv3 --kernel-trace --stats):  python scripts/pde2d_once.py [heat2d|foodweb] [size] [nb]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
os.environ.setdefault("DSH_LU_EXACT", "1")
import diffsol_amd as H

model = sys.argv[1] if len(sys.argv) > 1 else "heat2d"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
rng = np.random.default_rng(12345)
p = rng.uniform(0.6, 1.6, (nb, 1)) if model == "heat2d" else rng.uniform(0.9, 1.1, (nb, 2)) * [50.0, 1000.0]
tol = dict(rtol=1e-7, atol=[1e-7]) if model == "heat2d" else dict(rtol=1e-5, atol=[1e-5])
for _ in range(2):
    s = H.Solver(model, p, nbatch=nb, model_size=size, h0=1.0, fused=False, **tol)
    t0 = time.perf_counter()
    y, _ = s.solve_to_points([0.16 if model == "heat2d" else 0.1])
    print(model, size, nb, "wall ms %.1f" % (1e3 * (time.perf_counter() - t0)), s.stats()["number_of_steps"], "steps")
