#!/usr/bin/env python
"""The reference's 2-D PDE test models as ENSEMBLES through the host-driven lock-step BDF (trait operations; M - cJ assembled on the declared band): the general banded LU
route against the dense-LU route (DSH_LU_STRUCTURE=dense, exact kernels), end to end:   python scripts/pde2d_ensemble.py [nb=4096]      (GPU only)
Per case: wall ms of one whole solve (second of two), BDF steps / Newton iterations / LU setups, and whether the two routes return the same bits."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
os.environ.setdefault("DSH_LU_EXACT", "1")
import diffsol_amd as H

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(12345)
print("| model | grid | n | band | members | t_final | band route ms | dense route ms | steps | Newton its | LU setups | same bits |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for model, size, tf in (("heat2d", 10, 0.16), ("heat2d", 16, 0.16), ("heat2d", 22, 0.16), ("foodweb", 10, 0.1), ("foodweb", 14, 0.1)):
    p = rng.uniform(0.6, 1.6, (nb, 1)) if model == "heat2d" else rng.uniform(0.9, 1.1, (nb, 2)) * [50.0, 1000.0]
    tol = dict(rtol=1e-7, atol=[1e-7]) if model == "heat2d" else dict(rtol=1e-5, atol=[1e-5])
    res = {}
    for route in ("auto", "dense"):
        os.environ["DSH_LU_STRUCTURE"] = route
        best = None
        for _ in range(2):
            s = H.Solver(model, p, nbatch=nb, model_size=size, h0=1.0, fused=False, **tol)
            t0 = time.perf_counter()
            y, _ = s.solve_to_points([tf])
            best = time.perf_counter() - t0
            st = s.stats()
            del s
        res[route] = (best, y, st)
    os.environ.pop("DSH_LU_STRUCTURE", None)
    st = res["auto"][2]
    n = size * size * (1 if model == "heat2d" else 2)
    print(f"| {model} | {size} x {size} | {n} | {size if model == 'heat2d' else 2 * size} | {nb} | {tf} | {1e3 * res['auto'][0]:.1f} | {1e3 * res['dense'][0]:.1f} | {st['number_of_steps']} | "
          f"{st['number_of_nonlinear_solver_iterations']} | {st['number_of_linear_solver_setups']} | {np.array_equal(res['auto'][1], res['dense'][1]) and res['auto'][2] == res['dense'][2]} |", flush=True)
