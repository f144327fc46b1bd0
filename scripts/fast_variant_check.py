#!/usr/bin/env python
"""The opt-in fast-arithmetic variant of the headline kernel (deterministic_pow = 2: -ffp-contract=fast, reciprocal-math division, ocml pow, reciprocal Newton
weights) against the exact kernel: time of each on the bench's ensemble, and the states' relative difference at the bench's and at tight tolerances.
    python scripts/fast_variant_check.py [nb]      (GPU only)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from bench import robertson_params, T_EVAL, RTOL, ATOL

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
p = robertson_params(nb)
out = {"members": nb}
for name, tol in (("bench", dict(rtol=RTOL, atol=ATOL)), ("tight", dict(rtol=1e-9, atol=[1e-13, 1e-17, 1e-11]))):
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, **tol)
    res = {}
    for mode, label in ((1, "exact"), (0, "ocml_pow"), (2, "fast")):
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            s.solve_dense_adaptive(T_EVAL, want_host=False, group=64, deterministic_pow=mode)
            best = min(best, time.perf_counter() - t0)
        y, tot = s.solve_dense_adaptive(T_EVAL, group=64, deterministic_pow=mode)
        res[label] = (best, y, tot)
    y0 = res["exact"][1]
    row = {}
    for label in ("exact", "ocml_pow", "fast"):
        best, y, tot = res[label]
        rel = np.abs(y - y0) / (np.abs(y0) + 1e-300)
        big = np.abs(y0) > 1e-9  # components above the absolute tolerances
        row[label] = {"ms": best * 1e3, "steps": tot["number_of_steps"], "newton": tot["number_of_nonlinear_solver_iterations"], "failed": tot["failed_members"],
                      "steps_per_s": tot["number_of_steps"] / best, "max_rel_diff_vs_exact": float(rel[big].max()), "mass_err": float(np.abs(y.sum(axis=2) - 1).max())}
    out[name] = row
    print(name, json.dumps(row), flush=True)
json.dump(out, open("gpurun_out/r03_fast_variant.json", "w"), indent=1)
