#!/usr/bin/env python
"""Time dsh_bdf_solve_adaptive on the C2 Robertson sweep at several ensemble sizes / member orders:  python scripts/adaptive_bench.py   (GPU only)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from bench import robertson_params

T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]
ROB = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])


def run(p, tag, group=1, det=False):
    s = H.Solver("robertson_ode", p, nbatch=len(p), model_size=1, **ROB)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        y, tot = s.solve_dense_adaptive(T_EVAL, want_host=False, group=group, deterministic_pow=det)
        best = min(best, time.perf_counter() - t0)
    print(f"{tag} (group {group}{', deterministic pow' if det else ''}): {best*1e3:.3f} ms  steps/s {tot['number_of_steps']/best:.3e}  newton/s {tot['number_of_nonlinear_solver_iterations']/best:.3e}", flush=True)


if __name__ == "__main__":
    p0 = robertson_params(100_000)
    run(p0, "100k random order")
    run(np.repeat(p0[:1], 100_000, axis=0), "100k identical members")
    run(robertson_params(400_000), "400k random")
    run(robertson_params(1_600_000), "1.6M random")
    run(p0, "100k random order", group=64)
    run(robertson_params(400_000), "400k random", group=64)
    run(robertson_params(1_600_000), "1.6M random", group=64)
    run(p0, "100k random order", group=1, det=True)
    run(p0, "100k random order", group=64, det=True)
