#!/usr/bin/env python
"""How much of the per-member tax of the C2 Robertson sweep is divergence inside a wavefront — the part re-binning could recover.

Per-member control (every member its own step sizes and orders) of 100 000 members, four ensembles of the same size:
  sorted      the bench's ensemble in the library's launch-time Morton order (what `per_member` in bench.py times)
  replicated  1563 members drawn from the same distribution, each copied into the 64 lanes of one wavefront: NO divergence inside any wavefront,
              the same spread of work ACROSS wavefronts — what a perfect re-binning at every step would approach
  identical   one member copied 100 000 times: no divergence, no imbalance
  lock-step   the bench's headline mode (wavefront groups), for reference
python scripts/per_member_bound.py   (GPU only)"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from bench import robertson_params, T_EVAL, RTOL, ATOL

NB = 100_000


def run(p, tag, group):
    s = H.Solver("robertson_ode", p, nbatch=len(p), model_size=1, rtol=RTOL, atol=ATOL)
    best, tot = 1e9, None
    for _ in range(4):
        t0 = time.perf_counter()
        y, tot = s.solve_dense_adaptive(T_EVAL, want_host=False, group=group)
        best = min(best, time.perf_counter() - t0)
    print(f"{tag:<28s} {best * 1e3:7.3f} ms   member-steps {tot['number_of_steps']:>9d}   Newton iterations {tot['number_of_nonlinear_solver_iterations']:>9d}   "
          f"{tot['number_of_nonlinear_solver_iterations'] / best:.3e} Newton/s", flush=True)


if __name__ == "__main__":
    p = robertson_params(NB)
    run(p, "sorted, per member", 1)
    reps = (NB + 63) // 64
    base = robertson_params(reps, seed=777)
    run(np.repeat(base, 64, axis=0)[:NB], "replicated x64, per member", 1)
    run(np.repeat(p[:1], NB, axis=0), "identical, per member", 1)
    run(p, "sorted, wavefront lock-step", 64)
    run(np.repeat(base, 64, axis=0)[:NB], "replicated x64, lock-step", 64)
