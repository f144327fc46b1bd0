#!/usr/bin/env python
"""The bench's timed region reduced to a few launches (for rocprofv3 --pmc passes): K default-mode solve_dense calls of the C2 Robertson ensemble.
    python scripts/bench_kernel_once.py [nb] [K] [mode]      mode: auto (default) | member | lockstep"""
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from bench import robertson_params, T_EVAL, RTOL, ATOL

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = {"auto": None, "member": 1, "lockstep": 0, "wave": 64}[sys.argv[3] if len(sys.argv) > 3 else "auto"]
s = H.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL, ensemble_mode=mode, block_threads=256)
for _ in range(K):
    if mode == 0:
        s.reset()
    s.solve_dense(T_EVAL, want_host=False)
    print(s.last_solve_info())
