#!/usr/bin/env python
"""Wall clock of one headline solve (dshs_solve_dense, 100 000 Robertson members, output left on the device) against the kernel's own duration (HIP events):
what the host side adds per solve.    python scripts/host_overhead.py [nb] [reps]      (GPU only)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from bench import robertson_params, T_EVAL, RTOL, ATOL

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
s = H.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL, device=0, block_threads=256)
out = torch.empty((len(T_EVAL), 3, nb), dtype=torch.float64, device="cuda:0")
for timing in (True, False, True, False):  # alternating: the first batch also pays the clock ramp after idle
    s.set_kernel_timing(timing)
    for _ in range(50):
        s.solve_dense(T_EVAL, want_host=False, dev_ptr=out.data_ptr())
    if timing:
        s.kernel_timing()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.solve_dense(T_EVAL, want_host=False, dev_ptr=out.data_ptr())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    k = None
    if timing:
        n, ms = s.kernel_timing()
        k = ms / n * 1e3
    print(f"kernel timing {'on ' if timing else 'off'}: wall {wall * 1e6:.1f} us per solve" + (f", kernel {k:.1f} us, host adds {wall * 1e6 - k:.1f} us" if k else ""), flush=True)
