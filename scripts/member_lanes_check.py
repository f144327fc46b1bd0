#!/usr/bin/env python
"""Per-member control with fewer members per wavefront (DSH_MEMBER_LANES = 32 | 16 | 8; VERDICT r4 item 5: "nobody has tried the other axis"): BASELINE config 5
(RLC, ESDIRK34, events, 65 536 members) and config 2 per member (Robertson, BDF, 100 000 members), kernel time from HIP events and a checksum of the output
(the member -> lane mapping must not change any member's result).   One process per setting:   DSH_MEMBER_LANES=16 python scripts/member_lanes_check.py"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import diffsol_amd as H
from bench import rlc_params, robertson_params, T_EVAL, RTOL, ATOL

ml = os.environ.get("DSH_MEMBER_LANES", "64")
for name in (sys.argv[1:] or ["c5", "c2"]):
    if name == "c5":
        nb, t_eval = int(os.environ.get("C5_NB", "65536")), np.linspace(0.1, 1.0, 10)
        s = H.Solver("rlc", rlc_params(nb, 0.03), nbatch=nb, model_size=1, rtol=1e-6, atol=[1e-6], method=H.METHOD_ESDIRK34)
    else:
        nb, t_eval = int(os.environ.get("C2_NB", "100000")), np.asarray(T_EVAL)
        s = H.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL)
    out = torch.full((len(t_eval), s.n, nb), float("nan"), dtype=torch.float64, device="cuda:0")
    s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=out.data_ptr(), group=1)
    walls = []
    for _ in range(5):
        t0 = time.perf_counter(); _, tot = s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=out.data_ptr(), group=1); walls.append(time.perf_counter() - t0)
    s.set_kernel_timing(True); s.set_kernel_timing_target(0)
    for _ in range(3):
        s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=out.data_ptr(), group=1)
    nl, ms = s.kernel_timing()
    s.set_kernel_timing(False)
    h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{name} members_per_wavefront={ml:>2s}: kernel {ms / max(nl, 1):7.3f} ms, wall min {1e3 * min(walls):7.3f} ms, steps {tot['number_of_steps']}, failed {tot['failed_members']}, output sha {h}")
    del s, out
