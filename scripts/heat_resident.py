#!/usr/bin/env python
"""BASELINE config 3 (heat1d, n = 512, TR-BDF2) through the device-resident lane-per-member kernels (DSH_LANE_TWIN_MAX_N=512: one lane per member, the
state and the banded factors in per-lane scratch), against the oracle for a few members and against the Fourier series at full size:
    DSH_LANE_TWIN_MAX_N=512 python scripts/heat_resident.py [n] [nbatch] [method]        (GPU only)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffsol_amd as H
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
method = sys.argv[3] if len(sys.argv) > 3 else "tr_bdf2"
hm = {"bdf": H.METHOD_BDF, "tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
om = {"bdf": O.METHOD_BDF, "tr_bdf2": O.METHOD_TR_BDF2, "esdirk34": O.METHOD_ESDIRK34}[method]
t_final = 0.5
rng = np.random.default_rng(12345)
D = rng.uniform(0.5, 2.0, nb)
O.build()
O.set_det_pow(True)
# parity: 8 members, every member its own history, vs independent oracle solves (same deterministic pow)
s8 = H.Solver("heat1d", D[:8, None], nbatch=8, model_size=n, rtol=1e-6, atol=[1e-6], method=hm)
t0 = time.perf_counter()
y8, tot8, m8 = s8.solve_dense_adaptive([0.1, t_final], want_member_stats=True, group=1)
first = time.perf_counter() - t0
yo, so, failed = O.solve_dense_independent(O.MODEL_HEAT1D, D[:8, None], [0.1, t_final], model_size=n, rtol=1e-6, atol=[1e-6], method=om)
same = bool(np.array_equal(np.transpose(y8, (1, 0, 2)), yo))
print(json.dumps(dict(leg="parity", n=n, method=method, members=8, first_call_s=first, same_bits_as_oracle=same, steps=[int(v) for v in m8["stats"][0]], oracle_steps=[int(v) for v in so[:, 0]],
                      status=[int(v) for v in m8["status"]])), flush=True)
for group in (1, 64):
    s = H.Solver("heat1d", D[:, None], nbatch=nb, model_size=n, rtol=1e-6, atol=[1e-6], method=hm)
    s.solve_dense_adaptive([t_final], want_host=False, group=group)
    t0 = time.perf_counter()
    y, tot, mm = s.solve_dense_adaptive([t_final], want_member_stats=True, group=group)
    wall = time.perf_counter() - t0
    h = 1.0 / (n + 1)
    x = (np.arange(n) + 1) * h
    m = np.arange(1, 200)[:, None, None]
    ref = (np.sin((2 * m - 1) * np.pi * x[None, None, :]) * np.exp(-(2 * m - 1) ** 2 * np.pi ** 2 * D[None, :64, None] * t_final) / (2 * m - 1) ** 2).sum(0) * 8 / np.pi ** 2
    err = float(np.abs(y[0, :64] - ref).max())
    print(json.dumps(dict(leg="full", n=n, nbatch=nb, method=method, group=group, wall_s=wall, totals=tot, status_nonzero=int((mm["status"] != 0).sum()),
                          max_abs_err_vs_fourier_first64=err, mean_steps_per_member=tot["number_of_steps"] / nb)), flush=True)
