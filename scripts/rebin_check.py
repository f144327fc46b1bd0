#!/usr/bin/env python
"""Segmented per-member runs with re-binning between the segments (DSH_REBIN) against the single launch: every output bit for bit, and the time of each.
    python scripts/rebin_check.py [nb]      (GPU only)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from bench import robertson_params, T_EVAL, RTOL, ATOL

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
p = robertson_params(nb)
res = {}
MODES = [("DSH_REBIN", "0"), ("DSH_REBIN", "1"), ("DSH_REBIN", "2"), ("DSH_REBIN", "3"), ("DSH_REBIN_STEPS", "8"), ("DSH_REBIN_STEPS", "16"), ("DSH_REBIN_STEPS", "32"),
         ("DSH_REBIN_STEPS", "64"), ("DSH_REBIN_STEPS", "128")]
for var, rb in MODES:
    os.environ.pop("DSH_REBIN", None); os.environ.pop("DSH_REBIN_STEPS", None)
    os.environ[var] = rb
    rb = var + "=" + rb
    s = H.Solver("robertson_ode", p, nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        s.solve_dense_adaptive(T_EVAL, want_host=False, group=1)
        best = min(best, time.perf_counter() - t0)
    y, tot, mem = s.solve_dense_adaptive(T_EVAL, want_member_stats=True, group=1)
    res[rb] = (y, tot, mem)
    print(f"{rb}: {best * 1e3:.3f} ms  {tot}", flush=True)
os.environ.pop("DSH_REBIN", None); os.environ.pop("DSH_REBIN_STEPS", None)
y0, t0_, m0 = res["DSH_REBIN=0"]
ok = True
for rb in [v + "=" + k for v, k in MODES[1:]]:
    y, t, m = res[rb]
    same = np.array_equal(y, y0, equal_nan=True) and t == t0_ and all(np.array_equal(m[k], m0[k], equal_nan=True) for k in m0)
    print(f"{rb} bitwise equal to the single launch: {same}")
    ok = ok and same
sys.exit(0 if ok else 1)
