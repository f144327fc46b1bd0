#!/usr/bin/env python
"""Banded lane-per-member BDF (config 4, single-particle model n = 42): the memory-streaming kernel k_bdf_lane_banded against k_bdf_adaptive's banded
branch (DSH_LANE_BANDED_V1=1) — bitwise comparison of every output and the time of each:  python scripts/lane_banded_check.py [nb] [group]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
group = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cur = np.random.default_rng(12345).uniform(0.6, 1.4, (nb, 1))
res = {}
only = os.environ.get("LB_ONLY") == "1"  # time the streaming kernel only (tuning sweeps)
for v1 in (("0",) if only else ("0", "1")):
    os.environ["DSH_LANE_BANDED_V1"] = v1
    s = H.Solver("spm", cur, nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6])
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        out = s.solve_dense_adaptive([600.0, 1800.0, 3600.0], want_member_stats=not only, want_host=not only, group=group)
        best = min(best, time.perf_counter() - t0)
    res[v1] = out
    print("V1" if v1 == "1" else "V2", f"{best:.4f} s", out[1])
if only:
    sys.exit(0)
(ya, ta, ma), (yb, tb, mb) = res["0"], res["1"]
ok = np.array_equal(ya, yb, equal_nan=True) and ta == tb and all(np.array_equal(ma[k], mb[k], equal_nan=True) for k in ma)
print("bitwise equal:", ok)
sys.exit(0 if ok else 1)
