#!/usr/bin/env python
"""One device-resident solve of the C2 Robertson ensemble per control granularity (for rocprofv3 --pmc passes):  python scripts/resident_once.py [nb]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from bench import robertson_params, T_EVAL, RTOL, ATOL

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
s = H.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL)
for g in (1, 64):
    y, tot = s.solve_dense_adaptive(T_EVAL, want_host=False, group=g)
    print(g, tot)
