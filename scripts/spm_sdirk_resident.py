#!/usr/bin/env python
"""Battery model (spm, n = 42: BASELINE config 4's model) with TR-BDF2 / ESDIRK34, device-resident, every member its own history and cut-off event:
the lane-per-member banded form (default) against the wavefront-per-member form (DSH_RESIDENT_LANE=0).   python scripts/spm_sdirk_resident.py [nbatch] [method]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffsol_amd as H

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
method = sys.argv[2] if len(sys.argv) > 2 else "tr_bdf2"
hm = {"bdf": H.METHOD_BDF, "tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[method]
cur = np.random.default_rng(12345).uniform(0.6, 1.4, nb)
s = H.Solver("spm", cur[:, None], nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6], method=hm)
t_eval = np.linspace(360.0, 3600.0, 10)
s.solve_dense_adaptive(t_eval, want_host=False)
t0 = time.perf_counter()
y, tot, mm = s.solve_dense_adaptive(t_eval, want_member_stats=True)
wall = time.perf_counter() - t0
print(json.dumps(dict(model="spm n=42", nbatch=nb, method=method, form="wavefront per member" if os.environ.get("DSH_RESIDENT_LANE") == "0" else "lane per member (banded)",
                      wall_s=wall, totals=tot, members_stopped_by_event=int((mm["root_idx"] >= 0).sum()), status_nonzero=int((mm["status"] != 0).sum()),
                      steps_per_s=tot["number_of_steps"] / wall)))
