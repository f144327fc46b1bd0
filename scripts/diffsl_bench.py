"""DiffSL (run-time-compiled) models against the built-in registry models on the same ensembles: compile times and solve times.
Usage: python scripts/diffsl_bench.py  (needs a GPU)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import diffsol_amd as H  # noqa: E402
from diffsol_amd import diffsl  # noqa: E402
import diffsl_models as D  # noqa: E402
from helpers import robertson_params  # noqa: E402


def timed(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return min(ts)


out = {}
# ---- C2: Robertson, 100k members
nb = 100000
p = robertson_params(nb)
tol = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
t_eval = [0.4 * 10 ** k for k in range(7)]
t0 = time.perf_counter(); m = diffsl.DiffslModel(D.ROBERTSON_ODE); out["robertson_frontend_plus_operator_module_s"] = time.perf_counter() - t0
for fam, name in ((1, "fused_newton_modules_s"), (2, "resident_bdf_modules_s")):
    t0 = time.perf_counter(); m.precompile(fam); out["robertson_" + name] = time.perf_counter() - t0
for label, model, size in (("builtin", "robertson_ode", 1), ("diffsl", m, 0)):
    s = H.Solver(model, p, nbatch=nb, model_size=size, **tol)
    def lockstep():
        s.reset(); s.solve_dense(t_eval, want_host=False)
    out[f"robertson_{label}_lockstep_s"] = timed(lockstep)
    for g in (1, 64):
        out[f"robertson_{label}_resident_group{g}_s"] = timed(lambda: s.solve_dense_adaptive(t_eval, want_host=False, group=g))
# ---- C4-like: SPM n = 42, 32768 members, host-driven lock-step BDF to t = 600 s
nb = 32768
cur = np.random.default_rng(12345).uniform(0.6, 1.4, (nb, 1))
t0 = time.perf_counter(); ms = diffsl.DiffslModel(D.spm(20)); out["spm_frontend_plus_operator_module_s"] = time.perf_counter() - t0
for label, model, size in (("builtin", "spm", 20), ("diffsl", ms, 0)):
    s = H.Solver(model, cur, nbatch=nb, model_size=size, rtol=1e-6, atol=[1e-6])
    def run():
        s.reset(); s.solve_dense([60.0, 600.0], want_host=False)
    out[f"spm_{label}_lockstep_s"] = timed(run, reps=2)
    out[f"spm_{label}_wave_member_resident_s"] = timed(lambda: s.solve_dense_adaptive([600.0, 1800.0, 3600.0], want_host=False, group=1), reps=2)
print(json.dumps(out, indent=1))
