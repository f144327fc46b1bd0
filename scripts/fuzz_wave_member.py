#!/usr/bin/env python
"""Randomised parity sweep of the wavefront-per-member kernels (k_bdf_wave_member, k_sdirk_wave_member) against independent oracle solves, bit for bit:
run-time-sized built-in models (banded ones forced off their lane-per-member twin), BDF / TR-BDF2 / ESDIRK34, random sizes, tolerances, parameters and
output times, including members that stop at an event or fail.   DSH_RESIDENT_LANE=0 python scripts/fuzz_wave_member.py [ncases] [first_seed]   (GPU only)"""
import os
import sys

os.environ["DSH_RESIDENT_LANE"] = "0"
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffsol_amd as H
from helpers import ORACLE_MODEL
from oracle import oracle as O

O.build()
O.set_det_pow(True)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(first, first + ncases):
    rng = np.random.default_rng(7000 + seed)
    model = ["heat1d", "spm", "gaussian_decay", "robertson_ode"][seed % 4]
    method = int(rng.integers(0, 3))
    nb = int(rng.integers(8, 60))
    rtol = float(10.0 ** rng.uniform(-8, -3))
    if model == "heat1d":
        size = int(rng.integers(9, 64)); p = rng.uniform(0.2, 3.0, (nb, 1)); atol = [float(10.0 ** rng.uniform(-9, -5))]; t_eval = np.sort(rng.uniform(1e-3, 0.3, 3)).tolist()
    elif model == "spm":
        size = int(rng.integers(4, 31)); p = rng.uniform(0.5, 1.5, (nb, 1)); atol = [float(10.0 ** rng.uniform(-8, -5))]; t_eval = np.sort(rng.uniform(100.0, 6000.0, 4)).tolist()
    elif model == "gaussian_decay":
        size = int(rng.integers(9, 40)); p = rng.uniform(0.3, 3.0, (nb, size)); atol = [float(10.0 ** rng.uniform(-9, -5))]; t_eval = np.sort(rng.uniform(0.1, 4.0, 3)).tolist()
    else:
        size = int(rng.integers(2, 8)); p = np.exp(rng.uniform(np.log([0.004, 1e3, 3e6]), np.log([0.4, 1e5, 3e8]), (nb, 3)))
        atol = (10.0 ** rng.uniform(-14, -6, 3)).tolist() * size; t_eval = np.sort(10.0 ** rng.uniform(-2, 3, 4)).tolist()
    tol = dict(rtol=rtol, atol=atol)
    tag = f"seed {seed}: {model}(size {size}) method {method} nb {nb} rtol {rtol:.1e}"
    try:
        s = H.Solver(model, p, nbatch=nb, model_size=size, method=method, **tol)
        y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
    except H.DiffsolHipError as e:
        print("FAIL", tag, "device error", str(e)[:100]); bad += 1; continue
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL[model], p, t_eval, model_size=size, nthreads=16, group=1, method=method, **tol)
    ref = O.solve_dense_independent.last_roots
    okm = m["status"] == 0
    yy = np.transpose(yo, (1, 0, 2))
    good = (int((~okm).sum()) == failed and np.array_equal(y[:, okm], yy[:, okm], equal_nan=True) and np.array_equal(m["stats"].T[okm], so[okm])
            and np.array_equal(m["root_idx"][okm], ref["root_idx"][okm]) and np.array_equal(m["ncols"][okm], ref["ncols"][okm])
            and np.array_equal(m["t_root"][okm], ref["t_root"][okm], equal_nan=True))
    print("ok  " if good else "FAIL", tag, f"failed members {failed}, events {(m['root_idx'] >= 0).sum()}", flush=True)
    bad += 0 if good else 1
print(f"{ncases - bad} of {ncases} configurations bit-identical")
sys.exit(1 if bad else 0)
