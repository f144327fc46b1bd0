"""Randomised parity sweep of the HOST-DRIVEN lock-step integrators (fused kernels, trait operations, banded LU, difference-array kernels) against the CPU
oracle's lock-step batched run (libm pow on both sides), bit for bit: states at random output times, all counters, root stops.
python scripts/fuzz_lockstep.py [nseeds]   (needs a GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffsol_amd as H  # noqa: E402
from oracle import oracle as O  # noqa: E402
from helpers import ORACLE_MODEL  # noqa: E402

O.build()
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
base = int(os.environ.get("FUZZ_BASE", "2000"))
bad = 0
for seed in range(nseeds):
    rng = np.random.default_rng(base + seed)
    model = ["robertson_ode", "robertson", "rlc", "exponential_decay", "exponential_decay_with_algebraic", "heat1d", "spm", "gaussian_decay", "dydt_y2"][seed % 9]
    method = int(rng.integers(0, 3))
    nb = int(rng.choice([1, 2, 37, 64, 300, 1000])) if seed % 5 else 8192
    rtol = float(10.0 ** rng.uniform(-8, -3))
    fused = bool(rng.integers(0, 2))
    size = 0
    if model in ("robertson_ode", "robertson"):
        size = int(rng.choice([1, 1, 3])) if model == "robertson_ode" else 0
        p = np.exp(rng.uniform(np.log([0.01, 3e3, 1e7]), np.log([0.1, 3e4, 1e8]), (nb, 3)))
        atol = (10.0 ** rng.uniform(-12, -6, 3)).tolist() * max(size, 1)
        times = np.sort(10.0 ** rng.uniform(-1, 3, 3)).tolist()
    elif model == "rlc":
        p = np.stack([rng.uniform(50, 200, nb), np.ones(nb), np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), nb)), np.full(nb, 10.0), np.full(nb, 100.0), np.full(nb, 0.05)], axis=1)
        atol = [float(10.0 ** rng.uniform(-8, -5))] * 4
        times = np.sort(rng.uniform(1e-3, 0.1, 3)).tolist()
    elif model == "exponential_decay":
        p = np.stack([rng.uniform(0.05, 2.0, nb), rng.uniform(0.5, 3.0, nb)], axis=1)
        atol = [float(10.0 ** rng.uniform(-9, -5))] * 2
        times = np.sort(rng.uniform(0.1, 20, 4)).tolist()
    elif model == "exponential_decay_with_algebraic":
        p = rng.uniform(0.05, 3.0, (nb, 1))
        atol = [float(10.0 ** rng.uniform(-9, -5))] * 3
        times = np.sort(rng.uniform(0.1, 10, 3)).tolist()
    elif model == "heat1d":
        size = int(rng.choice([9, 16, 24, 40, 70]))
        p = rng.uniform(0.3, 2.5, (nb, 1))
        atol = [float(10.0 ** rng.uniform(-8, -5))]
        times = np.sort(rng.uniform(1e-3, 0.1, 2)).tolist()
    elif model == "spm":
        size = int(rng.choice([4, 8, 20]))
        p = rng.uniform(0.6, 0.9, (nb, 1))
        atol = [float(10.0 ** rng.uniform(-8, -5))]
        times = np.sort(rng.uniform(10.0, 400.0, 2)).tolist()
    elif model == "gaussian_decay":
        size = int(rng.integers(2, 20))
        p = rng.uniform(0.1, 2.0, (nb, size))
        atol = [float(10.0 ** rng.uniform(-8, -5))]
        times = np.sort(rng.uniform(0.1, 3.0, 3)).tolist()
    else:
        size = int(rng.integers(2, 20))
        p = np.zeros((nb, 0))
        atol = [float(10.0 ** rng.uniform(-8, -5))]
        times = np.sort(rng.uniform(0.001, 0.004, 2)).tolist()  # y' = y^2 from y0 = -200
    if nb == 8192 and model in ("heat1d",) and size > 24:
        size = 16
    tol = dict(rtol=rtol, atol=atol)
    tag = f"seed {seed}: {model}(size {size}) method {method} nb {nb} rtol {rtol:.1e} fused {fused}"
    s = H.Solver(model, p, nbatch=nb, model_size=size, method=method, fused=fused, **tol)
    o = O.OracleSolver(ORACLE_MODEL[model], p, nbatch=nb, model_size=size, method=method, **tol)
    y, r = s.solve_to_points(times)
    yo, ro = o.solve_to_points(times)
    good = np.array_equal(y, yo) and s.stats() == o.stats() and int(r) == int(ro)
    bad += 0 if good else 1
    print(tag, "->", "OK" if good else f"MISMATCH states {np.array_equal(y, yo)} stats {s.stats() == o.stats()} root {r} {ro}", flush=True)
print("mismatching configurations:", bad)
sys.exit(1 if bad else 0)
