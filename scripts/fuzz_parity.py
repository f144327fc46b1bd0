"""Randomised parity sweep of the device-resident integrators against the CPU oracle (bit for bit, deterministic pow): models x methods x control granularity x
tolerances x parameter ranges, including runs that FAIL (status codes must agree too).  python scripts/fuzz_parity.py [nseeds]   (needs a GPU)"""
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
O.set_det_pow(True)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bad = 0
for seed in range(nseeds):
    rng = np.random.default_rng(int(os.environ.get("FUZZ_BASE", "1000")) + seed)
    model = ["robertson_ode", "robertson", "rlc", "exponential_decay_with_root", "exponential_decay_with_algebraic", "heat1d", "spm"][seed % 7]
    method = int(rng.integers(0, 3))
    group = int(rng.choice([1, 64]))
    nb = int(rng.integers(65, 400))
    rtol = float(10.0 ** rng.uniform(-9, -3))
    size = 0
    if model in ("robertson_ode", "robertson"):
        size = 1 if model == "robertson_ode" else 0
        p = np.exp(rng.uniform(np.log([0.004, 1e3, 3e6]), np.log([0.4, 1e5, 3e8]), (nb, 3)))
        atol = (10.0 ** rng.uniform(-14, -6, 3)).tolist()
        t_eval = np.sort(10.0 ** rng.uniform(-2, 5, 5)).tolist()
    elif model == "rlc":
        size = 1
        p = np.stack([rng.uniform(20, 400, nb), rng.uniform(0.5, 2, nb), np.exp(rng.uniform(np.log(2e-4), np.log(5e-3), nb)), rng.uniform(5, 20, nb), rng.uniform(50, 200, nb),
                      rng.uniform(0.01, 0.2, nb) if group == 1 else np.full(nb, 1e3)], axis=1)
        atol = [float(10.0 ** rng.uniform(-9, -5))] * 4
        t_eval = np.sort(rng.uniform(1e-4, 0.2, 5)).tolist()
    elif model == "exponential_decay_with_root":
        p = np.stack([rng.uniform(0.01, 5.0, nb) if group == 1 else np.full(nb, 1e-9), rng.uniform(0.7, 3.0, nb)], axis=1)
        atol = [float(10.0 ** rng.uniform(-10, -5))] * 2
        t_eval = np.sort(rng.uniform(0.1, 30, 6)).tolist()
    elif model == "exponential_decay_with_algebraic":
        p = rng.uniform(0.05, 5.0, (nb, 1))
        atol = [float(10.0 ** rng.uniform(-10, -5))] * 3
        t_eval = np.sort(rng.uniform(0.1, 20, 4)).tolist()
    elif model == "heat1d":
        size = int(rng.integers(9, 30))
        p = rng.uniform(0.2, 3.0, (nb, 1))
        atol = [float(10.0 ** rng.uniform(-9, -5))]
        t_eval = np.sort(rng.uniform(1e-3, 0.3, 3)).tolist()
    else:
        size, method = int(rng.integers(4, 12)), 0 if group == 64 else method
        p = rng.uniform(0.5, 1.5, (nb, 1))
        atol = [float(10.0 ** rng.uniform(-8, -5))]
        t_eval = np.sort(rng.uniform(100.0, 6000.0 if group == 1 else 900.0, 4)).tolist()
    tol = dict(rtol=rtol, atol=atol)
    tag = f"seed {seed}: {model}(size {size}) method {method} group {group} nb {nb} rtol {rtol:.1e}"
    try:
        s = H.Solver(model, p, nbatch=nb, model_size=size, method=method, **tol)
        y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group)
    except H.DiffsolHipError as e:
        print(tag, "-> device error:", str(e)[:120])
        bad += 1
        continue
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL[model], p, t_eval, model_size=size, nthreads=16, group=group, method=method, **tol)
    ref = O.solve_dense_independent.last_roots
    ok_members = m["status"] == 0
    same_fail = int((~ok_members).sum()) == failed
    yy = np.transpose(yo, (1, 0, 2))
    states = np.array_equal(y[:, ok_members], yy[:, ok_members], equal_nan=True)
    stats = np.array_equal(m["stats"].T[ok_members], so[ok_members])
    roots = np.array_equal(m["t_root"][ok_members], ref["t_root"][ok_members], equal_nan=True) and np.array_equal(m["root_idx"][ok_members], ref["root_idx"][ok_members])
    good = same_fail and states and stats and roots
    bad += 0 if good else 1
    print(tag, "->", "OK" if good else f"MISMATCH fail {same_fail} states {states} stats {stats} roots {roots}", f"(failed members {failed}, events {(m['root_idx'] >= 0).sum()})", flush=True)
print("mismatching configurations:", bad)
sys.exit(1 if bad else 0)
