"""Randomised parity configurations shared by tests/test_gpu_fuzz.py (a seed-pinned subset in the GPU tier) and scripts/fuzz_parity.py /
scripts/fuzz_lockstep.py (long sweeps).  Each function builds ONE configuration from a seed, runs it on the GPU and on the CPU oracle and
returns (ok, message); bit-for-bit comparisons throughout."""
import numpy as np

from helpers import ORACLE_MODEL


def resident_case(H, O, seed, base=1000):
    """Device-resident integrators (dshs_solve_dense_adaptive) vs the oracle with the deterministic pow: models x methods x control granularity x
    tolerances x parameter ranges, including runs that FAIL (status codes must agree too).  The caller sets O.set_det_pow(True)."""
    rng = np.random.default_rng(base + seed)
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
        return False, tag + " -> device error: " + str(e)[:120]
    yo, so, failed = O.solve_dense_independent(ORACLE_MODEL[model], p, t_eval, model_size=size, nthreads=16, group=group, method=method, **tol)
    ref = O.solve_dense_independent.last_roots
    ok_members = m["status"] == 0
    same_fail = int((~ok_members).sum()) == failed
    yy = np.transpose(yo, (1, 0, 2))
    states = np.array_equal(y[:, ok_members], yy[:, ok_members], equal_nan=True)
    stats = np.array_equal(m["stats"].T[ok_members], so[ok_members])
    roots = np.array_equal(m["t_root"][ok_members], ref["t_root"][ok_members], equal_nan=True) and np.array_equal(m["root_idx"][ok_members], ref["root_idx"][ok_members])
    good = same_fail and states and stats and roots
    return bool(good), tag + " -> " + ("OK" if good else f"MISMATCH fail {same_fail} states {states} stats {stats} roots {roots}") + \
        f" (failed members {failed}, events {int((m['root_idx'] >= 0).sum())})"


def lockstep_case(H, O, seed, base=2000):
    """HOST-DRIVEN lock-step integrators (fused kernels, trait operations, banded LU, difference-array kernels) vs the oracle's lock-step batched run
    (libm pow on both sides): states at random output times, all counters, root stops."""
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
    return bool(good), tag + " -> " + ("OK" if good else f"MISMATCH states {np.array_equal(y, yo)} stats {s.stats() == o.stats()} root {r} {ro}")
