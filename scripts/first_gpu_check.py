"""First end-to-end GPU check: parity vs oracle (fused + trait modes) and a first timing at N=1e5."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import diffsol_amd
from diffsol_amd import Solver
from oracle import oracle as O

def params(nb, seed=12345):
    rng = np.random.default_rng(seed)
    return np.stack([np.exp(rng.uniform(np.log(0.02), np.log(0.08), nb)), np.exp(rng.uniform(np.log(0.5e4), np.log(2e4), nb)),
                     np.exp(rng.uniform(np.log(1.5e7), np.log(6e7), nb))], axis=1)

KEYS = diffsol_amd.STAT_NAMES
kw = dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
tp = [0.0] + [0.4 * 10 ** k for k in range(0, 12)]

# 1. single-system pins (must reproduce reference snapshot counters)
for fused in (True, False):
    s = Solver("robertson_ode", [0.04, 1e4, 3e7], model_size=1, fused=fused, **kw)
    y, _ = s.solve_to_points(tp)
    o = O.OracleSolver(O.MODEL_ROBERTSON_ODE, [0.04, 1e4, 3e7], model_size=1, **kw)
    yo, _ = o.solve_to_points(tp)
    print("fused" if fused else "trait", "single robertson: bit-equal", np.array_equal(y, yo), "maxabs", np.abs(y - yo).max(), s.stats() == o.stats())
    if s.stats() != o.stats():
        print(s.stats()); print(o.stats())

# 2. batched lock-step parity
for nb in (2, 67, 1000):
    p = params(nb)
    for fused in (True, False):
        s = Solver("robertson_ode", p, nbatch=nb, model_size=1, fused=fused, **kw)
        y, _ = s.solve_to_points(tp)
        o = O.OracleSolver(O.MODEL_ROBERTSON_ODE, p, nbatch=nb, model_size=1, **kw)
        yo, _ = o.solve_to_points(tp)
        print(f"nb={nb} fused={fused}: bit-equal {np.array_equal(y, yo)} maxrel {np.max(np.abs(y-yo)/(np.abs(yo)+1e-300)):.3e} stats equal {s.stats()==o.stats()}")

# 3. other models
for name, mid, p, kw2, pts in [
    ("exponential_decay", O.MODEL_EXPONENTIAL_DECAY, [0.1, 1.0], dict(h0=1.0), np.arange(10.0)),
    ("exponential_decay_with_algebraic", O.MODEL_EXPONENTIAL_DECAY_ALGEBRAIC, [0.1], dict(), np.arange(10.0) / 10),
    ("robertson", O.MODEL_ROBERTSON_DAE, [0.04, 1e4, 3e7], dict(rtol=1e-4, atol=[1e-8, 1e-6, 1e-6]), tp),
]:
    for method in (0, 1, 2):
        for fused in (True, False):
            s = Solver(name, p, method=method, fused=fused, **kw2)
            y, _ = s.solve_to_points(pts)
            o = O.OracleSolver(mid, p, method=method, **kw2)
            yo, _ = o.solve_to_points(pts)
            print(f"{name} method={method} fused={fused}: bit-equal {np.array_equal(y, yo)} maxabs {np.abs(y-yo).max():.3e} stats equal {s.stats()==o.stats()}")
            if s.stats() != o.stats(): print(s.stats(), o.stats())

# 4. timing at N = 1e5
nb = 100_000
p = params(nb)
for block in (64, 128, 256):
    s = Solver("robertson_ode", p, nbatch=nb, model_size=1, block_threads=block, **kw)
    t0 = time.perf_counter()
    y, ncols, reason = s.solve(4e5)
    dt = time.perf_counter() - t0
    st = s.stats()
    print(f"N={nb} block={block}: {dt*1e3:.1f} ms, steps={st['number_of_steps']} newton={st['number_of_nonlinear_solver_iterations']} "
          f"setups={st['number_of_linear_solver_setups']} -> {nb*st['number_of_nonlinear_solver_iterations']/dt:.3e} newton solves/s, "
          f"{nb*st['number_of_steps']/dt:.3e} steps/s, mass-conservation err {np.abs(y.sum(1)-1).max():.2e}")
