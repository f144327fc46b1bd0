import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import diffsol_amd as H
from bench import robertson_params
def timeit(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best
nb = 262144
cur = np.random.default_rng(12345).uniform(0.6, 1.4, (nb, 1))
for name, p in (("random", cur), ("sorted", np.sort(cur, axis=0))):
    s = H.Solver("spm", p, nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6])
    print("C4", name, timeit(lambda: s.solve_dense_adaptive([600.0, 1800.0, 3600.0], want_host=False, group=1)))
nb = 100000
p = robertson_params(nb)
T = [0.4, 4.0, 40.0, 400.0, 4e3, 4e4, 4e5]
lp = np.log(p)
q = ((lp - lp.min(0)) / (lp.max(0) - lp.min(0)) * 1023).astype(np.uint64)
def morton(q):
    out = np.zeros(len(q), dtype=np.uint64)
    for bit in range(10):
        for d in range(3):
            out |= ((q[:, d] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + d)
    return out
orders = {"random": np.arange(nb), "by k1": np.argsort(p[:, 0]), "by k3": np.argsort(p[:, 2]), "by k2": np.argsort(p[:, 1]), "morton": np.argsort(morton(q))}
for name, o in orders.items():
    s = H.Solver("robertson_ode", p[o], nbatch=nb, model_size=1, rtol=1e-4, atol=[1e-8, 1e-14, 1e-6])
    print("C2 per-member", name, timeit(lambda: s.solve_dense_adaptive(T, want_host=False, group=1)))
    if name in ("random", "morton"):
        print("C2 group-64  ", name, timeit(lambda: s.solve_dense_adaptive(T, want_host=False, group=64)))
