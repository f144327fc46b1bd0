#!/usr/bin/env python
"""Time dsh_lu_factor / dsh_lu_solve alone at a given (n, nbatch):  python scripts/lu_bench.py 512 4096 [reps]   (GPU only).
Reports ms per call, GFLOP/s of the factorisation (2/3 n^3 per system) and GB/s of the solve (8 n^2 bytes of factors per system)."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H

n, nb = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
kind = sys.argv[4] if len(sys.argv) > 4 else "dense"
ctx = H.HipContext(0, nbatch=nb)
rng = np.random.default_rng(0)
if kind == "dense":
    a1 = rng.standard_normal((n, n))
else:  # tridiagonal, diagonally dominant (heat1d-like: no pivoting)
    a1 = np.diag(np.full(n, 4.0)) + np.diag(np.full(n - 1, -1.0), 1) + np.diag(np.full(n - 1, -1.0), -1)
a = H.HipMat.from_array(np.broadcast_to(a1, (nb, n, n)).copy(), ctx)
b = H.HipVec.from_vec(rng.standard_normal((nb, n)), ctx)
lu = H.HipLU(ctx, n)
lu.factor(a); ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    lu.factor(a)
ctx.sync()
tf = (time.perf_counter() - t0) / reps
lu.solve_in_place(b); ctx.sync()
t0 = time.perf_counter()
for _ in range(reps * 4):
    lu.solve_in_place(b)
ctx.sync()
ts = (time.perf_counter() - t0) / (reps * 4)
print(f"n={n} nb={nb} {kind}: factor {tf*1e3:.3f} ms ({2/3*n**3*nb/tf/1e9:.1f} GFLOP/s)   solve {ts*1e3:.3f} ms ({8*n*n*nb/ts/1e9:.1f} GB/s of factors)")
