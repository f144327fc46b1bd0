#!/usr/bin/env python
"""General-bandwidth banded LU (dsh_lu_gband.hpp) against the dense route on the same operands (profiles/r06_band_general.md):
    python scripts/band_general_bench.py [nb=4096] [reps=5]          (GPU only)
Operands: 2-D 5-point Laplacian-like bands (heat2d: n = m^2, k = m) with random entries inside the band, dense containers.  Per (n, k): ms per factorisation and
per solve of both routes (wall clock over `reps` calls, stream drained), the algorithmic bytes of the banded route — factor: read (kl + ku + 1) n + write
(2 kl + ku + 1) n doubles + 4 n pivot bytes; solve: read (2 kl + ku + 1) n doubles + 4 n + 16 n of right-hand side — as TB/s, and whether the solutions agree bit for bit."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import diffsol_amd as H
from diffsol_amd import _ffi

L = _ffi.load_device_lib()


def check(rc):
    assert rc == 0, L.dsh_last_error()


nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = H.HipContext(0, nbatch=nb)
rng = np.random.default_rng(0)
print("| n | k | nb | band factor ms | TB/s | dense factor ms | band solve ms | TB/s | dense solve ms | same bits |")
print("|---|---|---|---|---|---|---|---|---|---|")
for n, k in ((100, 10), (200, 20), (400, 20), (400, 40), (900, 30), (1024, 64)):
    a1 = np.zeros((n, n))
    for d in range(-k, k + 1):
        i = np.arange(max(0, -d), min(n, n - d))
        a1[i, i + d] = rng.standard_normal(i.size)
    a1[np.arange(n), np.arange(n)] += 3.0
    a = H.HipMat.from_array(np.broadcast_to(a1, (nb, n, n)).copy(), ctx)
    b0 = rng.standard_normal((nb, n))
    out = {}
    for route in ("band", "dense"):
        lu = H.HipLU(ctx, n)
        lu.set_structure(route == "dense")
        # the band route with the band DECLARED (dsh_lu_factor_banded: what the integrators call for a model that declares its band) — dsh_lu_factor would first
        # probe the n^2 entries of every member for the bandwidth (one read of the dense container and a host round trip per factorisation)
        fac = (lambda: check(L.dsh_lu_factor_banded(lu._h, a.ptr, k, k))) if route == "band" else (lambda: lu.factor(a))
        fac(); ctx.sync()
        assert lu.band_width() == (k if route == "band" else 0), (route, lu.band_width())
        t0 = time.perf_counter()
        for _ in range(reps):
            fac()
        ctx.sync()
        tf = (time.perf_counter() - t0) / reps
        b = H.HipVec.from_vec(b0, ctx)
        lu.solve_in_place(b); ctx.sync()
        x = np.asarray(b.clone_as_vec()).copy()
        t0 = time.perf_counter()
        for _ in range(reps * 4):
            lu.solve_in_place(b)
        ctx.sync()
        ts = (time.perf_counter() - t0) / (reps * 4)
        out[route] = (tf, ts, x)
        del lu
    fb = (8 * ((2 * k + 1) + (3 * k + 1)) * n + 4 * n) * nb
    sb = (8 * (3 * k + 1) * n + 4 * n + 16 * n) * nb
    same = np.array_equal(out["band"][2], out["dense"][2])
    print(f"| {n} | {k} | {nb} | {out['band'][0]*1e3:.3f} | {fb/out['band'][0]/1e12:.3f} | {out['dense'][0]*1e3:.3f} | {out['band'][1]*1e3:.3f} | {sb/out['band'][1]/1e12:.3f} | "
          f"{out['dense'][1]*1e3:.3f} | {same} |", flush=True)
    del a
