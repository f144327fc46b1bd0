#!/usr/bin/env python
"""Randomised sweep of the general-bandwidth banded LU against the oracle's dense LU, bit for bit: random (n, kl, ku <= 64), ensemble sizes, row scalings (pivoting on
every step), dense and band containers, occasional singular members:   python scripts/fuzz_gband.py [nseeds] [first seed]      (GPU only)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("DSH_LU_EXACT", "1")
import diffsol_amd as H  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_gpu_lu_models import _banded, _pack_band  # noqa: E402

O.build()
L = H._ffi.load_device_lib()
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
bad = 0
for seed in range(first, first + nseeds):
    rng = np.random.default_rng(seed)
    kl, ku = int(rng.integers(0, 65)), int(rng.integers(0, 65))
    if max(kl, ku) < 5:
        kl = 5 + int(rng.integers(0, 60))
    w = 2 * (2 * kl + ku + 1)
    n = min(1024, int(rng.integers(w, w + 300)))
    if w > n:
        n = w if w <= 1024 else 0
    if n == 0:
        print(f"skip seed {seed}: ({kl}, {ku}) does not fit n <= 1024"); continue
    nb = int(rng.integers(1, 80)) if n <= 400 else int(rng.integers(1, 10))
    c = H.HipContext(nbatch=nb)
    a = _banded(rng, nb, n, kl, ku, bool(rng.integers(0, 2))) * np.exp(rng.uniform(-2, 2, (nb, n, 1)))
    sing = []
    if rng.integers(0, 4) == 0 and nb > 2:  # a singular member: a zero column
        sing = [int(rng.integers(0, nb))]
        a[sing[0], :, int(rng.integers(0, n))] = 0.0
    b = rng.standard_normal((nb, n))
    packed = bool(rng.integers(0, 2))
    lu = H.HipLU(c, n)
    if packed:
        Ab = H.HipVec.from_vec(_pack_band(a, kl, ku), c)
        assert L.dsh_lu_factor_packed(lu._h, Ab.ptr, kl, ku) == 0, L.dsh_last_error()
    else:
        lu.factor(H.HipMat.from_array(a, c))
    ok = lu.band_width() == max(kl, ku) and lu.n_singular() == len(sing)
    x = H.HipVec.from_vec(b, c)
    try:
        lu.solve_in_place(x)
        failed = False
    except H.DiffsolHipError:
        failed = True
    good = [i for i in range(nb) if i not in sing]
    xo, _, _, rc = O.lu_solve(a[good], b[good])
    ok = ok and failed == bool(sing) and rc == 0 and np.array_equal(np.asarray(x.clone_as_vec())[good], xo)
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} seed {seed}: n {n} kl {kl} ku {ku} nb {nb} {'band container' if packed else 'dense container'} singular {sing}", flush=True)
print("mismatching configurations:", bad)
sys.exit(1 if bad else 0)
