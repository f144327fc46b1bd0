#!/usr/bin/env python
"""Randomised parity sweep of the per-member device forms of DENSE run-time-compiled (DiffSL) models against independent oracle solves, bit for bit:
one wavefront per member (n <= 64) and one workgroup per member (64 < n <= 140), BDF / TR-BDF2 / ESDIRK34, as plain ODE models, as hybrid models
(stop_i + reset_i: every member its own event times) and with forward sensitivities (with and without sensitivity error control); random sizes, couplings,
tolerances, parameters and output times.      python scripts/fuzz_member_diffsl.py [ncases] [first_seed]     (GPU only)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffsol_amd as H
import diffsl_models as D
from diffsol_amd import diffsl as fe
from oracle import oracle as O

O.build()
O.set_det_pow(True)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for seed in range(first, first + ncases):
    rng = np.random.default_rng(9100 + seed)
    kind = ["plain", "hybrid", "sens"][seed % 3]
    big = bool(rng.integers(0, 2))
    n = int(rng.integers(65, 141)) if big else int(rng.integers(5, 65))
    method = int(rng.integers(0, 3))  # (the workgroup-per-member form has all three methods since k_sdirk_wave_member<.., TW>)
    nb = int(rng.integers(6, 40))
    rtol = float(10.0 ** rng.uniform(-7, -4))
    atol = [float(10.0 ** rng.uniform(-9, -6))]
    cpl = float(rng.uniform(0.02, 0.3)) / n
    w = ", ".join(f"({i}): {float(1.0 + rng.uniform(0.0, 1.5) * i / n)!r}" for i in range(n))
    base = (f"in = [k]\nk {{ 0.5 }}\nS_ij {{ (0:{n}, 0:{n}): {cpl!r} }}\nw_i {{ {w} }}\nu_i {{ (0:{n}): 1.0 }}\ncpl_i {{ S_ij * u_j }}\n")
    if kind == "hybrid":
        thr = float(rng.uniform(0.4, 0.7))
        code = base + f"F_i {{ -k * w_i * u_i - cpl_i }}\nstop_i {{ u_i[0:1] - {thr!r} }}\nreset_i {{ 0.5 * u_i + {float(rng.uniform(0.3, 0.5))!r} }}\n"
    elif kind == "sens":
        code = base + "F_i { -k * w_i * u_i - cpl_i + 0.05 * k * k }\n"
    else:
        code = base + "F_i { -k * w_i * u_i - cpl_i * u_i }\n"
    p = rng.uniform(0.2, 2.0, (nb, 1))
    t_eval = [0.0] + np.sort(rng.uniform(0.05, 8.0, 4)).tolist()
    tol = dict(rtol=rtol, atol=atol)
    hm = [H.METHOD_BDF, H.METHOD_TR_BDF2, H.METHOD_ESDIRK34][method]
    om = [O.METHOD_BDF, O.METHOD_TR_BDF2, O.METHOD_ESDIRK34][method]
    tag = f"seed {seed}: {kind} n {n} method {method} nb {nb} rtol {rtol:.1e} atol {atol[0]:.1e}"
    try:
        m, mid = fe.DiffslModel(code, form=fe.FORM_DYNAMIC if n <= 8 else None), D.host_model(O, code)  # n <= 8: the default (static) form has no per-member kernel above n = 4
        if kind == "sens":
            ec = bool(rng.integers(0, 2))
            kw = dict(sens_rtol=float(10.0 ** rng.uniform(-6, -4)), sens_atol=[float(10.0 ** rng.uniform(-8, -6))]) if ec else {}
            s = H.Solver(m, p, nbatch=nb, sens=True, method=hm, **kw, **tol)
            y, sens, tot, mm = s.solve_dense_adaptive_sens(t_eval[1:], group=1, want_member_stats=True)
            yo, so, sto, failed = O.solve_dense_independent_sens(mid, p, t_eval[1:], nthreads=8, group=1, method=om, **kw, **tol)
            ok = (np.array_equal(mm["stats"].T, sto) and np.array_equal(y, np.transpose(yo, (1, 0, 2)), equal_nan=True)
                  and np.array_equal(sens, np.transpose(so, (0, 2, 1, 3)), equal_nan=True) and tot["failed_members"] == failed)
            extra = f"sens error control {ec}"
        else:
            s = H.Solver(m, p, nbatch=nb, method=hm, **tol)
            y, tot, mm = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=1)
            yo, so, failed = O.solve_dense_independent(mid, p, t_eval, nthreads=8, method=om, **tol)
            lr = O.solve_dense_independent.last_roots
            ok = (np.array_equal(mm["stats"].T, so) and np.array_equal(y, np.transpose(yo, (1, 0, 2)), equal_nan=True) and tot["failed_members"] == failed
                  and np.array_equal(mm["t_root"], lr["t_root"], equal_nan=True) and np.array_equal(mm["root_idx"], lr["root_idx"]) and np.array_equal(mm["ncols"], lr["ncols"]))
            extra = f"events {(lr['root_idx'] >= 0).sum()}"
        print(("ok   " if ok else "FAIL ") + tag + f" | steps {int(mm['stats'][:, 0].sum())} failed {failed} {extra}", flush=True)
        bad += 0 if ok else 1
    except Exception as e:  # noqa: BLE001
        print("ERR  " + tag + f" | {type(e).__name__}: {e}", flush=True)
        bad += 1
print(f"{ncases - bad} of {ncases} configurations bit-identical")
sys.exit(1 if bad else 0)
