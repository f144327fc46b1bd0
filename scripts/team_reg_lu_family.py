"""The workgroup-per-member integrators on the reference's benchmark family (robertson_ode x 30 / x 40: n = 90 / 120; book/src/benchmarks/python_results.csv) with the LU
factors in registers (csrc/dsh_team_reg_lu.hpp, the default) and in LDS (DSH_TEAM_REG_LU=0): wall time of an ensemble and whether the two forms return the same bits.
    python scripts/team_reg_lu_family.py [members] [bdf|tr_bdf2|esdirk34]"""
import os, sys, time
sys.path.insert(0, '/root/repo')
import diffsol_amd as H
from bench import robertson_params
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]
os.environ["DSH_RESIDENT_LANE"] = "0"
import numpy as np
METHOD = {"bdf": H.METHOD_BDF, "tr_bdf2": H.METHOD_TR_BDF2, "esdirk34": H.METHOD_ESDIRK34}[sys.argv[2] if len(sys.argv) > 2 else "bdf"]
res = {}
for groups in (30, 40):
    n = 3 * groups
    for tol in (1e-4, 1e-8):
        for rl in ("0", "1"):
            os.environ["DSH_TEAM_REG_LU"] = rl
            p = robertson_params(nb)
            s = H.Solver("robertson_ode", p, nbatch=nb, model_size=groups, method=METHOD, rtol=tol, atol=[tol] * n)
            s.solve_dense_adaptive(T_EVAL, want_host=False, group=1)
            t0 = time.perf_counter()
            out = s.solve_dense_adaptive(T_EVAL, want_host=True, group=1)
            dt = time.perf_counter() - t0
            y = np.asarray(out[0] if isinstance(out, tuple) else out)
            res[(groups, tol, rl)] = y
            print(f"n={n} tol={tol:g} reg_lu={rl}: {dt:.4f} s for {nb} members; finite {np.isfinite(y).all()}", flush=True)
        a, b = res[(groups, tol, "0")], res[(groups, tol, "1")]
        print("   same bits:", np.array_equal(a, b), flush=True)
