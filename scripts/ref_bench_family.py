#!/usr/bin/env python
"""The reference's own benchmark family on the GPU: `robertson_ode` replicated ngroups times (n = 3 ngroups; book/src/benchmarks/python.md:9,
python_results.csv), BDF, rtol = atol = tol for tol in (1e-4, 1e-8), run as ENSEMBLES of parameter-sweep members through every device-resident route the library
has for that size, next to the published single-solve time of diffsol's CPU path (one EPYC 7343 core; the t_final / output grid of that benchmark live outside the
reference tree, so the published point is indicative — ours: t in [0, 4e5], 7 save points, bench.py's horizon).

    python scripts/ref_bench_family.py [nb]      -> gpurun_out/r06/ref_family.json + a markdown table on stdout"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffsol_amd as H
from bench import robertson_params

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]  # t in [0, 4e5]: with atol = 1e-4 on the 1e-5-sized component the algorithm (the oracle's CPU runs too) loses members beyond ~1e7
# book/src/benchmarks/python_results.csv, column diffsol_time (seconds per solve): (ngroups, tol) -> s
PUBLISHED = {(1, 1e-4): 3.1152e-05, (1, 1e-8): 8.3342e-05, (10, 1e-4): 1.9593e-04, (10, 1e-8): 4.2949e-04, (20, 1e-4): 3.2475e-04, (20, 1e-8): 6.4901e-04,
             (40, 1e-4): 5.4231e-04, (40, 1e-8): 1.0753e-03, (100, 1e-4): 1.2064e-03, (100, 1e-8): 2.3834e-03}
rows = []
for groups in (1, 10, 20, 40, 100):
    n = 3 * groups
    for tol in (1e-4, 1e-8):
        p = robertson_params(nb)
        if n <= 4:
            routes = [("register-resident, per member", {}, 1), ("register-resident, wavefront lock-step groups of 64", {}, 64)]
        else:
            routes = [("banded lane per member (block-diagonal Jacobian declared)", {"DSH_RESIDENT_LANE": "1"}, 1)]
            if n <= 140:
                routes.append(("wavefront per member" if n <= 32 else ("workgroup per member (LU in registers)" if n <= 128 else "workgroup per member (LU in LDS)"), {"DSH_RESIDENT_LANE": "0"}, 1))
        for name, env, group in routes:
            for k, v in env.items():
                os.environ[k] = v
            try:
                s = H.Solver("robertson_ode", p, nbatch=nb, model_size=groups, rtol=tol, atol=[tol] * n)
                s.solve_dense_adaptive(T_EVAL, want_host=False, group=group)
                t0 = time.perf_counter()
                _, tot = s.solve_dense_adaptive(T_EVAL, want_host=False, group=group)
                wall = time.perf_counter() - t0
                rows.append(dict(ngroups=groups, n=n, tol=tol, route=name, members=nb, wall_s=wall, seconds_per_member=wall / nb, steps_per_member=tot["number_of_steps"] / nb,
                                 newton_per_member=tot["number_of_nonlinear_solver_iterations"] / nb, failed=tot["failed_members"],
                                 published_diffsol_cpu_seconds_per_solve=PUBLISHED.get((groups, tol))))
            except Exception as e:  # noqa: BLE001
                rows.append(dict(ngroups=groups, n=n, tol=tol, route=name, error=str(e)[:200]))
            finally:
                for k in env:
                    os.environ.pop(k, None)
            print(json.dumps(rows[-1]), flush=True)
        if n >= 30:  # host-driven lock-step over the whole ensemble (the trait path; any size)
            try:
                s = H.Solver("robertson_ode", p[:512], nbatch=512, model_size=groups, rtol=tol, atol=[tol] * n, ensemble_mode=H.solver.ENSEMBLE_LOCKSTEP)
                t0 = time.perf_counter()
                s.solve_dense(T_EVAL, want_host=False)
                wall = time.perf_counter() - t0
                st = s.stats()
                rows.append(dict(ngroups=groups, n=n, tol=tol, route="host-driven lock-step over the trait operations (512 members, one step sequence)", members=512, wall_s=wall,
                                 seconds_per_member=wall / 512, steps_per_member=st["number_of_steps"], newton_per_member=st["number_of_nonlinear_solver_iterations"], failed=0,
                                 published_diffsol_cpu_seconds_per_solve=PUBLISHED.get((groups, tol))))
            except Exception as e:  # noqa: BLE001
                rows.append(dict(ngroups=groups, n=n, tol=tol, route="host-driven lock-step", error=str(e)[:200]))
            print(json.dumps(rows[-1]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r06", "ref_family.json"), "w"), indent=1)
lines = ["| ngroups | n | tol | route | members | wall (s) | s / member | steps / member | published diffsol CPU s / solve | members solved per published CPU-solve time |",
         "|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    if "error" in r:
        lines.append(f"| {r['ngroups']} | {r['n']} | {r['tol']:g} | {r['route']} | — | error: {r['error'][:80]} | | | | |")
        continue
    pub = r["published_diffsol_cpu_seconds_per_solve"]
    lines.append(f"| {r['ngroups']} | {r['n']} | {r['tol']:g} | {r['route']} | {r['members']} | {r['wall_s']:.4f} | {r['seconds_per_member']:.3e} | {r['steps_per_member']:.0f} | "
                 + (f"{pub:.3e} | {pub / r['seconds_per_member']:.1f} |" if pub else " | |"))
open(os.path.join(ROOT, "gpurun_out", "r06", "ref_family.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
