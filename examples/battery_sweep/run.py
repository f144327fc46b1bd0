"""A current sweep of the single-particle battery model written in DiffSL — the reference's physics-based-battery-simulation example
(examples/physics-based-battery-simulation: one solve per current, 0.6 ... 1.4 A) as ONE ensemble on the GPU.

    python examples/battery_sweep/run.py [nmembers]

Every member discharges at its own current until its terminal voltage reaches 3.105 V (stop_i) or one hour has passed; all members are integrated in a single
launch (banded lane-per-member BDF: the model's Jacobian is tridiagonal), each with its own step sizes and its own cut-off time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # the DiffSL text of the model lives with the tests (tests/diffsl_models.py)

from diffsol_amd import Solver, diffsl  # noqa: E402
import diffsl_models  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
currents = np.linspace(0.6, 1.4, nb)
model = diffsl.DiffslModel(diffsl_models.spm(20, voltage=True))          # DiffSL -> HIP source -> hiprtc (cached on disk)
print(f"model: {model.n} states, {model.nroots} stop conditions, Jacobian bandwidth {model.band[:2]}, lane-per-member form: {model.lane_model_id is not None}")
solver = Solver(model, currents[:, None], nbatch=nb, rtol=1e-6, atol=[1e-6])
t_eval = np.linspace(360.0, 3600.0, 10)
solver.solve_dense_adaptive(t_eval, want_host=False)                       # first call compiles the integrator for this model
t0 = time.perf_counter()
y, totals, member = solver.solve_dense_adaptive(t_eval, want_member_stats=True)
dt = time.perf_counter() - t0
stopped = member["root_idx"] >= 0
print(f"{nb} members in {dt * 1e3:.1f} ms: {totals['number_of_steps']} steps, {totals['number_of_nonlinear_solver_iterations']} Newton iterations, "
      f"{totals['number_of_linear_solver_setups']} LU factorisations, {totals['failed_members']} failures")
print(f"{int(stopped.sum())} members reached the cut-off voltage, between t = {np.nanmin(member['t_root']):.1f} s (I = {currents[np.nanargmin(member['t_root'])]:.3f} A) "
      f"and t = {np.nanmax(member['t_root']):.1f} s (I = {currents[np.nanargmax(member['t_root'])]:.3f} A)")
capacity = np.where(stopped, currents * member["t_root"] / 3600.0, currents)  # Ah delivered
print(f"delivered capacity: {capacity.min():.4f} ... {capacity.max():.4f} Ah; state 0 of the model integrates it: max |difference| at the last valid column "
      f"{np.nanmax(np.abs(np.array([y[member['ncols'][b] - 1, b, 0] for b in range(0, nb, max(nb // 64, 1))]) - capacity[::max(nb // 64, 1)])):.2e}")
