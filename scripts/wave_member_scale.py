"""How the wavefront-per-member BDF (8 < n <= 64) fills the device: wall time of robertson_ode x ngroups ensembles of 256 ... 4096 members (one wavefront each).
    python scripts/wave_member_scale.py [ngroups ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DSH_RESIDENT_LANE"] = "0"
import diffsol_amd as H
from bench import robertson_params
T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]
for groups in [int(a) for a in sys.argv[1:]] or [10, 20]:
    n = 3 * groups
    for nb in (256, 512, 1024, 2048, 4096):
        s = H.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=groups, rtol=1e-4, atol=[1e-4] * n)
        s.solve_dense_adaptive(T_EVAL, want_host=False, group=1)
        t0 = time.perf_counter(); out = s.solve_dense_adaptive(T_EVAL, group=1); dt = time.perf_counter() - t0
        tot = out[1]
        print(f"n = {n}, {nb} members: {dt * 1e3:.2f} ms; per member: steps {tot['number_of_steps'] / nb:.0f}, Newton iterations {tot['number_of_nonlinear_solver_iterations'] / nb:.0f}, "
              f"factorisations {tot['number_of_linear_solver_setups'] / nb:.0f}", flush=True)
        del s
