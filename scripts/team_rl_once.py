"""One ensemble through the workgroup-per-member BDF with the LU in registers, for rocprofv3 (profiles/r06_team_rl_kernel_stats.md):
    rocprofv3 --kernel-trace --stats -d <dir> -o trace -- python scripts/team_rl_once.py [ngroups] [members]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DSH_RESIDENT_LANE"] = "0"
import diffsol_amd as H
from bench import robertson_params
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = 3 * groups
T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]
s = H.Solver("robertson_ode", robertson_params(nb), nbatch=nb, model_size=groups, rtol=1e-4, atol=[1e-4] * n)
for rep in range(3):
    t0 = time.perf_counter(); y, tot = s.solve_dense_adaptive(T_EVAL, group=1); dt = time.perf_counter() - t0
print(f"robertson_ode x {groups} (n = {n}), {nb} members: {dt * 1e3:.2f} ms wall; failed {tot['failed_members']}")
