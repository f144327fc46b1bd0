#!/bin/bash
# Where a member's solve spends its cycles in the workgroup-per-member BDF (k_bdf_team_member, -DDSH_TEAM_MEMBER_PROF: thread 0 of workgroup 0 prints its phase clocks).
#   build:  bash scripts/team_member_prof.sh build      run (GPU box):  bash scripts/team_member_prof.sh run
cd "$(dirname "$0")/.."
CS=diffsol_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function"
d=diffsol_amd/lib_exp/team_prof
if [ "$1" = build ]; then
  mkdir -p $d
  /opt/rocm/bin/hipcc $FL -DDSH_TEAM_MEMBER_PROF -c $CS/dsh_wave_member.hip -o $d/dsh_wave_member.o 2> $d/build.log &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libdiffsol_hip.so $d/dsh_wave_member.o $(ls diffsol_amd/lib/obj/*.o | grep -v "/dsh_wave_member.o") -L/opt/rocm/lib -lhiprtc -ldl &&
  cp diffsol_amd/lib/libdiffsol_hip_host.so $d/ && rm $d/dsh_wave_member.o && echo "built $d"
else
  DSH_RESIDENT_LANE=0 DSH_LIB_DIR=$PWD/$d python - <<'PY'
import numpy as np, time
import diffsol_amd as H
import os
CASES = {"120": (("robertson_ode", 40, 120, dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6] * 40)), ("gaussian_decay", 120, 120, dict(rtol=1e-6, atol=[1e-6]))), "300": (("robertson_ode", 100, 300, dict(rtol=1e-4, atol=[1e-8, 1e-14, 1e-6] * 100)),)}
for model, size, n, tol in CASES[os.environ.get("TEAM_PROF_N", "120")]:
    rng = np.random.default_rng(12345)
    nb = 256
    if model == "robertson_ode":
        p = np.stack([0.04 * 2 ** rng.uniform(-1, 1, nb), 1e4 * 2 ** rng.uniform(-1, 1, nb), 3e7 * 2 ** rng.uniform(-1, 1, nb)], axis=1); te = [0.4, 4.0, 40.0, 400.0, 4e3, 4e4, 4e5]
    else:
        p = rng.uniform(0.5, 2.0, (nb, n)); te = [0.5, 1.0, 2.0]
    s = H.Solver(model, p, nbatch=nb, model_size=size, **tol)
    for rep in range(2):
        t0 = time.perf_counter(); y, tot = s.solve_dense_adaptive(te, group=1); dt = time.perf_counter() - t0
    print(model, "n", n, "256 members (one per CU):", round(dt * 1e3, 2), "ms;", {k: tot[k] // nb for k in ("number_of_steps", "number_of_nonlinear_solver_iterations", "number_of_linear_solver_setups")}, "per member", flush=True)
PY
fi
