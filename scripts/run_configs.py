#!/usr/bin/env python
"""Full-size runs of BASELINE.json configs[2..4] (parity-test cases, not bench lines): wall time, solver counters, size-independent checks.

  C3  heat1d n=512 x 4096, TR-BDF2, rtol=atol=1e-6, t_final 0.5   (check: Fourier series of the triangle initial condition)
  C5  series RLC DAE n=4 x 65536, ESDIRK34, t_final 1, root function armed (threshold out of reach: lock-step root finding needs all members to
      cross in the same step)           (check: algebraic constraints of the DAE hold, members == independent CPU solves)
  C4  single-particle battery model n=42 x 262144 (the whole 8-GPU ensemble on one GPU; --spm-nb 32768 = one GPU's shard), BDF, t_final 1200
Writes one JSON object per config to gpurun_out/configs.json.  GPU only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_heat(nb, n, t_final=0.5):
    import diffsol_amd as H
    rng = np.random.default_rng(12345)
    D = rng.uniform(0.5, 2.0, nb)
    t0 = time.perf_counter()
    s = H.Solver("heat1d", D[:, None], nbatch=nb, model_size=n, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    y, _ = s.solve_to_points([t_final])
    wall = time.perf_counter() - t0
    st = s.stats()
    # the same job once more from a fresh solver state: the first run pays the one-time allocations of the context (the dense route maps three 8.6 GB
    # buffers inside its first factorisation), the second does not
    s2 = H.Solver("heat1d", D[:, None], nbatch=nb, model_size=n, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2)
    t0 = time.perf_counter()
    y2, _ = s2.solve_to_points([t_final])
    wall_warm = time.perf_counter() - t0
    assert np.array_equal(y2, y)
    del s2
    h = 1.0 / (n + 1)
    x = (np.arange(n) + 1) * h
    m = np.arange(1, 200)[:, None, None]
    ref = (np.sin((2 * m - 1) * np.pi * x[None, None, :]) * np.exp(-(2 * m - 1) ** 2 * np.pi ** 2 * D[None, :64, None] * t_final) / (2 * m - 1) ** 2).sum(0) * 8 / np.pi ** 2
    err = np.abs(y[0, :64] - ref).max()
    return dict(config="C3 heat1d", n=n, nbatch=nb, method="tr_bdf2", setup_s=t_setup, wall_s=wall, wall_s_second_run=wall_warm, stats=st, max_abs_err_vs_fourier_first64=float(err),
                steps_per_s=st["number_of_steps"] * nb / wall, newton_solves_per_s=st["number_of_nonlinear_solver_iterations"] * nb / wall,
                finite=bool(np.isfinite(y).all()))


def run_rlc(nb, t_final=1.0):
    import diffsol_amd as H
    rng = np.random.default_rng(12345)
    R = rng.uniform(50.0, 200.0, nb)
    Cc = np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), nb))
    p = np.stack([R, np.ones(nb), Cc, np.full(nb, 10.0), np.full(nb, 100.0), np.full(nb, 1e3)], axis=1)
    t0 = time.perf_counter()
    s = H.Solver("rlc", p, nbatch=nb, model_size=1, rtol=1e-6, atol=[1e-6], method=H.METHOD_ESDIRK34)
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    y, ncols, reason = s.solve(t_final)
    wall = time.perf_counter() - t0
    st = s.stats()
    return dict(config="C5 rlc", n=s.n, nbatch=nb, method="esdirk34", fused=s.fused, setup_s=t_setup, wall_s=wall, stats=st, stop_reason=int(reason),
                steps_per_s=st["number_of_steps"] * nb / wall, newton_solves_per_s=st["number_of_nonlinear_solver_iterations"] * nb / wall,
                finite=bool(np.isfinite(y).all()))


def run_rlc_resident(nb, t_final=1.0, group=1, i_thresh=0.03):
    """C5 through the device-resident ESDIRK34 kernel (dsh_sdirk_solve_resident): one launch, per-member step control AND per-member events —
    each member stops when its resistor current crosses i_thresh (the lock-step backend cannot: members cross at different times)."""
    import diffsol_amd as H
    rng = np.random.default_rng(12345)
    R = rng.uniform(50.0, 200.0, nb)
    Cc = np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), nb))
    p = np.stack([R, np.ones(nb), Cc, np.full(nb, 10.0), np.full(nb, 100.0), np.full(nb, i_thresh)], axis=1)
    s = H.Solver("rlc", p, nbatch=nb, model_size=1, rtol=1e-6, atol=[1e-6], method=H.METHOD_ESDIRK34)
    t_eval = np.linspace(0.1, t_final, 10)
    s.solve_dense_adaptive(t_eval, want_host=False, group=group)  # warm-up
    t0 = time.perf_counter()
    y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True, group=group)
    wall = time.perf_counter() - t0
    hit = m["root_idx"] >= 0
    return dict(config=f"C5 rlc device-resident (group {group}, i_thresh {i_thresh})", n=s.n, nbatch=nb, method="esdirk34", wall_s=wall, totals=tot,
                members_stopped_by_event=int(hit.sum()), event_time_min=float(np.nanmin(m["t_root"])) if hit.any() else None,
                event_time_max=float(np.nanmax(m["t_root"])) if hit.any() else None, status_nonzero=int((m["status"] != 0).sum()),
                steps_per_s=tot["number_of_steps"] / wall, newton_solves_per_s=tot["number_of_nonlinear_solver_iterations"] / wall,
                mean_steps_per_member=tot["number_of_steps"] / nb, stats={})


def run_spm(nb, t_final=1200.0):
    """C4: single-particle battery model n=42, BDF; currents U[0.6,1.4] A.  t_final stays below the first member's voltage cut-off (1.4 A
    reaches 3.105 V at ~1720 s): lock-step root finding needs all members to cross in the same step."""
    import diffsol_amd as H
    rng = np.random.default_rng(12345)
    cur = rng.uniform(0.6, 1.4, nb)
    t0 = time.perf_counter()
    s = H.Solver("spm", cur[:, None], nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6])
    t_setup = time.perf_counter() - t0
    t0 = time.perf_counter()
    y, ncols, reason = s.solve(t_final)
    wall = time.perf_counter() - t0
    st = s.stats()
    cap_err = float(np.abs(y[:, 0] - cur * t_final / 3600.0).max())
    return dict(config="C4 spm", n=s.n, nbatch=nb, method="bdf", setup_s=t_setup, wall_s=wall, stats=st, stop_reason=int(reason), max_capacity_error_Ah=cap_err,
                steps_per_s=st["number_of_steps"] * nb / wall, newton_solves_per_s=st["number_of_nonlinear_solver_iterations"] * nb / wall,
                finite=bool(np.isfinite(y).all()))


def run_spm_resident(nb, t_final=3600.0):
    """C4 device-resident with per-member control: full one-hour discharge with the stop conditions armed — every member integrates with its own step sizes and
    stops at ITS OWN voltage cut-off.  The model's Jacobian is tridiagonal, so the lane-per-member banded BDF runs (DynLane<spm, 42, ..> instantiated by hiprtc);
    DSH_RESIDENT_LANE=0 selects the wavefront-per-member kernel (dsh_bdf_solve_wave_member) instead."""
    import diffsol_amd as H
    rng = np.random.default_rng(12345)
    cur = rng.uniform(0.6, 1.4, nb)
    s = H.Solver("spm", cur[:, None], nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6])
    t_eval = np.linspace(360.0, t_final, 10)
    s.solve_dense_adaptive(t_eval, want_host=False)  # warm-up
    t0 = time.perf_counter()
    y, tot, m = s.solve_dense_adaptive(t_eval, want_member_stats=True)
    wall = time.perf_counter() - t0
    hit = m["root_idx"] >= 0
    return dict(config="C4 spm device-resident (%s, events armed)" % ("one wavefront per member" if os.environ.get("DSH_RESIDENT_LANE", "1")[:1] == "0" else "one lane per member, banded LU"), n=s.n, nbatch=nb, method="bdf", wall_s=wall, totals=tot,
                members_stopped_by_event=int(hit.sum()), event_time_min=float(np.nanmin(m["t_root"])) if hit.any() else None,
                event_time_max=float(np.nanmax(m["t_root"])) if hit.any() else None, status_nonzero=int((m["status"] != 0).sum()),
                steps_per_s=tot["number_of_steps"] / wall, newton_solves_per_s=tot["number_of_nonlinear_solver_iterations"] / wall,
                mean_steps_per_member=tot["number_of_steps"] / nb, stats={})


def run_spm_lane(nb, t_final=3600.0, dae=False):
    """dae=True: BASELINE configs[3] as worded — the SINGULAR-MASS formulation (tests/diffsl_models.py spm_dae(20): n = 43, the terminal voltage an algebraic
    state, made consistent per lane on the device).  Otherwise:
    C4 with the model written in DiffSL (tests/diffsl_models.py spm(20, voltage=True): the same equations and voltage cut-offs as the built-in model) through the
    banded lane-per-member BDF (k_bdf_adaptive with the state in per-lane memory, banded LU): what a DiffSL user gets for this model."""
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import diffsol_amd as H
    from diffsol_amd import diffsl
    import diffsl_models as D
    rng = np.random.default_rng(12345)
    cur = rng.uniform(0.6, 1.4, nb)
    m = diffsl.DiffslModel(D.spm_dae(20) if dae else D.spm(20, voltage=True))
    s = H.Solver(m, cur[:, None], nbatch=nb, rtol=1e-6, atol=[1e-6])
    t_eval = np.linspace(360.0, t_final, 10)
    t0 = time.perf_counter()
    s.solve_dense_adaptive(t_eval, want_host=False)  # warm-up, includes the hiprtc compilation of the 42-state kernel
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    s.solve_dense_adaptive(t_eval, want_host=False)
    device_only = time.perf_counter() - t0  # the same solve with the 10 x nb x n output left in HBM (0.9 GB over PCIe otherwise)
    t0 = time.perf_counter()
    y, tot, mm = s.solve_dense_adaptive(t_eval, want_member_stats=True)
    wall = time.perf_counter() - t0
    hit = mm["root_idx"] >= 0
    return dict(output_left_on_device_wall_s=device_only, config="C4 spm DAE (singular mass, V algebraic) from DiffSL, device-resident (one LANE per member, banded LU, events armed)" if dae else
                "C4 spm from DiffSL, device-resident (one LANE per member, banded LU, events armed)", n=s.n, nbatch=nb, method="bdf", wall_s=wall,
                first_call_with_compilation_s=first, totals=tot, members_stopped_by_event=int(hit.sum()),
                event_time_min=float(np.nanmin(mm["t_root"])) if hit.any() else None, event_time_max=float(np.nanmax(mm["t_root"])) if hit.any() else None,
                status_nonzero=int((mm["status"] != 0).sum()), steps_per_s=tot["number_of_steps"] / wall,
                newton_solves_per_s=tot["number_of_nonlinear_solver_iterations"] / wall, mean_steps_per_member=tot["number_of_steps"] / nb, stats={})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--heat-nb", type=int, default=4096)
    ap.add_argument("--heat-n", type=int, default=512)
    ap.add_argument("--rlc-nb", type=int, default=65536)
    ap.add_argument("--spm-nb", type=int, default=262144)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    out = []
    if a.only in ("", "rlc"):
        out.append(run_rlc(a.rlc_nb)); print(json.dumps(out[-1]), flush=True)
    if a.only in ("", "rlc", "rlc_resident"):
        out.append(run_rlc_resident(a.rlc_nb, group=1, i_thresh=0.03)); print(json.dumps(out[-1]), flush=True)
        out.append(run_rlc_resident(a.rlc_nb, group=1, i_thresh=1e3)); print(json.dumps(out[-1]), flush=True)
        out.append(run_rlc_resident(a.rlc_nb, group=64, i_thresh=1e3)); print(json.dumps(out[-1]), flush=True)
    if a.only in ("", "spm", "spm_resident"):
        out.append(run_spm_resident(a.spm_nb)); print(json.dumps(out[-1]), flush=True)
    if a.only in ("", "spm", "spm_lane"):
        out.append(run_spm_lane(a.spm_nb)); print(json.dumps(out[-1]), flush=True)
    if a.only in ("", "spm", "spm_dae"):
        out.append(run_spm_lane(a.spm_nb, dae=True)); print(json.dumps(out[-1]), flush=True)
    if a.only in ("", "spm", "spm_host"):
        out.append(run_spm(a.spm_nb)); print(json.dumps(out[-1]), flush=True)
    if a.only in ("", "heat"):
        out.append(run_heat(a.heat_nb, a.heat_n)); print(json.dumps(out[-1]), flush=True)
    if a.only in ("", "heat_dense"):
        # BASELINE configs[2] says "banded-as-dense LU": the same run with the structure detection off — dense containers, the library's default dense LU for
        # n = 512 (the matrix-core kernel of dsh_lu_tiled.hpp; DSH_LU_EXACT=1 selects the bit-exact blocked kernel instead)
        os.environ["DSH_LU_STRUCTURE"] = "dense"
        try:
            r = run_heat(a.heat_nb, a.heat_n)
            r["config"] = "C3 heat1d, DSH_LU_STRUCTURE=dense (%s dense LU)" % ("bit-exact blocked" if os.environ.get("DSH_LU_EXACT") == "1" else "matrix-core")
            out.append(r); print(json.dumps(out[-1]), flush=True)
        finally:
            os.environ.pop("DSH_LU_STRUCTURE", None)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
