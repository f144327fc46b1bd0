#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on BASELINE.json's config, on N MI355X of one node.

Workload (config.workload): configs[1] — a 100 000-member Robertson (n=3, fp64) parameter sweep per GPU, integrated with BDF from t=0 to
4e5 by the lock-step ensemble solver (host-side step control, fused HIP kernels), interpolated output at 7 decades, trajectories gathered
with one RCCL all-gather when N>1 (weak scaling: every rank owns its own 100 000 members; no collective inside the integration).
One bench "step" = one whole ensemble solve (fresh `.bdf()` state -> t_final) with the parameters already resident in HBM.

Prints ONE JSON line: metric = accepted ODE steps/s summed over the ensemble (newton_solves_per_sec alongside), plus
  roofline     — the dominant kernel (fused Newton iteration): algorithmic bytes per launch / mean launch duration measured live with
                 HIP events on the solver's stream during the timed region; HBM traffic from the committed rocprofv3 PMC summary if present
  cpu_baseline — the CPU oracle (restatement of the reference's algorithm, one independent IVP per solve like diffsol's CPU path) timed
                 on this box's host cores over a bounded sample of the same sweep (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NB_PER_GPU = 100_000
T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]  # 0.4 ... 4e5
RTOL, ATOL = 1e-4, [1e-8, 1e-14, 1e-6]
# Algorithmic bytes of one fused Newton launch per system, n=3, np=3: reads 8n^2+4n (LU+piv) + 8(4n+np) (y, psi-y0, y_predict, y_old, p) = 204 B,
# writes 8n per iteration it performs; NIT=1 gives SURVEY §8(d)'s 228 B "fused Newton iteration".
NEWTON_READ_BYTES, NEWTON_WRITE_BYTES_PER_ITER = 204, 24
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec


def robertson_params(nb, seed=12345):
    """k1~logU[0.02,0.08], k2~logU[0.5e4,2e4], k3~logU[1.5e7,6e7], numpy default_rng(12345)."""
    rng = np.random.default_rng(seed)
    return np.stack([np.exp(rng.uniform(np.log(0.02), np.log(0.08), nb)), np.exp(rng.uniform(np.log(0.5e4), np.log(2e4), nb)),
                     np.exp(rng.uniform(np.log(1.5e7), np.log(6e7), nb))], axis=1)


def cpu_baseline(params, sample):
    from oracle import oracle as O
    O.build()
    cores = os.cpu_count() or 1
    p = params[:sample]
    r = O.solve_ensemble_independent(O.MODEL_ROBERTSON_ODE, p, model_size=1, rtol=RTOL, atol=ATOL, t_final=T_EVAL[-1], nthreads=cores, want_y=False)
    return {
        "value": r["steps"] / r["seconds"], "unit": "ODE steps/s", "cores": cores, "kind": "port",
        "newton_solves_per_sec": r["newton_iterations"] / r["seconds"], "seconds": r["seconds"],
        "sample": f"first {sample} members of the same Robertson sweep, one independent BDF solve per member to t={T_EVAL[-1]:g} "
                  f"(oracle = C++ restatement of diffsol Bdf+NalgebraLU), {cores} std::threads, static partition",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--block", type=int, default=256, help="threads per workgroup of the one-lane-per-system kernels")
    ap.add_argument("--nb", type=int, default=NB_PER_GPU)
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the Newton kernel with HIP events (measures their overhead)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adaptive", action="store_true", help="skip the extra passes through the device-resident kernel")
    ap.add_argument("--cpu-sample", type=int, default=100_000)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, (world, args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)  # nccl == RCCL on ROCm

    import diffsol_amd
    from diffsol_amd.dist import gather_batch_axis, shard_bounds

    nb = args.nb
    n_total = nb * world
    params = robertson_params(n_total)
    lo, hi = shard_bounds(n_total, rank, world)
    solver = diffsol_amd.Solver("robertson_ode", params[lo:hi], nbatch=hi - lo, model_size=1, rtol=RTOL, atol=ATOL, device=local_rank,
                                block_threads=args.block)
    assert solver.fused, "fused HIP kernels not active"
    out = torch.empty((len(T_EVAL), solver.n, hi - lo), dtype=torch.float64, device=f"cuda:{local_rank}")

    def one_step():
        solver.reset()
        solver.solve_dense(T_EVAL, want_host=False, dev_ptr=out.data_ptr())
        st = solver.stats()
        y = gather_batch_axis(out, n_total, rank, world) if world > 1 else out
        return st, y

    for _ in range(args.warmup):
        one_step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    steps = newton = setups = 0
    for _ in range(args.steps):
        st, y = one_step()
        steps += st["number_of_steps"]
        newton += st["number_of_nonlinear_solver_iterations"]
        setups += st["number_of_linear_solver_setups"]
    barrier()
    elapsed = time.perf_counter() - t0

    # Roofline pass: the same K solves again with every launch of the dominant kernel bracketed by HIP events on the solver's stream.
    # Bracketed launches are synchronous (the events must complete), so this pass is NOT the one `value` is taken from.
    launches, kernel_ms, events_elapsed, bracket_ms, clock_ms = 0, 0.0, 0.0, 0.0, 0.0
    if not args.no_kernel_events:
        solver.set_kernel_timing(True)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        barrier()
        events_elapsed = time.perf_counter() - t1
        launches, kernel_ms = solver.kernel_timing()
        bracket_ms, clock_ms = solver.kernel_timing_overhead_ms()
        solver.set_kernel_timing(False)

    # Extra passes (not `value`): the same K solves through the device-resident kernel — one launch per ensemble solve, no host
    # in the loop — in both control granularities: every member its own step-size/order history (diffsol's CPU semantics for a sweep), and
    # wavefront-sized lock-step groups (the reference's batched semantics with nbatch = 64 per group).
    def device_resident_pass(group):
        solver.solve_dense_adaptive(T_EVAL, want_host=False, dev_ptr=out.data_ptr(), group=group)
        barrier()
        t2 = time.perf_counter()
        a_steps = a_newton = 0
        for _ in range(args.steps):
            _, tot = solver.solve_dense_adaptive(T_EVAL, want_host=False, dev_ptr=out.data_ptr(), group=group)
            if world > 1:
                gather_batch_axis(out, n_total, rank, world)
            a_steps += tot["number_of_steps"]
            a_newton += tot["number_of_nonlinear_solver_iterations"]
        barrier()
        a_elapsed = time.perf_counter() - t2
        a = torch.tensor([a_elapsed, a_steps, a_newton, tot["failed_members"]], dtype=torch.float64, device=f"cuda:{local_rank}")
        if world > 1:
            amax = a[:1].clone()
            dist.all_reduce(amax, op=dist.ReduceOp.MAX)
            dist.all_reduce(a, op=dist.ReduceOp.SUM)
            a[0] = amax[0]
        a_el, a_st, a_nw, a_failed = (float(v) for v in a.tolist())
        return {"ms_per_step": 1e3 * a_el / args.steps, "ode_steps_per_sec": a_st / a_el, "newton_solves_per_sec": a_nw / a_el,
                "mean_steps_per_member": a_st / args.steps / n_total, "failed_members": int(a_failed)}

    adaptive = None
    if not args.no_adaptive:
        adaptive = {"note": "dsh_bdf_solve_adaptive: one kernel launch per ensemble solve, solver state in registers/LDS, no host round trips; same job as "
                            "`value` (same members, tolerances, save points); pow() = include/diffsol_detpow.h, i.e. results bit-identical to the CPU oracle",
                    "per_member": device_resident_pass(1), "wavefront_lockstep_64": device_resident_pass(64)}

    # whole-job aggregates: max time over ranks, units summed over ranks
    agg = torch.tensor([elapsed, (hi - lo) * steps, (hi - lo) * newton, (hi - lo) * setups], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        agg[0] = tmax[0]
    elapsed_max, member_steps, member_newton, member_setups = (float(v) for v in agg.tolist())

    finite = bool(torch.isfinite(y).all().item())
    mass_err = float((y.sum(dim=1) - 1.0).abs().max().item())

    if rank == 0:
        rec = {
            "metric": "ODE steps/sec (and Newton solves/sec) per ensemble",
            "value": member_steps / elapsed_max,
            "unit": "accepted ODE steps/s summed over ensemble members",
            "newton_solves_per_sec": member_newton / elapsed_max,
            "lu_refactors_per_sec": member_setups / elapsed_max,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1]: Robertson stiff ODE (n=3, fp64) ensemble, 100k parameter-sweep members per GPU, BDF, "
                            "batched dense LU, t in [0, 4e5], rtol 1e-4, atol (1e-8,1e-14,1e-6), output at 7 decades",
                "members_per_gpu": nb, "members_total": n_total, "t_final": T_EVAL[-1], "method": "bdf", "lockstep_steps_per_solve": steps / args.steps,
                "newton_iterations_per_solve": newton / args.steps, "block_threads": args.block, "parallelism": f"ensemble-shard x{world}",
                "newton_iterations_per_launch": int(os.environ.get("DSH_NEWTON_NIT", "3")),
            },
            "checks": {"finite": finite, "max_mass_conservation_error": mass_err},
        }
        if launches > 0:
            nit = int(os.environ.get("DSH_NEWTON_NIT", "3"))
            bytes_per_launch = (NEWTON_READ_BYTES + NEWTON_WRITE_BYTES_PER_ITER * nit) * (hi - lo)
            # HIP-event bracket = kernel + part of the marker-packet processing (an EMPTY bracket measures `empty_bracket_us`); the in-kernel
            # device clock (max workgroup end - min workgroup start) is the quantity rocprofv3's kernel trace reports.  `achieved` uses the
            # raw HIP-event figure (conservative); both other figures are reported next to it.
            avg_s = kernel_ms * 1e-3 / launches
            clock_avg_s = clock_ms * 1e-3 / launches
            achieved = bytes_per_launch / avg_s / 1e9
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_newton_iter.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            rec["roofline"] = {"bound": "hbm", "kernel": f"k_newton_iter<RobertsonOde1,...,NIT={nit}> (fused BDF Newton launch, {nit} iterations)", "achieved": achieved,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_s * 1e6, "empty_bracket_us": bracket_ms * 1e3,
                               "avg_launch_us_device_clock": clock_avg_s * 1e6, "achieved_device_clock": bytes_per_launch / max(clock_avg_s, 1e-12) / 1e9,
                               "frac_device_clock": bytes_per_launch / max(clock_avg_s, 1e-12) / 1e9 / HBM_PEAK_GBS, "launches_timed": launches,
                               "measured": "HIP events on the solver stream, second pass over the same K solves (bracketed launches are synchronous)",
                               "events_pass_ms_per_step": 1e3 * events_elapsed / args.steps}
        else:
            rec["roofline"] = None
        rec["device_resident"] = adaptive
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(params, min(args.cpu_sample, n_total))
        print(json.dumps(rec))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
