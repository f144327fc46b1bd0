#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on BASELINE.json's config, on N MI355X of one node.

Workload (config.workload): configs[1] — a 100 000-member Robertson (n=3, fp64) parameter sweep per GPU, integrated with BDF from t=0 to
4e5, interpolated output at 7 decades, trajectories gathered with one RCCL all-gather when N>1 (weak scaling: every rank owns its own
100 000 members; no collective inside the integration).  One bench "step" = one whole ensemble solve — `OdeSolverMethod::solve_dense`
(method.rs:467-520) through the drop-in boundary `dshs_solve_dense` in its default ensemble mode, i.e. the device-resident integrator
(one launch: state initialisation, initial step size, first Jacobian, every BDF step, interpolation) with the parameters resident in HBM.

`python bench.py --gpus N` launches its own N ranks (torch.distributed.run, RCCL) when it is not already running under a launcher.

Prints ONE JSON line: metric = accepted ODE steps/s summed over the ensemble (newton_solves_per_sec alongside), plus
  roofline     — the dominant (only) kernel of the timed region, k_bdf_adaptive: bound by VALU issue (state in registers, no HBM traffic besides
                 parameters in / save points out).  achieved = VALU lane-operations per launch (SQ_INSTS_VALU x 64, counters in profiles/) /
                 launch duration measured live with HIP events on the solver's stream; peak = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz.
  cpu_baseline — the CPU oracle (restatement of the reference's algorithm, one independent IVP per solve like diffsol's CPU path) timed on
                 this box's host cores over a bounded sample of the same sweep (rank 0, N=1 only): all cores, one core, and the
                 reference's published single-solve time beside them.
Extra keys (not `value`): the same job host-driven over the trait operations (lock-step over the whole ensemble), per-member control,
and a 1.6M-member ensemble that fills the chip.
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NB_PER_GPU = 100_000
T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]  # 0.4 ... 4e5
RTOL, ATOL = 1e-4, [1e-8, 1e-14, 1e-6]
N_STATES, N_PARAMS = 3, 3
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 at 2.4 GHz — a wavefront's VALU instruction issues over 2 cycles (32 lanes/clk), an FP64 one over 4
# (78.6 TFLOP/s FP64 vector = 16 lanes/clk x 2 flop); HBM3E 8.0 TB/s spec.
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9
HBM_PEAK_GBS = 8000.0
PUBLISHED_SINGLE_SOLVE_S = 3.115e-05  # /root/reference book/src/benchmarks/python_results.csv:2 (robertson_ode n=3, BDF, rtol=atol=1e-4, EPYC 7343)
# Host-driven fused Newton launch (extra key `host_lockstep`): reads 8n^2+4n (LU+piv) + 8(4n+np) = 204 B per member, writes 8n per iteration.
NEWTON_READ_BYTES, NEWTON_WRITE_BYTES_PER_ITER = 204, 24


KERNEL_SOURCES = ["diffsol_amd/csrc/dsh_adaptive_kernel.hpp", "diffsol_amd/csrc/dsh_resident.hpp", "diffsol_amd/csrc/dsh_device.hpp", "diffsol_amd/csrc/dsh_lu_dev.hpp",
                  "diffsol_amd/csrc/dsh_models.hpp", "include/diffsol_detpow.h"]


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the sources that define the headline kernel: the committed instruction counters (profiles/*_pmc_resident.json) carry
    the hash they were measured with, and the roofline refuses counters of another kernel (a stale fraction after a kernel edit)."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def robertson_params(nb, seed=12345):
    """k1~logU[0.02,0.08], k2~logU[0.5e4,2e4], k3~logU[1.5e7,6e7], numpy default_rng(12345)."""
    rng = np.random.default_rng(seed)
    return np.stack([np.exp(rng.uniform(np.log(0.02), np.log(0.08), nb)), np.exp(rng.uniform(np.log(0.5e4), np.log(2e4), nb)),
                     np.exp(rng.uniform(np.log(1.5e7), np.log(6e7), nb))], axis=1)


def cpu_baseline(params, sample):
    """Reference algorithm on the host cores, one independent IVP per member (how diffsol's CPU path runs a sweep).  The timed code is
    oracle/oracle_fast.hpp — the oracle's BDF with fixed-size stack arrays (no per-operation allocation, -O3), verified bit for bit against
    the line-by-line restatement oracle_ode.hpp by tests/test_oracle_golden.py — on a bounded sample of the same sweep."""
    from oracle import oracle as O
    O.build()
    try:
        cores = len(os.sched_getaffinity(0))  # the CPUs this process may run on (a container is often confined to fewer than os.cpu_count())
    except AttributeError:
        cores = os.cpu_count() or 1
    p = params[:sample]
    kw = dict(model_size=1, rtol=RTOL, atol=ATOL, t_final=T_EVAL[-1], want_y=False)
    fast = hasattr(O, "solve_ensemble_independent_fast")
    run = O.solve_ensemble_independent_fast if fast else O.solve_ensemble_independent
    r = run(O.MODEL_ROBERTSON_ODE, p, nthreads=cores, **kw)
    n1 = max(1, min(sample, 2000 if fast else 200))
    r1 = run(O.MODEL_ROBERTSON_ODE, p[:n1], nthreads=1, **kw)
    rec = {
        "value": r["steps"] / r["seconds"], "unit": "ODE steps/s", "cores": cores, "os_cpu_count": os.cpu_count(), "kind": "port",
        "newton_solves_per_sec": r["newton_iterations"] / r["seconds"], "seconds": r["seconds"],
        "sample": f"first {sample} members of the same Robertson sweep, one independent BDF solve per member to t={T_EVAL[-1]:g} "
                  f"({'oracle_fast.hpp: stack-array build of the' if fast else ''} C++ restatement of diffsol Bdf+NalgebraLU), {cores} std::threads, static partition",
        "single_core": {"value": r1["steps"] / r1["seconds"], "unit": "ODE steps/s", "cores": 1, "seconds_per_solve": r1["seconds"] / n1,
                        "newton_solves_per_sec": r1["newton_iterations"] / r1["seconds"], "sample": f"first {n1} members, one thread"},
        "reference_published": {"seconds_per_solve": PUBLISHED_SINGLE_SOLVE_S,
                                "what": "diffsol BDF+nalgebra LU via pydiffsol, robertson_ode n=3, rtol=atol=1e-4, one EPYC 7343 core "
                                        "(book/src/benchmarks/python_results.csv:2); t_final/output grid of that benchmark are defined outside the "
                                        "reference tree, so the point is indicative, not the same job"},
    }
    if fast:
        rs = O.solve_ensemble_independent(O.MODEL_ROBERTSON_ODE, p[:n1], nthreads=1, **kw)
        rec["fidelity_build_single_core_seconds_per_solve"] = rs["seconds"] / n1
    return rec


class _CpuStub:
    """TEST HOOK (--cpu-stub, used by tests/test_bench_cli.py): stands in for the HIP solver on a box without a GPU so that the argument /
    launcher / aggregation path of this file can be exercised under gloo.  Never a measurement: the JSON line says data = "cpu-stub"."""
    n = N_STATES

    def __init__(self, p):
        self.p = p
        self.nbatch = p.shape[0]

    def solve_dense_into(self, out):
        t = np.asarray(T_EVAL)[:, None, None]
        y = np.exp(-self.p.T[None, :, :] * 1e-9 * t)
        out.copy_(__import__("torch").from_numpy(y))
        return {"number_of_steps": 300 * self.nbatch, "number_of_nonlinear_solver_iterations": 700 * self.nbatch,
                "number_of_linear_solver_setups": 70 * self.nbatch, "failed_members": 0}


def _relaunch_under_torchrun(args):
    """`python bench.py --gpus N` outside a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...` (one rank per GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--spinup-seconds", type=float, default=0.75,
                    help="untimed solves before the warm-up steps until the device and the host cores have left their idle clocks (scripts/host_overhead.py: the first "
                         "~0.5 s after idle run 3 %% slower in the kernel and with 0.1 ms more host time per solve); 0 disables")
    ap.add_argument("--nb", type=int, default=NB_PER_GPU, help="ensemble members per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-gather", action="store_true",
                    help="N > 1: wait for every step's trajectory all-gather before the next solve starts (default: the gather of step k runs on RCCL's stream while "
                         "step k + 1 integrates, two output buffers in turn)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra passes (host-driven lock-step, per-member control, 1.6M members)")
    ap.add_argument("--cpu-sample", type=int, default=100_000)
    ap.add_argument("--large-nb", type=int, default=1_600_000)
    ap.add_argument("--cpu-stub", action="store_true", help="TEST HOOK: no GPU, gloo backend, stub solver (exercises launcher + aggregation only)")
    args = ap.parse_args()
    assert args.gpus >= 1 and args.steps >= 1 and args.warmup >= 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _relaunch_under_torchrun(args)  # does not return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
    stub = args.cpu_stub
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    dev = "cpu" if stub else f"cuda:{local_rank}"
    if not stub:
        torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("gloo" if stub else "nccl", rank=rank, world_size=world)  # nccl == RCCL on ROCm
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from diffsol_amd.dist import gather_batch_axis, gather_batch_axis_async, shard_bounds

    nb = args.nb
    n_total = nb * world
    params = robertson_params(n_total)
    lo, hi = shard_bounds(n_total, rank, world)
    out = torch.empty((len(T_EVAL), N_STATES, hi - lo), dtype=torch.float64, device=dev)

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    def allreduce(vals, maxfirst=True):
        """[time, counters...] -> max over ranks of the time, sum over ranks of the counters."""
        a = torch.tensor(vals, dtype=torch.float64, device=dev)
        if world > 1:
            tmax = a[:1].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(a, op=dist.ReduceOp.SUM)
            a[0] = tmax[0]
        return [float(v) for v in a.tolist()]

    if stub:
        solver = _CpuStub(params[lo:hi])
        resolved = 64

        def solve_once(buf=None):
            return solver.solve_dense_into(out if buf is None else buf)
    else:
        import diffsol_amd
        from diffsol_amd.solver import ENSEMBLE_LOCKSTEP, ENSEMBLE_PER_MEMBER, ENSEMBLE_WAVEFRONT, ENSEMBLE_AUTO
        solver = diffsol_amd.Solver("robertson_ode", params[lo:hi], nbatch=hi - lo, model_size=1, rtol=RTOL, atol=ATOL, device=local_rank, block_threads=256)
        assert solver.fused, "fused HIP kernels not active"
        _, resolved = solver.ensemble_mode()
        assert resolved == ENSEMBLE_WAVEFRONT, f"solve_dense did not resolve to the device-resident integrator (mode {resolved})"

        def solve_once(buf=None):
            solver.solve_dense(T_EVAL, want_host=False, dev_ptr=(out if buf is None else buf).data_ptr())
            mode, tot = solver.last_solve_info()
            assert mode == solver.ensemble_mode()[1], (mode, solver.ensemble_mode())
            return tot

    # N > 1: the all-gather of step k's trajectories overlaps the integration of step k + 1 (RCCL runs on its own stream; the solver's launch is on the solver's
    # stream): two output buffers in turn, a buffer is handed to the solver again only after the gather that read it has finished.  The timed region ends when the
    # last gather has finished too (drain).  --no-overlap-gather: gather and wait inside every step.
    overlap = world > 1 and not args.no_overlap_gather
    bufs = [out, torch.empty_like(out)] if overlap else [out]
    pend = [None, None]
    turn = [0]

    def one_step():
        if not overlap:
            tot = solve_once()
            return tot, (gather_batch_axis(out, n_total, rank, world) if world > 1 else out)
        i = turn[0] % 2
        turn[0] += 1
        if pend[i] is not None:
            pend[i].finish()
        tot = solve_once(bufs[i])
        pend[i] = gather_batch_axis_async(bufs[i], n_total, rank, world)
        return tot, None

    def drain():
        """wait for the gathers still in flight; returns the newest gathered trajectory (None if there is none)"""
        newest = pend[(turn[0] - 1) % 2] if turn[0] > 0 else None
        y_last = None
        for q in (pend[turn[0] % 2], newest):  # older first
            if q is not None:
                y_last = q.finish()
        pend[0] = pend[1] = None
        return y_last

    def timed(k, step):
        barrier()
        t0 = time.perf_counter()
        acc = {}
        y = None
        for _ in range(k):
            tot, y = step()
            for key, v in tot.items():
                acc[key] = acc.get(key, 0) + v
        if overlap:
            yd = drain()
            y = yd if yd is not None else y
        barrier()
        return time.perf_counter() - t0, acc, y

    spun = 0
    if not stub and args.spinup_seconds > 0:  # not steps of the measurement: the job is timed at its steady clocks, like any long-running ensemble service
        t_end = time.perf_counter() + args.spinup_seconds
        while time.perf_counter() < t_end:
            solve_once()  # the rank's own solves only — no collective: ranks may leave this loop after different numbers of solves
            spun += 1
    for _ in range(args.warmup):
        one_step()
    if overlap:
        drain()
    if not stub:
        solver.set_kernel_timing(True)  # HIP events around the one launch of every solve (the launch is synchronous anyway: the counters come back)
    elapsed, acc, y = timed(args.steps, one_step)
    launches, kernel_ms = (0, 0.0) if stub else solver.kernel_timing()
    bracket_ms = 0.0 if stub else solver.kernel_timing_overhead_ms()[0]
    if not stub:
        solver.set_kernel_timing(False)
    elapsed_max, member_steps, member_newton, member_setups, failed = allreduce(
        [elapsed, acc["number_of_steps"], acc["number_of_nonlinear_solver_iterations"], acc["number_of_linear_solver_setups"], acc["failed_members"]])
    finite = bool(torch.isfinite(y).all().item())
    mass_err = float((y.sum(dim=1) - 1.0).abs().max().item()) if not stub else 0.0

    y_value_pass = y.clone() if not stub else None  # the extras below reuse the output buffer

    # ------------------------------------------------------------------ extra passes (never `value`)
    extras = {}
    if not stub and not args.no_extras:
        k_x = min(args.steps, 5)

        def mode_pass(mode, with_reset):
            solver.set_ensemble_mode(mode)

            def step():
                if with_reset:
                    solver.reset()  # host-driven: a fresh .bdf() state (the device-resident integrators initialise the state inside the launch)
                return one_step()
            step()
            el, a, _ = timed(k_x, step)
            solver.set_ensemble_mode(ENSEMBLE_AUTO)
            el, st, nw, fl = allreduce([el, a["number_of_steps"], a["number_of_nonlinear_solver_iterations"], a["failed_members"]])
            return {"ms_per_step": 1e3 * el / k_x, "ode_steps_per_sec": st / el, "newton_solves_per_sec": nw / el,
                    "mean_steps_per_member": st / k_x / n_total, "failed_members": int(fl)}

        extras["per_member"] = dict(mode_pass(ENSEMBLE_PER_MEMBER, False), note="every member its own step-size/order history (diffsol's CPU semantics for a sweep)")
        try:  # its roofline entry, from the committed counters of the same kernel (MODE=member scripts/profile_r03.sh); refused when the kernel sources changed since
            pm = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_per_member.json"))).get("bench_kernel", {})
            if pm.get("kernel_source_sha16") == kernel_source_hash() and pm.get("members") == nb and world == 1:
                t_s = extras["per_member"]["ms_per_step"] * 1e-3
                extras["per_member"]["roofline"] = {
                    "bound": "valu", "kernel": pm.get("kernel"), "avg_launch_us": t_s * 1e6, "measured": "wall clock of the per-member solves of this pass (one launch each)",
                    "achieved": pm["valu_insts_per_launch"] * 64 / t_s / 1e12, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                    "frac": pm["valu_insts_per_launch"] * 64 / t_s / 1e12 / VALU_PEAK_TLANEOPS, "valu_wave_instructions_per_launch": pm["valu_insts_per_launch"],
                    "fp64_wave_instructions_per_launch": pm.get("f64_insts_per_launch"), "traffic": pm.get("hbm_bytes_per_launch"), "counters_from": "profiles/r03_pmc_per_member.json",
                    "note": "lane-operations ISSUED, most of them masked off: 3.3x the wave-instructions of the lock-step kernel for fewer member-steps is divergence inside "
                            "wavefronts (profiles/r03_per_member.md: 8.3 ms against 3.2 ms for the same ensemble size with no divergence inside any wavefront)"}
            else:
                extras["per_member"]["roofline"] = None
        except Exception:
            extras["per_member"]["roofline"] = None
        # the opt-in fast-arithmetic variant of the same kernel (dsh_adaptive_fast.hip: fused multiply-adds, reciprocal-math division, ocml pow; NOT bit-comparable
        # with the oracle, held to 1e-6 relative by tests/test_gpu_adaptive.py) — an extra key, never `value`
        def fast_step():
            _, tot = solver.solve_dense_adaptive(T_EVAL, want_host=False, dev_ptr=out.data_ptr(), group=64, deterministic_pow=2)
            return tot, (gather_batch_axis(out, n_total, rank, world) if world > 1 else out)
        try:
            y_exact = y_value_pass
            fast_step()
            el, a, yf = timed(k_x, fast_step)
            el, st, nw, fl = allreduce([el, a["number_of_steps"], a["number_of_nonlinear_solver_iterations"], a["failed_members"]])
            big = y_exact.abs() > 1e-9
            extras["fast_variant"] = {"ms_per_step": 1e3 * el / k_x, "ode_steps_per_sec": st / el, "newton_solves_per_sec": nw / el, "failed_members": int(fl),
                                      "max_rel_diff_vs_exact_states": float(((yf - y_exact).abs() / y_exact.abs().clamp_min(1e-300))[big].max().item()),
                                      "note": "opt-in (deterministic_pow = 2): -ffp-contract=fast, reciprocal-math division, ocml pow, reciprocal Newton weights; "
                                              "wavefront lock-step groups like `value`, not bit-comparable with the oracle"}
        except Exception as e:  # noqa: BLE001 — an extra must not take the bench line down
            extras["fast_variant"] = {"error": str(e)[:200]}
        hl = mode_pass(ENSEMBLE_LOCKSTEP, True)
        hl["note"] = ("DSHS_ENSEMBLE_LOCKSTEP: host-driven, one (t, h, order) for all members over the trait-boundary operations (fused Newton / accept kernels); "
                      "round 1's `value` path")
        # its dominant kernel, the fused 3-iteration Newton launch, is HBM-bound: bracketed pass for its roofline (bracketed launches are synchronous)
        solver.set_ensemble_mode(ENSEMBLE_LOCKSTEP)
        solver.set_kernel_timing(True)
        solver.reset(); one_step()
        if overlap:
            drain()
        nl, nms = solver.kernel_timing()
        solver.set_kernel_timing(False)
        solver.set_ensemble_mode(ENSEMBLE_AUTO)
        if nl > 0:
            nit = int(os.environ.get("DSH_NEWTON_NIT", "3"))
            bpl = (NEWTON_READ_BYTES + NEWTON_WRITE_BYTES_PER_ITER * nit) * (hi - lo)
            hl["newton_kernel_roofline"] = {"bound": "hbm", "kernel": f"k_newton_iter<RobertsonOde1,...,NIT={nit}>", "algorithmic_bytes_per_launch": bpl,
                                            "avg_launch_us": 1e3 * nms / nl, "achieved": bpl / (nms * 1e-3 / nl) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": bpl / (nms * 1e-3 / nl) / 1e9 / HBM_PEAK_GBS, "launches_timed": nl}
        extras["host_lockstep"] = hl
        # chip-filling ensemble: 25 000 wavefronts (2 per SIMD is the kernel's occupancy), same job per member
        nbl = args.large_nb
        pl = robertson_params(nbl * world, seed=54321)[rank * nbl:(rank + 1) * nbl]
        big = diffsol_amd.Solver("robertson_ode", pl, nbatch=nbl, model_size=1, rtol=RTOL, atol=ATOL, device=local_rank, block_threads=256)
        outl = torch.empty((len(T_EVAL), N_STATES, nbl), dtype=torch.float64, device=dev)

        def big_step():
            big.solve_dense(T_EVAL, want_host=False, dev_ptr=outl.data_ptr())
            return big.last_solve_info()[1], outl
        big_step()
        big.set_kernel_timing(True)
        el, a, yl = timed(2, big_step)
        bl, bms = big.kernel_timing()
        big.set_kernel_timing(False)
        el, st, nw, fl = allreduce([el, a["number_of_steps"], a["number_of_nonlinear_solver_iterations"], a["failed_members"]])
        extras["large_ensemble"] = {"members_per_gpu": nbl, "ms_per_step": 1e3 * el / 2, "ode_steps_per_sec": st / el, "newton_solves_per_sec": nw / el,
                                    "kernel_ms": bms / max(bl, 1), "failed_members": int(fl), "finite": bool(torch.isfinite(yl).all().item()),
                                    "hbm_algorithmic_gbs": 8 * (N_PARAMS + N_STATES * len(T_EVAL)) * nbl / (bms * 1e-3 / max(bl, 1)) / 1e9,
                                    "note": "same job per member on a chip-filling ensemble (wavefront lock-step groups of 64); the kernel stays VALU-issue bound — "
                                            "its HBM traffic is parameters in + save points out"}
        del big, outl

    if rank == 0:
        rec = {
            "metric": "ODE steps/sec (and Newton solves/sec) per ensemble",
            "value": member_steps / elapsed_max,
            "unit": "accepted ODE steps/s summed over ensemble members",
            "newton_solves_per_sec": member_newton / elapsed_max,
            "lu_refactors_per_sec": member_setups / elapsed_max,
            "n_gpus": world, "ranks": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "cpu-stub (test hook: launcher/aggregation path only, not a measurement)" if stub else "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1]: Robertson stiff ODE (n=3, fp64) ensemble, 100k parameter-sweep members per GPU, BDF, "
                            "batched dense LU, t in [0, 4e5], rtol 1e-4, atol (1e-8,1e-14,1e-6), output at 7 decades",
                "members_per_gpu": nb, "members_total": n_total, "t_final": T_EVAL[-1], "method": "bdf",
                "path": "dshs_solve_dense, default ensemble mode -> device-resident BDF (dsh_bdf_solve_adaptive), wavefront lock-step groups of 64 members "
                        "(the reference's batched semantics with nbatch = 64 per group)",
                "ensemble_mode": resolved, "untimed_spinup_solves": spun, "mean_steps_per_member": member_steps / args.steps / n_total,
                "mean_newton_iterations_per_member": member_newton / args.steps / n_total, "parallelism": f"ensemble-shard x{world}",
                "backend": "gloo" if stub else ("nccl" if world > 1 else "none"), "gather": ("overlapped with the next solve" if overlap else ("per step" if world > 1 else "none")),
            },
            "checks": {"finite": finite, "max_mass_conservation_error": mass_err, "failed_members": int(failed)},
        }
        if launches > 0:
            avg_s = kernel_ms * 1e-3 / launches
            pmc = {}
            stale = None
            path = os.path.join(ROOT, "profiles", "r03_pmc_resident.json")
            if os.path.exists(path):
                try:
                    pmc = json.load(open(path)).get("bench_kernel", {}) or {}
                except Exception:
                    pmc = {}
                if pmc:
                    pmc["file"] = "profiles/r03_pmc_resident.json"
                    if pmc.get("kernel_source_sha16") != kernel_source_hash():  # counters of another kernel: no fraction rather than a stale one
                        stale = f"profiles/r03_pmc_resident.json was measured on kernel sources {pmc.get('kernel_source_sha16')}, this tree has {kernel_source_hash()}: re-run scripts/profile_r03.sh"
                        pmc = {}
            algo_hbm = 8 * (N_PARAMS + N_STATES * len(T_EVAL)) * (hi - lo)
            roof = {"bound": "valu", "kernel": "dsh::k_bdf_adaptive<RobertsonOde1, BA=true, WAVE=true> (the whole ensemble solve, one launch)",
                    "avg_launch_us": avg_s * 1e6, "launches_timed": launches, "empty_bracket_us": bracket_ms * 1e3,
                    "measured": "HIP events on the solver stream around the launch of every timed solve (same pass as `value`)",
                    "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                    "hbm": {"algorithmic_bytes_per_launch": algo_hbm, "achieved_gbs": algo_hbm / avg_s / 1e9, "frac_of_8TBs": algo_hbm / avg_s / 1e9 / HBM_PEAK_GBS,
                            "note": "parameters in + save points out; the solver state never leaves registers/LDS, so HBM does not bound this kernel"}}
            valu = pmc.get("valu_insts_per_launch") if nb == pmc.get("members", NB_PER_GPU) else None
            if valu:
                roof["achieved"] = valu * 64 / avg_s / 1e12
                roof["frac"] = roof["achieved"] / VALU_PEAK_TLANEOPS
                roof["valu_wave_instructions_per_launch"] = valu
                f64 = pmc.get("f64_insts_per_launch")
                if f64:  # issue-slot utilisation: an FP64 VALU instruction occupies the SIMD for 4 cycles, any other for 2
                    roof["fp64_wave_instructions_per_launch"] = f64
                    roof["issue_slot_frac"] = (4 * f64 + 2 * (valu - f64)) / (SIMD_CYCLES_PER_S * avg_s)
                    roof["fp64_flop_per_launch"] = pmc.get("f64_flop_per_launch")
                    if pmc.get("f64_flop_per_launch"):
                        roof["fp64_tflops"] = pmc["f64_flop_per_launch"] / avg_s / 1e12
                roof["traffic"] = pmc.get("hbm_bytes_per_launch")
                roof["counters_from"] = pmc.get("file")
            else:
                roof.update({"achieved": None, "frac": None, "traffic": None, "note": stale or "no PMC summary for this ensemble size under profiles/"})
            rec["roofline"] = roof
        else:
            rec["roofline"] = None
        rec.update(extras)
        if world == 1 and not args.no_cpu_baseline and not stub:
            rec["cpu_baseline"] = cpu_baseline(params, min(args.cpu_sample, n_total))
        print(json.dumps(rec))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
