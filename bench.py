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
T_START = time.perf_counter()
IMPORT_SECONDS = [0.0]  # time spent in `import torch`: 1 - 2 minutes on a box that has never paged the image in, ~1.5 s afterwards — not the bench's work, not charged to --time-budget
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NB_PER_GPU = 100_000
T_EVAL = [0.4 * 10 ** k for k in range(0, 7)]  # 0.4 ... 4e5
RTOL, ATOL = 1e-4, [1e-8, 1e-14, 1e-6]
N_STATES, N_PARAMS = 3, 3
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 at 2.4 GHz — a wavefront's VALU instruction issues over 2 cycles (32 lanes/clk), an FP64 one over 4
# (FP64 vector peak derived from that, not printed in the guide: 16 lanes/clk x 2 flop x 1024 SIMDs x 2.4 GHz = 78.6 TFLOP/s); HBM3E 8.0 TB/s spec.
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9
HBM_PEAK_GBS = 8000.0
PUBLISHED_SINGLE_SOLVE_S = 3.115e-05  # /root/reference book/src/benchmarks/python_results.csv:2 (robertson_ode n=3, BDF, rtol=atol=1e-4, EPYC 7343)
# Host-driven fused Newton launch (extra key `host_lockstep`): reads 8n^2+4n (LU+piv) + 8(4n+np) = 204 B per member, writes 8n per iteration.
NEWTON_READ_BYTES, NEWTON_WRITE_BYTES_PER_ITER = 204, 24
# SURVEY 8(d) algorithmic bytes per unit for n = 3, np = 3, q = 3: full fused Newton iteration 8n^2+4n+8(5n+np) = 228; accepted step 2*8*n*(q+3)+16n+8n = 360;
# LU refactorisation 8n^2 read + 8n^2 + 4n written = 156
MODEL_8D_NEWTON_BYTES, MODEL_8D_STEP_BYTES, MODEL_8D_REFACTOR_BYTES = 228, 360, 156


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


DETAIL_FILE = "bench_detail.json"
LINE_LIMIT = 4096  # the driver keeps an 8 KB tail of stdout: the LAST line must fit with room to spare (a 23.7 KB line did not parse)


def _sig(x, nd=5):
    """floats to nd significant digits (the compact line only; the detail file keeps full precision)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{nd}g}") if np.isfinite(x) else None
    return x


def _pick(d, keys):
    return {k: _sig(d.get(k)) for k in keys if isinstance(d, dict) and k in d}


def compact_line(rec):
    """The contract line: every key the driver / judge reads (metric, value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype,
    data, config, roofline, cpu_baseline) with short strings, the other BASELINE configs as `configs: {name: [ms_per_solve, roofline_frac, bound, cpu_steps_per_s]}`,
    the extra passes as one number each.  Everything else lives in bench_detail.json (also printed on an earlier stdout line)."""
    out = {k: _sig(rec.get(k)) for k in ("metric", "value", "unit", "newton_solves_per_sec", "lu_refactors_per_sec", "n_gpus", "ranks", "steps", "warmup", "ms_per_step",
                                         "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in rec}
    cfg = rec.get("config") or {}
    out["config"] = _pick(cfg, ("members_per_gpu", "members_total", "t_final", "method", "ensemble_mode", "mean_steps_per_member", "parallelism", "backend", "gather",
                                "gathered_bytes_per_solve"))
    if cfg.get("arithmetic"):
        out["config"]["arithmetic"] = str(cfg["arithmetic"])[:150]
    out["config"]["workload"] = str(cfg.get("workload_short") or cfg.get("workload", ""))[:200]
    if "checks" in rec:
        out["checks"] = {k: _sig(v) for k, v in rec["checks"].items()}
    roof = rec.get("roofline")
    if isinstance(roof, dict):
        r = _pick(roof, ("bound", "avg_launch_us", "launches_timed", "achieved", "peak", "unit", "frac", "fp64_frac", "hbm_frac_actual", "hbm_model_8d_frac", "traffic", "fp64_tflops",
                         "counters_from"))
        if roof.get("hbm_model_8d_frac") is not None:
            r["hbm_model_8d_note"] = "8(d) model assumes LU/state in HBM; they are in registers, so >1 is expected"
        r["kernel"] = str(roof.get("kernel", ""))[:80]
        if isinstance(roof.get("lane_ops"), dict):
            r["lane_ops_frac"] = _sig(roof["lane_ops"].get("frac"))
        if isinstance(roof.get("hbm"), dict):
            r["algorithmic_hbm_bytes"] = roof["hbm"].get("algorithmic_bytes_per_launch")
        if roof.get("achieved") is None and roof.get("note"):
            r["note"] = str(roof["note"])[:160]
        out["roofline"] = r
    elif "roofline" in rec:
        out["roofline"] = None
    cpu = rec.get("cpu_baseline")
    if isinstance(cpu, dict):
        c = _pick(cpu, ("value", "unit", "cores", "usable_cpus", "kind", "newton_solves_per_sec", "parallel_efficiency", "seconds"))
        c["threads"] = cpu.get("cores")
        c["sample"] = str(cpu.get("sample_short") or cpu.get("sample", ""))[:160]
        if isinstance(cpu.get("single_core"), dict):
            c["single_core_value"] = _sig(cpu["single_core"].get("value"))
            c["single_core_seconds"] = _sig(cpu["single_core"].get("seconds"))
        out["cpu_baseline"] = c
    ex = {}
    for k in ("per_member", "exact_variant", "host_lockstep", "large_ensemble", "trait_path"):
        v = rec.get(k)
        if isinstance(v, dict):
            ex[k + "_ms"] = _sig(v.get("ms_per_step")) if "error" not in v else "error"
    if ex:
        out["extras"] = ex
    cfgs = rec.get("configs")
    if isinstance(cfgs, dict):
        out["configs_columns"] = ["ms_per_solve", "roofline_frac", "bound", "cpu_steps_per_s"]
        cc = {}
        for name, v in cfgs.items():
            if not isinstance(v, dict):
                continue
            if "error" in v or "skipped" in v:
                cc[name] = [None, None, ("error: " + str(v.get("error"))[:60]) if "error" in v else "skipped", None]
                continue
            rf = v.get("roofline") or {}
            cc[name] = [_sig(v.get("ms_per_solve")), _sig(rf.get("frac")), rf.get("bound"), _sig((v.get("cpu_baseline") or {}).get("value"))]
            rff = v.get("roofline_factor")
            if isinstance(rff, dict):
                cc[name + ":factor"] = [_sig(rff.get("avg_launch_us", 0.0) / 1e3), _sig(rff.get("frac")), rff.get("bound"), None]
        out["configs"] = cc
    if "bench_seconds" in rec:
        out["bench_seconds"] = _sig(rec["bench_seconds"])
    if "jit_compiles_this_run" in rec:
        out["jit_compiles_this_run"] = rec["jit_compiles_this_run"]
    out["detail"] = DETAIL_FILE
    line = json.dumps(out, separators=(",", ":"))
    if len(line) > LINE_LIMIT:  # never print a line the driver cannot hold: drop the optional blocks, keep the contract keys
        for k in ("extras", "configs_columns", "checks"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    assert len(line) <= LINE_LIMIT, len(line)
    return line


def emit(rec):
    """full record -> bench_detail.json (+ gpurun_out/ when it exists) and an earlier stdout line; the compact contract line LAST."""
    full = json.dumps(rec)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAIL_FILE), "w") as f:
                    f.write(full + "\n")
            except OSError:
                pass
    print("bench_detail: " + full)
    print(compact_line(rec))
    sys.stdout.flush()


def robertson_params(nb, seed=12345):
    """k1~logU[0.02,0.08], k2~logU[0.5e4,2e4], k3~logU[1.5e7,6e7], numpy default_rng(12345)."""
    rng = np.random.default_rng(seed)
    return np.stack([np.exp(rng.uniform(np.log(0.02), np.log(0.08), nb)), np.exp(rng.uniform(np.log(0.5e4), np.log(2e4), nb)),
                     np.exp(rng.uniform(np.log(1.5e7), np.log(6e7), nb))], axis=1)


def cpu_limits():
    """CPUs this process may use: affinity mask, and the cgroup-v2 quota (cpu.max "quota period": quota/period CPUs) when the container has one."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    raw = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            raw = open(path).read().strip()
        except OSError:
            continue
        try:
            if path.endswith("cpu.max"):
                q, per = raw.split()
                quota = None if q == "max" else float(q) / float(per)
            else:
                q = float(raw)
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                quota = None if q < 0 else q / per
        except (ValueError, OSError):
            quota = None
        break
    return {"affinity_cpus": aff, "os_cpu_count": os.cpu_count(), "cgroup_cpu_max": raw, "cgroup_quota_cpus": quota}


def thread_counts(lim):
    """Thread counts of the sweep: with a cgroup CPU quota q (the GPU boxes of this pool: 16 of 256 CPUs) q, 2q, 4q — more threads than that only time-slice;
    without one 16, 32, 64, ... up to the CPUs of the affinity mask (the sweep VERDICT r3 asked for)."""
    aff, q = lim["affinity_cpus"], lim["cgroup_quota_cpus"]
    if q and q < aff:
        q = max(1, int(round(q)))
        return sorted({min(aff, q), min(aff, 2 * q), min(aff, 4 * q)})
    c, out = 16, []
    while c < aff:
        out.append(c)
        c *= 2
    out.append(aff)
    return out if aff >= 16 else [aff]


def usable_cpus(lim):
    q = lim["cgroup_quota_cpus"]
    return min(lim["affinity_cpus"], q) if q else lim["affinity_cpus"]


def cpu_sweep(run, single, counts, lim):
    """run(nthreads) -> (units, newton, seconds) for the whole sample; single = units/s measured on one thread.  Returns the best thread count; parallel efficiency =
    rate / (min(threads, usable CPUs) x single-thread rate), usable CPUs = min(affinity mask, cgroup quota)."""
    rows = []
    for c in counts:
        u, nw, sec = run(c)
        rows.append({"threads": c, "value": u / sec, "newton_solves_per_sec": nw / sec, "seconds": sec, "parallel_efficiency": (u / sec) / (min(c, usable_cpus(lim)) * single)})
    best = max(rows, key=lambda r: r["value"])
    for r in rows:
        # more than the quota's worth of single-core rates: the cgroup CPU quota is enforced per accounting period, a sub-second sample on more threads than the quota
        # bursts above it (the single-core denominator is a >= 0.5 s sample, best of three).  The baseline is the faster for it, never the slower.
        r["cores_worth_of_single_core_rate"] = r["value"] / single
        if r["parallel_efficiency"] > 1.05:
            r["note"] = "above 1: a sub-second sample on more threads than the cgroup quota bursts above the quota"
    return best, rows


def cpu_baseline(params, sample):
    """Reference algorithm on the host cores, one independent IVP per member (how diffsol's CPU path runs a sweep).  The timed code is
    oracle/oracle_fast.hpp — the oracle's BDF with fixed-size stack arrays (no per-operation allocation, -O3), verified bit for bit against
    the line-by-line restatement oracle_ode.hpp by tests/test_oracle_golden.py — on a bounded sample of the same sweep.  The thread count is SWEPT
    (16 ... all CPUs) and the best is reported with its parallel efficiency against the single-core rate; cgroup CPU quota stated."""
    from oracle import oracle as O
    O.build()
    lim = cpu_limits()
    cores = lim["affinity_cpus"]
    p = params[:sample]
    kw = dict(model_size=1, rtol=RTOL, atol=ATOL, t_final=T_EVAL[-1], want_y=False)
    fast = hasattr(O, "solve_ensemble_independent_fast")
    run = O.solve_ensemble_independent_fast if fast else O.solve_ensemble_independent
    n1 = max(1, min(sample, 2000 if fast else 200))
    run(O.MODEL_ROBERTSON_ODE, p[:n1], nthreads=1, **kw)  # page in, leave the idle clock
    while True:  # single-core denominator from a sample of >= 0.5 s (a cold, tiny sample gave parallel efficiencies above 1)
        r1 = run(O.MODEL_ROBERTSON_ODE, p[:n1], nthreads=1, **kw)
        if r1["seconds"] >= 0.5 or n1 >= sample:
            break
        n1 = min(sample, max(2 * n1, int(n1 * 0.6 / max(r1["seconds"], 1e-3))))
    for _ in range(2):  # the denominator is the BEST of three such samples: the single-core rate must not be the thing that is slow
        r1b = run(O.MODEL_ROBERTSON_ODE, p[:n1], nthreads=1, **kw)
        if r1b["steps"] / r1b["seconds"] > r1["steps"] / r1["seconds"]:
            r1 = r1b
    single = r1["steps"] / r1["seconds"]

    def go(c):
        r = run(O.MODEL_ROBERTSON_ODE, p, nthreads=c, **kw)
        return r["steps"], r["newton_iterations"], r["seconds"]
    go(min(cores, 16))  # page in, spin the cores up
    best, rows = cpu_sweep(go, single, thread_counts(lim), lim)
    rec = {
        "value": best["value"], "unit": "ODE steps/s", "cores": best["threads"], "usable_cpus": usable_cpus(lim), "kind": "port",
        "newton_solves_per_sec": best["newton_solves_per_sec"], "seconds": best["seconds"],
        "parallel_efficiency": best["parallel_efficiency"], "thread_sweep": rows, "cpu_limits": lim,
        "sample": f"first {sample} members of the same Robertson sweep, one independent BDF solve per member to t={T_EVAL[-1]:g} "
                  f"({'oracle_fast.hpp: stack-array build of the' if fast else ''} C++ restatement of diffsol Bdf+NalgebraLU), std::threads with a static partition; "
                  f"thread count swept, best reported",
        "single_core": {"value": single, "unit": "ODE steps/s", "cores": 1, "seconds_per_solve": r1["seconds"] / n1,
                        "newton_solves_per_sec": r1["newton_iterations"] / r1["seconds"], "seconds": r1["seconds"], "sample": f"first {n1} members, one thread"},
        "sample_short": f"{sample} members, same distribution/seed as the GPU's {NB_PER_GPU} (longer draw; rate per member-step compared), independent BDF solves, best thread count",
        "reference_published": {"seconds_per_solve": PUBLISHED_SINGLE_SOLVE_S,
                                "what": "diffsol BDF+nalgebra LU via pydiffsol, robertson_ode n=3, rtol=atol=1e-4, one EPYC 7343 core "
                                        "(book/src/benchmarks/python_results.csv:2); t_final/output grid of that benchmark are defined outside the "
                                        "reference tree, so the point is indicative, not the same job"},
    }
    if fast:
        rs = O.solve_ensemble_independent(O.MODEL_ROBERTSON_ODE, p[:n1], nthreads=1, **kw)
        rec["fidelity_build_single_core_seconds_per_solve"] = rs["seconds"] / n1
    return rec


# ------------------------------------------------------------------ BASELINE.json configs[2..4] (extra keys of the line; `value` stays configs[1])
# Workloads as SURVEY §8(d) defines them (seed 12345): C3 heat1d n = 512 x 4096, D ~ U[0.5, 2], rtol = atol = 1e-6, TR-BDF2, t in [0, 0.5];
# C4 single-particle battery model n = 42 (ODE form) / 43 (singular mass: terminal voltage algebraic), I ~ U[0.6, 1.4] A, BDF, t in [0, 3600 s], voltage
# cut-offs armed, 32 768 (one GPU's shard of the 8-GPU job) and 262 144 members; C5 series RLC DAE n = 4 x 65 536, R ~ U[50, 200], C ~ logU[5e-4, 2e-3],
# ESDIRK34, t in [0, 1], root i_R = i_thresh.
TIMING_RESIDENT, TIMING_LU_SOLVE, TIMING_LU_FACTOR = 0, 1, 2
# FP64 peak, DERIVED (MI355X_MICROARCH.md prints no FP64 row): 256 CUs x 4 SIMDs x 16 FP64 lanes/clk x 2 flop (FMA) x 2.4 GHz = 78.6 TFLOP/s — half the
# guide's 157.3 TFLOP/s FP32 vector figure (SIMD-32), and the rate of v_mfma_f64_16x16x4_f64 as well (one instruction = 2048 flop per 64 cycles and SIMD).
FP64_MATRIX_PEAK_TFLOPS = 256 * 4 * 16 * 2 * 2.4e9 / 1e12
CFG_SOURCES = {
    "c3_banded": ["diffsol_amd/csrc/dsh_lu_band_team.hpp", "diffsol_amd/csrc/dsh_lu_band.hpp"],
    "c3_dense": ["diffsol_amd/csrc/dsh_lu_tiled.hpp", "diffsol_amd/csrc/dsh_lu_coop.hpp"],
    "c4": ["diffsol_amd/csrc/dsh_lane_banded_kernel.hpp", "diffsol_amd/csrc/dsh_lu_band.hpp", "diffsol_amd/csrc/dsh_resident.hpp"],
    "c5": ["diffsol_amd/csrc/dsh_sdirk_kernel.hpp", "diffsol_amd/csrc/dsh_resident.hpp", "diffsol_amd/csrc/dsh_models.hpp", "include/diffsol_detpow.h"],
}


def source_hash(files):
    import hashlib
    h = hashlib.sha256()
    for rel in files:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def latest_profile(suffix):
    """newest committed round of a profile summary: profiles/rNN_<suffix> with the largest NN (None when there is none).  bench.py reads counters only from the newest
    file of a kind and refuses them when the kernel's sources changed since (tests/test_profiles_fresh.py fails the CPU tier in that case: re-run scripts/profile_r06.sh)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)):
        m = re.match(r"r(\d\d)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), os.path.basename(f))
    return best[1] if best else None


def heat_params(nb):
    return np.random.default_rng(12345).uniform(0.5, 2.0, nb)[:, None]


def spm_params(nb):
    return np.random.default_rng(12345).uniform(0.6, 1.4, nb)[:, None]


def rlc_params(nb, i_thresh):
    rng = np.random.default_rng(12345)
    R = rng.uniform(50.0, 200.0, nb)
    Cc = np.exp(rng.uniform(np.log(5e-4), np.log(2e-3), nb))
    return np.stack([R, np.ones(nb), Cc, np.full(nb, 10.0), np.full(nb, 100.0), np.full(nb, i_thresh)], axis=1)


def _counters(name, key, sources):
    """committed PMC summary of a config's kernel (the newest profiles/rNN_pmc_configs.json), refused when the kernel's sources changed since it was taken"""
    for cand in (latest_profile("pmc_configs.json"),):
        if not cand:
            continue
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", cand))).get(key)
        except Exception:
            continue
        if d and d.get("kernel_source_sha16") == source_hash(sources):
            d = dict(d)
            d["file"] = "profiles/" + cand
            return d
    return None


def cfg_cpu(model, p, t_eval, method, what, single_n, **kw):
    """CPU leg of a config: the oracle's solve_dense per member (independent IVPs, the reference's CPU usage) on a bounded sample; thread count swept."""
    from oracle import oracle as O
    lim = cpu_limits()
    cores = lim["affinity_cpus"]

    def run(pp, c):
        t0 = time.perf_counter()
        _, st, failed = O.solve_dense_independent(model, pp, t_eval, method=method, nthreads=c, **kw)
        return int(st[:, 0].sum()), int(st[:, 1].sum()), time.perf_counter() - t0, failed
    run(p[:min(single_n, 8)], 1)  # page in
    while True:  # single-core denominator from a sample of >= 0.5 s
        u1, n1, s1, _ = run(p[:single_n], 1)
        if s1 >= 0.5 or single_n >= p.shape[0]:
            break
        single_n = min(p.shape[0], max(2 * single_n, int(single_n * 0.6 / max(s1, 1e-3))))
    for _ in range(2):  # best of three
        u1b, n1b, s1b, _ = run(p[:single_n], 1)
        if u1b / s1b > u1 / s1:
            u1, n1, s1 = u1b, n1b, s1b
    single = u1 / s1
    counts = [c for c in thread_counts(lim) if c <= p.shape[0]] or [min(cores, p.shape[0])]
    if len(counts) > 3:
        counts = counts[-3:]  # bounded: the three largest thread counts
    best, rows = cpu_sweep(lambda c: run(p, c)[:3], single, counts, lim)
    return {"value": best["value"], "unit": "ODE steps/s", "cores": best["threads"], "usable_cpus": usable_cpus(lim), "kind": "port", "newton_solves_per_sec": best["newton_solves_per_sec"],
            "seconds": best["seconds"], "seconds_per_solve_single_core": s1 / single_n, "single_core_seconds": s1, "parallel_efficiency": best["parallel_efficiency"],
            "thread_sweep": rows, "sample": f"{what}: first {p.shape[0]} members, one independent solve_dense per member (oracle restatement), best of the thread sweep"}


def bench_configs(device, want_cpu, quick=False, budget_s=0.0):
    """BASELINE configs[2..4] on one GPU, each with its own roofline entry (dominant kernel, live launch durations from HIP-event brackets on the solver's stream)
    and CPU leg.  Every entry is guarded: a failing config reports {"error": ...} and the line still prints."""
    import diffsol_amd as H
    ARITH = 2 if H.get_resident_arithmetic() == H.ARITH_FAST else 1  # the library's default arithmetic of the device-resident integrators (fast builds where they exist)
    from diffsol_amd import diffsl
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    out = {}

    def guarded(name, fn):
        t0 = time.perf_counter()
        if budget_s and t0 - T_START - IMPORT_SECONDS[0] > budget_s:  # the contract line must come out within the driver's patience: later configs are reported as skipped, not run
            out[name] = {"skipped": f"time budget: {t0 - T_START - IMPORT_SECONDS[0]:.0f} s since start (without the {IMPORT_SECONDS[0]:.0f} s of `import torch`) > --time-budget {budget_s:g} s"}
            return
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        out[name]["bench_seconds"] = time.perf_counter() - t0

    O = None
    cpu_memo = {}
    if want_cpu:
        from oracle import oracle as O
        O.build()

    # ---------------------------------------------------------------- C3
    def c3(dense):
        nb, n, tf = (512 if quick else 4096), 512, 0.5
        D = heat_params(nb)
        if dense:
            os.environ["DSH_LU_STRUCTURE"] = "dense"
        try:
            s = H.Solver("heat1d", D, nbatch=nb, model_size=n, rtol=1e-6, atol=[1e-6], method=H.METHOD_TR_BDF2, device=device)
            t0 = time.perf_counter(); y, _ = s.solve_to_points([tf]); first = time.perf_counter() - t0
            st = s.stats()
            walls = []
            for _ in range(2 if dense else 3):  # from a fresh .tr_bdf2() state on the same problem (dshs_reset): the context keeps its device blocks
                s.reset()
                t0 = time.perf_counter(); y2, _ = s.solve_to_points([tf]); walls.append(time.perf_counter() - t0)
            wall = min(walls)
            assert np.array_equal(y2, y)
            timed = {}
            for tgt, nm in ((TIMING_LU_SOLVE, "solve"), (TIMING_LU_FACTOR, "factor")):
                s.reset()
                s.set_kernel_timing(True); s.set_kernel_timing_target(tgt)
                s.solve_to_points([tf])
                timed[nm] = s.kernel_timing()
                s.set_kernel_timing(False)
            del s
        finally:
            os.environ.pop("DSH_LU_STRUCTURE", None)
        h = 1.0 / (n + 1)
        x = (np.arange(n) + 1) * h
        m = np.arange(1, 200)[:, None, None]
        ref = (np.sin((2 * m - 1) * np.pi * x[None, None, :]) * np.exp(-(2 * m - 1) ** 2 * np.pi ** 2 * D[None, :64, 0, None] * tf) / (2 * m - 1) ** 2).sum(0) * 8 / np.pi ** 2
        steps, newton = st["number_of_steps"] * nb, st["number_of_nonlinear_solver_iterations"] * nb
        rec = {"workload": f"BASELINE configs[2]: heat1d n={n} x {nb}, TR-BDF2, rtol=atol=1e-6, t in [0, {tf}]; host-driven lock-step over the trait operations, "
                           + ("dense containers and the default dense LU (DSH_LU_STRUCTURE=dense: the literal 'banded-as-dense')" if dense else "band containers + banded LU (bit-identical to the dense route)"),
               "ms_per_solve": 1e3 * wall, "first_solve_ms": 1e3 * first, "ode_steps_per_sec": steps / wall, "newton_solves_per_sec": newton / wall,
               "lockstep_steps": st["number_of_steps"], "newton_iterations": st["number_of_nonlinear_solver_iterations"], "lu_setups": st["number_of_linear_solver_setups"],
               "max_abs_err_vs_fourier_first64": float(np.abs(y[0, :64] - ref).max()), "finite": bool(np.isfinite(y).all())}
        ns, ms_s = timed["solve"]
        nf, ms_f = timed["factor"]
        if dense:
            by = nb * (8 * n * n + 20 * n)
            fl = nb * 2.0 / 3.0 * n ** 3
            rec["roofline"] = {"bound": "hbm", "kernel": "dsh_lu_solve (k_lu_solve_blocked: forward + backward substitution on the dense factors)", "launches_timed": ns,
                               "avg_launch_us": 1e3 * ms_s / max(ns, 1), "algorithmic_bytes_per_launch": by, "achieved": by / (ms_s * 1e-3 / max(ns, 1)) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": by / (ms_s * 1e-3 / max(ns, 1)) / 1e9 / HBM_PEAK_GBS, "share_of_solve_wall": ms_s * 1e-3 / wall,
                               "measured": "HIP events on the solver stream around every dsh_lu_solve launch of one whole solve", "kernel_source_sha16": source_hash(CFG_SOURCES["c3_dense"])}
            rec["roofline_factor"] = {"bound": "mfma", "kernel": "dsh_lu_factor (staging copy + dense LU factor kernel, FP64 matrix cores)", "launches_timed": nf,
                                      "avg_launch_us": 1e3 * ms_f / max(nf, 1), "algorithmic_flop_per_launch": fl, "achieved": fl / (ms_f * 1e-3 / max(nf, 1)) / 1e12,
                                      "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / (ms_f * 1e-3 / max(nf, 1)) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                                      "share_of_solve_wall": ms_f * 1e-3 / wall}
        else:
            K = 1
            by = nb * (8 * n * (3 * K + 1) + 4 * n + 16 * n)
            rec["roofline"] = {"bound": "hbm", "kernel": "k_lu_band_solve_team<1,16,EPI>: banded solve + norm (+ Newton update) in one launch", "launches_timed": ns,
                               "avg_launch_us": 1e3 * ms_s / max(ns, 1), "algorithmic_bytes_per_launch": by, "achieved": by / (ms_s * 1e-3 / max(ns, 1)) / 1e9, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": by / (ms_s * 1e-3 / max(ns, 1)) / 1e9 / HBM_PEAK_GBS, "share_of_solve_wall": ms_s * 1e-3 / wall,
                               "measured": "HIP events on the solver stream around every dsh_lu_solve launch of one whole solve", "kernel_source_sha16": source_hash(CFG_SOURCES["c3_banded"]),
                               "note": "the chain of one wavefront's dependent FP64 operations bounds this kernel (profiles/r03_band_solve.md), not HBM; bytes = the solve's own "
                                       "(factors, pivots, right-hand side in and out) — the fused epilogue's y / x_in / x_out ride along"}
        if want_cpu:
            if "c3" not in cpu_memo:  # the reference's CPU path has one route for this config (NalgebraLU on the dense matrix): one measurement serves both GPU routes
                ns_cpu = max(1, min(nb, 4 * int(usable_cpus(cpu_limits()))))
                cpu_memo["c3"] = cfg_cpu(O.MODEL_HEAT1D, D[:ns_cpu], [tf], O.METHOD_TR_BDF2, "heat1d n=512 TR-BDF2 with the dense LU (the reference's CPU path: NalgebraLU on a dense matrix)",
                                         1, model_size=n, rtol=1e-6, atol=[1e-6])
            rec["cpu_baseline"] = cpu_memo["c3"]
        return rec


    # ---------------------------------------------------------------- C4
    # algorithmic bytes (SURVEY §8(d) with the banded factors): per accepted step read + write the difference array 2 x 8 n (q + 3) at the mean order q = 4, write
    # y, dy (16 n), read atol (8 n); per Newton iteration read the banded factors 8 n (3K + 1) + pivots 4 n, and 8 (5 n + np) of vectors
    def c4_bytes(n, steps, newton, K=1, q=4, npar=1):
        return steps * (16 * n * (q + 3) + 24 * n) + newton * (8 * n * (3 * K + 1) + 4 * n + 8 * (5 * n + npar))

    def c4(nb, dae):
        import diffsl_models as DM
        cur = spm_params(nb)
        t_eval = np.linspace(360.0, 3600.0, 10)
        if dae:
            model = diffsl.DiffslModel(DM.spm_dae(20))
            s = H.Solver(model, cur, nbatch=nb, rtol=1e-6, atol=[1e-6], device=device)
        else:
            s = H.Solver("spm", cur, nbatch=nb, model_size=20, rtol=1e-6, atol=[1e-6], device=device)
        import torch
        outb = torch.empty((len(t_eval), s.n, nb), dtype=torch.float64, device=f"cuda:{device}")
        t0 = time.perf_counter(); s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=outb.data_ptr(), deterministic_pow=ARITH); first = time.perf_counter() - t0
        walls = []
        for _ in range(3):
            t0 = time.perf_counter(); _, tot = s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=outb.data_ptr(), deterministic_pow=ARITH); walls.append(time.perf_counter() - t0)
        wall = min(walls)
        s.set_kernel_timing(True); s.set_kernel_timing_target(TIMING_RESIDENT)
        _, tot, mm = s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=outb.data_ptr(), want_member_stats=True, deterministic_pow=ARITH)
        nl, ms = s.kernel_timing()
        s.set_kernel_timing(False)
        steps, newton = tot["number_of_steps"], tot["number_of_nonlinear_solver_iterations"]
        by = c4_bytes(s.n, steps, newton)
        rec = {"workload": f"BASELINE configs[3]: single-particle battery model, {'singular mass (terminal voltage algebraic), n=43, DiffSL' if dae else 'identity-mass form, n=42'}, {nb} members"
                           f"{' (one GPU shard of the 8-GPU job)' if nb == 32768 else ' (the whole 8-GPU ensemble on one GPU)'}, BDF, rtol=atol=1e-6, t in [0, 3600 s], voltage cut-offs armed; "
                           "device-resident, one lane per member, banded LU, output left in HBM",
               "ms_per_solve": 1e3 * wall, "first_solve_ms_incl_compilation": 1e3 * first, "ode_steps_per_sec": steps / wall, "newton_solves_per_sec": newton / wall,
               "mean_steps_per_member": steps / nb, "failed_members": tot["failed_members"], "members_stopped_by_event": int((mm["root_idx"] >= 0).sum()),
               "finite": bool(torch.isfinite(outb[:1]).all().item()),
               "roofline": {"bound": "hbm", "kernel": "dsh::k_bdf_lane_banded (the whole ensemble solve, one launch; state, difference arrays, band and factors in per-lane HBM)",
                            "launches_timed": nl, "avg_launch_us": 1e3 * ms / max(nl, 1), "algorithmic_bytes_per_launch": by, "achieved": by / (ms * 1e-3 / max(nl, 1)) / 1e9,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (ms * 1e-3 / max(nl, 1)) / 1e9 / HBM_PEAK_GBS,
                            "bytes_model": "steps x (16 n (q+3) + 24 n) + newton x (8 n (3K+1) + 4 n + 8 (5 n + np)), q = 4, K = 1",
                            "measured": "HIP events on the solver stream around the launch", "kernel_source_sha16": source_hash(CFG_SOURCES["c4"])}}
        pm = _counters("r04_pmc_configs.json", "c4_dae" if dae else "c4_ode", CFG_SOURCES["c4"])
        if pm and pm.get("members") == nb:
            rec["roofline"]["traffic"] = pm.get("hbm_bytes_per_launch")
            rec["roofline"]["counters_from"] = pm.get("file")
            if pm.get("hbm_bytes_per_launch"):  # against what the kernel actually moves (calibrated counters, profiles/r05_c4_traffic.md): the rate its access pattern sustains
                rec["roofline"]["traffic_gbs"] = pm["hbm_bytes_per_launch"] / (ms * 1e-3 / max(nl, 1)) / 1e9
                rec["roofline"]["traffic_frac_of_8TBs"] = rec["roofline"]["traffic_gbs"] / HBM_PEAK_GBS
        if want_cpu and nb == 32768:
            mid = DM.host_model(O, DM.spm_dae(20)) if dae else O.MODEL_SPM
            ns_cpu = 8192 if dae else 32768
            rec["cpu_baseline"] = cfg_cpu(mid, cur[:ns_cpu], t_eval, O.METHOD_BDF, "SPM BDF, dense LU n=%d" % s.n, 64, **({} if dae else {"model_size": 20}), rtol=1e-6, atol=[1e-6])
        del s, outb
        return rec


    # ---------------------------------------------------------------- C5
    def c5(group):
        nb = 4096 if quick else 65536
        i_thresh = 0.03 if group == 1 else 1e3  # lock-step groups must agree on every event (the reference's batched root finding): threshold out of reach there
        p = rlc_params(nb, i_thresh)
        t_eval = np.linspace(0.1, 1.0, 10)
        s = H.Solver("rlc", p, nbatch=nb, model_size=1, rtol=1e-6, atol=[1e-6], method=H.METHOD_ESDIRK34, device=device)
        import torch
        outb = torch.empty((len(t_eval), s.n, nb), dtype=torch.float64, device=f"cuda:{device}")
        s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=outb.data_ptr(), group=group, deterministic_pow=ARITH)
        walls = []
        for _ in range(5):
            t0 = time.perf_counter(); _, tot = s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=outb.data_ptr(), group=group, deterministic_pow=ARITH); walls.append(time.perf_counter() - t0)
        wall = min(walls)
        _, tot_exact = s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=outb.data_ptr(), group=group, deterministic_pow=1)
        t0 = time.perf_counter(); s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=outb.data_ptr(), group=group, deterministic_pow=1); wall_exact = time.perf_counter() - t0
        s.set_kernel_timing(True); s.set_kernel_timing_target(TIMING_RESIDENT)
        _, tot, mm = s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=outb.data_ptr(), want_member_stats=True, group=group, deterministic_pow=ARITH)
        nl, ms = s.kernel_timing()
        s.set_kernel_timing(False)
        steps, newton = tot["number_of_steps"], tot["number_of_nonlinear_solver_iterations"]
        rec = {"workload": f"BASELINE configs[4]: series RLC DAE n=4 x {nb}, ESDIRK34, rtol=atol=1e-6, t in [0, 1], root i_R = {i_thresh:g}; device-resident, "
                           + ("every member its own steps and its own event" if group == 1 else "wavefront lock-step groups of 64 (threshold out of reach)"),
               "arithmetic": "fast build (library default): same counters as the exact kernel" if ARITH == 2 else "exact",
               "exact_kernel_ms_per_solve": 1e3 * wall_exact, "same_totals_as_exact_kernel": bool(tot_exact["number_of_steps"] == tot["number_of_steps"] and
                                                                                             tot_exact["number_of_nonlinear_solver_iterations"] == tot["number_of_nonlinear_solver_iterations"]),
               "ms_per_solve": 1e3 * wall, "ode_steps_per_sec": steps / wall, "newton_solves_per_sec": (newton + steps) / wall, "newton_iterations_per_sec": newton / wall,
               "mean_steps_per_member": steps / nb, "failed_members": tot["failed_members"], "members_stopped_by_event": int((mm["root_idx"] >= 0).sum()),
               "roofline": {"bound": "valu", "kernel": f"dsh::k_sdirk_resident<RlcModel, ESDIRK34, group {group}> (the whole ensemble solve, one launch; state in registers)", "launches_timed": nl,
                            "avg_launch_us": 1e3 * ms / max(nl, 1), "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "measured": "HIP events on the solver stream around the launch",
                            "kernel_source_sha16": source_hash(CFG_SOURCES["c5"]),
                            "hbm": {"algorithmic_bytes_per_launch": 8 * (p.shape[1] + s.n * len(t_eval)) * nb,
                                    "note": "parameters in + save points out; the solver state never leaves registers, HBM does not bound this kernel"}}}
        pm = _counters("r04_pmc_configs.json", "c5_per_member" if group == 1 else "c5_group64", CFG_SOURCES["c5"])
        avg_s = ms * 1e-3 / max(nl, 1)
        if pm and pm.get("members") == nb:
            valu, f64 = pm["valu_insts_per_launch"], pm.get("f64_insts_per_launch") or 0
            rec["roofline"].update({"achieved": valu * 64 / avg_s / 1e12, "frac": valu * 64 / avg_s / 1e12 / VALU_PEAK_TLANEOPS, "valu_wave_instructions_per_launch": valu,
                                    "issue_slot_frac": (4 * f64 + 2 * (valu - f64)) / (SIMD_CYCLES_PER_S * avg_s), "traffic": pm.get("hbm_bytes_per_launch"),
                                    "counters_from": pm.get("file")})
        else:
            rec["roofline"].update({"achieved": None, "frac": None, "traffic": None, "note": "no PMC summary of this kernel build under profiles/ (scripts/profile_configs.sh)"})
        if want_cpu:
            ns_cpu = 16384
            rec["cpu_baseline"] = cfg_cpu(O.MODEL_RLC, p[:ns_cpu], t_eval, O.METHOD_ESDIRK34, "RLC ESDIRK34 with the root function armed", 256, model_size=1, rtol=1e-6, atol=[1e-6])
            rec["cpu_baseline"]["note"] = "independent solves = the per-member semantics; the group-of-64 GPU entry runs the batched (lock-step) semantics" if group != 1 else "same semantics as this entry"
        del s, outb
        return rec

    # most informative first: a time budget that runs out drops the tail of this list
    nb4 = 4096 if quick else 32768
    guarded("c3_banded", lambda: c3(False))
    guarded(f"c4_ode_{nb4}", lambda: c4(nb4, False))
    guarded("c5_per_member", lambda: c5(1))
    guarded(f"c4_dae_{nb4}", lambda: c4(nb4, True))
    guarded("c3_dense", lambda: c3(True))
    guarded("c5_group64", lambda: c5(64))
    if not quick:
        guarded("c4_ode_262144", lambda: c4(262144, False))
        guarded("c4_dae_262144", lambda: c4(262144, True))
    return out


def bench_trait_path(params, device, k):
    """BASELINE configs[1] through the pure 1:1 trait composition: fused=False (every algorithm step one Vector/Matrix/LinearSolver call, one launch each — what an unchanged
    OdeBuilder....bdf::<HipLU>() of the Rust shim executes), host-driven lock-step over the whole ensemble."""
    import diffsol_amd as H
    from diffsol_amd.solver import ENSEMBLE_LOCKSTEP
    nb = params.shape[0]
    s = H.Solver("robertson_ode", params, nbatch=nb, model_size=1, rtol=RTOL, atol=ATOL, device=device, fused=False, ensemble_mode=ENSEMBLE_LOCKSTEP)
    assert not s.fused
    s.solve_dense(T_EVAL, want_host=False)
    walls = []
    for _ in range(k):
        s.reset()
        t0 = time.perf_counter(); s.solve_dense(T_EVAL, want_host=False); walls.append(time.perf_counter() - t0)
    st = s.stats()
    wall = min(walls)
    return {"ms_per_step": 1e3 * wall, "ode_steps_per_sec": st["number_of_steps"] * nb / wall, "newton_solves_per_sec": st["number_of_nonlinear_solver_iterations"] * nb / wall,
            "lockstep_steps": st["number_of_steps"], "failed_members": 0,
            "note": "use_fused_kernels = 0 + DSHS_ENSEMBLE_LOCKSTEP: the literal drop-in composition (HipVec/HipMat/HipLU operations only, one launch each, the reference's own "
                    "sequence of trait calls); the Rust shim's OdeBuilder/.bdf() executes exactly these dsh_* calls"}


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
    ap.add_argument("--no-configs", action="store_true", help="skip BASELINE configs[2..4] (the `configs` object) and the pure trait-path pass")
    ap.add_argument("--quick-configs", action="store_true", help="configs at reduced ensemble sizes (smoke test of the bench itself; never a measurement)")
    ap.add_argument("--time-budget", type=float, default=75.0,
                    help="seconds since process start (not counting `import torch`) after which no further BASELINE config of the `configs` object is started (reported as skipped); 0 = no limit")
    ap.add_argument("--cpu-sample", type=int, default=400_000)
    ap.add_argument("--large-nb", type=int, default=1_600_000)
    ap.add_argument("--config", default="c2", choices=["c2", "c4"],
                    help="c2 (default): BASELINE configs[1], weak scaling (100 000 members per GPU).  c4: BASELINE configs[3] as worded — 262 144 battery-model members SPLIT over "
                         "the N GPUs (strong scaling), one gather of the trajectories at the end of every solve")
    ap.add_argument("--members-total", type=int, default=262_144, help="--config c4: ensemble size (split over the ranks)")
    ap.add_argument("--gather", default="torch", choices=["torch", "cabi"],
                    help="--config c4: torch = torch.distributed all_gather_into_tensor (RCCL); cabi = the library's own dsh_gather_batch_axis (csrc/dsh_dist.hip: librccl bound "
                         "by libdiffsol_hip.so, what a Rust / C caller uses), the 128-byte id broadcast through torch's store")
    ap.add_argument("--cpu-stub", action="store_true", help="TEST HOOK: no GPU, gloo backend, stub solver (exercises launcher + aggregation only)")
    args = ap.parse_args()
    assert args.gpus >= 1 and args.steps >= 1 and args.warmup >= 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _relaunch_under_torchrun(args)  # does not return

    _t_imp = time.perf_counter()
    import torch
    import torch.distributed as dist
    IMPORT_SECONDS[0] += time.perf_counter() - _t_imp

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

    if args.config == "c4":
        return main_c4(args, rank, local_rank, world, stub)
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
        arithmetic = "cpu-stub"

        def solve_once(buf=None):
            return solver.solve_dense_into(out if buf is None else buf)
    else:
        import diffsol_amd
        from diffsol_amd.solver import ENSEMBLE_LOCKSTEP, ENSEMBLE_PER_MEMBER, ENSEMBLE_WAVEFRONT, ENSEMBLE_AUTO
        solver = diffsol_amd.Solver("robertson_ode", params[lo:hi], nbatch=hi - lo, model_size=1, rtol=RTOL, atol=ATOL, device=local_rank, block_threads=256)
        assert solver.fused, "fused HIP kernels not active"
        _, resolved = solver.ensemble_mode()
        assert resolved == ENSEMBLE_WAVEFRONT, f"solve_dense did not resolve to the device-resident integrator (mode {resolved})"
        fast_arith = diffsol_amd.get_resident_arithmetic() == diffsol_amd.ARITH_FAST
        arithmetic = ("f64, library default (DSHS_ARITH_FAST): fused multiply-adds, reciprocal-math division, ocml pow; same step decisions as the exact kernel, states within 1e-9 "
                      "(tests/test_gpu_adaptive.py); exact kernel under extras.exact_variant_ms") if fast_arith else "f64 exact (DSH_RESIDENT_ARITH=exact): bit-identical to the CPU oracle"

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
            pm_name = latest_profile("pmc_per_member.json") or "r06_pmc_per_member.json"
            pm = json.load(open(os.path.join(ROOT, "profiles", pm_name))).get("bench_kernel", {})
            if pm.get("kernel_source_sha16") == kernel_source_hash() and pm.get("members") == nb and world == 1:
                t_s = extras["per_member"]["ms_per_step"] * 1e-3
                extras["per_member"]["roofline"] = {
                    "bound": "valu", "kernel": pm.get("kernel"), "avg_launch_us": t_s * 1e6, "measured": "wall clock of the per-member solves of this pass (one launch each)",
                    "achieved": pm["valu_insts_per_launch"] * 64 / t_s / 1e12, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s",
                    "frac": pm["valu_insts_per_launch"] * 64 / t_s / 1e12 / VALU_PEAK_TLANEOPS, "valu_wave_instructions_per_launch": pm["valu_insts_per_launch"],
                    "fp64_wave_instructions_per_launch": pm.get("f64_insts_per_launch"), "traffic": pm.get("hbm_bytes_per_launch"), "counters_from": "profiles/" + pm_name,
                    "note": "lane-operations ISSUED, most of them masked off: 3.3x the wave-instructions of the lock-step kernel for fewer member-steps is divergence inside "
                            "wavefronts (profiles/r03_per_member.md: 8.3 ms against 3.2 ms for the same ensemble size with no divergence inside any wavefront)"}
            else:
                extras["per_member"]["roofline"] = None
        except Exception:
            extras["per_member"]["roofline"] = None
        # the EXACT kernel of the same solve (deterministic_pow = 1: bit-identical to the CPU oracle; what `value` was through round 5).  `value` is the library
        # default since round 6: the fast-arithmetic build of the same source (dsh_adaptive_fast.hip: fused multiply-adds, reciprocal-math division, ocml pow), which
        # makes the same step / order / refactorisation decisions for every member of this ensemble (tests/test_gpu_adaptive.py) — checked again here on the counters
        def exact_step():
            _, tot = solver.solve_dense_adaptive(T_EVAL, want_host=False, dev_ptr=out.data_ptr(), group=64, deterministic_pow=1)
            return tot, (gather_batch_axis(out, n_total, rank, world) if world > 1 else out)
        try:
            y_default = y_value_pass
            exact_step()
            solver.set_kernel_timing(True)
            el, a, ye = timed(k_x, exact_step)
            xl, xms = solver.kernel_timing()
            solver.set_kernel_timing(False)
            el, st, nw, fl = allreduce([el, a["number_of_steps"], a["number_of_nonlinear_solver_iterations"], a["failed_members"]])
            big = ye.abs() > 1e-9
            extras["exact_variant"] = {"ms_per_step": 1e3 * el / k_x, "kernel_ms": xms / max(xl, 1), "ode_steps_per_sec": st / el, "newton_solves_per_sec": nw / el, "failed_members": int(fl),
                                       "same_step_and_newton_totals_as_value_pass": bool(abs(st / k_x - member_steps / args.steps) < 0.5 and abs(nw / k_x - member_newton / args.steps) < 0.5),
                                       "max_rel_diff_of_value_pass_states": float(((y_default - ye).abs() / ye.abs().clamp_min(1e-300))[big].max().item()),
                                       "note": "deterministic_pow = 1 / DSH_RESIDENT_ARITH=exact: no contraction, IEEE division, portable pow; bit-identical to the oracle "
                                               "(the GPU test tier runs in this mode)"}
        except Exception as e:  # noqa: BLE001 — an extra must not take the bench line down
            extras["exact_variant"] = {"error": str(e)[:200]}
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
                "workload_short": "BASELINE configs[1]: Robertson (n=3, f64) ensemble, 100k members per GPU, BDF, batched dense LU, t in [0,4e5], 7 save points",
                "workload": "BASELINE.json configs[1]: Robertson stiff ODE (n=3, fp64) ensemble, 100k parameter-sweep members per GPU, BDF, "
                            "batched dense LU, t in [0, 4e5], rtol 1e-4, atol (1e-8,1e-14,1e-6), output at 7 decades",
                "members_per_gpu": nb, "members_total": n_total, "t_final": T_EVAL[-1], "method": "bdf",
                "path": "dshs_solve_dense, default ensemble mode -> device-resident BDF (dsh_bdf_solve_adaptive), wavefront lock-step groups of 64 members "
                        "(the reference's batched semantics with nbatch = 64 per group)",
                "arithmetic": arithmetic,
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
            pmc_name = latest_profile("pmc_resident.json") or "r06_pmc_resident.json"
            path = os.path.join(ROOT, "profiles", pmc_name)
            if os.path.exists(path):
                try:
                    pmc = json.load(open(path)).get("bench_kernel", {}) or {}
                except Exception:
                    pmc = {}
                if pmc:
                    pmc["file"] = "profiles/" + pmc_name
                    if pmc.get("kernel_source_sha16") != kernel_source_hash():  # counters of another kernel: no fraction rather than a stale one
                        stale = f"profiles/{pmc_name} was measured on kernel sources {pmc.get('kernel_source_sha16')}, this tree has {kernel_source_hash()}: re-run scripts/profile_r06.sh"
                        pmc = {}
            algo_hbm = 8 * (N_PARAMS + N_STATES * len(T_EVAL)) * (hi - lo)
            roof = {"bound": "valu", "kernel": "dsh::k_bdf_adaptive<RobertsonOde1, BA=true, WAVE=true" + (", FAST=true" if not stub and fast_arith else "") + "> (the whole ensemble solve, one launch)",
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
                    # the headline fraction is the issue-slot one (the honest one for a kernel whose VALU stream is mixed FP64 / other); the lane-operation
                    # rate against the FP64 VALU peak stays next to it
                    roof["lane_ops"] = {"achieved": roof["achieved"], "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "frac": roof["frac"]}
                    roof.update({"bound": "valu-issue", "achieved": (4 * f64 + 2 * (valu - f64)) / avg_s / 1e12, "peak": SIMD_CYCLES_PER_S / 1e12, "unit": "T issue-cycles/s (all SIMDs)",
                                 "frac": roof["issue_slot_frac"],
                                 "frac_definition": "VALU issue cycles of the launch (4 per FP64 wave-instruction, 2 per other VALU wave-instruction, from the committed counters) / (1024 SIMDs x clock x launch time)"})
                    roof["fp64_flop_per_launch"] = pmc.get("f64_flop_per_launch")
                    if pmc.get("f64_flop_per_launch"):
                        roof["fp64_tflops"] = pmc["f64_flop_per_launch"] / avg_s / 1e12
                roof["traffic"] = pmc.get("hbm_bytes_per_launch")
                roof["counters_from"] = pmc.get("file")
                # the four fractions side by side: `frac` = VALU issue cycles (the bound); fp64_frac = FP64 flop against the derived 78.6 TFLOP/s;
                # hbm_frac_actual = counter traffic against 8 TB/s (the state never leaves registers: ~0.001 by design); hbm_model_8d_frac = SURVEY 8(d)'s byte model
                # (LU, state and history re-read from HBM per unit of work), which this kernel does not follow and which therefore exceeds 1
                if roof.get("fp64_tflops"):
                    roof["fp64_frac"] = roof["fp64_tflops"] / FP64_MATRIX_PEAK_TFLOPS
                if roof["traffic"]:
                    roof["hbm_frac_actual"] = roof["traffic"] / avg_s / 1e9 / HBM_PEAK_GBS
            # the 8(d) model needs no counters: units executed by one launch of this rank (the totals are sums over ranks and timed solves)
            per = 1.0 / args.steps / world
            model_bytes = (member_newton * MODEL_8D_NEWTON_BYTES + member_steps * MODEL_8D_STEP_BYTES + member_setups * MODEL_8D_REFACTOR_BYTES) * per
            roof["hbm_model_8d_bytes_per_launch"] = model_bytes
            roof["hbm_model_8d_frac"] = model_bytes / avg_s / 1e9 / HBM_PEAK_GBS
            roof["hbm_model_8d_note"] = ("SURVEY 8(d) bytes (228 B per Newton iteration, 360 B per accepted step, 156 B per refactorisation) assume LU/state/history in HBM; "
                                         "here they live in registers, so the model does not apply (> 1 is expected) — hbm_frac_actual is the measured traffic")
            if not valu:
                roof.update({"achieved": None, "frac": None, "traffic": None, "note": stale or "no PMC summary for this ensemble size under profiles/"})
            rec["roofline"] = roof
        else:
            rec["roofline"] = None
        rec.update(extras)
        if world == 1 and not args.no_cpu_baseline and not stub:
            big_sample = robertson_params(max(args.cpu_sample, n_total))  # same distribution, same seed, a longer draw
            rec["cpu_baseline"] = cpu_baseline(big_sample, args.cpu_sample)
        if world == 1 and not stub and not args.no_configs:
            del solver
            try:
                rec["trait_path"] = bench_trait_path(params[lo:hi], local_rank, 3)
            except Exception as e:  # noqa: BLE001
                rec["trait_path"] = {"error": str(e)[:300]}
            rec["configs"] = bench_configs(local_rank, not args.no_cpu_baseline, quick=args.quick_configs, budget_s=args.time_budget)
        rec["bench_seconds"] = time.perf_counter() - T_START
        rec["import_torch_seconds"] = IMPORT_SECONDS[0]
        if not stub:
            try:  # modules hiprtc compiled during THIS run (0 when build() has replayed the committed manifests: first-use compilation inside the bench is a bug)
                from diffsol_amd import _ffi
                rec["jit_compiles_this_run"] = int(_ffi.load_device_lib().dsh_jit_compile_count())
            except Exception:
                pass
        emit(rec)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_c4(args, rank, local_rank, world, stub):
    """BASELINE configs[3] as worded: the 262 144-member battery ensemble SPLIT over the N ranks (strong scaling), device-resident BDF per rank on its shard
    (no collective inside the integration), ONE gather of the [save points x states x members] trajectories per solve.  One JSON line, `scaling: "strong"`."""
    import torch
    import torch.distributed as dist

    from diffsol_amd.dist import CabiCommunicator, gather_batch_axis, shard_bounds
    n_total = args.members_total
    lo, hi = shard_bounds(n_total, rank, world)
    cur = spm_params(n_total)
    t_eval = np.linspace(360.0, 3600.0, 10)
    dev = "cpu" if stub else f"cuda:{local_rank}"

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()
    comm = None
    if stub:
        n = 42

        def solve(buf):
            buf.copy_(torch.from_numpy(np.broadcast_to(cur[lo:hi, 0], (len(t_eval), n, hi - lo)).copy()))
            return {"number_of_steps": 90 * (hi - lo), "number_of_nonlinear_solver_iterations": 100 * (hi - lo), "failed_members": 0}
    else:
        import diffsol_amd as H
        s = H.Solver("spm", cur[lo:hi], nbatch=hi - lo, model_size=20, rtol=1e-6, atol=[1e-6], device=local_rank)
        n = s.n

        def solve(buf):
            return s.solve_dense_adaptive(t_eval, want_host=False, dev_ptr=buf.data_ptr())[1]
        if args.gather == "cabi":
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt = torch.frombuffer(bytearray(CabiCommunicator.unique_id()), dtype=torch.uint8).to(dev)
            if world > 1:
                dist.broadcast(idt, src=0)
            comm = CabiCommunicator(s.context_handle(), rank, world, bytes(idt.cpu().numpy().tobytes()), owner=s)
    out = torch.empty((len(t_eval), n, hi - lo), dtype=torch.float64, device=dev)
    full = torch.empty((len(t_eval), n, n_total), dtype=torch.float64, device=dev) if comm is not None else None

    def step():
        tot = solve(out)
        if comm is not None:
            return tot, comm.gather(out, n_total, out=full)
        return tot, (gather_batch_axis(out, n_total, rank, world) if world > 1 else out)
    for _ in range(max(args.warmup, 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        tot, y = step()
        for k, v in tot.items():
            acc[k] = acc.get(k, 0) + v
    barrier()
    el = time.perf_counter() - t0
    a = torch.tensor([el, acc["number_of_steps"], acc["number_of_nonlinear_solver_iterations"], acc["failed_members"]], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = a[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        a[0] = tmax[0]
    el, steps, newton, failed = [float(v) for v in a.tolist()]
    ok = bool(torch.isfinite(y[0]).all().item()) and y.shape[-1] == n_total
    if rank == 0:
        emit({
            "metric": "ODE steps/sec (and Newton solves/sec) per ensemble", "value": steps / el, "unit": "accepted ODE steps/s summed over ensemble members",
            "newton_solves_per_sec": newton / el, "n_gpus": world, "ranks": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "cpu-stub (test hook: launcher/aggregation path only, not a measurement)" if stub else "synthetic",
            "config": {"workload": f"BASELINE.json configs[3]: single-particle battery model (n = {n}), {n_total} members SPLIT over {world} GPU(s), BDF, rtol=atol=1e-6, t in [0, 3600 s], "
                                   "voltage cut-offs armed, device-resident per-member control, one trajectory gather per solve",
                       "members_total": n_total, "members_per_gpu": hi - lo, "parallelism": f"ensemble-shard x{world}",
                       "gather": "none" if world == 1 and comm is None else ("dsh_gather_batch_axis (library-bound RCCL)" if comm is not None else "torch.distributed all_gather_into_tensor (RCCL)"),
                       "gathered_bytes_per_solve": 8 * len(t_eval) * n * n_total},
            "checks": {"finite_and_complete": ok, "failed_members": int(failed)}})
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
