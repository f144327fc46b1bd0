#!/usr/bin/env python
"""Copy a profiling summary (gpurun_out/<round>/pmc_resident_<tag>.json from scripts/profile_r03.sh) into the committed profiles/:
    python scripts/publish_profile.py <tag> [round, default r03]
writes profiles/<round>_pmc_resident.json (what bench.py's roofline reads: `bench_kernel`, stamped with the hash of the kernel's sources so that bench.py can
refuse counters of another kernel) and profiles/<round>_kernel_stats.md (the kernel trace)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r03"
member = len(sys.argv) > 3 and sys.argv[3] == "member"  # the per-member kernel (MODE=member bash scripts/profile_r03.sh <tag>) -> profiles/<round>_pmc_per_member.json
sys.path.insert(0, ROOT)
from bench import kernel_source_hash
d = json.load(open(os.path.join(ROOT, "gpurun_out", rnd, f"pmc_resident_{tag}.json")))
name, k = next((n, v) for n, v in d["kernels"].items() if "k_bdf_adaptive" in n and ("true, false" if member else "true, true") in n)
tr = [r for r in d.get("kernel_trace", []) if "k_bdf_adaptive" in r["name"]]
out = {
    "_note": "rocprofv3 --kernel-trace --pmc <counters> -- python scripts/bench_kernel_once.py 100000 3  (scripts/profile_r03.sh; one MI355X; separate passes: "
             "SQ instruction counters, SQ wait/active counters, FETCH_SIZE, WRITE_SIZE; means over the 3 dispatches).  The kernel is bench.py's whole timed region: one "
             "launch = one solve_dense of the 100 000-member C2 Robertson ensemble (seed 12345), wavefront lock-step groups of 64.  SQ_WAVE_CYCLES / SQ_WAIT_* / "
             "SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md).  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the guide's gfx950 correction for reads).  "
             "f64_flop = 64 x (ADD + MUL + 2 FMA + TRANS) F64 wave-instructions.",
    "bench_kernel": {
        "kernel": name, "members": 100000, "kernel_source_sha16": kernel_source_hash(),
        "valu_insts_per_launch": k["SQ_INSTS_VALU"], "f64_insts_per_launch": k["f64_insts"], "f64_flop_per_launch": k["f64_flop"],
        "salu_insts_per_launch": k.get("SQ_INSTS_SALU"), "lds_insts_per_launch": k.get("SQ_INSTS_LDS"), "smem_insts_per_launch": k.get("SQ_INSTS_SMEM"),
        "vmem_rd_insts_per_launch": k.get("SQ_INSTS_VMEM_RD"), "waves": k.get("SQ_WAVES"),
        "wave_quad_cycles": k.get("SQ_WAVE_CYCLES"), "active_frac": k["SQ_ACTIVE_INST_ANY"] / k["SQ_WAVE_CYCLES"],
        "wait_any_frac": k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], "wait_inst_frac": k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"],
        "hbm_bytes_per_launch": k["hbm_bytes_per_launch_corrected"], "FETCH_SIZE_KB_raw": k["FETCH_SIZE"], "WRITE_SIZE_KB": k["WRITE_SIZE"],
        "kernel_trace_avg_us": tr[0]["avg_us"] if tr else None, "kernel_trace_calls": tr[0]["calls"] if tr else None,
    },
    "raw": k,
}
if member:
    out["_note"] = out["_note"].replace("bench_kernel_once.py 100000 3 ", "bench_kernel_once.py 100000 3 member ").replace(
        "wavefront lock-step groups of 64", "PER-MEMBER control (every member its own step sizes and orders; members in the library's launch-time Morton order) — bench.py's `per_member` extra")
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_per_member.json"), "w"), indent=1)
    print(json.dumps(out["bench_kernel"], indent=1))
    sys.exit(0)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_resident.json"), "w"), indent=1)
with open(os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats.md"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2   (1 x MI355X)\n\n"
            "The timed region of bench.py is ONE kernel: `dshs_solve_dense` in its default ensemble mode launches `dsh::k_bdf_adaptive<RobertsonOde1, BA, WAVE>` once per "
            "ensemble solve (100 000 Robertson members, ~303 BDF steps and ~690 Newton iterations per wavefront group inside the launch).  Durations in microseconds.\n\n"
            "| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
    for r in d.get("kernel_trace", []):
        f.write(f"| `{r['name'][:110]}` | {r['calls']} | {r['total_us']:.1f} | {r['avg_us']:.3f} | {r['pct']:.2f} |\n")
print(json.dumps(out["bench_kernel"], indent=1))
