#!/usr/bin/env python
"""gpurun_out/r04/pmc_<cfg>.json (scripts/profile_configs.sh) -> profiles/r04_pmc_configs.json: one entry per config with the counters bench.py's `configs`
rooflines read, stamped with the hash of the kernel's sources (bench.py refuses counters of another kernel build).
    python scripts/publish_configs_profile.py c4_ode:262144 c4_dae:262144 c5_per_member:65536 c5_group64:65536"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CFG_SOURCES, source_hash

RND = os.environ.get("RND", "r05")
path = os.path.join(ROOT, "profiles", f"{RND}_pmc_configs.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out["_note"] = ("rocprofv3 --kernel-trace --pmc <counters> -- python scripts/config_once.py <cfg> <members> (scripts/profile_configs.sh; one MI355X; separate passes: SQ instruction "
                "counters, SQ wait/active counters, FETCH_SIZE, WRITE_SIZE; means over the dispatches).  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
                "(MI355X_MICROARCH.md's gfx950 read correction).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.")
for arg in sys.argv[1:]:
    cfg, nb = arg.split(":")
    d = json.load(open(os.path.join(ROOT, "gpurun_out", RND, f"pmc_{cfg}.json")))
    name, k = max(d["kernels"].items(), key=lambda kv: kv[1].get("SQ_INSTS_VALU", 0))
    tr = [r for r in d.get("kernel_trace", []) if r["name"][:60] == name[:60]]
    out[cfg] = {"kernel": name, "members": int(nb), "kernel_source_sha16": source_hash(CFG_SOURCES["c4" if cfg.startswith("c4") else "c5"]),
                "valu_insts_per_launch": k.get("SQ_INSTS_VALU"), "f64_insts_per_launch": k.get("f64_insts"), "f64_flop_per_launch": k.get("f64_flop"),
                "waves": k.get("SQ_WAVES"), "active_frac": k["SQ_ACTIVE_INST_ANY"] / k["SQ_WAVE_CYCLES"] if k.get("SQ_WAVE_CYCLES") and k.get("SQ_ACTIVE_INST_ANY") is not None else None,
                "wait_any_frac": k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"] if k.get("SQ_WAVE_CYCLES") and k.get("SQ_WAIT_ANY") is not None else None,
                "hbm_bytes_per_launch": k.get("hbm_bytes_per_launch_corrected"), "FETCH_SIZE_KB_raw": k.get("FETCH_SIZE"), "WRITE_SIZE_KB": k.get("WRITE_SIZE"),
                "kernel_trace_avg_us": tr[0]["avg_us"] if tr else None, "kernel_trace_calls": tr[0]["calls"] if tr else None}
    print(cfg, json.dumps(out[cfg])[:400])
json.dump(out, open(path, "w"), indent=1)
