"""Turn the rocprofv3 sqlite outputs merged back from the GPU box (gpurun_out/) into the small committed summaries under profiles/.

    python scripts/summarize_profiles.py <kernel-trace .db> <fetch-pmc .db> <write-pmc .db> <tag>
"""
import collections
import json
import os
import sqlite3
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_stats(db):
    con = sqlite3.connect(db)
    return list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))


def pmc(db, counter):
    con = sqlite3.connect(db)
    d = collections.defaultdict(list)
    for name, val in con.execute("select kernel_name,value from counters_collection where counter_name=?", (counter,)):
        d[name].append(val)
    return {k: (len(v), float(np.mean(v))) for k, v in d.items()}


def main():
    trace_db, fetch_db, write_db, tag = sys.argv[1:5]
    cmd = sys.argv[5] if len(sys.argv) > 5 else "python bench.py --no-cpu-baseline --no-kernel-events"
    rows = kernel_stats(trace_db)
    out = os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.md")
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats — {tag}\n\nCommand: `rocprofv3 --kernel-trace --stats -- {cmd}` on one MI355X "
                "(2 warm-up + 5 timed solves of the 100k-member Robertson ensemble; durations in microseconds).\n\n"
                "| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:14]:
            f.write(f"| `{r[0][:120]}` | {r[1]} | {r[2]:.1f} | {r[3]:.3f} | {r[4]:.2f} |\n")
    print("wrote", out)
    fe, wr = pmc(fetch_db, "FETCH_SIZE"), pmc(write_db, "WRITE_SIZE")
    newton = [k for k in fe if "k_newton_iter" in k and "RobertsonOde1" in k]
    summ = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only), KB per dispatch averaged over all dispatches of the "
                     "kernel in one 100k-member Robertson solve.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports 1/2 of the bytes of a coalesced "
                     "streaming read — confirmed here on kernels with known traffic (k_binary<FAxpy> reads 2 x 2343.75 KB, reports 2346 KB; k_transpose reads 2343.75 KB, "
                     "reports 1186 KB) — so read bytes = 2 x FETCH_SIZE; WRITE_SIZE matches known store traffic 1:1.",
            "kernels": {}}
    for k in sorted(fe, key=lambda k: -fe[k][0]):
        if k in wr and fe[k][0] >= 5:
            summ["kernels"][k[:140]] = {"dispatches": fe[k][0], "FETCH_SIZE_KB_raw": fe[k][1], "WRITE_SIZE_KB": wr[k][1],
                                        "hbm_bytes_per_launch_corrected": (2.0 * fe[k][1] + wr[k][1]) * 1024.0}
    if newton:
        k = max(newton, key=lambda k: fe[k][0])
        summ["hbm_bytes_per_launch"] = (2.0 * fe[k][1] + wr[k][1]) * 1024.0
        summ["kernel"] = k[:140]
    with open(os.path.join(ROOT, "profiles", "pmc_newton_iter.json"), "w") as f:
        json.dump(summ, f, indent=1)
    print(json.dumps(summ, indent=1)[:1500])


if __name__ == "__main__":
    main()
