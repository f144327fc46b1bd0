"""Idle gaps between consecutive kernels of a rocprofv3 rocpd database:  python scripts/kernel_gaps.py <db> [min_gap_us]
prints the total busy / idle time of the traced interval and the gaps above the threshold with the kernels on both sides."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 1e6
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
view = next(t for t in tabs if t == "kernels" or t.endswith("kernels") and "top" not in t)
cols = [r[1] for r in con.execute(f"pragma table_info({view})")]
rows = list(con.execute(f"select name, start, end from {view} order by start"))
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
print(f"{len(rows)} kernels, span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms, idle {(span-busy)/1e6:.1f} ms")
agg = {}
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g > thr:
        k = (n0[:50], n1[:50])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += g
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"{c:5d} gaps, {t/1e6:9.1f} ms total, {t/c/1e3:9.1f} us avg:  {k[0]}  ->  {k[1]}")
