"""Print the top_kernels view of a rocprofv3 rocpd database as a markdown table:  python scripts/top_kernels.py <db> [rows]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for r in rows[:n]:
    print(f"| `{r[0][:110]}` | {r[1]} | {r[2]:.1f} | {r[3]:.3f} | {r[4]:.2f} |")
