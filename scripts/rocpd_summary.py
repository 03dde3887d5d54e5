"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel table (markdown)."""
import sqlite3, sys
db = sys.argv[1]
con = sqlite3.connect(db)
rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for n, c, s, a, mn, mx in rows:
    print(f"| {n[:110]} | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.1f} |")
