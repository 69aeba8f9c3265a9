#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace --stats) as text:
per-kernel calls / total / average / min / max, and per (kernel, grid) groups so that the four GEMV
shapes of a decode layer can be told apart.  Usage: tools/rocpd_summary.py results.db [out.txt]"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'%':>6}  kernel", file=out)
for n, c, s, a, mn, mx in rows[:25]:
    print(f"{c:8d} {s/1e6:10.3f} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.1f}  {n[:150]}", file=out)
print("\nper (kernel, grid, workgroup) for the top kernel family:", file=out)
rows = db.execute("select name, grid_x, workgroup_x, lds_size, count(*), avg(duration), min(duration), max(duration) from kernels group by name, grid_x, workgroup_x, lds_size order by sum(duration) desc").fetchall()
for n, g, w, l, c, a, mn, mx in rows[:16]:
    print(f"{c:8d} calls  grid {g:7d} wg {w:4d} lds {l:6d}  avg {a/1e3:8.2f} us  min {mn/1e3:8.2f}  max {mx/1e3:8.2f}  {n[:90]}", file=out)
# inter-kernel gaps on the busiest stream
rows = db.execute("select start, end from kernels order by start").fetchall()
gaps = [b[0] - a[1] for a, b in zip(rows, rows[1:]) if 0 <= b[0] - a[1] < 20000]
if gaps:
    gaps.sort()
    print(f"\ninter-kernel gaps (< 20 us, n={len(gaps)}): median {gaps[len(gaps)//2]/1e3:.2f} us, p10 {gaps[len(gaps)//10]/1e3:.2f}, p90 {gaps[9*len(gaps)//10]/1e3:.2f}", file=out)
