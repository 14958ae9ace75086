#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 (rocpd sqlite) kernel trace: tools/rocpd_stats.py <results.db> [n_runs]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
runs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = c.execute("select name, grid_x/workgroup_x, grid_y, count(*), sum(end-start)/1e3, avg(end-start)/1e3 "
                 "from kernels group by name, grid_x, grid_y order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
print(f"total kernel time {tot / runs:.1f} us per run ({runs:g} runs)")
for r in rows[:40]:
    print(f"{r[0][:78]:78s} blocks=({r[1]},{r[2]}) n={r[3]:4d} total/run={r[4] / runs:9.1f}us avg={r[5]:8.1f}us {100 * r[4] / tot:5.1f}%")
