#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into the per-kernel table
committed under profiles/.  usage: tools/rocpd_summary.py results.db > profiles/<name>.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
print(f"# total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'Name':100s} {'Calls':>7s} {'TotalDurationNs':>16s} {'AverageNs':>12s} {'Percentage':>10s} {'MinNs':>10s} {'MaxNs':>10s}")
for name, n, tot, avg, mn, mx in rows:
    print(f"{name[:100]:100s} {n:7d} {tot:16d} {avg:12.1f} {100.0 * tot / total:10.2f} {mn:10d} {mx:10d}")
