#!/usr/bin/env python3
"""Idle time inside the steps of a bench.py run traced with rocprofv3 --kernel-trace (rocpd db):
lists the largest gaps between consecutive kernels (any stream) and what ran either side.
usage: tools/step_gaps.py results.db [min_gap_us]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# steady state: the last 60 % of the trace
t0 = rows[0][1]
t1 = max(r[2] for r in rows)
lo = t0 + 0.4 * (t1 - t0)
rows = [r for r in rows if r[1] >= lo]
busy_end = rows[0][2]
gaps = []
idle = 0
for prev, cur in zip(rows, rows[1:]):
    if cur[1] > busy_end:
        g = (cur[1] - busy_end) / 1e3
        idle += g
        if g >= min_gap:
            gaps.append((g, prev[0][:60], cur[0][:60], (cur[1] - rows[0][1]) / 1e6))
    busy_end = max(busy_end, cur[2])
span = (busy_end - rows[0][1]) / 1e3
print(f"span {span / 1e3:.2f} ms, idle {idle / 1e3:.2f} ms ({100 * idle / span:.1f} %), kernels {len(rows)}")
small = idle - sum(g[0] for g in gaps)
print(f"gaps < {min_gap} us: {small / 1e3:.2f} ms in total")
for g in sorted(gaps, key=lambda g: g[3]):
    print(f"  t={g[3]:8.2f} ms  gap {g[0]:8.1f} us   after {g[1]}   before {g[2]}")
