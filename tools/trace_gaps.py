#!/usr/bin/env python3
"""Timeline accounting of a rocprofv3 --kernel-trace rocpd database: per steady-state frame of
tools/latency_probe.py, how much is kernel time and how much is gaps between kernels.
usage: tools/trace_gaps.py results.db [frames_to_skip]"""
import sqlite3
import sys

import numpy as np

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0] for r in rows]
st = np.array([r[1] for r in rows], dtype=np.int64)
en = np.array([r[2] for r in rows], dtype=np.int64)
# a frame starts at each loc_scatter kernel
starts = [i for i, n in enumerate(names) if "loc_scatter" in n]
if len(starts) < 8:
    sys.exit("too few frames in the trace")
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
fr = []
for a, b in zip(starts[skip:-1], starts[skip + 1:]):
    dur = en[a:b] - st[a:b]
    span = en[b - 1] - st[a]
    busy = 0
    # union of intervals (streams may overlap)
    cur_s, cur_e = st[a], en[a]
    for s, e in zip(st[a + 1:b], en[a + 1:b]):
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    fr.append((b - a, span, dur.sum(), busy))
fr = np.array(fr, dtype=np.float64)
print(f"frames {len(fr)}: kernels/frame {fr[:, 0].mean():.0f}  span {fr[:, 1].mean() / 1e3:.1f} us  "
      f"sum of kernel durations {fr[:, 2].mean() / 1e3:.1f} us  busy (union) {fr[:, 3].mean() / 1e3:.1f} us  "
      f"idle gaps {(fr[:, 1] - fr[:, 3]).mean() / 1e3:.1f} us")
# per-kernel-name totals over the analysed frames
a, b = starts[skip], starts[-1]
agg = {}
for n, s, e in zip(names[a:b], st[a:b], en[a:b]):
    k = n[:90]
    t = agg.setdefault(k, [0, 0])
    t[0] += 1
    t[1] += e - s
nf = len(fr)
for k, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {tot / nf / 1e3:8.1f} us/frame {cnt / nf:6.1f} calls  {tot / cnt / 1e3:7.2f} us avg  {k}")
