#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (rocpd db): per kernel name, mean of each counter."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {k: i for i, k in enumerate(cols)}
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    agg[r[ix['kernel_name']] if 'kernel_name' in ix else r[ix['name']]][r[ix['counter_name']]].append(r[ix['value']])
for kname, d in agg.items():
    if len(sys.argv) > 2 and sys.argv[2] not in kname:
        continue
    print(kname[:110])
    for cn, vals in sorted(d.items()):
        print(f"    {cn:32s} mean {sum(vals) / len(vals):16.1f}  n={len(vals)}")
