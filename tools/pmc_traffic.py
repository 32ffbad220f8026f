#!/usr/bin/env python3
"""HBM-side traffic of the convolution kernels from two rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes) -> profiles/rNN_pmc_conv_traffic.json.
usage: tools/pmc_traffic.py fetch.db write.db out.json "<command that was profiled>" """
import json
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (the kernel sources this measurement belongs to)


def per_launch(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ix = {k: i for i, k in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ix else "name"
    disp_col = next((k for k in ("dispatch_id", "dispatch_idx", "event_id", "id") if k in ix), None)
    per_dispatch = defaultdict(float)
    n_rows = 0
    for r in c.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != counter or "conv_" not in r[ix[name_col]]:
            continue
        key = r[ix[disp_col]] if disp_col else n_rows
        per_dispatch[key] += float(r[ix["value"]])  # instances (XCD / SE) of one dispatch add up
        n_rows += 1
    if not per_dispatch:
        sys.exit(f"{db}: no {counter} rows for conv kernels (columns: {cols})")
    return sum(per_dispatch.values()) / len(per_dispatch), len(per_dispatch), cols


fetch_kb, n_f, cols = per_launch(sys.argv[1], "FETCH_SIZE")
write_kb, n_w, _ = per_launch(sys.argv[2], "WRITE_SIZE")
out = {
    "round": 2,
    "source_hash": source_hash(),
    "kernel": "conv_* (all instantiations of conv_igemm / conv_dma / conv_halo / conv_t32 / conv_ws / conv_ws_s2 / conv_pw / conv_stem / conv_direct)",
    "command": sys.argv[4] if len(sys.argv) > 4 else "",
    "launches_counted": n_f,
    "FETCH_SIZE_kb_per_launch": fetch_kb,
    "WRITE_SIZE_kb_per_launch": write_kb,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, "
                  "HBM section) -> doubled; WRITE_SIZE uncalibrated, taken as is; counters tally L2->fabric requests, "
                  "Infinity-Cache hits included",
    "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
