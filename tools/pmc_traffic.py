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
    sym_of = {}
    n_rows = 0
    for r in c.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != counter or "conv_" not in r[ix[name_col]]:
            continue
        key = r[ix[disp_col]] if disp_col else n_rows
        per_dispatch[key] += float(r[ix["value"]])  # instances (XCD / SE) of one dispatch add up
        sym_of[key] = r[ix[name_col]]
        n_rows += 1
    if not per_dispatch:
        sys.exit(f"{db}: no {counter} rows for conv kernels (columns: {cols})")
    by_sym = defaultdict(lambda: [0.0, 0])
    for key, v in per_dispatch.items():
        e = by_sym[sym_of[key]]
        e[0] += v
        e[1] += 1
    # time per symbol from the kernel trace of the same run (--kernel-trace rides along with --pmc)
    dur = {}
    try:
        for name, n, tot in c.execute("select name, count(*), sum(end-start) from kernels group by name"):
            dur[name] = (n, tot)
    except sqlite3.Error:
        pass
    return sum(per_dispatch.values()) / len(per_dispatch), len(per_dispatch), cols, {k: (v[0] / v[1], v[1]) for k, v in by_sym.items()}, dur


fetch_kb, n_f, cols, fetch_sym, dur = per_launch(sys.argv[1], "FETCH_SIZE")
write_kb, n_w, _, write_sym, _ = per_launch(sys.argv[2], "WRITE_SIZE")
# the dominant kernel of the command = the conv symbol with the most time in the trace (bench.py's roofline names the same
# one from its HIP-event profile: a template instantiation is one symbol)
by_kernel = {}
for sym, (fkb, n) in fetch_sym.items():
    wkb = write_sym.get(sym, (0.0, 0))[0]
    n_tr, tot = dur.get(sym, (0, 0))
    by_kernel[sym] = {"launches": n, "traffic_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0,
                      "avg_us_under_pmc": round(tot / n_tr / 1e3, 2) if n_tr else None, "total_ms_under_pmc": round(tot / 1e6, 3)}
dominant = max(by_kernel.items(), key=lambda kv: kv[1]["total_ms_under_pmc"])[0] if by_kernel else None
out = {
    "round": 3,
    "dominant_kernel": dominant,
    "dominant_traffic_bytes_per_launch": by_kernel[dominant]["traffic_bytes_per_launch"] if dominant else None,
    "by_kernel": dict(sorted(by_kernel.items(), key=lambda kv: -kv[1]["total_ms_under_pmc"])[:16]),
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
