#!/usr/bin/env python3
"""HBM-side traffic of the convolution kernels from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected
separately, as MI355X_MICROARCH.md prescribes) -> profiles/rNN_pmc_conv_traffic.json.

Round 4: per LAYER, not only per symbol.  The passes run bench.py under the committed pinned plan (profiles/plans/), so
every step launches the same convolution kernels in the same order; that order -- layer name, algorithmic bytes and FLOPs of
each launch -- is written by the library's profiler (RMR_PROFILE_ORDER=<file>, bench.py --launch-order) during the
bench's own profiled step.  The k-th conv dispatch of a PMC pass is therefore launch k mod L of the step, and its bytes
belong to that layer.  The mapping is checked: the kernel symbol at a step position must be the same in every step.

usage: tools/pmc_traffic.py fetch.db write.db out.json "<command that was profiled>" [launch_order.txt]"""
import json
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import instantiation_of, source_hash  # noqa: E402  (the kernel sources this measurement belongs to)


def dispatches(db, counter):
    """conv dispatches of a pass in dispatch order: [(symbol, counter value summed over its instances)], + time per symbol"""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    ix = {k: i for i, k in enumerate(cols)}
    name_col = "kernel_name" if "kernel_name" in ix else "name"
    disp_col = next((k for k in ("dispatch_id", "dispatch_idx", "event_id", "id") if k in ix), None)
    if disp_col is None:
        sys.exit(f"{db}: no dispatch column in {cols}")
    val, sym = defaultdict(float), {}
    for r in c.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != counter or "conv_" not in r[ix[name_col]]:
            continue
        val[r[ix[disp_col]]] += float(r[ix["value"]])  # instances (XCD / SE) of one dispatch add up
        sym[r[ix[disp_col]]] = r[ix[name_col]]
    if not val:
        sys.exit(f"{db}: no {counter} rows for conv kernels (columns: {cols})")
    dur = {}
    try:  # time per symbol from the kernel trace of the same run (--kernel-trace rides along with --pmc)
        for name, n, tot in c.execute("select name, count(*), sum(end-start) from kernels group by name"):
            dur[name] = (n, tot)
    except sqlite3.Error:
        pass
    return [(sym[k], val[k]) for k in sorted(val)], dur


def step_order(path):
    """One step's conv launches in enqueue order from the profiler's log: the flops-only profiled steps (level 2) of
    bench.py repeat the same sequence; returns [(stage, layer, flops, bytes)] of one period."""
    rows = []
    for line in open(path):
        level, rest = line.rstrip("\n").split(" ", 1)
        stage, name, flops, nbytes = rest.split("|")[:4]
        # (the head's fused tail, tag y0, declares FLOPs but is not a conv_* symbol: dispatches() does not list it either)
        if level == "2" and float(flops) > 0 and not name.endswith(" y0"):
            rows.append((stage, name, float(flops), float(nbytes)))
    names = [r[1] for r in rows]
    for period in range(1, len(rows) + 1):
        if len(rows) % period == 0 and all(names[i] == names[i % period] for i in range(len(rows))):
            return rows[:period]
    return rows


def main():
    fetch, dur = dispatches(sys.argv[1], "FETCH_SIZE")
    write, _ = dispatches(sys.argv[2], "WRITE_SIZE")
    KB = 1024.0
    order = step_order(sys.argv[5]) if len(sys.argv) > 5 and os.path.exists(sys.argv[5]) else None

    by_sym = defaultdict(lambda: [0.0, 0.0, 0])
    for (s, f) in fetch:
        by_sym[s][0] += f
        by_sym[s][2] += 1
    for (s, w) in write:
        by_sym[s][1] += w
    by_kernel = {}
    for s, (f, w, n) in by_sym.items():
        n_tr, tot = dur.get(s, (0, 0))
        by_kernel[s] = {"launches": n, "traffic_bytes_per_launch": (2.0 * f + w) * KB / n,
                        "avg_us_under_pmc": round(tot / n_tr / 1e3, 2) if n_tr else None, "total_ms_under_pmc": round(tot / 1e6, 3)}
    dominant = max(by_kernel.items(), key=lambda kv: kv[1]["total_ms_under_pmc"])[0] if by_kernel else None

    by_layer, by_inst, mapping = None, None, "no launch-order file: per-symbol figures only"
    if order:
        L = len(order)
        ok = len(fetch) % L == 0 and len(write) == len(fetch)
        sym_at = {}
        for i, (s, _) in enumerate(fetch):
            ok = ok and sym_at.setdefault(i % L, s) == s
        for i, (s, _) in enumerate(write):
            ok = ok and sym_at.get(i % L) == s
        if ok:
            steps = len(fetch) // L
            acc = defaultdict(lambda: {"launches_per_step": 0, "fetch_kb": 0.0, "write_kb": 0.0, "alg_bytes": 0.0, "flops": 0.0, "symbol": None})
            for i in range(len(fetch)):
                stage, name, flops, nbytes = order[i % L]
                e = acc[name]
                e["fetch_kb"] += fetch[i][1]
                e["write_kb"] += write[i][1]
                e["symbol"] = fetch[i][0]
                if i < L:
                    e["launches_per_step"] += 1
                    e["alg_bytes"] += nbytes
                    e["flops"] += flops
            by_layer = {}
            for name, e in acc.items():
                n = e["launches_per_step"] * steps
                t = (2.0 * e["fetch_kb"] + e["write_kb"]) * KB / n
                alg = e["alg_bytes"] / e["launches_per_step"]
                by_layer[name] = {"launches_per_step": e["launches_per_step"], "traffic_bytes_per_launch": round(t),
                                  "algorithmic_bytes_per_launch": round(alg), "traffic_over_algorithmic": round(t / alg, 3) if alg else None,
                                  "symbol": e["symbol"]}
            inst = defaultdict(lambda: [0.0, 0.0, 0])
            for name, v in by_layer.items():
                e = inst[instantiation_of(name)]
                e[0] += v["traffic_bytes_per_launch"] * v["launches_per_step"]
                e[1] += v["algorithmic_bytes_per_launch"] * v["launches_per_step"]
                e[2] += v["launches_per_step"]
            by_inst = {t: {"launches_per_step": n, "traffic_bytes_per_launch": round(tr / n), "algorithmic_bytes_per_launch": round(al / n),
                           "traffic_over_algorithmic": round(tr / al, 3) if al else None}
                       for t, (tr, al, n) in sorted(inst.items(), key=lambda kv: -kv[1][0])}
            by_layer = dict(sorted(by_layer.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches_per_step"]))
            mapping = f"{steps} steps x {L} conv launches, symbol at every step position identical across steps"
        else:
            mapping = (f"launch order has {L} conv launches per step, the passes {len(fetch)} / {len(write)} conv dispatches with differing symbols "
                       "at some step position: the passes did not run under a pinned plan -- per-symbol figures only")

    n = len(fetch)
    fetch_kb, write_kb = sum(v for _, v in fetch) / n, sum(v for _, v in write) / max(len(write), 1)
    out = {
        "round": 4,
        "dominant_kernel": dominant,
        "dominant_traffic_bytes_per_launch": by_kernel[dominant]["traffic_bytes_per_launch"] if dominant else None,
        "launch_mapping": mapping,
        "conv_launches_per_step": len(order) if order else None,
        "by_instantiation": by_inst,
        "by_layer": by_layer,
        "by_kernel": dict(sorted(by_kernel.items(), key=lambda kv: -kv[1]["total_ms_under_pmc"])[:16]),
        "source_hash": source_hash(),
        "kernel": "conv_* (every convolution launch of the step, all instantiations)",
        "command": sys.argv[4] if len(sys.argv) > 4 else "",
        "launches_counted": n,
        "FETCH_SIZE_kb_per_launch": fetch_kb,
        "WRITE_SIZE_kb_per_launch": write_kb,
        "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, "
                      "HBM section) -> doubled; WRITE_SIZE uncalibrated, taken as is; counters tally L2->fabric requests, "
                      "Infinity-Cache hits included",
        "traffic_bytes_per_launch": (2.0 * fetch_kb + write_kb) * KB,
    }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ("by_layer", "by_kernel")}))


if __name__ == "__main__":
    main()
