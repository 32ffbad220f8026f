#!/usr/bin/env python3
"""Batch-1 latency of one frame (host inputs): locate + two-stage detect + search.  Prints p50 and a
coarse host-side phase breakdown."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import rm_radar_amd as rmr  # noqa: E402
import scenes  # noqa: E402
from rm_radar_amd import weights as W  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
d = "/tmp/rmr_packs"
os.makedirs(d, exist_ok=True)
car, armor = d + "/car_lat.rmrw", d + "/armor_lat.rmrw"
if not os.path.exists(car):
    W.make_synthetic_pack(car, "m", 1, seed=1, cls_bias=-6.0)
    W.make_synthetic_pack(armor, "m", 12, seed=2, cls_bias=-6.0)
import bench  # noqa: E402
plan = bench.apply_plan(bench.parse(sys.argv[3:4] and ["--plan", sys.argv[3]] or []), (car, armor))   # the committed pinned plan (profiles/plans/), when there is one; argv[3] = tune: autotune here
rng = np.random.default_rng(0)
img = scenes.synthetic_image(0)
cloud = scenes.make_cloud(rng, 30000, scenes.K640, scenes.SAMPLE_L2C, (640, 640), [((100, 300, 120, 90), 2000, 200)])
rects = [[(10 + 150 * i, 200, 120, 100) for i in range(K)]]
rd = rmr.RobotDetector(car, armor, (640, 640), 12, max_cars=max(K, 1), opt_cars=max(K, 1))
loc = rmr.Locator(640, 640, scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32))
try:
    rd.detect_batch([img], forced_crops=rects)
except rmr.RmrError as e:   # the committed plan does not cover this build or these batch sizes: tune here, and say so
    if "pinned plan" not in str(e):
        raise
    rd.close()
    os.environ.pop("RMR_PLAN", None)
    for pk in (car, armor):
        if os.path.exists(pk + ".tune"):
            os.remove(pk + ".tune")
    print("(the committed plan does not cover this build: autotuned here)")
    rd = rmr.RobotDetector(car, armor, (640, 640), 12, max_cars=max(K, 1), opt_cars=max(K, 1))
lat, ph = [], np.zeros(3)
for i in range(reps):
    t0 = time.perf_counter()
    loc.update(cloud)
    loc.cluster()
    t1 = time.perf_counter()
    rb = rd.detect_batch([img], forced_crops=rects)[0]
    t2 = time.perf_counter()
    loc.search(rb)
    t3 = time.perf_counter()
    if i >= 10:
        lat.append((t3 - t0) * 1e3)
        ph += [t1 - t0, t2 - t1, t3 - t2]
lat = np.array(lat)
print(f"K={K}: p50 {np.percentile(lat, 50):.3f} ms  p99 {np.percentile(lat, 99):.3f} ms  "
      f"phases ms: locate-enqueue {ph[0] / len(lat) * 1e3:.3f} detect {ph[1] / len(lat) * 1e3:.3f} "
      f"search {ph[2] / len(lat) * 1e3:.3f}")
if os.environ.get("RMR_PROBE_PROFILE"):
    # per-kernel-family HIP-event times of the same frame (events add a little launch overhead)
    with rmr.profile(0) as prof:
        for i in range(20):
            loc.update(cloud)
            loc.cluster()
            rb = rd.detect_batch([img], forced_crops=rects)[0]
            loc.search(rb)
        st = prof.read()
    tot = 0.0
    for name, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
        tot += v["total_ms"] / 20
        print(f"  {name:28s} {v['launches'] // 20:4d} launches/frame  {v['total_ms'] / 20 * 1e3:8.1f} us/frame  "
              f"{v['total_ms'] / max(v['launches'], 1) * 1e3:7.1f} us/launch")
    print(f"  sum {tot * 1e3:.1f} us/frame")
