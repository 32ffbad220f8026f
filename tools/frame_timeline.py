#!/usr/bin/env python3
"""Host-side timeline of the batch-1 frame bench.py times (rmr.run_batch with one frame, host inputs): the medians of the phases
api_pipeline.cpp prints under RMR_STEP_TIMING=1, beside the p50 of the whole call.
usage (GPU box): python tools/frame_timeline.py [frames=200]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("RMR_STEP_TIMING") != "1":   # the switch is read once by the library: run the measurement in a child with it set
    env = dict(os.environ, RMR_STEP_TIMING="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
    sys.stdout.write(p.stdout)
    rows = [[int(v) for v in re.findall(r"(-?\d+)", l.split("]", 1)[1])] for l in p.stderr.splitlines() if l.startswith("[rmr step]")]
    import numpy as np
    a = np.array(rows[len(rows) // 4:], dtype=np.float64)
    names = ["between calls", "entry -> cars known, armor stage enqueued (H2D, car stage, D2H)", "after_cars (search enqueue, under the armor stage)", "rest of the armor stage + assembly", "search end", "merge",
             "  of the first: frame staged + car stage enqueued", "  locator enqueued (under the car stage)", "  wait for the car stage + heads + crops + armor enqueue"]
    for i, n in enumerate(names):
        print(f"  {n:60s} median {np.median(a[:, i]):8.1f} us   p90 {np.percentile(a[:, i], 90):8.1f} us")
    print(f"  sum of the medians inside a call: {np.median(a[:, 1:6], axis=0).sum():.1f} us over {len(a)} frames")
    sys.exit(p.returncode)

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import time  # noqa: E402

import numpy as np  # noqa: E402

import bench  # noqa: E402
import rm_radar_amd as rmr  # noqa: E402
import scenes  # noqa: E402
from rm_radar_amd import weights as W  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d = "/tmp/rmr_packs"
os.makedirs(d, exist_ok=True)
car, armor = d + "/car_lat.rmrw", d + "/armor_lat.rmrw"
if not os.path.exists(car):
    W.make_synthetic_pack(car, "m", 1, seed=1, cls_bias=-6.0)
    W.make_synthetic_pack(armor, "m", 12, seed=2, cls_bias=-6.0)
bench.apply_plan(bench.parse([]), (car, armor))
rng = np.random.default_rng(0)
img = scenes.synthetic_image(0)
cloud = scenes.make_cloud(rng, 30000, scenes.K640, scenes.SAMPLE_L2C, (640, 640), [((100, 300, 120, 90), 2000, 200)])
rects = [(10 + 150 * i, 200, 120, 100) for i in range(4)]
rd = rmr.RobotDetector(car, armor, (640, 640), 12, max_cars=4, opt_cars=4)
loc = rmr.Locator(640, 640, scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32), max_frames=1)
lat = []
for i in range(reps + 20):
    t0 = time.perf_counter()
    rmr.run_batch(rd, loc, [img], [cloud], [rects])
    lat.append((time.perf_counter() - t0) * 1e3)
lat = np.array(lat[20:])
print(f"rmr.run_batch, one frame, host inputs: p50 {np.percentile(lat, 50):.3f} ms  p99 {np.percentile(lat, 99):.3f} ms over {len(lat)} frames")
