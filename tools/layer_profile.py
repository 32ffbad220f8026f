#!/usr/bin/env python3
"""Per-GEMM-shape timing of one YOLOv8m forward (RMR_PROFILE_LAYERS=1), HIP events.
usage: python tools/layer_profile.py [batch] [nc]"""
import os
import sys

os.environ["RMR_PROFILE_LAYERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

import rm_radar_amd as rmr  # noqa: E402
import scenes  # noqa: E402
from rm_radar_amd import weights as W  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 12
reps = 3
path = f"/tmp/lp_{nc}.rmrw"
W.make_synthetic_pack(path, "m", nc, seed=1, cls_bias=-6.0)
import bench  # noqa: E402
plan = bench.apply_plan(bench.parse([]), (path,), ("armor" if nc == 12 else "car",)) if os.environ.get("RMR_LAYER_PLAN", "1") != "0" else None
import torch  # noqa: E402
imgs = [torch.from_numpy(scenes.synthetic_image(i)).cuda() for i in range(batch)]
det = rmr.Detector(path, nc, (640, 640), batch)
try:
    det.detect(imgs)
except rmr.RmrError as e:   # the committed plan has no entry for this batch size: tune here
    if "pinned plan" not in str(e):
        raise
    det.close()
    os.environ.pop("RMR_PLAN", None)
    os.remove(path + ".tune")
    plan = None
    det = rmr.Detector(path, nc, (640, 640), batch)
    det.detect(imgs)
with rmr.profile() as p:
    for _ in range(reps):
        det.detect(imgs)
    st = p.read()
tot_ms = sum(v["total_ms"] for v in st.values()) / reps
tot_fl = sum(v["flops"] for v in st.values()) / reps
print(f"kernel plan: {'pinned ' + plan[0] if plan else 'autotuned on this box'}")
print(f"batch {batch} nc {nc}: {tot_ms:.3f} ms/forward (sum of kernels), {tot_fl / tot_ms / 1e9:.1f} TFLOP/s overall")
rows = sorted(st.items(), key=lambda kv: -kv[1]["total_ms"])
print(f"{'kernel':44s} {'n':>4s} {'ms':>9s} {'%':>6s} {'TFLOP/s':>9s} {'GB/s':>8s}")
for k, v in rows:
    ms = v["total_ms"] / reps
    n = v["launches"] // reps
    tf = v["flops"] / v["total_ms"] / 1e9 if v["total_ms"] > 0 else 0
    gb = v["bytes"] / v["total_ms"] / 1e6 if v["total_ms"] > 0 else 0
    print(f"{k:44s} {n:4d} {ms:9.3f} {100 * ms / tot_ms:6.1f} {tf:9.1f} {gb:8.0f}")
