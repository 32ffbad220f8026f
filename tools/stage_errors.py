#!/usr/bin/env python3
"""Per-stage error of the HIP network against the f16-emulating and fp32 oracles (what the budget of
tests/test_gpu_network.py::test_stage_error_budget is set from).  usage: stage_errors.py [nc]"""
import os
import sys

os.environ["RMR_ARENA_REUSE"] = "0"  # stage outputs must survive the forward

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import netutil  # noqa: E402
import oracle  # noqa: E402
import rm_radar_amd as rmr  # noqa: E402
from oracle import yolov8_ref as R  # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
fp8 = "--fp8" in sys.argv
images = [netutil.test_image(1), netutil.test_image(2, 810, 1080)]
path = f"/tmp/stage_{nc}.rmrw"
netutil.tuned_pack(path, nc, 11, 0.25, 0.01, images)
det = rmr.Detector(path, nc, (1920, 1080), 2, precision="fp8" if fp8 else "f16")
det.infer(images)
blobs = np.stack([oracle.preprocess(im)[0] for im in images])
f16 = R.load(path, True, fp8=fp8).features(blobs)
f32 = R.load(path, False).features(blobs)
print(f"{'stage':10s} {'shape':>16s} {'rms':>8s} | vs f16-emulating: {'max':>9s} {'mean':>9s} {'max/rms':>8s} | vs fp32: {'max':>9s} {'mean':>9s} | f16 vs fp32 oracle: {'max':>9s}")
for name in f16:
    for img in range(2):
        got = det.read_feature(name, img)
        w16, w32 = f16[name][img], f32[name][img]
        rms = float(np.sqrt((w32 ** 2).mean()))
        e16, e32, eo = np.abs(got - w16), np.abs(got - w32), np.abs(w16 - w32)
        print(f"{name:10s} {str(got.shape):>16s} {rms:8.4f} | {e16.max():9.5f} {e16.mean():9.6f} {e16.max() / rms:8.4f} | {e32.max():9.5f} {e32.mean():9.6f} | {eo.max():9.5f}")
