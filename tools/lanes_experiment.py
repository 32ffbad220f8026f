#!/usr/bin/env python3
"""Does the GPU run faster when a step's frames go through L independent detector lanes (own streams, B / L frames
each) than through one?  Kernels of different lanes overlap, so the tail round of one layer's tiles (1.56 rounds
of 256 x 192 tiles on the 40 x 40 level at 64 images) is filled by the other lane's tiles.
usage: lanes_experiment.py [lanes ...]   (default 1 2 4)
RMR_LANES_FULL=1: every lane runs the WHOLE batch (L batches in flight, same launch sizes as one lane): the upper bound of what
overlapping one step's car stage with the previous step's armor stage could give."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

sys.argv = sys.argv[:1]
args = bench.parse()
import torch  # noqa: E402

import rm_radar_amd as rmr  # noqa: E402
import scenes  # noqa: E402
from rm_radar_amd import weights as W  # noqa: E402

lanes_list = [int(v) for v in sys.orig_argv[2:]] if len(sys.orig_argv) > 2 else [1, 2, 4]
dev = torch.device("cuda", 0)
pack_dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rmr_packs")
os.makedirs(pack_dir, exist_ok=True)
images, clouds, rects = bench.make_inputs(args, 0)
size = bench.frame_size(args)
d_images = torch.from_numpy(images).to(dev)
d_clouds = torch.from_numpy(clouds).to(dev)
B, K = args.batch, args.crops
forced_all = np.ascontiguousarray(np.asarray(rects, np.int32).reshape(B, -1, 4))
FULL = os.environ.get("RMR_LANES_FULL", "0") not in ("", "0")
for L in lanes_list:
    n = B if FULL else B // L
    lanes = []
    for l in range(L):
        packs = (os.path.join(pack_dir, f"car_l{L}_{l}.rmrw"), os.path.join(pack_dir, f"armor_l{L}_{l}.rmrw"))
        W.make_synthetic_pack(packs[0], "m", 1, seed=1, cls_bias=-6.0)
        W.make_synthetic_pack(packs[1], "m", 12, seed=2, cls_bias=-6.0)
        rdet = rmr.RobotDetector(packs[0], packs[1], size, 12, max_cars=K, opt_cars=K, device=0, max_frames=n)
        loc = rmr.Locator(size[0], size[1], bench.intrinsic(args), scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32), device=0, max_frames=n)
        lo = 0 if FULL else l * n
        fb = rmr.FrameBatch([d_images[f] for f in range(lo, lo + n)], [d_clouds[f] for f in range(lo, lo + n)])
        lanes.append((rdet, loc, fb, np.ascontiguousarray(forced_all[lo:lo + n])))
    pool = ThreadPoolExecutor(max_workers=L)

    def step():
        futs = [pool.submit(rmr.run_batch, rdet, loc, fb, None, fc) for rdet, loc, fb, fc in lanes]
        return [f.result() for f in futs]

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 6.0:
        step()
        steps += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fr = B * (L if FULL else 1)
    print(f"lanes {L}{' (full batches)' if FULL else ''}: {steps * fr / dt:8.1f} frames/s  ({dt / steps * 1e3:.2f} ms per {fr}-frame step)", flush=True)
    for rdet, loc, fb, fc in lanes:
        rdet.close()
        loc.close()
    pool.shutdown()
