#!/usr/bin/env python3
"""How close can two exact f16 implementations of a YOLOv8 of another size be on seeded, uncalibrated weights?  For the m and l
packs of the tests: the engine against the f16-emulating oracle, beside that oracle against itself with every convolution result
moved by 2^-22 of its value before its f16 rounding (another f32 summation order), and both against f32.  The bar of
tests/test_gpu_network.py::test_other_sizes_of_the_family_match_the_oracle comes from here.
usage (GPU box): python tools/size_probe.py   -> stdout (profiles/r05_size_probe.txt)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import netutil, oracle
import rm_radar_amd as rmr
from oracle import yolov8_ref as R
from rm_radar_amd import weights as W
images = [netutil.test_image(1), netutil.test_image(2, 810, 1080), netutil.test_image(3, 1280, 720)]
blobs = np.stack([oracle.preprocess(im)[0] for im in images])
for scale in ("s", "m", "l"):
    pack = W.make_synthetic_pack(f"/tmp/{scale}.rmrw", scale, 12, seed=21, cls_bias=-4.0)
    A = R.load(pack, True).forward(blobs)
    A1 = R.load(pack, True, jitter=2.0 ** -22, jitter_seed=1).forward(blobs)
    A32 = R.load(pack, False).forward(blobs)
    det = rmr.Detector(pack, 12, (2592, 2048), 3, conf_thresh=0.5)
    E, _ = det.infer(images)
    det.close()
    def d(a, b):
        x = np.abs(a[:, :4] - b[:, :4]); s = np.abs(a[:, 4:] - b[:, 4:])
        return f"box max {x.max():.3f} mean {x.mean():.4f} p99.9 {np.quantile(x, 0.999):.3f} | score max {s.max():.5f}"
    print(scale, "engine vs f16 oracle   ", d(E, A))
    print(scale, "f16 oracle vs jittered ", d(A, A1))
    print(scale, "engine vs jittered     ", d(E, A1))
    print(scale, "f16 oracle vs f32      ", d(A, A32))
    print(scale, "engine vs f32          ", d(E, A32), flush=True)
