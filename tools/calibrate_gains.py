#!/usr/bin/env python3
"""Dev-time tool: derive the per-layer gain table baked into rm_radar_amd/weights.py.

LSUV-style sequential calibration on the torch CPU oracle: walk the convs in execution order and
scale each one so its post-activation output has unit std on seeded uniform-noise 640x640 input.
The resulting gains (relative to var(w) = 1/fan_in) are printed as a Python dict.  Deterministic
packs are then generated from the table with numpy only (no torch at generation time)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import yolov8_ref as R  # noqa: E402
from rm_radar_amd import weights as W  # noqa: E402

scale, nc = "m", 12
tensors = W.synthesize(scale, nc, seed=0, gains={})
meta = dict(scale=scale, nc=nc)
ref = R.YoloV8Ref(tensors, meta)
gains = {}
orig = ref.conv
target_act, target_dfl, target_cls = 1.0, 1.5, 1.5


def conv(name, x, k, s=1, act=True, residual=None, keep_f32=False):
    w = ref.t[name + ".weight"]
    y = orig(name, x, k, s, act, None, keep_f32)
    tgt = target_act if act else (target_dfl if ".cv2." in name else target_cls)
    g = 1.0
    for _ in range(6):
        sd = float((y - (0 if act else ref.t[name + ".bias"].view(1, -1, 1, 1))).std())
        f = tgt / sd
        g *= f
        w.mul_(f)
        y = orig(name, x, k, s, act, None, keep_f32)
        if abs(f - 1) < 1e-3:
            break
    gains[name] = g * g * gains_base(name)
    if residual is not None:
        y = y + residual
    return y


def gains_base(name):
    return W.DEFAULT_GAIN


ref.conv = conv
rng = np.random.default_rng(1234)
blob = rng.integers(0, 256, (2, 3, 640, 640)).astype(np.float32) / 255.0
ref.forward(blob)
print("{")
for k, v in gains.items():
    print(f'    "{k}": {v:.4f},')
print("}")
