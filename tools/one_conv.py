#!/usr/bin/env python3
"""Run one conv layer a few times (for rocprofv3 --pmc).  usage: one_conv.py n h w cin cout k stride [tile]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import rm_radar_amd as rmr  # noqa: E402

n, h, w, cin, cout, k, stride = [int(v) for v in sys.argv[1:8]]
tile = int(sys.argv[8]) if len(sys.argv) > 8 else -1
rng = np.random.default_rng(0)
x = rng.normal(0, 1, (n, h, w, cin)).astype(np.float32)
wt = (rng.normal(0, 1, (cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
b = rng.normal(0, 0.5, cout).astype(np.float32)
for _ in range(3):
    rmr.conv2d(x, wt, b, stride, k // 2, True, tile=tile)
print("done")
