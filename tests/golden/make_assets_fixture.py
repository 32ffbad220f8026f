#!/usr/bin/env python3
"""Convert the reference's sample clouds (assets/clouds/{0..9}.pcd, ASCII PCD v0.7, fields x y z,
f32, millimetres, 10 000 points each; MIT licence) into tests/golden/assets_clouds.npz.
DATA only.  Run in the build container (needs /root/reference)."""
import os

import numpy as np

REF = "/root/reference/assets/clouds"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets_clouds.npz")

out = {}
for i in range(10):
    lines = open(f"{REF}/{i}.pcd").read().split("\n")
    k = lines.index("DATA ascii")
    pts = np.array([[float(v) for v in l.split()] for l in lines[k + 1:] if l.strip()], np.float32)
    assert pts.shape == (10000, 3)
    assert np.all(pts == np.rint(pts))  # integer millimetres: store losslessly as int32
    out[f"cloud{i}"] = pts.astype(np.int32)
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT))
