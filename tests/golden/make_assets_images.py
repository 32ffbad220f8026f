#!/usr/bin/env python3
"""Decode the reference's sample frames (assets/images/{0..9}.jpg, 2592 x 2048, MIT licence) and store them
down-scaled 4x (648 x 512, Lanczos) as JPEG quality 90 under tests/golden/assets_images/ -- ~0.1 MB each.
DATA only (pixels).  Run in the build container (needs /root/reference and PIL).  The tests decode them with
PIL and scale the calibration of samples/main.cpp:12-22 by the same factor."""
import os

from PIL import Image

REF = "/root/reference/assets/images"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets_images")
os.makedirs(OUT, exist_ok=True)
total = 0
for i in range(10):
    im = Image.open(f"{REF}/{i}.jpg").convert("RGB")
    assert im.size == (2592, 2048)
    im = im.resize((648, 512), Image.LANCZOS)
    path = os.path.join(OUT, f"{i}.jpg")
    im.save(path, quality=90)
    total += os.path.getsize(path)
print("wrote", OUT, total, "bytes")
# ... and frames at their own size (2592 x 2048, re-encoded at quality 80: ~0.5 MB each), so that configs[1] also runs on
# reference frames at the reference's resolution and calibration (the letterbox of a 2592 x 2048 frame has the Q2 row quirk)
# (round 4: three frames -- 0, 4 and 9 -- so that the full-size case is a SEQUENCE: the Locator's background and depth ring
# carry over from frame to frame, as in the sample application)
for i in (0, 4, 9):
    im = Image.open(f"{REF}/{i}.jpg").convert("RGB")
    path = os.path.join(OUT, f"full_{i}.jpg")
    im.save(path, quality=80)
    print("wrote", path, os.path.getsize(path), "bytes")
