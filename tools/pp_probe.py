import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import rm_radar_amd as rmr, scenes
d = "/tmp/rmr_packs"
img = scenes.synthetic_image(0)
for name, nc, th in (("car_lat", 1, 0.25), ("armor_lat", 12, 0.5)):
    det = rmr.Detector(f"{d}/{name}.rmrw", nc, (640, 640), 4, conf_thresh=th)
    out, pps = det.infer([img] * 4)
    best = out[:, 4:].max(1)
    print(name, "candidates per image", (best >= th).sum(1), "max", best.max())
    # direct postprocess timing through the C-ABI on device-resident input
    with rmr.profile(0) as prof:
        for _ in range(50):
            dets = det.detect([img] * 4)
        st = prof.read()
    for k, v in st.items():
        if k in ("postprocess", "head_decode", "letterbox"):
            print("   ", k, v["total_ms"] / v["launches"] * 1e3, "us/launch")
    print("   dets", [len(x) for x in dets])
