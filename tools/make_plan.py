#!/usr/bin/env python3
"""Writes the committed kernel plans (profiles/plans/yolov8m_{car,armor}_{f16,fp8}.tune) on an MI355X: the autotuner's
choice for every layer at the batch sizes the bench and the profiling tools launch -- 64-image car chunk + 256-image armor
chunk (BASELINE configs[2] / [3]), 1 + 4 images (batch-1 latency), 256 + 256 (configs[4], fp8).  bench.py, tools/round_profile.sh
and tools/pmc_refresh.sh then run under RMR_PLAN with these files, so that the driver's line, the rocprofv3 kernel stats and
the PMC traffic passes describe the same launches (VERDICT r03 "Pin the plan for every artefact").

usage (GPU box):  python tools/make_plan.py [f16] [fp8]      -> gpurun_out/plans/*.tune  (copy into profiles/plans/)"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.pop("RMR_PLAN", None)
os.environ.setdefault("RMR_TUNE_ROUNDS", "6")   # a committed plan deserves a longer run-off than a first call in the field
import numpy as np  # noqa: E402

import bench  # noqa: E402
import rm_radar_amd as rmr  # noqa: E402
import scenes  # noqa: E402
from rm_radar_amd import weights as W  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "plans")
os.makedirs(out, exist_ok=True)
for dtype in (sys.argv[1:] or ["f16", "fp8"]):
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rmr_plan_{dtype}")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    packs = (os.path.join(d, "car.rmrw"), os.path.join(d, "armor.rmrw"))
    W.make_synthetic_pack(packs[0], "m", 1, seed=1, cls_bias=-6.0)
    W.make_synthetic_pack(packs[1], "m", 12, seed=2, cls_bias=-6.0)
    merged = [{}, {}]   # (op, n) -> choice per pack; a detector re-writes its cache with the entries of ITS batch sizes only
    headers = [None, None]
    for batch in ((64, 1) if dtype == "f16" else (256, 64, 1)):
        args = bench.parse(["--batch", str(batch), "--dtype", dtype])
        images, clouds, rects = bench.make_inputs(args, 0)
        for pk in packs:
            if os.path.exists(pk + ".tune"):
                os.remove(pk + ".tune")
        rdet = rmr.RobotDetector(packs[0], packs[1], (640, 640), 12, max_cars=4, opt_cars=4, max_frames=batch, precision=dtype)
        loc = rmr.Locator(640, 640, scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32), max_frames=batch)
        for _ in range(2):
            rmr.run_batch(rdet, loc, list(images), list(clouds), np.ascontiguousarray(rects, np.int32))
        rdet.close()
        loc.close()
        for i, pk in enumerate(packs):
            lines = open(pk + ".tune").read().splitlines()
            assert headers[i] in (None, lines[0]), "the plan signature changed between batch sizes"
            headers[i] = lines[0]
            for ln in lines[1:]:
                op, n, choice = (int(v) for v in ln.split())
                merged[i][(op, n)] = choice
    for i, f in enumerate(bench.plan_files(args, out)):
        with open(f, "w") as fh:
            fh.write(headers[i] + "\n")
            for (op, n), choice in sorted(merged[i].items()):
                fh.write(f"{op} {n} {choice}\n")
        print(f, len(merged[i]), "entries, batch sizes", sorted({n for _, n in merged[i]}))
