"""Headless counterpart of the reference's sample application (samples/sample_radar.h:41-127):
the caller that defines the order of the hot path, with the tracker stage behind it.  No GUI
(`visualize`, sample_radar.h:160-281, is out of scope).

    radar = SampleRadar(car_pack, armor_pack, image_size, K, lidar_to_camera, world_to_camera,
                        lidar_noise=(0.1, 0.1, 0.1))
    radar.update_background_cloud(background)          # sample_radar.h:94-97, main.cpp:87
    robots = radar.run_once(image, cloud, timestamp)   # sample_radar.h:106-127
"""
from __future__ import annotations

import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from typing import List

from . import Locator, Robot, RobotDetector, Tracker

# sample_radar.h:32-34
K_CLASS_NUM = 12
K_MAX_BATCH_SIZE = 20
K_OPT_BATCH_SIZE = 4


class SampleRadar:
    def __init__(self, car_engine_path, armor_engine_path, image_size, intrinsic, lidar_to_camera,
                 world_to_camera, lidar_noise=None, device=0, detector_kwargs=None, **locator_kwargs):
        # sample_radar.h:73-92: RobotDetector(car, armor, image_size, kClassNum, kMaxBatchSize,
        # kOptBatchSize) and Locator(image_size.width, image_size.height, K, L2C, W2C)
        self.detector = RobotDetector(car_engine_path, armor_engine_path, image_size, K_CLASS_NUM,
                                      K_MAX_BATCH_SIZE, K_OPT_BATCH_SIZE, device=device, **(detector_kwargs or {}))
        self.locator = Locator(image_size[0], image_size[1], intrinsic, lidar_to_camera,
                               world_to_camera, device=device, **locator_kwargs)
        # sample_radar.h:68: Tracker(lidar_noise, kClassNum); None = stop after search()
        self.tracker = None if lidar_noise is None else Tracker(lidar_noise, K_CLASS_NUM)
        self._pool = ThreadPoolExecutor(max_workers=2)

    def update_background_cloud(self, cloud) -> None:
        """sample_radar.h:94-97: the background cloud goes through the same update() (Q21)."""
        self.locator.update(cloud)

    def run_once(self, image, cloud, timestamp=None) -> List[Robot]:
        """sample_radar.h:106-127: update+cluster on one thread while detect runs on another,
        join, search, then the tracker (timestamp in seconds or integer nanoseconds; now if None)."""
        def locate():
            self.locator.update(cloud)
            self.locator.cluster()

        fa = self._pool.submit(locate)
        fb = self._pool.submit(self.detector.detect, image)
        fa.result()
        robots = fb.result()
        self.locator.search(robots)
        if self.tracker is not None:
            self.tracker.update(robots, time.time_ns() if timestamp is None else timestamp)
        return robots

    def close(self):
        self._pool.shutdown(wait=True)
        self.detector.close()
        self.locator.close()
        if self.tracker is not None:
            self.tracker.close()


# ------------------------------------------------------------------------------- headless CLI
# samples/main.cpp: the reference's constants and call order, printing instead of imshow.

MAIN_IMAGE_SIZE = (2592, 2048)                                              # main.cpp:12
MAIN_INTRINSIC = [[1685.51538398561, 0, 1278.99324114319], [0, 1685.26471848220, 1037.21273138299], [0, 0, 1]]
MAIN_LIDAR_TO_CAMERA = [[0, -1, 0, 0.85443], [0, 0, -1, -37.6845], [1, 0, 0, 12.2631], [0, 0, 0, 1]]
MAIN_WORLD_TO_CAMERA = [[0.05975021, 0.99807031, 0.01689906, -7179.65399136],
                        [0.28962566, -0.00113262, -0.95713933, -4671.34956587],
                        [-0.9552732, 0.06208368, -0.28913445, 28286.8920291], [0, 0, 0, 1]]
MAIN_LIDAR_NOISE = (0.4, 0.4, 0.4)                                          # main.cpp:22


def main(argv=None) -> int:
    import argparse

    import numpy as np

    from . import assets

    ap = argparse.ArgumentParser(prog="python -m rm_radar_amd.sample",
                                 description="headless counterpart of the reference's sample (samples/main.cpp)")
    ap.add_argument("--models", default="../models", help="folder with car.rmrw / armor.rmrw (or car.onnx / armor.onnx)")
    ap.add_argument("--assets", default="../assets", help="folder with images/<i>.jpg|.npy and clouds/<i>.pcd [+ background.pcd]")
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)

    radar = SampleRadar(os.path.join(args.models, "car.rmrw"), os.path.join(args.models, "armor.rmrw"), MAIN_IMAGE_SIZE,
                        np.array(MAIN_INTRINSIC, np.float32), np.array(MAIN_LIDAR_TO_CAMERA, np.float32),
                        np.array(MAIN_WORLD_TO_CAMERA, np.float32), lidar_noise=MAIN_LIDAR_NOISE, device=args.device)
    img_dir, cloud_dir = os.path.join(args.assets, "images"), os.path.join(args.assets, "clouds")
    background = os.path.join(cloud_dir, "background.pcd")
    if os.path.exists(background):
        radar.update_background_cloud(assets.read_pcd(background))               # main.cpp:87
    else:
        print(f"warning: {background} does not exist; the first frames build the background", file=sys.stderr)
    start_ns = time.time_ns()
    for i in range(args.frames):
        ip = assets.find_frame(img_dir, i, (".jpg", ".png", ".npy"))
        cp = assets.find_frame(cloud_dir, i, (".pcd",))
        if ip is None or cp is None:
            raise FileNotFoundError(f"frame {i}: image or cloud missing under {args.assets}")  # main.cpp:33-35,55-57
        robots = radar.run_once(assets.read_image(ip), assets.read_pcd(cp), start_ns + i * 100_000_000)  # main.cpp:85
        print(f"frame {i}: {len(robots)} robot(s)")
        for r in robots:
            loc = "None" if r.location is None else "[%.3f, %.3f, %.3f]" % tuple(r.location)
            state = {None: "None", 1: "Tentative", 2: "Confirmed", 3: "Deleted"}[r.track_state]
            print(f"  label {r.label} rect {[round(v, 1) for v in r.rect]} confidence {r.confidence} state {state} location {loc}")
    radar.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
