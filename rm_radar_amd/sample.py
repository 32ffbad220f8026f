"""Headless counterpart of the reference's sample application (samples/sample_radar.h:41-127):
the caller that defines the order of the hot path, with the tracker stage behind it.  No GUI
(`visualize`, sample_radar.h:160-281, is out of scope).

    radar = SampleRadar(car_pack, armor_pack, image_size, K, lidar_to_camera, world_to_camera,
                        lidar_noise=(0.1, 0.1, 0.1))
    radar.update_background_cloud(background)          # sample_radar.h:94-97, main.cpp:87
    robots = radar.run_once(image, cloud, timestamp)   # sample_radar.h:106-127
"""
from __future__ import annotations

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
                 world_to_camera, lidar_noise=None, device=0, **locator_kwargs):
        # sample_radar.h:73-92: RobotDetector(car, armor, image_size, kClassNum, kMaxBatchSize,
        # kOptBatchSize) and Locator(image_size.width, image_size.height, K, L2C, W2C)
        self.detector = RobotDetector(car_engine_path, armor_engine_path, image_size, K_CLASS_NUM,
                                      K_MAX_BATCH_SIZE, K_OPT_BATCH_SIZE, device=device)
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
