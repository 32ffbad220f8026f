#!/usr/bin/env python3
"""Extract the golden vectors the reference's own tests hold for the hot path.

Run once in the build container (needs /root/reference); writes kat_reference.json next to
this script.  Only DATA is extracted (expected byte vectors and PreParam constants):
  * ResizeDouble / ResizeHalf truth        test/detect/kernel_test.cu:71-90
  * CopyMakeBorder truth                   test/detect/kernel_test.cu:125-139
  * PreParam goldens (bus / zidane)        test/detect/detector_test.cpp:38-41,57-67
  * Locator test calibration + constants   test/locate/locator_test.cpp:14-29,43-74
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_reference.json")


def truth_arrays(text):
    out = []
    for m in re.finditer(r"std::vector<unsigned char>\s+truth\{([^}]*)\}", text, re.S):
        out.append([int(v) for v in re.findall(r"\d+", m.group(1))])
    return out


def main():
    kt = open(os.path.join(REF, "test/detect/kernel_test.cu")).read()
    arrays = truth_arrays(kt)
    assert [len(a) for a in arrays] == [192, 12, 144], [len(a) for a in arrays]
    dt = open(os.path.join(REF, "test/detect/detector_test.cpp")).read()
    # ASSERT_FLOAT_EQ(pparam.height, 1080) ... in file order: bus(single), bus, zidane (multi)
    vals = re.findall(r"ASSERT_FLOAT_EQ\((\w+)\.(\w+), (\d+)\)", dt)
    pre = {}
    for var, field, v in vals:
        key = "zidane" if "zidane" in var else "bus"
        pre.setdefault(key, {})[field] = int(v)
    kat = {
        "source": "zmsbruce/rm_radar test/detect/kernel_test.cu, test/detect/detector_test.cpp, "
                  "test/locate/locator_test.cpp",
        "ramp_src": {"w": 4, "h": 4, "c": 3, "note": "bytes 0..47 (kernel_test.cu:30-33)"},
        "resize_double": {"dst_w": 8, "dst_h": 8, "truth": arrays[0]},
        "resize_half": {"dst_w": 2, "dst_h": 2, "truth": arrays[1]},
        "copy_make_border": {"top": 2, "bottom": 2, "left": 1, "right": 1, "truth": arrays[2]},
        "blob": {"scale": 0.01, "note": "== cv::dnn::blobFromImage(src, scale, Size(), Scalar(), "
                                        "swapRB=true): planar f32 RGB, u8*scale (kernel_test.cu:141-173)"},
        "transpose": {"rows": 2, "cols": 36, "note": "f32 ramp 0..71 == cv::transpose (kernel_test.cu:175-205)"},
        "preparam": pre,
        "locator_test": {
            "image_width": 640, "image_height": 480, "zoom_factor": 0.5, "queue_size": 5,
            "min_depth_diff": 0.05, "max_depth_diff": 5.0, "cluster_tolerance": 100.0,
            "min_cluster_size": 10, "max_cluster_size": 1000, "max_distance": 20.0,
            "zoom_rect": [100, 100, 50, 50],
            "transform_point": [1.0, 2.0, 3.0],
            "search_rect": [140, 100, 40, 40],
            "blob1": {"x": [160.0, 10.0], "y": [120.0, 10.0], "depth": [5.0, 6.0]},
            "blob2": {"x": [80.0, 10.0], "y": [60.0, 10.0], "depth": [1.0, 2.0]},
            "points_per_blob": 500, "expect_clusters": 2,
        },
    }
    assert kat["preparam"]["bus"] == {"height": 1080, "width": 810, "dw": 80, "dh": 0}
    assert kat["preparam"]["zidane"] == {"height": 720, "width": 1280, "dw": 0, "dh": 140}
    json.dump(kat, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
