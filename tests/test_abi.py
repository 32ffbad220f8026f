"""CPU checks of the drop-in boundary: librmr.so builds, loads without a GPU and exports every
symbol include/rmr.h declares; the binding's struct layouts match the header; compute entry
points fail loudly (RMR_ERR_DEVICE) when no gfx950 device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from rm_radar_amd import _lib
    return _lib


def test_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "rmr.h")).read()
    declared = set(re.findall(r"\b(rmr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rmr_status"}
    L = built.lib()
    assert built.MISSING == []
    assert declared == set(built.SYMBOLS), declared ^ set(built.SYMBOLS)
    for name in declared:
        assert hasattr(L, name)
    assert L.rmr_abi_version() == 6   # 6: rmr_upload_* / rmr_pinned_*; 5: rmr_robot_detector_read_heads; 4: rmr_locator_update_cluster_batch; 3: precision in the detector cfgs, 64-byte kernel names


def test_struct_layouts(built):
    assert C.sizeof(built.Detection) == 24          # detection.h:62-67: six f32
    assert C.sizeof(built.PreParam) == 20
    assert C.sizeof(built.Robot) == 16 + 4 * 4 + 64 * 24 + 4 + 12 + 4   # + track_state (ABI 2)
    assert C.sizeof(built.RobotRecord) == 48
    assert built.DET_DTYPE.itemsize == 24


def test_host_only_entry_points_work_without_gpu(built, oracle):
    import rm_radar_amd as r
    p = r.preparam(810, 1080)
    assert (p.dw, p.dh) == (80.0, 0.0)
    assert r.letterbox_geometry(r.preparam(2592, 2048)) == (640, 505, 67, 0)
    d = r.restore_detection((100, 100, 50, 50, 0, 0.9), r.preparam(1280, 720))
    assert d[:4] == (200.0, 0.0, 100.0, 100.0)
    rng = np.random.default_rng(1)
    for _ in range(200):
        w, h = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
        det = tuple(float(v) for v in rng.uniform(-50, 700, 4)) + (1.0, 0.5)
        assert r.preparam(w, h).astuple() == oracle.preparam(w, h).astuple()
        assert r.restore_detection(det, r.preparam(w, h)) == oracle.restore(det, oracle.preparam(w, h))
        g = oracle.letterbox_geometry(oracle.preparam(w, h))
        assert r.letterbox_geometry(r.preparam(w, h)) == (g[0], g[1], g[2], g[4])


def test_host_robot_assembly_matches_oracle(built, oracle):
    import rm_radar_amd as r
    D = r.DET_DTYPE
    rng = np.random.default_rng(0)
    for trial in range(50):
        n_cars = int(rng.integers(0, 7))
        got_in, want_in = [], []
        for _ in range(n_cars):
            car = (float(rng.uniform(0, 500)), float(rng.uniform(0, 500)), float(rng.uniform(20, 300)),
                   float(rng.uniform(20, 300)), 0.0, float(rng.uniform(0.3, 1)))
            na = int(rng.integers(0, 5))
            armors = np.zeros(na, D)
            for a in armors:
                a["x"], a["y"] = rng.uniform(0, 100, 2)
                a["width"], a["height"] = rng.uniform(5, 30, 2)
                a["label"] = float(rng.integers(0, 4))
                a["confidence"] = np.float32(rng.choice([0.5, 0.6, 0.75, 0.9]))
            got_in.append(r.Robot.from_detection(car, armors))
            want_in.append(oracle.make_robot(car, armors))
        got = r.group_robots(got_in, 0.75)
        want = oracle.group_robots(want_in, 0.75)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert g.rect == tuple(w.rect)
            assert (g.label is not None) == bool(w.has_label)
            if w.has_label:
                assert g.label == w.label and np.float32(g.confidence) == np.float32(w.confidence)
                assert g.armors.tobytes() == np.array(
                    [(a.x, a.y, a.width, a.height, a.label, a.confidence) for a in w.armors[:w.n_armors]], D).tobytes()
    a, b = (10.0, 10.0, 100.0, 50.0), (60.0, 20.0, 100.0, 50.0)
    assert r.compute_iou(a, b) == oracle.compute_iou_bounding(a, b)


def test_compute_fails_loudly_without_gpu(built):
    import rm_radar_amd as r
    if r.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(r.DeviceError):
        r.preprocess([np.zeros((8, 8, 3), np.uint8)])
    with pytest.raises(r.DeviceError):
        r.Locator(640, 480, np.eye(3), np.eye(4), np.eye(4))
    with pytest.raises(r.DeviceError):
        r.postprocess(np.zeros((1, 5, 8400), np.float32), 1, 0.65, 0.25, [r.preparam(640, 640)])


def test_product_never_imports_oracle():
    import subprocess, sys
    code = ("import sys; import rm_radar_amd, rm_radar_amd.weights; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rm_radar_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "rmr_oracle" not in txt
