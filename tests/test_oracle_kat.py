"""Pin the CPU oracle to every golden vector the reference's own tests hold for the path
(fixture: tests/golden/kat_reference.json, made by make_kat.py)."""
import numpy as np
import pytest


def ramp():
    return np.arange(48, dtype=np.uint8).reshape(4, 4, 3)


def test_resize_double(kat, oracle):
    # test/detect/kernel_test.cu:71-85
    out = oracle.resize(ramp(), 8, 8)
    assert out.reshape(-1).tolist() == kat["resize_double"]["truth"]


def test_resize_half(kat, oracle):
    # test/detect/kernel_test.cu:87-90
    out = oracle.resize(ramp(), 2, 2)
    assert out.reshape(-1).tolist() == kat["resize_half"]["truth"]


def test_copy_make_border(kat, oracle):
    # test/detect/kernel_test.cu:125-139
    k = kat["copy_make_border"]
    out = oracle.copy_make_border(ramp(), k["top"], k["bottom"], k["left"], k["right"])
    assert out.shape == (8, 6, 3)
    assert out.reshape(-1).tolist() == k["truth"]


def test_blob_matches_blob_from_image(kat, oracle):
    # test/detect/kernel_test.cu:141-173: == blobFromImage(src, 0.01, swapRB=true), bit-equal
    scale = np.float32(kat["blob"]["scale"])
    src = ramp()
    out = oracle.blob(src, float(scale))
    want = (src[:, :, ::-1].astype(np.float32) * scale).transpose(2, 0, 1)
    assert out.dtype == np.float32 and np.array_equal(out, want)


def test_transpose(kat, oracle):
    # test/detect/kernel_test.cu:175-205
    t = kat["transpose"]
    src = np.arange(t["rows"] * t["cols"], dtype=np.float32).reshape(t["rows"], t["cols"])
    assert np.array_equal(oracle.transpose(src), src.T)


@pytest.mark.parametrize("name", ["bus", "zidane"])
def test_preparam_goldens(kat, oracle, name):
    # test/detect/detector_test.cpp:38-41, 57-67
    g = kat["preparam"][name]
    p = oracle.preparam(g["width"], g["height"], 640, 640)
    assert p.width == g["width"] and p.height == g["height"]
    assert p.dw == g["dw"] and p.dh == g["dh"]


def test_preparam_survey_rows(oracle):
    # SURVEY.md section 7 fixture 2 (survey-computed f32 rows)
    for (w, h), (ratio, dw, dh, rw, rh) in {
        (2592, 2048): (4.05, 0, 67, 640, 505),
        (1920, 1080): (3.0, 0, 140, 640, 360),
        (2560, 1440): (4.0, 0, 140, 640, 360),
    }.items():
        p = oracle.preparam(w, h)
        assert abs(p.ratio - ratio) < 1e-6 and p.dw == dw and p.dh == dh
        g = oracle.letterbox_geometry(p)
        assert g[0] == rw and g[1] == rh


def _test_locator(kat, oracle):
    k = kat["locator_test"]
    eye3, eye4 = np.eye(3, dtype=np.float32), np.eye(4, dtype=np.float32)
    return oracle.Locator(k["image_width"], k["image_height"], eye3, eye4, eye4,
                          k["zoom_factor"], k["queue_size"], k["min_depth_diff"],
                          k["max_depth_diff"], k["cluster_tolerance"], k["min_cluster_size"],
                          k["max_cluster_size"], k["max_distance"])


def test_locator_zoom(kat, oracle):
    # test/locate/locator_test.cpp:43-51
    loc = _test_locator(kat, oracle)
    r = kat["locator_test"]["zoom_rect"]
    z = loc.zoom(r)
    assert z[2] == int(r[2] * 0.5) and z[3] == int(r[3] * 0.5)
    assert (loc.wz, loc.hz) == (320, 240)


def test_locator_coordinate_transform(kat, oracle):
    # test/locate/locator_test.cpp:53-74 (identity calibration)
    loc = _test_locator(kat, oracle)
    p = np.array(kat["locator_test"]["transform_point"], np.float32)
    assert np.array_equal(loc.lidar_to_world(p), p)
    cam = loc.lidar_to_camera(p)
    assert cam[0] == np.float32(p[0] * np.float32(0.5) / cam[2])
    assert cam[1] == np.float32(p[1] * np.float32(0.5) / cam[2])
    assert cam[2] == p[2]
    back = loc.camera_to_lidar(cam)
    np.testing.assert_allclose(back, p, rtol=4e-7)  # EXPECT_FLOAT_EQ = 4 ulp


def _two_blobs(k, seed):
    rng = np.random.default_rng(seed)
    img = np.zeros((240, 320), np.float32)
    for b in ("blob1", "blob2"):
        xs = np.clip(rng.normal(*k[b]["x"], k["points_per_blob"]).astype(int), 0, 640 - 1)
        ys = np.clip(rng.normal(*k[b]["y"], k["points_per_blob"]).astype(int), 0, 480 - 1)
        ds = rng.uniform(*k[b]["depth"], k["points_per_blob"]).astype(np.float32)
        for x, y, d in zip(xs, ys, ds):
            if y < 240 and x < 320:
                img[y, x] = d
    return img


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_locator_two_blob_cluster_and_search(kat, oracle, seed):
    # test/locate/locator_test.cpp:76-168 re-created with a fixed seed (the reference uses
    # std::random_device): two Gaussian pixel blobs written straight into the diff image ->
    # exactly 2 clusters, and rect (140,100,40,40) is located.
    k = kat["locator_test"]
    loc = _test_locator(kat, oracle)
    loc.diff_image[:] = _two_blobs(k, seed)
    loc.cluster()
    assert loc.num_clusters == k["expect_clusters"]
    xyz = loc.search(k["search_rect"])
    assert xyz is not None


def test_inverse_matches_float64(oracle):
    rng = np.random.default_rng(3)
    a = rng.normal(size=(4, 4)).astype(np.float32) + 3 * np.eye(4, dtype=np.float32)
    np.testing.assert_allclose(oracle.inv4x4(a), np.linalg.inv(a.astype(np.float64)), rtol=2e-5, atol=2e-6)
    b = rng.normal(size=(3, 3)).astype(np.float32) + 3 * np.eye(3, dtype=np.float32)
    np.testing.assert_allclose(oracle.inv3x3(b), np.linalg.inv(b.astype(np.float64)), rtol=2e-5, atol=2e-6)
