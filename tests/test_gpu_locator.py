"""GPU parity: Locator (update / cluster / search / zoom / transforms) vs the CPU oracle through
the C-ABI.  Images, foreground lists and cluster partitions are bit-exact; located XYZ within
1e-3 m (BASELINE.json north_star) -- in practice ~1e-5 m (only the centroid sum associates
differently)."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu

XYZ_TOL_M = 1e-3


@pytest.fixture(scope="module")
def rmr():
    import rm_radar_amd as r
    assert r.device_count() >= 1
    return r


def _pair(rmr, oracle, size, K, L2C, W2C, **kw):
    a = rmr.Locator(size[0], size[1], K, L2C, W2C, **kw)
    b = oracle.Locator(size[0], size[1], K, L2C, W2C, **kw)
    return a, b


def _same_partition(cid_a, cid_b):
    return np.array_equal(cid_a, cid_b)


def _check_frame(rmr, gpu, cpu, cloud, rects):
    gpu.update(cloud)
    cpu.update(cloud)
    assert np.array_equal(gpu.read_image(gpu.BACKGROUND), cpu.background_image)
    assert np.array_equal(gpu.read_image(gpu.DEPTH), cpu.depth_image)
    assert np.array_equal(gpu.read_image(gpu.DIFF), cpu.diff_image)
    gpu.cluster()
    cpu.cluster()
    gx, gp, gc = gpu.foreground()
    cx, cp, cc = cpu.foreground()
    assert np.array_equal(gp, cp)
    assert gx.tobytes() == cx.tobytes()
    assert gpu.num_clusters == cpu.num_clusters
    assert _same_partition(gc, cc)
    robots = [rmr.Robot(rect=tuple(float(v) for v in r)) for r in rects]
    gpu.search(robots)
    n_loc = 0
    for rb, r in zip(robots, rects):
        want = cpu.search(r)
        assert (want is None) == (rb.location is None)
        if want is not None:
            n_loc += 1
            assert np.max(np.abs(np.array(rb.location) - want)) <= XYZ_TOL_M
    return len(gp), cpu.num_clusters, n_loc


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_scene_640(rmr, oracle, seed):
    clouds, rects = scenes.scene(seed, 30000, (640, 640), scenes.K640, scenes.SAMPLE_L2C)
    gpu, cpu = _pair(rmr, oracle, (640, 640), scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32))
    tot_fg = tot_loc = 0
    for f, cloud in enumerate(clouds):
        nfg, ncl, nloc = _check_frame(rmr, gpu, cpu, cloud, rects[f])
        tot_fg += nfg
        tot_loc += nloc
    assert tot_fg > 100 and tot_loc >= 4  # the scene really exercises cluster + search


def test_scene_sample_calibration_100k(rmr, oracle):
    # config-4-like: 1920x1080 stream, 100k points, the sample's world transform
    K = scenes.SAMPLE_K.copy()
    K[0] *= 1920 / 2592
    K[1] *= 1080 / 2048
    clouds, rects = scenes.scene(7, 100000, (1920, 1080), K, scenes.SAMPLE_L2C, n_frames=5)
    gpu, cpu = _pair(rmr, oracle, (1920, 1080), K, scenes.SAMPLE_L2C, scenes.SAMPLE_W2C)
    for f, cloud in enumerate(clouds):
        _check_frame(rmr, gpu, cpu, cloud, rects[f])


def test_assets_clouds_plumbing(rmr, oracle):
    # BASELINE config 1: the reference's own sample clouds (fixture tests/golden/assets_clouds.npz)
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "assets_clouds.npz")
    data = np.load(path)
    gpu, cpu = _pair(rmr, oracle, scenes.SAMPLE_SIZE, scenes.SAMPLE_K, scenes.SAMPLE_L2C, scenes.SAMPLE_W2C)
    for i in range(10):
        c = np.zeros((10000, 4), np.float32)
        c[:, :3] = data[f"cloud{i}"]
        _check_frame(rmr, gpu, cpu, c, [(1000, 800, 400, 300), (200, 1200, 300, 300)])


def test_empty_and_null_cloud(rmr, oracle):
    clouds, _ = scenes.scene(3, 5000, (640, 640), scenes.K640, scenes.SAMPLE_L2C, n_frames=3)
    gpu, cpu = _pair(rmr, oracle, (640, 640), scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32))
    for c in (clouds[0], None, clouds[1], np.zeros((0, 4), np.float32), clouds[2]):
        gpu.update(c)
        cpu.update(c)
        assert np.array_equal(gpu.read_image(gpu.DIFF), cpu.diff_image)
        assert np.array_equal(gpu.read_image(gpu.BACKGROUND), cpu.background_image)
        gpu.cluster()
        cpu.cluster()
        assert np.array_equal(gpu.foreground()[1], cpu.foreground()[1])


def test_duplicate_pixels_last_index_wins(rmr, oracle):
    # many points per pixel, out-of-range / behind-camera / NaN-producing points, strides 12 and 16
    rng = np.random.default_rng(4)
    eye3, eye4 = np.eye(3, dtype=np.float32), np.eye(4, dtype=np.float32)
    kw = dict(zoom_factor=1.0, queue_size=2, min_depth_diff=0.5, max_depth_diff=50.0, max_distance=1e9)
    gpu, cpu = _pair(rmr, oracle, (32, 24), eye3, eye4, eye4, **kw)
    for stride in (3, 4):
        n = 4000
        d = rng.uniform(1, 100, n).astype(np.float32)
        u = rng.uniform(-2, 34, n).astype(np.float32)
        v = rng.uniform(-2, 26, n).astype(np.float32)
        c = np.zeros((n, stride), np.float32)
        c[:, 0], c[:, 1], c[:, 2] = u * d, v * d, d
        c[::50, 2] = -c[::50, 2]         # behind the camera
        c[::97, :3] = 0                  # exact zeros are dropped
        c[5, :3] = (1.0, 1.0, 0.0)       # d == 0 -> inf / NaN pixel coordinates
        gpu.update(c)
        cpu.update(c)
        assert np.array_equal(gpu.read_image(gpu.DEPTH), cpu.depth_image)
        assert np.array_equal(gpu.read_image(gpu.BACKGROUND), cpu.background_image)
        assert np.array_equal(gpu.read_image(gpu.DIFF), cpu.diff_image)


# ---- the reference's own locator tests (test/locate/locator_test.cpp) on the GPU -----------

def _test_pair(rmr, oracle, kat):
    k = kat["locator_test"]
    eye3, eye4 = np.eye(3, dtype=np.float32), np.eye(4, dtype=np.float32)
    kw = dict(zoom_factor=k["zoom_factor"], queue_size=k["queue_size"],
              min_depth_diff=k["min_depth_diff"], max_depth_diff=k["max_depth_diff"],
              cluster_tolerance=k["cluster_tolerance"], min_cluster_size=k["min_cluster_size"],
              max_cluster_size=k["max_cluster_size"], max_distance=k["max_distance"])
    return _pair(rmr, oracle, (k["image_width"], k["image_height"]), eye3, eye4, eye4, **kw)


def test_ref_zoom(rmr, oracle, kat):
    gpu, cpu = _test_pair(rmr, oracle, kat)
    r = kat["locator_test"]["zoom_rect"]
    assert gpu.zoom(r) == cpu.zoom(r)
    assert gpu.zoom(r)[2:] == (int(r[2] * 0.5), int(r[3] * 0.5))
    for rect in [(0, 0, 640, 480), (-50, -50, 100, 100), (600, 440, 100, 100), (700, 500, 10, 10),
                 (101, 77, 33, 55), (5, 5, 1, 1)]:
        assert gpu.zoom(rect) == cpu.zoom(rect)


def test_ref_coordinate_transform(rmr, oracle, kat):
    gpu, cpu = _test_pair(rmr, oracle, kat)
    p = np.array(kat["locator_test"]["transform_point"], np.float32)
    assert np.array_equal(gpu.lidar_to_world(p), p)
    cam = gpu.lidar_to_camera(p)
    assert cam.tobytes() == cpu.lidar_to_camera(p).tobytes()
    np.testing.assert_allclose(gpu.camera_to_lidar(cam), p, rtol=4e-7)
    g2, c2 = _pair(rmr, oracle, scenes.SAMPLE_SIZE, scenes.SAMPLE_K, scenes.SAMPLE_L2C, scenes.SAMPLE_W2C)
    for q in ([19427, 2560, 1833], [6100, -3000, 250], [28000, 9000, -1500]):
        q = np.array(q, np.float32)
        for name in ("lidar_to_world", "lidar_to_camera", "camera_to_lidar"):
            assert getattr(g2, name)(q).tobytes() == getattr(c2, name)(q).tobytes()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ref_two_blob_cluster_and_search(rmr, oracle, kat, seed):
    from test_oracle_kat import _two_blobs
    k = kat["locator_test"]
    gpu, cpu = _test_pair(rmr, oracle, kat)
    img = _two_blobs(k, seed)
    gpu.write_image(gpu.DIFF, img)
    cpu.diff_image[:] = img
    gpu.cluster()
    cpu.cluster()
    assert gpu.num_clusters == k["expect_clusters"] == cpu.num_clusters
    gx, gp, gc = gpu.foreground()
    cx, cp, cc = cpu.foreground()
    assert np.array_equal(gp, cp) and np.array_equal(gc, cc) and gx.tobytes() == cx.tobytes()
    rb = rmr.Robot(rect=tuple(float(v) for v in k["search_rect"]))
    gpu.search([rb])
    want = cpu.search(k["search_rect"])
    assert rb.location is not None and want is not None
    assert np.max(np.abs(np.array(rb.location) - want)) <= XYZ_TOL_M


def test_keep_and_search_kept(rmr, oracle):
    clouds, rects = scenes.scene(5, 20000, (640, 640), scenes.K640, scenes.SAMPLE_L2C, n_frames=6)
    eye4 = np.eye(4, dtype=np.float32)
    gpu = rmr.Locator(640, 640, scenes.K640, scenes.SAMPLE_L2C, eye4, max_frames=6)
    cpu = oracle.Locator(640, 640, scenes.K640, scenes.SAMPLE_L2C, eye4)
    wants = []
    for f, c in enumerate(clouds):
        gpu.update(c)
        gpu.cluster()
        gpu.keep(f)
        cpu.update(c)
        cpu.cluster()
        wants.append([cpu.search(r) for r in rects[f]])
    for f in range(len(clouds)):
        robots = [rmr.Robot(rect=tuple(float(v) for v in r)) for r in rects[f]]
        gpu.search(robots, frame=f)
        for rb, want in zip(robots, wants[f]):
            assert (want is None) == (rb.location is None)
            if want is not None:
                assert np.max(np.abs(np.array(rb.location) - want)) <= XYZ_TOL_M


def test_state_snapshot_resumes_a_stream(rmr, oracle, tmp_path):
    # background image + depth-image queue saved after 7 frames (the queue of 3 has
    # wrapped) and restored into a NEW locator; both then see the same frames and must agree with
    # each other and with the uninterrupted CPU oracle bit for bit.
    size = (640, 640)
    rng = np.random.default_rng(11)
    spec = [((100, 300, 120, 90), 2000, 200), ((400, 200, 80, 120), 1500, 150)]
    gpu, cpu = _pair(rmr, oracle, size, scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32))
    bg = scenes.make_cloud(rng, 30000, scenes.K640, scenes.SAMPLE_L2C, size)  # background pass
    gpu.update(bg), cpu.update(bg)
    for i in range(6):
        c = scenes.make_cloud(rng, 20000, scenes.K640, scenes.SAMPLE_L2C, size, spec if i % 2 else spec[:1])
        gpu.update(c), cpu.update(c)
    path = tmp_path / "stream0.rmrl"
    blob = gpu.save_state(str(path))
    assert path.stat().st_size == len(blob) == 32 + 4 * 320 * 320 * 4  # header, background, queue_size = 3 depth images
    resumed = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32))
    resumed.load_state(str(path))
    assert np.array_equal(resumed.read_image(resumed.BACKGROUND), cpu.background_image)
    for i in range(4):
        c = scenes.make_cloud(rng, 20000, scenes.K640, scenes.SAMPLE_L2C, size, spec if i % 2 == 0 else ())
        for loc in (gpu, resumed):
            loc.update(c)
            loc.cluster()
        cpu.update(c)
        for which in (gpu.BACKGROUND, gpu.DEPTH, gpu.DIFF):
            assert np.array_equal(gpu.read_image(which), resumed.read_image(which))
        assert np.array_equal(resumed.read_image(resumed.DIFF), cpu.diff_image)
        a, b = gpu.foreground(), resumed.foreground()
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # a half-filled queue round-trips too, and mismatched geometry is refused
    fresh = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32))
    fresh.update(scenes.make_cloud(rng, 1000, scenes.K640, scenes.SAMPLE_L2C, size))
    twin = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32))
    twin.load_state(fresh.save_state())
    c = scenes.make_cloud(rng, 5000, scenes.K640, scenes.SAMPLE_L2C, size, spec)
    fresh.update(c), twin.update(c)
    assert np.array_equal(fresh.read_image(fresh.DIFF), twin.read_image(twin.DIFF))
    other = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32), queue_size=5)
    with pytest.raises(rmr.InvalidArgument):
        other.load_state(blob)
    with pytest.raises(rmr.InvalidArgument):
        twin.load_state(blob[:100])


def test_search_batch_equals_per_frame_search(rmr):
    # throughput mode: kept frames searched in one pass == Locator::search frame by frame
    import ctypes as C
    from rm_radar_amd import _lib
    size, nf, cap = (640, 640), 5, 3
    rng = np.random.default_rng(21)
    loc = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32), max_frames=nf)
    loc.update(scenes.make_cloud(rng, 30000, scenes.K640, scenes.SAMPLE_L2C, size))
    rects = [[(100, 300, 120, 90), (400, 200, 80, 120), (10, 10, 30, 30)][: f % 4] for f in range(nf)]
    for f in range(nf):
        spec = [(r, 2000.0, 250) for r in rects[f][:2]]
        loc.update(scenes.make_cloud(rng, 20000, scenes.K640, scenes.SAMPLE_L2C, size, spec))
        loc.cluster()
        loc.keep(f)
    def fill():
        arr = (_lib.Robot * (nf * cap))()
        for f in range(nf):
            for i, r in enumerate(rects[f]):
                arr[f * cap + i].rect[:] = [float(v) for v in r]
        return arr
    counts = np.array([len(r) for r in rects], np.int32)
    a, b = fill(), fill()
    loc.search_batch_raw(a, counts, cap)
    for f in range(nf):
        if counts[f]:
            loc.search_raw(C.cast(C.addressof(b) + f * cap * C.sizeof(_lib.Robot), C.POINTER(_lib.Robot)), int(counts[f]), frame=f)
    located = 0
    for k in range(nf * cap):
        assert a[k].has_location == b[k].has_location and tuple(a[k].location) == tuple(b[k].location)
        located += a[k].has_location
    assert located >= 4
    with pytest.raises(rmr.InvalidArgument):
        loc.search_batch_raw(a, np.array([cap + 1] * nf, np.int32), cap)


@pytest.mark.parametrize("max_fg", [4096, 32768])
def test_update_cluster_batch_equals_per_frame_calls(rmr, max_fg):
    # throughput mode: one cluster pass over a batch of frames == update(); cluster(); keep(f) frame by frame, bit for bit
    # (max_fg 4096: the single-workgroup forest only; 32768: the grid kernels of the large-list path are launched too)
    import ctypes as C
    import torch
    from rm_radar_amd import _lib
    size, nf, cap = (640, 640), 7, 3
    eye = np.eye(4, dtype=np.float32)
    rects = [[(100, 300, 120, 90), (400, 200, 80, 120), (10, 10, 30, 30)][: (f + 1) % 4] for f in range(nf)]
    for device_clouds in (False, True):
        rng = np.random.default_rng(33)
        first = scenes.make_cloud(rng, 30000, scenes.K640, scenes.SAMPLE_L2C, size)
        clouds = [scenes.make_cloud(rng, 20000, scenes.K640, scenes.SAMPLE_L2C, size, [(r, 2000.0 + 100 * f, 250) for r in rects[f][:2]])
                  for f in range(nf)]
        clouds[3] = None   # a null cloud inside the batch: that frame's foreground is empty, nothing is queued
        if device_clouds:
            clouds = [None if c is None else torch.from_numpy(np.ascontiguousarray(c, np.float32)).cuda() for c in clouds]
        a = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, eye, max_frames=nf, max_foreground=max_fg)
        b = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, eye, max_frames=nf, max_foreground=max_fg)
        a.update(first)
        b.update(first)
        for f in range(nf):
            a.update(clouds[f])
            a.cluster()
            a.keep(f)
        b.update_cluster_batch(clouds)
        for which in (a.DIFF, a.BACKGROUND, a.DEPTH):
            assert np.array_equal(a.read_image(which), b.read_image(which))
        for x, y in zip(a.foreground(), b.foreground()):   # the current frame is the batch's last one
            assert np.array_equal(x, y)
        assert a.num_clusters == b.num_clusters
        def fill():
            arr = (_lib.Robot * (nf * cap))()
            for f in range(nf):
                for i, r in enumerate(rects[f]):
                    arr[f * cap + i].rect[:] = [float(v) for v in r]
            return arr
        counts = np.array([len(r) for r in rects], np.int32)
        ra, rb = fill(), fill()
        a.search_batch_raw(ra, counts, cap)
        b.search_batch_raw(rb, counts, cap)
        located = 0
        for k in range(nf * cap):
            assert ra[k].has_location == rb[k].has_location and tuple(ra[k].location) == tuple(rb[k].location)
            located += ra[k].has_location
        assert located >= 5
        # the stream goes on after a batch exactly as after the per-frame calls
        nxt = scenes.make_cloud(rng, 20000, scenes.K640, scenes.SAMPLE_L2C, size, [(rects[0][0], 2500.0, 300)])
        for loc in (a, b):
            loc.update(nxt)
            loc.cluster()
        assert np.array_equal(a.read_image(a.DIFF), b.read_image(b.DIFF))
        for x, y in zip(a.foreground(), b.foreground()):
            assert np.array_equal(x, y)
        a.close()
        b.close()
    loc = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, eye, max_frames=2)
    with pytest.raises(rmr.InvalidArgument):
        loc.update_cluster_batch([first, first, first])   # more frames than max_frames
    loc.close()


def test_update_cluster_batch_reports_an_overflow_in_any_frame(rmr):
    # the foreground of ONE frame inside a batch exceeds max_foreground: the batched search reports it; a per-frame search
    # reports the flag of the frame it searches (kept with the frame's slot)
    import ctypes as C
    from rm_radar_amd import _lib
    size, nf, cap = (640, 640), 4, 2
    eye = np.eye(4, dtype=np.float32)
    rng = np.random.default_rng(5)
    first = scenes.make_cloud(rng, 30000, scenes.K640, scenes.SAMPLE_L2C, size)
    rect = (100, 300, 300, 200)   # queue_size 1 below: a frame's foreground is its own points only
    clouds = [scenes.make_cloud(rng, 20000, scenes.K640, scenes.SAMPLE_L2C, size, [(rect, 2000.0, 3000 if f == 1 else 150)]) for f in range(nf)]
    # foreground sizes frame by frame, to put max_foreground between the dense frame's and the others'
    probe = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, eye, queue_size=1)
    probe.update(first)
    n_fg = []
    for c in clouds:
        probe.update(c)
        probe.cluster()
        n_fg.append(len(probe.foreground()[1]))
    probe.close()
    others = max(n for f, n in enumerate(n_fg) if f != 1)
    assert n_fg[1] > others + 64, n_fg
    max_fg = (n_fg[1] + others) // 2
    loc = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, eye, max_frames=nf, max_foreground=max_fg, queue_size=1)
    loc.update(first)
    loc.update_cluster_batch(clouds)
    arr = (_lib.Robot * (nf * cap))()
    for f in range(nf):
        arr[f * cap].rect[:] = [float(v) for v in rect]
    with pytest.raises(rmr.CapacityError):
        loc.search_batch_raw(arr, np.ones(nf, np.int32), cap)
    # ... while a search of ONE kept frame reports that frame's flag only, and the current frame is the batch's last
    one = [rmr.Robot(rect=tuple(float(v) for v in rect))]
    loc.search(one, frame=0)
    with pytest.raises(rmr.CapacityError):
        loc.search(one, frame=1)
    loc.search(one, frame=2)
    loc.search(one)
    loc.close()
    # the same stream without the dense frame: nothing overflows
    clouds[1] = clouds[0]
    loc2 = rmr.Locator(size[0], size[1], scenes.K640, scenes.SAMPLE_L2C, eye, max_frames=nf, max_foreground=max_fg, queue_size=1)
    loc2.update(first)
    loc2.update_cluster_batch(clouds)
    loc2.search_batch_raw(arr, np.ones(nf, np.int32), cap)
    loc2.close()


def test_dense_foreground_beyond_one_workgroup(rmr, oracle):
    """K = 20 robots of ~600 points (kMaxBatchSize cars, sample_radar.h:34): more than 4096 foreground points,
    so the pair phase of the clustering runs over the whole chip on a global union-find forest (cc_init_grid /
    cc_pairs_grid) instead of inside the single-workgroup kernel.  Same foreground list, partition and cluster
    ids as the oracle (the PCL Euclidean clustering of locate.cpp:255-257)."""
    size = (1280, 1024)
    K = np.array([[832, 0, 640], [0, 832, 512], [0, 0, 1]], np.float32)
    rng = np.random.default_rng(21)
    gpu, cpu = _pair(rmr, oracle, size, K, scenes.SAMPLE_L2C, np.eye(4, dtype=np.float32), max_cluster_size=4000)
    for _ in range(2):   # warm the background image and the depth ring
        bg = scenes.make_cloud(rng, 150000, K, scenes.SAMPLE_L2C, size)
        _check_frame(rmr, gpu, cpu, bg, [])
    rects = []
    for i in range(20):
        w, h = rng.uniform(90, 140), rng.uniform(90, 140)
        rects.append((40 + (i % 5) * 240 + rng.uniform(0, 60), 300 + (i // 5) * 170 + rng.uniform(0, 20), w, h))
    robots = [(r, float(rng.uniform(1500, 3000)), 900) for r in rects]
    cloud = scenes.make_cloud(rng, 150000 + 20 * 900, K, scenes.SAMPLE_L2C, size, robots)
    nfg, ncl, nloc = _check_frame(rmr, gpu, cpu, cloud, rects)
    assert nfg > 4096, nfg          # the grid path was taken
    assert ncl >= 10 and nloc >= 10
