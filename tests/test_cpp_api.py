"""The C++20 drop-in headers (include/radar/) compile with g++ -std=c++20, link against librmr.so
and behave like the reference's classes: constructors throw, hot path works on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rm_radar_amd", "_build", "api_smoke")
EXE_DETECT = os.path.join(ROOT, "rm_radar_amd", "_build", "api_detect")


def _build(src="api_smoke.cpp", exe=EXE):
    import __graft_entry__ as g
    g.build()
    libdir = os.path.join(ROOT, "rm_radar_amd", "_build")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", src), "-L", libdir, "-lrmr",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])


def _run():
    return subprocess.run([EXE], capture_output=True, text=True, timeout=300)


def test_cpp_api_compiles_and_fails_loudly_without_gpu():
    _build()
    _build("api_detect.cpp", EXE_DETECT)  # Detector::detect<T> is a template: it only compiles where it is called
    import rm_radar_amd as r
    if r.device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test")
    res = _run()
    assert res.returncode == 0, res.stdout + res.stderr
    assert "api_smoke ok" in res.stdout


@pytest.mark.gpu
def test_cpp_api_on_gpu():
    _build()
    res = _run()
    assert res.returncode == 0, res.stdout + res.stderr
    assert "located at" in res.stdout


@pytest.mark.gpu
def test_cpp_detect_and_locate_equal_the_ctypes_path(tmp_path):
    """radar::Detector::detect (both overloads), radar::RobotDetector::detect and radar::Locator::update /
    cluster / search, called from C++ through include/radar/*.h (tests/cpp/api_detect.cpp), give byte for
    byte what the ctypes mirror gives on the same packs, frames and clouds.  The kernels per layer come from
    the packs' tuning caches, written by the Python pass and pinned (RMR_PLAN=1) for the C++ one, so both
    launch the same kernels."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import netutil
    import scenes
    import rm_radar_amd as rmr

    _build("api_detect.cpp", EXE_DETECT)
    w, h = 1280, 720
    frames = [netutil.test_image(20 + i, w, h) for i in range(3)]
    car = netutil.tuned_pack(str(tmp_path / "car.rmrw"), 1, 11, 0.25, 0.01, frames)
    armor = netutil.tuned_pack(str(tmp_path / "armor.rmrw"), 12, 12, 0.50, 0.01, frames)
    car_conf, armor_conf = 0.25, 0.5

    # ---- the ctypes path (also writes <pack>.tune for every layer and batch size the C++ pass will use)
    det = rmr.Detector(car, 1, (w, h), len(frames), conf_thresh=car_conf)
    one = det.detect(frames[0])
    many = det.detect(frames)
    det.close()
    rd = rmr.RobotDetector(car, armor, (w, h), 12, max_cars=6, opt_cars=4, car_conf_thresh=car_conf,
                           armor_conf_thresh=armor_conf)
    robots = rd.detect(frames[0])
    rd.close()
    clouds, rects = scenes.scene(3, 20000, (w, h), K=scenes.K640, n_frames=4)
    eye4 = np.eye(4, dtype=np.float32)
    loc = rmr.Locator(w, h, scenes.K640, scenes.SAMPLE_L2C, eye4)
    for c in clouds:
        loc.update(c)
        loc.cluster()
    rb = [rmr.Robot(rect=tuple(float(v) for v in r)) for r in rects[-1]]
    loc.search(rb)

    # ---- the same through the C++ classes
    with open(tmp_path / "frames.bin", "wb") as f:
        f.write(np.array([len(frames), w, h], np.int32).tobytes())
        for im in frames:
            f.write(np.ascontiguousarray(im).tobytes())
    with open(tmp_path / "clouds.bin", "wb") as f:
        f.write(np.array([len(clouds), clouds[0].shape[0]], np.int32).tobytes())
        for c in clouds:
            f.write(np.ascontiguousarray(c[:, :3], np.float32).tobytes())
        f.write(scenes.K640.astype(np.float32).tobytes() + scenes.SAMPLE_L2C.astype(np.float32).tobytes() + eye4.tobytes())
        rr = np.array(rects[-1], np.float32).reshape(-1, 4)
        f.write(np.array([len(rr)], np.int32).tobytes() + rr.tobytes())
    env = dict(os.environ, RMR_PLAN="1")
    res = subprocess.run([EXE_DETECT, car, armor, str(tmp_path / "frames.bin"), str(tmp_path / "clouds.bin"),
                          repr(car_conf), repr(armor_conf)], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = res.stdout.strip().split("\n")
    assert lines[-1] == "api_detect ok"
    it = iter(lines)

    def dets(n, tag):
        out = np.zeros(n, rmr.DET_DTYPE)
        for k in range(n):
            tok = next(it).split()
            assert tok[0] == tag
            out[k] = tuple(np.float32(float.fromhex(v)) for v in tok[1:7])
        return out

    tok = next(it).split()
    assert tok[0] == "detect_one" and int(tok[1]) == len(one) and len(one) >= 1
    assert dets(len(one), "d").tobytes() == one.tobytes()
    tok = next(it).split()
    assert tok[0] == "detect_many" and int(tok[1]) == len(frames)
    for want in many:
        tok = next(it).split()
        assert tok[0] == "image" and int(tok[1]) == len(want)
        assert dets(len(want), "d").tobytes() == want.tobytes()
    tok = next(it).split()
    assert tok[0] == "robots" and int(tok[1]) == len(robots) and len(robots) >= 1
    for want in robots:
        tok = next(it).split()
        assert tok[0] == "robot"
        assert tuple(np.float32(float.fromhex(v)) for v in tok[1:5]) == tuple(np.float32(v) for v in want.rect)
        assert int(tok[6]) == (-1 if want.label is None else want.label)
        n_arm = 0 if want.armors is None else len(want.armors)
        assert int(tok[10]) == n_arm
        if n_arm:
            assert np.float32(float.fromhex(tok[8])) == np.float32(want.confidence)
            assert dets(n_arm, "a").tobytes() == np.asarray(want.armors).tobytes()
    tok = next(it).split()
    assert tok[0] == "located" and int(tok[1]) == len(rb)
    n_loc = 0
    for want in rb:
        tok = next(it).split()
        if want.location is None:
            assert tok[1] == "none"
        else:
            n_loc += 1
            assert tuple(np.float32(float.fromhex(v)) for v in tok[1:4]) == tuple(np.float32(v) for v in want.location)
    assert n_loc >= 1


EXE_SAMPLE = os.path.join(ROOT, "rm_radar_amd", "_build", "sample_calls")
STUBS = os.path.join(ROOT, "tests", "cpp", "stubs")


def _build_sample_calls():
    """tests/cpp/sample_calls.cpp: the reference application's call sequence with cv::Mat / cv::Size / cv::Matx /
    cv::Point3f / pcl::PointCloud<pcl::PointXYZ>::Ptr arguments, compiled against the test-only stand-in headers under
    tests/cpp/stubs (this image has no OpenCV / PCL) -- with them on the include path include/radar/views.h switches
    the classes to the real types."""
    import __graft_entry__ as g
    g.build()
    libdir = os.path.join(ROOT, "rm_radar_amd", "_build")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-Wall", "-Werror", "-I", STUBS, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "sample_calls.cpp"), "-L", libdir, "-lrmr", "-pthread",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE_SAMPLE])


def test_reference_call_sites_compile_with_opencv_and_pcl_types():
    """INTEGRATION.md's claim, compiled: Detector::detect(cv::Mat) / (std::vector<cv::Mat>) / (std::span<cv::Mat>),
    RobotDetector(.., cv::Size, ..)::detect(const cv::Mat&), Locator(int, int, cv::Matx33f, cv::Matx44f, cv::Matx44f),
    Locator::update(const pcl::PointCloud<pcl::PointXYZ>::Ptr&), Tracker(cv::Point3f, int), Robot::rect() -> cv::Rect."""
    _build_sample_calls()
    res = subprocess.run([EXE_SAMPLE], capture_output=True, text=True, timeout=60)
    assert res.returncode == 2 and "usage" in res.stderr   # no arguments: nothing touches a GPU


@pytest.mark.gpu
def test_reference_call_sequence_equals_the_python_mirror(tmp_path):
    """The cycle of samples/sample_radar.h:94-127 (background update, then per frame update + cluster on one thread while
    detect runs on another, join, search, tracker) through the C++ classes with OpenCV / PCL argument types, against
    rm_radar_amd.sample.SampleRadar on the same packs, frames and clouds: rects, labels, confidences, track states and
    located XYZ equal byte for byte (kernels pinned to the tuning caches the Python pass wrote)."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import netutil
    import scenes
    import rm_radar_amd as rmr
    from rm_radar_amd.sample import SampleRadar

    _build_sample_calls()
    w, h = 1280, 720
    n = 4
    frames = [netutil.test_image(40 + i, w, h) for i in range(n)]
    car = netutil.tuned_pack(str(tmp_path / "car.rmrw"), 1, 11, 0.25, 0.01, frames)
    armor = netutil.tuned_pack(str(tmp_path / "armor.rmrw"), 12, 12, 0.50, 0.01, frames)
    car_conf, armor_conf = 0.25, 0.5
    clouds, _ = scenes.scene(5, 20000, (w, h), K=scenes.K640, n_frames=n + 1)
    eye4 = np.eye(4, dtype=np.float32)
    t0, dt = 1000 * 10 ** 9, 100 * 10 ** 6

    radar = SampleRadar(car, armor, (w, h), scenes.K640, scenes.SAMPLE_L2C, eye4, lidar_noise=(0.4, 0.4, 0.4),
                        detector_kwargs=dict(car_conf_thresh=car_conf, armor_conf_thresh=armor_conf))
    radar.update_background_cloud(clouds[0])
    want = [radar.run_once(frames[i], clouds[i + 1], t0 + i * dt) for i in range(n)]
    radar.close()
    det = rmr.Detector(car, 1, (w, h), n, conf_thresh=car_conf)   # the batch sizes the C++ pass's Detector block will use
    one, many = det.detect(frames[0]), det.detect(frames)
    det.close()

    with open(tmp_path / "frames.bin", "wb") as f:
        f.write(np.array([n, w, h], np.int32).tobytes())
        for im in frames:
            f.write(np.ascontiguousarray(im).tobytes())
    with open(tmp_path / "clouds.bin", "wb") as f:
        f.write(np.array([len(clouds), clouds[0].shape[0]], np.int32).tobytes())
        for c in clouds:
            f.write(np.ascontiguousarray(c[:, :3], np.float32).tobytes())
        f.write(scenes.K640.astype(np.float32).tobytes() + scenes.SAMPLE_L2C.astype(np.float32).tobytes() + eye4.tobytes())
    env = dict(os.environ, RMR_PLAN="1")
    res = subprocess.run([EXE_SAMPLE, car, armor, str(tmp_path / "frames.bin"), str(tmp_path / "clouds.bin"),
                          repr(car_conf), repr(armor_conf)], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "cloud is null." in res.stderr and "cloud is empty." in res.stderr   # locate.cpp:160-171
    lines = res.stdout.strip().split("\n")
    assert lines[-1] == "sample_calls ok"
    assert lines[-2].split() == ["detect", str(len(one)), str(len(many[0])), str(len(many))]
    it = iter(lines)
    n_robots = n_located = 0
    for i in range(n):
        tok = next(it).split()
        assert tok[:2] == ["frame", str(i)] and int(tok[3]) == len(want[i])
        for r in want[i]:
            tok = next(it).split()
            assert tok[0] == "robot"
            assert tuple(np.float32(float.fromhex(v)) for v in tok[1:5]) == tuple(np.float32(v) for v in r.rect)
            # Robot::rect(): cv::Rect2f -> cv::Rect rounds half to even (robot.h:111)
            assert [int(v) for v in tok[6:10]] == [int(np.rint(np.float32(v))) for v in r.rect]
            assert int(tok[11]) == (-1 if r.label is None else r.label)
            if r.confidence is not None:
                assert np.float32(float.fromhex(tok[13])) == np.float32(r.confidence)
            assert int(tok[15]) == (r.track_state or 0)
            if r.location is None:
                assert tok[17] == "none"
            else:
                assert tuple(np.float32(float.fromhex(v)) for v in tok[17:20]) == tuple(np.float32(v) for v in r.location)
                n_located += 1
            n_robots += 1
    assert n_robots >= 1
    print(f"{n_robots} robots over {n} frames, {n_located} located")
