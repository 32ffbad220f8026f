"""rm_radar_amd -- MI355X-native detect + locate hot path of zmsbruce/rm_radar.

Host-side mirror (Python) of the reference's C++ interface for this path -- ``Detector``,
``RobotDetector``, ``Locator``, ``Robot``, ``Detection``, ``PreParam`` -- over the C-ABI of
``librmr.so`` (include/rmr.h).  All compute is hand-written HIP for gfx950 inside that
library; there is no CPU fallback and importing this package never touches ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import (CapacityError, DET_DTYPE, DeviceError, InvalidArgument, PreParam, RmrError,
                   check, lib)

__all__ = ["Detector", "RobotDetector", "Locator", "Robot", "PreParam", "preparam", "FrameBatch", "UploadRing", "PinnedArray", "DeviceArray",
           "letterbox_geometry", "letterbox", "preprocess", "postprocess", "transpose",
           "conv2d", "pin_plan", "restore_detection", "device_count", "profile", "DET_DTYPE", "RmrError",
           "run_batch", "Tracker", "KalmanFilter", "SingerEKF", "auction", "TRACK_TENTATIVE", "TRACK_CONFIRMED", "TRACK_DELETED",
           "InvalidArgument", "DeviceError", "CapacityError", "Label"]

# radar::Label (src/robot/robot.h:32-45)
Label = {"BlueHero": 0, "BlueEngineer": 1, "BlueInfantryThree": 2, "BlueInfantryFour": 3,
         "BlueInfantryFive": 4, "RedHero": 5, "RedEngineer": 6, "RedInfantryThree": 7,
         "RedInfantryFour": 8, "RedInfantryFive": 9, "BlueSentry": 10, "RedSentry": 11}


def device_count() -> int:
    return lib().rmr_device_count()


# ------------------------------------------------------------------------------- images

def _is_device_tensor(obj) -> bool:
    return hasattr(obj, "data_ptr") and getattr(obj, "is_cuda", False)


def _as_image(obj, keep: list) -> _lib.Image:
    """numpy HxWx3 u8 (host) or a torch CUDA tensor of that shape (device) -> rmr_image."""
    if _is_device_tensor(obj):
        if obj.dim() != 3 or obj.shape[2] != 3 or obj.element_size() != 1:
            raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "device image must be HxWx3 uint8")
        if obj.stride(2) != 1 or obj.stride(1) != 3:
            raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "device image pixels must be packed BGR")
        keep.append(obj)
        return _lib.Image(obj.data_ptr(), obj.shape[1], obj.shape[0], obj.stride(0), _lib.MEM_DEVICE)
    a = np.asarray(obj)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "image must be HxWx3 uint8 (BGR)")
    if a.strides[2] != 1 or a.strides[1] != 3:
        a = np.ascontiguousarray(a)
    keep.append(a)
    return _lib.Image(a.ctypes.data, a.shape[1], a.shape[0], a.strides[0], _lib.MEM_HOST)


def _images(images, crops):
    keep: list = []
    arr = (_lib.Image * len(images))(*[_as_image(im, keep) for im in images])
    cr = None
    if crops is not None:
        cr = np.ascontiguousarray(np.asarray(crops, np.int32).reshape(len(images), 4))
        keep.append(cr)
    return arr, (_lib.ip(cr) if cr is not None else None), keep


# ------------------------------------------------------------------------------- geometry

def preparam(in_w: int, in_h: int, out_w: int = 640, out_h: int = 640) -> PreParam:
    """radar::detect::PreParam(cv::Size input, cv::Size output)  (preparam.h:46-52)"""
    p = PreParam()
    check(lib().rmr_preparam_make(in_w, in_h, out_w, out_h, C.byref(p)))
    return p


def letterbox_geometry(p: PreParam):
    v = [C.c_int() for _ in range(4)]
    check(lib().rmr_letterbox_geometry(C.byref(p), *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)  # resized_w, resized_h, top, left


def restore_detection(det, p: PreParam):
    d = _lib.Detection(*[float(v) for v in det])
    check(lib().rmr_restore_detection(C.byref(d), C.byref(p)))
    return (d.x, d.y, d.width, d.height, d.label, d.confidence)


# ------------------------------------------------------------------------------- unit kernels

def letterbox(images, resized_w, resized_h, top, left, out_w, out_h, fill=128, scale=1.0,
              fmt="u8", crops=None, device=0):
    """Fused resize + border (+ blob) with explicit geometry (detector.cu:40-171)."""
    arr, cr, keep = _images(images, crops)
    n = len(images)
    if fmt == "u8":
        out = np.empty((n, out_h, out_w, 3), np.uint8)
        code = _lib.FMT_U8_HWC
    else:
        out = np.empty((n, 3, out_h, out_w), np.float32)
        code = _lib.FMT_F32_NCHW
    check(lib().rmr_letterbox(device, arr, cr, n, resized_w, resized_h, top, left, out_w, out_h,
                              fill, scale, code, out.ctypes.data))
    return out


def preprocess(images, crops=None, out_w=640, out_h=640, device=0):
    """Detector::preprocess (detector.cu:380-502) -> (blob [n,3,h,w] f32, [PreParam])."""
    arr, cr, keep = _images(images, crops)
    n = len(images)
    blob = np.empty((n, 3, out_h, out_w), np.float32)
    pps = (PreParam * n)()
    check(lib().rmr_preprocess(device, arr, cr, n, out_w, out_h, _lib.fp(blob), pps))
    return blob, list(pps)


def postprocess(net_out, classes, nms_thresh, conf_thresh, pps, cap=None, device=0):
    """Detector::postprocess (detector.cu:522-582): [n, 4+classes, anchors] -> list of arrays."""
    net_out = np.ascontiguousarray(net_out, np.float32)
    n, ch, a = net_out.shape
    cap = cap or a
    out = np.empty((n, cap), DET_DTYPE)
    counts = np.zeros(n, np.int32)
    pp = (PreParam * n)(*pps)
    check(lib().rmr_postprocess(device, _lib.fp(net_out), n, ch, a, classes, nms_thresh,
                                conf_thresh, pp, out.ctypes.data, _lib.ip(counts), cap))
    return [out[i, :counts[i]].copy() for i in range(n)]


def transpose(src, device=0):
    src = np.ascontiguousarray(src, np.float32)
    r, c = src.shape
    dst = np.empty((c, r), np.float32)
    check(lib().rmr_transpose(device, _lib.fp(src), _lib.fp(dst), r, c))
    return dst


def conv2d(x_nhwc, w_oihw, bias, stride, pad, silu, residual=None, tile=-1, device=0):
    """One conv(+bias)(+SiLU)(+residual) layer through the MFMA implicit-GEMM engine."""
    x = np.ascontiguousarray(x_nhwc, np.float32)
    w = np.ascontiguousarray(w_oihw, np.float32)
    n, h, wd, cin = x.shape
    cout, cin2, kh, kw = w.shape
    assert cin2 == cin
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    y = np.empty((n, ho, wo, cout), np.float32)
    b = np.ascontiguousarray(bias, np.float32) if bias is not None else np.zeros(cout, np.float32)
    r = np.ascontiguousarray(residual, np.float32) if residual is not None else None
    check(lib().rmr_conv2d(device, _lib.fp(x), n, h, wd, cin, _lib.fp(w), _lib.fp(b), cout, kh, kw,
                           stride, pad, int(bool(silu)), _lib.fp(r) if r is not None else None,
                           _lib.fp(y), tile))
    return y


def pin_plan(tune_path, plan_path, batches):
    """Write a pinned plan (RMR_PLAN=<plan_path>) from a tuning cache: every layer runs, at each batch size in
    `batches`, the kernel the cache chose at ONE tuned batch size -- the largest at which the cache has every layer.
    One kernel per layer whatever the batch means one f32 summation order per output value, so an image gives
    bit-identical results alone or inside a batch; and nothing is timed when a plan is pinned, so two boxes launch
    the same kernels.  All entries come from the same batch size, so a grouped launch (200000 + v) and its members'
    398 markers, or a fused bottleneck (340..) and its 399, stay together.  The small-batch family (100000 + v, and the
    grouped launches) has a hard limit of tiles per launch that grows with the batch: a plan that carries such entries is
    only written for batch sizes up to the one they were tuned at (ValueError beyond), instead of a plan whose entries
    the library would drop at load time."""
    with open(tune_path) as f:
        header = f.readline()
        by_n = {}
        for line in f:
            op, n, choice = (int(v) for v in line.split())
            by_n.setdefault(n, {})[op] = choice
    if not by_n:
        raise ValueError(f"{tune_path}: no tuned layers")
    all_ops = set().union(*[set(d) for d in by_n.values()])
    full = [n for n, d in by_n.items() if set(d) == all_ops]
    if not full:
        raise ValueError(f"{tune_path}: no batch size at which every layer is tuned")
    n_ref = max(full)
    best = {}
    for op, choice in by_n[n_ref].items():
        # split-K (1000 * split + tile) reduces partial sums in an order that depends on the split: the
        # plan takes the same tile without it
        if choice < 100000:   # (100000 + variant = the small-batch family conv_sb: no split-K form)
            choice %= 1000
        best[op] = choice
    if any(c >= 100000 for c in best.values()) and max(batches) > n_ref:
        raise ValueError(f"{tune_path}: the cache's choices at {n_ref} images include small-batch kernels (conv_sb), which cannot be "
                         f"pinned for larger batches ({max(batches)}): tune at the largest batch the plan is for")
    with open(plan_path, "w") as f:
        f.write(header)
        for op, choice in sorted(best.items()):
            for n in batches:
                f.write(f"{op} {n} {choice}\n")
    return plan_path


def conv_bench(n, h, w, cin, cout, k, stride, kernel, residual=False, reps=10, device=0):
    """Mean launch time in ms of one conv layer (f16 in / out, bias + SiLU) on device-resident random data
    with the tiled kernel `kernel` (ids as conv2d's `tile`).  Development hook behind tools/conv_bench.py."""
    ms = C.c_float(0.0)
    check(lib().rmr_conv_bench(device, n, h, w, cin, cout, k, stride, int(bool(residual)), kernel, reps, C.byref(ms)))
    return ms.value


# ------------------------------------------------------------------------------- Robot

@dataclass
class Robot:
    """radar::Robot as filled by detect + locate (src/robot/robot.h:53-164)."""
    rect: tuple = (0.0, 0.0, 0.0, 0.0)
    label: Optional[int] = None
    confidence: Optional[float] = None
    armors: Optional[np.ndarray] = None
    location: Optional[tuple] = None
    track_state: Optional[int] = None  # TRACK_TENTATIVE / TRACK_CONFIRMED once a Tracker has seen it

    def is_detected(self) -> bool:
        return self.armors is not None

    def is_located(self) -> bool:
        return self.location is not None

    def is_tracked(self) -> bool:
        return self.track_state is not None

    @staticmethod
    def from_c(r: _lib.Robot) -> "Robot":
        armors = None
        if r.n_armors > 0:  # isDetected(); a label alone can also come from a track (robot.cpp:81-94)
            armors = np.array([(a.x, a.y, a.width, a.height, a.label, a.confidence)
                               for a in r.armors[: r.n_armors]], DET_DTYPE)
        return Robot(rect=tuple(r.rect), label=r.label if r.has_label else None,
                     confidence=r.confidence if r.n_armors > 0 else None, armors=armors,
                     location=tuple(r.location) if r.has_location else None,
                     track_state=r.track_state if r.track_state else None)

    def to_c(self) -> _lib.Robot:
        r = _lib.Robot()
        r.rect[:] = [float(v) for v in self.rect]
        r.has_label = int(self.label is not None)
        r.label = self.label if self.label is not None else -1
        r.confidence = self.confidence or 0.0
        if self.armors is not None:
            r.n_armors = len(self.armors)
            for i, a in enumerate(self.armors):
                r.armors[i] = _lib.Detection(*[float(v) for v in a])
        if self.location is not None:
            r.has_location = 1
            r.location[:] = [float(v) for v in self.location]
        r.track_state = self.track_state or 0
        return r

    def feature(self, class_num: int) -> np.ndarray:
        """Robot::feature (robot.cpp:102-122)"""
        out = np.zeros(class_num, np.float32)
        c = self.to_c()
        check(lib().rmr_robot_feature(C.byref(c), class_num, _lib.fp(out)))
        return out

    @staticmethod
    def from_detection(car, armors) -> "Robot":
        """Robot::Robot(const Detection& car, const std::vector<Detection>& armors)"""
        r = _lib.Robot()
        c = _lib.Detection(*[float(v) for v in car])
        armors = np.ascontiguousarray(armors, DET_DTYPE)
        check(lib().rmr_robot_set_detection(C.byref(r), C.byref(c), armors.ctypes.data, len(armors)))
        return Robot.from_c(r)


def compute_iou(a, b) -> float:
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return lib().rmr_compute_iou(_lib.fp(a), _lib.fp(b))


def group_robots(robots: Sequence[Robot], iou_thresh: float) -> List[Robot]:
    n = len(robots)
    arr = (_lib.Robot * max(n, 1))(*[r.to_c() for r in robots])
    out = (_lib.Robot * max(n, 1))()
    m = C.c_int()
    check(lib().rmr_group_robots(arr, n, iou_thresh, out, C.byref(m)))
    return [Robot.from_c(out[i]) for i in range(m.value)]


# ------------------------------------------------------------------------------- Detector

class Detector:
    """radar::Detector (src/detect/detector.h:84-169).  ``engine_path`` names a weight pack
    (rm_radar_amd.weights) instead of a TensorRT engine."""

    def __init__(self, engine_path, classes, image_size, max_batch_size, opt_batch_size=None,
                 nms_thresh=0.65, conf_thresh=0.25, input_width=640, input_height=640,
                 input_name="images", input_channels=3, opt_level=3, device=0, precision="f16"):
        cfg = _lib.DetectorCfg()
        lib().rmr_detector_cfg_default(C.byref(cfg))
        from .onnx_import import ensure_pack  # detector.cpp:74-99: build from the sibling .onnx when missing
        self._path = ensure_pack(engine_path).encode()
        cfg.engine_path = self._path
        cfg.classes = classes
        cfg.image_width, cfg.image_height = image_size
        cfg.max_batch_size = max_batch_size
        cfg.opt_batch_size = opt_batch_size or 0
        cfg.nms_thresh, cfg.conf_thresh = nms_thresh, conf_thresh
        cfg.input_width, cfg.input_height = input_width, input_height
        cfg.input_channels = input_channels
        cfg.device = device
        cfg.precision = {"f16": 0, "fp8": 1}[precision]
        self.classes = classes
        self._h = C.c_void_p()
        check(lib().rmr_detector_create(C.byref(cfg), C.byref(self._h)))
        self.anchors = lib().rmr_detector_anchors(self._h)
        self.channels = lib().rmr_detector_channels(self._h)
        self.flops_per_image = lib().rmr_detector_flops_per_image(self._h)

    def close(self):
        if getattr(self, "_h", None) and lib is not None:  # module globals are gone at interpreter exit
            lib().rmr_detector_destroy(self._h)
            self._h = None

    __del__ = close

    def detect(self, images, crops=None, cap=1024):
        """Detector::detect<T> (detector.h:117-134): one image -> array, a list -> list."""
        single = not isinstance(images, (list, tuple))
        imgs = [images] if single else list(images)
        arr, cr, keep = _images(imgs, crops)
        n = len(imgs)
        out = np.empty((n, cap), DET_DTYPE)
        counts = np.zeros(n, np.int32)
        check(lib().rmr_detector_detect(self._h, arr, cr, n, out.ctypes.data, _lib.ip(counts), cap))
        res = [out[i, :counts[i]].copy() for i in range(n)]
        return res[0] if single else res

    def arena_bytes(self) -> float:
        """Activation memory held on the GPU (for `chunk()` images per launch)."""
        return lib().rmr_detector_arena_bytes(self._h)

    def chunk(self) -> int:
        return lib().rmr_detector_chunk(self._h)

    def read_feature(self, name, img=0):
        """Output of backbone / neck stage `name` ("model.0" ... "model.21") for image `img` of the last call,
        f32 [h, w, c] (parity hook)."""
        dims = np.zeros(3, np.int32)
        check(lib().rmr_detector_read_feature(self._h, name.encode(), img, None, _lib.ip(dims)))
        out = np.empty(tuple(int(v) for v in dims), np.float32)
        check(lib().rmr_detector_read_feature(self._h, name.encode(), img, _lib.fp(out), _lib.ip(dims)))
        return out

    def infer(self, images, crops=None):
        """preprocess + network: the [n, 4+classes, anchors] tensor handed to postprocess."""
        imgs = list(images)
        arr, cr, keep = _images(imgs, crops)
        n = len(imgs)
        out = np.empty((n, self.channels, self.anchors), np.float32)
        pps = (PreParam * n)()
        check(lib().rmr_detector_infer(self._h, arr, cr, n, _lib.fp(out), pps))
        return out, list(pps)


class RobotDetector:
    """radar::RobotDetector (src/detect/detector.h:171-190)."""

    def __init__(self, car_engine_path, armor_engine_path, image_size, armor_classes, max_cars,
                 opt_cars, iou_thresh=0.75, car_nms_thresh=0.65, car_conf_thresh=0.25,
                 armor_nms_thresh=0.65, armor_conf_thresh=0.50, input_width=640,
                 input_height=640, input_name="images", input_channels=3, opt_level=5,
                 device=0, max_frames=1, precision="f16"):
        cfg = _lib.RobotDetectorCfg()
        lib().rmr_robot_detector_cfg_default(C.byref(cfg))
        from .onnx_import import ensure_pack
        self._paths = (ensure_pack(car_engine_path).encode(), ensure_pack(armor_engine_path).encode())
        cfg.car_engine_path, cfg.armor_engine_path = self._paths
        cfg.image_width, cfg.image_height = image_size
        cfg.armor_classes = armor_classes
        cfg.max_cars, cfg.opt_cars = max_cars, opt_cars
        cfg.iou_thresh = iou_thresh
        cfg.car_nms_thresh, cfg.car_conf_thresh = car_nms_thresh, car_conf_thresh
        cfg.armor_nms_thresh, cfg.armor_conf_thresh = armor_nms_thresh, armor_conf_thresh
        cfg.input_width, cfg.input_height = int(input_width), int(input_height)
        cfg.input_channels = input_channels
        cfg.device = device
        cfg.max_frames = max_frames
        cfg.precision = {"f16": 0, "fp8": 1}[precision]
        self.max_cars = max_cars
        self.armor_classes = armor_classes
        self.input_size = (int(input_width), int(input_height))
        self._h = C.c_void_p()
        check(lib().rmr_robot_detector_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and lib is not None:  # module globals are gone at interpreter exit
            lib().rmr_robot_detector_destroy(self._h)
            self._h = None

    __del__ = close

    def arena_bytes(self) -> float:
        """Activation memory both networks hold on the GPU."""
        return lib().rmr_robot_detector_arena_bytes(self._h)

    def detect(self, image) -> List[Robot]:
        """RobotDetector::detect(const cv::Mat&) (detector.cpp:413-455)"""
        keep: list = []
        img = _as_image(image, keep)
        cap = self.max_cars
        out = (_lib.Robot * cap)()
        n = C.c_int()
        check(lib().rmr_robot_detector_detect(self._h, C.byref(img), out, C.byref(n), cap))
        return [Robot.from_c(out[i]) for i in range(n.value)]

    def detect_batch_raw(self, images, forced_crops=None):
        """Throughput mode; returns (ctypes Robot array [n_frames*cap], counts)."""
        imgs = list(images)
        arr, _, keep = _images(imgs, None)
        n = len(imgs)
        cap = self.max_cars
        out = (_lib.Robot * (cap * n))()
        counts = np.zeros(n, np.int32)
        fc, per = None, 0
        if forced_crops is not None:
            fc = np.ascontiguousarray(np.asarray(forced_crops, np.int32).reshape(n, -1, 4))
            per = fc.shape[1]
        check(lib().rmr_robot_detector_detect_batch(self._h, arr, n, _lib.ip(fc) if fc is not None else None,
                                                    per, out, _lib.ip(counts), cap))
        return out, counts

    def read_heads(self, stage, first=0, n=None):
        """Parity hook (rmr_robot_detector_read_heads): the network output of images [first, first + n) of the LAST call's
        car (stage 0) or armor (stage 1) batch -- f32 [n, 4 + classes, anchors] -- and their PreParams."""
        last = C.c_int()
        check(lib().rmr_robot_detector_read_heads(self._h, stage, 0, 0, None, None, C.byref(last)))
        n = last.value - first if n is None else n
        ch = 5 if stage == 0 else 4 + self.armor_classes
        out = np.empty((n, ch, 8400 * self.input_size[0] * self.input_size[1] // (640 * 640)), np.float32)
        pps = (PreParam * max(n, 1))()
        if n > 0:
            check(lib().rmr_robot_detector_read_heads(self._h, stage, first, n, _lib.fp(out), pps, None))
        return out, list(pps)[:n]

    def detect_batch(self, images, forced_crops=None) -> List[List[Robot]]:
        out, counts = self.detect_batch_raw(images, forced_crops)
        cap = self.max_cars
        return [[Robot.from_c(out[f * cap + i]) for i in range(counts[f])]
                for f in range(len(counts))]


# ------------------------------------------------------------------------------- Locator

class Locator:
    """radar::Locator (src/locate/locator.h:53-98).  Clouds: [n, >=3] f32 (mm), numpy (host) or a
    torch CUDA tensor (device)."""

    DEPTH, BACKGROUND, DIFF = 0, 1, 2

    def __init__(self, image_width, image_height, intrinsic, lidar_to_camera, world_to_camera,
                 zoom_factor=0.5, queue_size=3, min_depth_diff=500.0, max_depth_diff=4000.0,
                 cluster_tolerance=400.0, min_cluster_size=8, max_cluster_size=1000,
                 max_distance=29300.0, device=0, max_points=262144, max_foreground=32768,
                 max_frames=1):
        cfg = _lib.LocatorCfg()
        lib().rmr_locator_cfg_default(C.byref(cfg))
        cfg.image_width, cfg.image_height = image_width, image_height
        cfg.intrinsic[:] = np.asarray(intrinsic, np.float32).reshape(9).tolist()
        cfg.lidar_to_camera[:] = np.asarray(lidar_to_camera, np.float32).reshape(16).tolist()
        cfg.world_to_camera[:] = np.asarray(world_to_camera, np.float32).reshape(16).tolist()
        cfg.zoom_factor = zoom_factor
        cfg.queue_size = queue_size
        cfg.min_depth_diff, cfg.max_depth_diff = min_depth_diff, max_depth_diff
        cfg.cluster_tolerance = cluster_tolerance
        cfg.min_cluster_size, cfg.max_cluster_size = min_cluster_size, max_cluster_size
        cfg.max_distance = max_distance
        cfg.device = device
        cfg.max_points, cfg.max_foreground, cfg.max_frames = max_points, max_foreground, max_frames
        self._h = C.c_void_p()
        check(lib().rmr_locator_create(C.byref(cfg), C.byref(self._h)))
        self.wz = lib().rmr_locator_width(self._h)
        self.hz = lib().rmr_locator_height(self._h)

    def close(self):
        if getattr(self, "_h", None) and lib is not None:  # module globals are gone at interpreter exit
            lib().rmr_locator_destroy(self._h)
            self._h = None

    __del__ = close

    def update(self, cloud):
        """Locator::update (locate.cpp:158-220); None / empty = the null / empty cloud."""
        if cloud is None or len(cloud) == 0:
            check(lib().rmr_locator_update(self._h, None, 0, 0, _lib.MEM_HOST))
            return
        if _is_device_tensor(cloud):
            if cloud.dim() != 2 or cloud.shape[1] < 3 or cloud.element_size() != 4 or cloud.stride(1) != 1:
                raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "device cloud must be [n, >=3] f32")
            check(lib().rmr_locator_update(self._h, cloud.data_ptr(), cloud.shape[0],
                                           cloud.stride(0) * 4, _lib.MEM_DEVICE))
            return
        a = np.ascontiguousarray(cloud, np.float32)
        check(lib().rmr_locator_update(self._h, a.ctypes.data, a.shape[0], a.strides[0], _lib.MEM_HOST))

    def cluster(self):
        check(lib().rmr_locator_cluster(self._h))

    def keep(self, frame: int):
        check(lib().rmr_locator_keep(self._h, frame))

    def update_cluster_batch(self, clouds):
        """Throughput mode: update + cluster + keep(f) for the consecutive frames `clouds` of this stream (all host
        arrays or all device tensors of [n, >=3] f32 with one row stride; None / empty = the null cloud), the
        cluster stage as one pass over all frames.  Same results as the three calls per frame."""
        nf = len(clouds)
        ptrs = (C.c_void_p * nf)()
        counts = np.zeros(nf, np.int32)
        keep_alive, stride, mem = [], 0, None
        for f, c in enumerate(clouds):
            if c is None or len(c) == 0:
                ptrs[f] = None
                continue
            if _is_device_tensor(c):
                if c.dim() != 2 or c.shape[1] < 3 or c.element_size() != 4 or c.stride(1) != 1:
                    raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "device cloud must be [n, >=3] f32")
                ptr, n, st, m = c.data_ptr(), c.shape[0], c.stride(0) * 4, _lib.MEM_DEVICE
            else:
                a = np.ascontiguousarray(c, np.float32)
                keep_alive.append(a)
                ptr, n, st, m = a.ctypes.data, a.shape[0], a.strides[0], _lib.MEM_HOST
            if mem not in (None, m) or stride not in (0, st):
                raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "the clouds of a batch share one memory kind and row stride")
            ptrs[f], counts[f], stride, mem = ptr, n, st, m
        check(lib().rmr_locator_update_cluster_batch(self._h, ptrs, _lib.ip(counts), stride or 16,
                                                     _lib.MEM_HOST if mem is None else mem, nf))

    def _search(self, robots, frame):
        n = len(robots)
        if n == 0:
            return robots
        arr = (_lib.Robot * n)(*[r.to_c() for r in robots])
        if frame is None:
            check(lib().rmr_locator_search(self._h, arr, n))
        else:
            check(lib().rmr_locator_search_kept(self._h, frame, arr, n))
        for r, c in zip(robots, arr):
            if c.has_location:
                r.location = tuple(c.location)
        return robots

    def search(self, robots: List[Robot], frame=None) -> List[Robot]:
        """Locator::search(std::vector<Robot>&) (locate.cpp:323-326)"""
        return self._search(robots, frame)

    def search_raw(self, robots_c, n: int, frame=None):
        if n <= 0:
            return
        if frame is None:
            check(lib().rmr_locator_search(self._h, robots_c, n))
        else:
            check(lib().rmr_locator_search_kept(self._h, frame, robots_c, n))

    def search_batch_raw(self, robots_c, counts, cap: int):
        """Throughput mode: search the kept frames 0..len(counts)-1 in one pass; `robots_c` is the
        ctypes array RobotDetector.detect_batch_raw returned (cap robots per frame)."""
        counts = np.ascontiguousarray(counts, np.int32)
        check(lib().rmr_locator_search_batch(self._h, robots_c, _lib.ip(counts), len(counts), cap))

    # -- private members the reference's tests reach (locator_test.cpp:6-13)
    def read_image(self, which) -> np.ndarray:
        out = np.empty((self.hz, self.wz), np.float32)
        check(lib().rmr_locator_read_image(self._h, which, _lib.fp(out)))
        return out

    def write_image(self, which, img):
        img = np.ascontiguousarray(img, np.float32)
        assert img.shape == (self.hz, self.wz)
        check(lib().rmr_locator_write_image(self._h, which, _lib.fp(img)))

    def save_state(self, path=None) -> bytes:
        """Background image + depth-image queue as one blob (optionally written to `path`): the
        temporal state of locator.h:90-91, which the reference never persists."""
        n = C.c_size_t()
        check(lib().rmr_locator_state_bytes(self._h, C.byref(n)))
        buf = (C.c_char * n.value)()
        check(lib().rmr_locator_save_state(self._h, buf, n.value))
        blob = bytes(buf)
        if path is not None:
            with open(path, "wb") as f:
                f.write(blob)
        return blob

    def load_state(self, blob_or_path) -> None:
        """Restores what save_state produced (same image size and queue_size required)."""
        blob = blob_or_path
        if not isinstance(blob, (bytes, bytearray)):
            with open(blob_or_path, "rb") as f:
                blob = f.read()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        check(lib().rmr_locator_load_state(self._h, buf, len(blob)))

    def _xf(self, which, p):
        a = np.asarray(p, np.float32)
        o = np.zeros(3, np.float32)
        check(lib().rmr_locator_transform(self._h, which, _lib.fp(a), _lib.fp(o)))
        return o

    def lidar_to_world(self, p):
        return self._xf(0, p)

    def camera_to_lidar(self, p):
        return self._xf(1, p)

    def lidar_to_camera(self, p):
        return self._xf(2, p)

    def zoom(self, rect):
        r = np.asarray(rect, np.int32)
        o = np.zeros(4, np.int32)
        check(lib().rmr_locator_zoom(self._h, _lib.ip(r), _lib.ip(o)))
        return tuple(int(v) for v in o)

    def foreground(self, cap=32768):
        xyz = np.empty((cap, 3), np.float32)
        pix = np.empty(cap, np.int32)
        cid = np.empty(cap, np.int32)
        n = C.c_int()
        check(lib().rmr_locator_foreground(self._h, _lib.fp(xyz), _lib.ip(pix), _lib.ip(cid), cap, C.byref(n)))
        m = min(n.value, cap)
        return xyz[:m].copy(), pix[:m].copy(), cid[:m].copy()

    @property
    def num_clusters(self) -> int:
        return lib().rmr_locator_num_clusters(self._h)


# ------------------------------------------------------------------------------- profiling

# ------------------------------------------------------------------------------- input staging

class DeviceArray:
    """A typed view of device memory this package did not get from torch (an UploadRing slot): quacks like the
    torch CUDA tensors that _as_image / FrameBatch accept (data_ptr, shape, stride in elements, element_size)."""
    is_cuda = True

    def __init__(self, ptr: int, shape, dtype):
        self.ptr, self.shape, self.dtype = int(ptr), tuple(int(v) for v in shape), np.dtype(dtype)
        st, acc = [], 1
        for n in reversed(self.shape):
            st.append(acc)
            acc *= n
        self._strides = tuple(reversed(st))

    def __getitem__(self, i):
        """a[i]: the i-th slice along the first axis (a frame of a batch block)."""
        i = int(i)
        if not 0 <= i < self.shape[0]:
            raise IndexError(i)
        return DeviceArray(self.ptr + i * self._strides[0] * self.dtype.itemsize, self.shape[1:], self.dtype)

    def data_ptr(self):
        return self.ptr

    def dim(self):
        return len(self.shape)

    def stride(self, i):
        return self._strides[i]

    def element_size(self):
        return self.dtype.itemsize


class PinnedArray:
    """Page-locked host memory (rmr_pinned_alloc) as a numpy array: `.a`.  Copies from it to the GPU are single DMAs
    that really run asynchronously; a capture pipeline would write its frames straight into such buffers."""

    def __init__(self, shape, dtype, device=None):
        self.dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * self.dtype.itemsize
        self._p = C.c_void_p()
        if device is None:   # the calling thread's current device
            check(lib().rmr_pinned_alloc(max(n, 1), C.byref(self._p)))
        else:                # the device whose upload ring will read it (one rank per GPU: never GPU 0 by default)
            check(lib().rmr_pinned_alloc_on(int(device), max(n, 1), C.byref(self._p)))
        self.a = np.frombuffer((C.c_char * max(n, 1)).from_address(self._p.value), dtype=self.dtype, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if getattr(self, "_p", None) and lib is not None:
            self.a = None
            lib().rmr_pinned_free(self._p)
            self._p = None

    __del__ = close


class UploadRing:
    """rmr_upload_*: `slots` device buffers filled on a copy stream of their own, so the inputs of step i + 1 travel
    while step i computes (the reference uploads inside its cycle, detector.cu:388-399).
        ring = UploadRing(2, bytes_per_slot)
        dev = ring.begin(slot, [images_block, clouds_block])   # numpy arrays (ideally PinnedArray.a); returns at once
        ring.wait(slot)                                        # before the step that reads the slot
    begin() returns one DeviceArray per block, shaped like the block."""

    def __init__(self, slots, bytes_per_slot, device=0):
        self._h = C.c_void_p()
        self.slots = slots
        check(lib().rmr_upload_create(device, slots, int(bytes_per_slot), C.byref(self._h)))
        self._keep = [None] * slots

    def begin(self, slot, blocks):
        n = len(blocks)
        arrs = [b if (b.flags.c_contiguous) else np.ascontiguousarray(b) for b in blocks]
        src = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        nb = (C.c_size_t * n)(*[a.nbytes for a in arrs])
        out = (C.c_void_p * n)()
        check(lib().rmr_upload_begin(self._h, slot, src, nb, n, out))
        self._keep[slot] = arrs   # the sources must stay alive until the copies have landed
        return [DeviceArray(out[i], a.shape, a.dtype) for i, a in enumerate(arrs)]

    def wait(self, slot):
        check(lib().rmr_upload_wait(self._h, slot))
        self._keep[slot] = None

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().rmr_upload_destroy(self._h)
            self._h = None

    __del__ = close


# ------------------------------------------------------------------------------- whole path

class FrameBatch:
    """The frames and clouds of one run_batch() call, marshalled once: the rmr_image array, the cloud
    pointer table and the point counts that rmr_pipeline_run_batch takes.  A caller that keeps its
    frames in the same buffers (a capture ring, the synthetic bench) builds this once and passes it
    to run_batch() every step; a C++ host fills the same arrays in place and pays nothing per step.
    `clouds`: per frame a [n, >=3] f32 device tensor or numpy array (all of one kind, same row stride)."""

    def __init__(self, images, clouds):
        imgs = list(images)
        self.n = len(imgs)
        if len(clouds) != self.n:
            raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "run_batch: one cloud per image")
        self.images, _, self._keep = _images(imgs, None)
        self.device = _is_device_tensor(clouds[0]) if self.n else False
        self.ptrs = (_lib._fp * max(self.n, 1))()
        self.npts = np.zeros(max(self.n, 1), np.int32)
        stride = None
        for f, c in enumerate(clouds):
            if _is_device_tensor(c) != self.device:
                raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "run_batch: clouds must be all device or all host")
            if self.device:
                if c.dim() != 2 or c.shape[1] < 3 or c.element_size() != 4 or c.stride(1) != 1:
                    raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "device cloud must be [n, >=3] f32")
                p, st = c.data_ptr(), c.stride(0) * 4
                self._keep.append(c)
            else:
                c = np.ascontiguousarray(c, np.float32)
                self._keep.append(c)
                p, st = c.ctypes.data, c.strides[0]
            if stride is None:
                stride = st
            elif st != stride and c.shape[0] > 0:
                raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "run_batch: clouds must share one row stride")
            self.ptrs[f] = C.cast(p, _lib._fp)
            self.npts[f] = c.shape[0]
        self.stride = stride or 16
        self._out = None  # (cap, Robot array, counts): reused from call to call


def run_batch(robot_detector: "RobotDetector", locator, images, clouds=None, forced_crops=None):
    """Throughput mode of SampleRadar::runOnce (sample_radar.h:106-127) over the frames of ONE
    stream -- or, with a list of Locators, of len(locator) streams sharing this GPU and its detector (frames
    stream-major, the same number per stream) -- in one native call: update + cluster of every cloud on a helper thread while the
    two-stage detect runs, then one batched search.  `images` / `clouds`: lists (see FrameBatch), or
    a FrameBatch built once for buffers that are refilled in place (then the returned arrays are
    reused by the next call with that batch too).  Returns (ctypes Robot array [n_frames * max_cars],
    counts) like RobotDetector.detect_batch_raw, robots located."""
    fb = images if isinstance(images, FrameBatch) else FrameBatch(images, clouds)
    n = fb.n
    cap = robot_detector.max_cars
    if fb is images and fb._out is not None and fb._out[0] == cap:
        out, counts = fb._out[1], fb._out[2]
    else:
        out = (_lib.Robot * (cap * n))()
        counts = np.zeros(n, np.int32)
        if fb is images:
            fb._out = (cap, out, counts)
    fc, per = None, 0
    if forced_crops is not None:
        fc = forced_crops if isinstance(forced_crops, np.ndarray) and forced_crops.dtype == np.int32 and \
            forced_crops.ndim == 3 and forced_crops.flags.c_contiguous else \
            np.ascontiguousarray(np.asarray(forced_crops, np.int32).reshape(n, -1, 4))
        per = fc.shape[1]
    if isinstance(locator, (list, tuple)):
        handles = (C.c_void_p * len(locator))(*[l._h for l in locator])
        check(lib().rmr_pipeline_run_streams(robot_detector._h, handles, len(locator), fb.images, fb.ptrs, _lib.ip(fb.npts),
                                             fb.stride, _lib.MEM_DEVICE if fb.device else _lib.MEM_HOST, n,
                                             _lib.ip(fc) if fc is not None else None, per, out, _lib.ip(counts), cap))
        return out, counts
    check(lib().rmr_pipeline_run_batch(robot_detector._h, locator._h, fb.images, fb.ptrs, _lib.ip(fb.npts), fb.stride,
                                       _lib.MEM_DEVICE if fb.device else _lib.MEM_HOST, n,
                                       _lib.ip(fc) if fc is not None else None, per, out, _lib.ip(counts), cap))
    return out, counts


# ------------------------------------------------------------------------------- Tracker (host)

TRACK_TENTATIVE, TRACK_CONFIRMED, TRACK_DELETED = 1, 2, 3


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, np.float32)
    if shape is not None and a.shape != shape:
        raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, f"expected an array of shape {shape}, got {a.shape}")
    return a


class KalmanFilter:
    """radar::track::KalmanFilter / ExtendedKalmanFilter (src/track/kalman_filter.h:77-296).
    Without F, Q, H it is the extended filter: predict(F, Q) and update(z, hx, H) take the
    transition, process noise, predicted measurement and Jacobian evaluated by the caller."""

    def __init__(self, x0, P0, R, F=None, Q=None, H=None):
        x0 = _f32(x0)
        self.n, self.m = len(x0), len(np.atleast_2d(np.asarray(R)))
        P0, R = _f32(P0, (self.n, self.n)), _f32(R, (self.m, self.m))
        model = [None if a is None else _f32(a, sh) for a, sh in
                 ((F, (self.n, self.n)), (Q, (self.n, self.n)), (H, (self.m, self.n)))]
        self._h = C.c_void_p()
        check(lib().rmr_kalman_create(self.n, self.m, _lib.fp(x0), _lib.fp(P0),
                                      *[None if a is None else _lib.fp(a) for a in model],
                                      _lib.fp(R), C.byref(self._h)))

    def predict(self, F=None, Q=None):
        if F is None:
            check(lib().rmr_kalman_predict(self._h))
        else:
            check(lib().rmr_kalman_predict_ekf(self._h, _lib.fp(_f32(F, (self.n, self.n))),
                                               _lib.fp(_f32(Q, (self.n, self.n)))))

    def update(self, z, hx=None, H=None):
        z = _f32(z, (self.m,))
        if hx is None:
            check(lib().rmr_kalman_update(self._h, _lib.fp(z)))
        else:
            check(lib().rmr_kalman_update_ekf(self._h, _lib.fp(z), _lib.fp(_f32(hx, (self.m,))),
                                              _lib.fp(_f32(H, (self.m, self.n)))))

    @property
    def state(self):
        x = np.empty(self.n, np.float32)
        check(lib().rmr_kalman_state(self._h, _lib.fp(x), None))
        return x

    @property
    def covariance(self):
        P = np.empty((self.n, self.n), np.float32)
        check(lib().rmr_kalman_state(self._h, None, _lib.fp(P)))
        return P

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().rmr_kalman_destroy(self._h)
            self._h = None

    __del__ = close


class SingerEKF:
    """radar::track::SingerEKF (src/track/singer.h:33-132): state [x vx ax y vy ay z vz az]."""

    def __init__(self, initial_state, initial_covariance, max_a, tau, observation_noise):
        self._h = C.c_void_p()
        check(lib().rmr_singer_create(_lib.fp(_f32(initial_state, (9,))), _lib.fp(_f32(initial_covariance, (9, 9))),
                                      float(max_a), float(tau), _lib.fp(_f32(observation_noise, (3, 3))),
                                      C.byref(self._h)))

    def predict(self, dt):
        check(lib().rmr_singer_predict(self._h, float(dt)))

    def update(self, measurement):
        check(lib().rmr_singer_update(self._h, _lib.fp(_f32(measurement, (3,)))))

    @property
    def state(self):
        x = np.empty(9, np.float32)
        check(lib().rmr_singer_state(self._h, _lib.fp(x), None))
        return x

    @property
    def covariance(self):
        P = np.empty((9, 9), np.float32)
        check(lib().rmr_singer_state(self._h, None, _lib.fp(P)))
        return P

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().rmr_singer_destroy(self._h)
            self._h = None

    __del__ = close


def auction(value_matrix, max_iter=100):
    """radar::track::auction (src/track/auction.h:49-127): rows are agents, columns tasks;
    returns the task of each agent, -1 where not matched."""
    v = np.ascontiguousarray(value_matrix, np.float32)
    if v.ndim != 2:
        raise InvalidArgument(_lib.ERR_INVALID_ARGUMENT, "auction: value_matrix must be 2-D")
    out = np.full(v.shape[0], -1, np.int32)
    check(lib().rmr_auction(_lib.fp(v), v.shape[0], v.shape[1], int(max_iter), _lib.ip(out)))
    return out


class Tracker:
    """radar::Tracker (src/track/tracker.h:23-54): Singer-EKF tracks, auction matching on a
    distance + class-feature score, tentative / confirmed / deleted bookkeeping.  CPU code, as
    in the reference."""

    def __init__(self, observation_noise, class_num, init_thresh=4, miss_thresh=10, max_acceleration=2.0,
                 acceleration_correlation_time=1.0, distance_weight=0.40, feature_weight=0.60, max_iter=100,
                 distance_thresh=0.8):
        cfg = _lib.TrackerCfg()
        lib().rmr_tracker_cfg_default(C.byref(cfg))
        cfg.observation_noise[:] = [float(v) for v in observation_noise]
        cfg.class_num, cfg.init_thresh, cfg.miss_thresh = class_num, init_thresh, miss_thresh
        cfg.max_acceleration = max_acceleration
        cfg.acceleration_correlation_time = acceleration_correlation_time
        cfg.distance_weight, cfg.feature_weight = distance_weight, feature_weight
        cfg.max_iter, cfg.distance_thresh = max_iter, distance_thresh
        self._h = C.c_void_p()
        check(lib().rmr_tracker_create(C.byref(cfg), C.byref(self._h)))

    def update(self, robots: List[Robot], timestamp) -> List[Robot]:
        """Tracker::update(robots, timestamp) (tracker.cpp:126-220).  `timestamp`: seconds (float)
        or nanoseconds (int).  The robots are updated in place and returned."""
        t_ns = int(timestamp) if isinstance(timestamp, (int, np.integer)) else int(round(float(timestamp) * 1e9))
        arr = (_lib.Robot * max(len(robots), 1))()
        for i, r in enumerate(robots):
            arr[i] = r.to_c()
        check(lib().rmr_tracker_update(self._h, arr, len(robots), t_ns))
        for i, r in enumerate(robots):
            u = Robot.from_c(arr[i])
            r.label, r.location, r.track_state = u.label, u.location, u.track_state
        return robots

    def tracks(self):
        """Live tracks as dicts (id, state, label, init_count, miss_count, location, state_vector)."""
        n = C.c_int()
        check(lib().rmr_tracker_tracks(self._h, None, 0, C.byref(n)))
        arr = (_lib.TrackInfo * max(n.value, 1))()
        check(lib().rmr_tracker_tracks(self._h, arr, n.value, C.byref(n)))
        return [dict(id=t.id, state=t.state, label=t.label, init_count=t.init_count, miss_count=t.miss_count,
                     location=tuple(t.location), state_vector=np.array(t.state_vector[:], np.float32))
                for t in arr[: n.value]]

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().rmr_tracker_destroy(self._h)
            self._h = None

    __del__ = close


class profile:
    """HIP-event timing of the library's own launches (on the streams they run on)."""

    def __init__(self, device=0, flops_only=False):
        self.device = device
        self.level = 2 if flops_only else 1  # 2: only launches that declare FLOPs (the conv family)

    def __enter__(self):
        check(lib().rmr_profile_reset(self.device))
        check(lib().rmr_profile_enable(self.device, self.level))
        return self

    def __exit__(self, *exc):
        check(lib().rmr_profile_enable(self.device, 0))
        return False

    def read(self, by_stage=False):
        """{name: {launches, total_ms, flops, bytes}}.  Launches enqueued by a RobotDetector carry their stage in the name
        ("car|conv n64 ...", "armor|conv n256 ..."); by_stage=False (default) merges the stages under the bare name."""
        n = C.c_int()
        check(lib().rmr_profile_read(self.device, None, 0, C.byref(n)))
        arr = (_lib.KernelStat * max(n.value, 1))()
        check(lib().rmr_profile_read(self.device, arr, n.value, C.byref(n)))
        out = {}
        for i in range(n.value):
            name = arr[i].name.decode()
            if not by_stage and "|" in name:
                name = name.split("|", 1)[1]
            e = out.setdefault(name, {"launches": 0, "total_ms": 0.0, "flops": 0.0, "bytes": 0.0})
            e["launches"] += arr[i].launches
            e["total_ms"] += arr[i].total_ms
            e["flops"] += arr[i].flops
            e["bytes"] += arr[i].bytes
        return out
