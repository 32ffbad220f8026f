"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes binding of ``oracle/rmr_oracle.c`` (a plain-C restatement of the reference's
detect+locate hot path; see the header of ``rmr_oracle.h`` for the parity pin).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product package ``rm_radar_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "rmr_oracle.c")
    hdr = os.path.join(_HERE, "rmr_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class PreParam(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("width", "height", "ratio", "dw", "dh")]

    def astuple(self):
        return (self.width, self.height, self.ratio, self.dw, self.dh)


class Detection(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("x", "y", "width", "height", "label", "confidence")]


DET_DTYPE = np.dtype(
    [(n, np.float32) for n in ("x", "y", "width", "height", "label", "confidence")]
)

MAX_ARMORS = 64


class Robot(C.Structure):
    _fields_ = [
        ("rect", C.c_float * 4),
        ("has_label", C.c_int),
        ("label", C.c_int),
        ("confidence", C.c_float),
        ("n_armors", C.c_int),
        ("armors", Detection * MAX_ARMORS),
        ("has_location", C.c_int),
        ("location", C.c_float * 3),
        ("track_state", C.c_int),
    ]


class LocatorCfg(C.Structure):
    _fields_ = [
        ("image_width", C.c_int),
        ("image_height", C.c_int),
        ("intrinsic", C.c_float * 9),
        ("lidar_to_camera", C.c_float * 16),
        ("world_to_camera", C.c_float * 16),
        ("zoom_factor", C.c_float),
        ("queue_size", C.c_int),
        ("min_depth_diff", C.c_float),
        ("max_depth_diff", C.c_float),
        ("cluster_tolerance", C.c_float),
        ("min_cluster_size", C.c_int),
        ("max_cluster_size", C.c_int),
        ("max_distance", C.c_float),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        u8p = C.POINTER(C.c_uint8)
        ip = C.POINTER(C.c_int)
        L.orc_preparam_make.argtypes = [C.c_int] * 4 + [C.POINTER(PreParam)]
        L.orc_letterbox_geometry.argtypes = [C.POINTER(PreParam)] + [ip] * 6
        L.orc_resize_u8.argtypes = [u8p, u8p] + [C.c_int] * 5
        L.orc_copy_make_border_u8.argtypes = [u8p, u8p] + [C.c_int] * 7
        L.orc_blob.argtypes = [u8p, fp, C.c_int, C.c_int, C.c_int, C.c_float]
        L.orc_transpose.argtypes = [fp, fp, C.c_int, C.c_int]
        L.orc_decode.argtypes = [fp, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_iou.argtypes = [C.c_float] * 8
        L.orc_iou.restype = C.c_float
        L.orc_nms.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int]
        L.orc_restore.argtypes = [C.POINTER(Detection), C.POINTER(PreParam)]
        L.orc_preprocess.argtypes = [u8p] + [C.c_int] * 7 + [fp, C.POINTER(PreParam)]
        L.orc_postprocess.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                      C.POINTER(PreParam), C.c_void_p, C.c_int]
        L.orc_postprocess.restype = C.c_int
        L.orc_robot_set_detection.argtypes = [C.POINTER(Robot), C.POINTER(Detection),
                                              C.c_void_p, C.c_int]
        L.orc_rect_round.argtypes = [fp, ip]
        L.orc_compute_iou_bounding.argtypes = [fp, fp]
        L.orc_compute_iou_bounding.restype = C.c_float
        L.orc_group_robots.argtypes = [C.POINTER(Robot), C.c_int, C.c_float, C.POINTER(Robot)]
        L.orc_group_robots.restype = C.c_int
        L.orc_crop_rect.argtypes = [C.POINTER(Detection), ip]
        L.orc_locator_cfg_default.argtypes = [C.POINTER(LocatorCfg)]
        L.orc_locator_create.argtypes = [C.POINTER(LocatorCfg)]
        L.orc_locator_create.restype = C.c_void_p
        L.orc_locator_destroy.argtypes = [C.c_void_p]
        L.orc_locator_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_locator_cluster.argtypes = [C.c_void_p]
        L.orc_locator_search.argtypes = [C.c_void_p, fp, fp]
        L.orc_locator_search.restype = C.c_int
        L.orc_locator_zoom.argtypes = [C.c_void_p, ip, ip]
        for name in ("lidar_to_world", "camera_to_lidar", "lidar_to_camera"):
            getattr(L, "orc_locator_" + name).argtypes = [C.c_void_p, fp, fp]
        for name in ("width", "height", "num_foreground", "num_clusters"):
            f = getattr(L, "orc_locator_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = C.c_int
        for name in ("depth_image", "background_image", "diff_image", "foreground_xyz"):
            f = getattr(L, "orc_locator_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = fp
        for name in ("foreground_pixel", "foreground_cluster"):
            f = getattr(L, "orc_locator_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = ip
        L.orc_locator_cluster_size.argtypes = [C.c_void_p, C.c_int]
        L.orc_locator_cluster_size.restype = C.c_int
        L.orc_inv3x3.argtypes = [fp, fp]
        L.orc_inv4x4.argtypes = [fp, fp]
        L.orc_conv2d_nchw.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


# --------------------------------------------------------------------------- detect

def preparam(in_w, in_h, out_w=640, out_h=640) -> PreParam:
    p = PreParam()
    lib().orc_preparam_make(in_w, in_h, out_w, out_h, C.byref(p))
    return p


def letterbox_geometry(p: PreParam):
    v = [C.c_int() for _ in range(6)]
    lib().orc_letterbox_geometry(C.byref(p), *[C.byref(x) for x in v])
    return tuple(x.value for x in v)  # rw, rh, top, bottom, left, right


def resize(src: np.ndarray, dst_w: int, dst_h: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    h, w, c = src.shape
    dst = np.empty((dst_h, dst_w, c), np.uint8)
    lib().orc_resize_u8(_u8(src), _u8(dst), c, w, h, dst_w, dst_h)
    return dst


def copy_make_border(src: np.ndarray, top, bottom, left, right) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    h, w, c = src.shape
    dst = np.empty((h + top + bottom, w + left + right, c), np.uint8)
    lib().orc_copy_make_border_u8(_u8(src), _u8(dst), c, w, h, top, bottom, left, right)
    return dst


def blob(src: np.ndarray, scale: float) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    h, w, c = src.shape
    dst = np.empty((c, h, w), np.float32)
    lib().orc_blob(_u8(src), _fp(dst), w, h, c, scale)
    return dst


def transpose(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src, np.float32)
    r, c = src.shape
    dst = np.empty((c, r), np.float32)
    lib().orc_transpose(_fp(src), _fp(dst), r, c)
    return dst


def decode(src_ac: np.ndarray, classes: int) -> np.ndarray:
    src = np.ascontiguousarray(src_ac, np.float32)
    a, ch = src.shape
    out = np.empty(a, DET_DTYPE)
    lib().orc_decode(_fp(src), out.ctypes.data, ch, a, classes)
    return out


def iou(a, b) -> float:
    return lib().orc_iou(*[float(v) for v in a], *[float(v) for v in b])


def nms(dets: np.ndarray, nms_thresh: float, score_thresh: float) -> np.ndarray:
    dets = np.ascontiguousarray(dets.copy())
    lib().orc_nms(dets.ctypes.data, nms_thresh, score_thresh, len(dets))
    return dets


def preprocess(img: np.ndarray, crop=None, out_w=640, out_h=640):
    """img: HxWx3 BGR u8 -> (blob f32 [3,out_h,out_w], PreParam)."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    assert img.strides[1] == 3 and img.strides[2] == 1
    h, w, _ = img.shape
    if crop is None:
        crop = (0, 0, w, h)
    out = np.empty((3, out_h, out_w), np.float32)
    p = PreParam()
    lib().orc_preprocess(_u8(img), img.strides[0], *[int(v) for v in crop], out_w, out_h,
                         _fp(out), C.byref(p))
    return out, p


def postprocess(net_out: np.ndarray, classes: int, nms_thresh: float, conf_thresh: float,
                p: PreParam) -> np.ndarray:
    """net_out: [4+classes, anchors] f32 -> structured array of surviving detections."""
    net_out = np.ascontiguousarray(net_out, np.float32)
    ch, a = net_out.shape
    out = np.empty(a, DET_DTYPE)
    n = lib().orc_postprocess(_fp(net_out), ch, a, classes, nms_thresh, conf_thresh,
                              C.byref(p), out.ctypes.data, a)
    return out[:n].copy()


def restore(det, p: PreParam):
    d = Detection(*[float(v) for v in det])
    lib().orc_restore(C.byref(d), C.byref(p))
    return (d.x, d.y, d.width, d.height, d.label, d.confidence)


# --------------------------------------------------------------------------- grouping

def make_robot(car, armors: np.ndarray) -> Robot:
    r = Robot()
    c = Detection(*[float(v) for v in car])
    armors = np.ascontiguousarray(armors, DET_DTYPE)
    lib().orc_robot_set_detection(C.byref(r), C.byref(c), armors.ctypes.data, len(armors))
    return r


def group_robots(robots, iou_thresh: float):
    n = len(robots)
    arr = (Robot * max(n, 1))(*robots)
    out = (Robot * max(n, 1))()
    m = lib().orc_group_robots(arr, n, iou_thresh, out)
    return [out[i] for i in range(m)]


def rect_round(rect):
    r = np.asarray(rect, np.float32)
    o = np.empty(4, np.int32)
    lib().orc_rect_round(_fp(r), _ip(o))
    return tuple(int(v) for v in o)


def compute_iou_bounding(a, b) -> float:
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return lib().orc_compute_iou_bounding(_fp(a), _fp(b))


def crop_rect(car):
    c = Detection(*[float(v) for v in car])
    o = np.empty(4, np.int32)
    lib().orc_crop_rect(C.byref(c), _ip(o))
    return tuple(int(v) for v in o)


# --------------------------------------------------------------------------- locator

class Locator:
    """Mirror of radar::Locator (src/locate/locator.h:53-98) on the CPU oracle."""

    def __init__(self, image_width, image_height, intrinsic, lidar_to_camera, world_to_camera,
                 zoom_factor=0.5, queue_size=3, min_depth_diff=500.0, max_depth_diff=4000.0,
                 cluster_tolerance=400.0, min_cluster_size=8, max_cluster_size=1000,
                 max_distance=29300.0):
        cfg = LocatorCfg()
        lib().orc_locator_cfg_default(C.byref(cfg))
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
        self._h = lib().orc_locator_create(C.byref(cfg))
        self.wz = lib().orc_locator_width(self._h)
        self.hz = lib().orc_locator_height(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_locator_destroy(self._h)
            self._h = None

    def update(self, cloud):
        if cloud is None or len(cloud) == 0:
            lib().orc_locator_update(self._h, None, 0, 0)
            return
        cloud = np.ascontiguousarray(cloud, np.float32)
        lib().orc_locator_update(self._h, cloud.ctypes.data, cloud.shape[0], cloud.strides[0])

    def cluster(self):
        lib().orc_locator_cluster(self._h)

    def search(self, rect):
        r = np.asarray(rect, np.float32)
        o = np.zeros(3, np.float32)
        ok = lib().orc_locator_search(self._h, _fp(r), _fp(o))
        return o if ok else None

    def zoom(self, rect):
        r = np.asarray(rect, np.int32)
        o = np.zeros(4, np.int32)
        lib().orc_locator_zoom(self._h, _ip(r), _ip(o))
        return tuple(int(v) for v in o)

    def _xf(self, name, p):
        a = np.asarray(p, np.float32)
        o = np.zeros(3, np.float32)
        getattr(lib(), "orc_locator_" + name)(self._h, _fp(a), _fp(o))
        return o

    def lidar_to_world(self, p):
        return self._xf("lidar_to_world", p)

    def camera_to_lidar(self, p):
        return self._xf("camera_to_lidar", p)

    def lidar_to_camera(self, p):
        return self._xf("lidar_to_camera", p)

    def _img(self, name):
        ptr = getattr(lib(), "orc_locator_" + name)(self._h)
        return np.ctypeslib.as_array(ptr, shape=(self.hz, self.wz))

    @property
    def depth_image(self):
        return self._img("depth_image")

    @property
    def background_image(self):
        return self._img("background_image")

    @property
    def diff_image(self):
        return self._img("diff_image")

    @property
    def num_clusters(self):
        return lib().orc_locator_num_clusters(self._h)

    def cluster_size(self, i):
        return lib().orc_locator_cluster_size(self._h, i)

    def foreground(self):
        n = lib().orc_locator_num_foreground(self._h)
        if n == 0:
            return (np.zeros((0, 3), np.float32), np.zeros(0, np.int32), np.zeros(0, np.int32))
        xyz = np.ctypeslib.as_array(lib().orc_locator_foreground_xyz(self._h), shape=(n, 3)).copy()
        pix = np.ctypeslib.as_array(lib().orc_locator_foreground_pixel(self._h), shape=(n,)).copy()
        cid = np.ctypeslib.as_array(lib().orc_locator_foreground_cluster(self._h), shape=(n,)).copy()
        return xyz, pix, cid


def inv3x3(a):
    a = np.ascontiguousarray(a, np.float32).reshape(9)
    o = np.empty(9, np.float32)
    lib().orc_inv3x3(_fp(a), _fp(o))
    return o.reshape(3, 3)


def inv4x4(a):
    a = np.ascontiguousarray(a, np.float32).reshape(16)
    o = np.empty(16, np.float32)
    lib().orc_inv4x4(_fp(a), _fp(o))
    return o.reshape(4, 4)


def conv2d_nchw(x, w, b, stride, pad, silu):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    n, cin, h, wd = x.shape
    cout, _, kh, kw = w.shape
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    y = np.empty((n, cout, ho, wo), np.float32)
    bp = _fp(np.ascontiguousarray(b, np.float32)) if b is not None else None
    lib().orc_conv2d_nchw(_fp(x), n, cin, h, wd, _fp(w), bp, cout, kh, kw, stride, pad,
                          int(silu), _fp(y))
    return y


def num_threads() -> int:
    return lib().orc_num_threads()
