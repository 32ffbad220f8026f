"""ctypes binding of librmr.so (include/rmr.h).  No torch, no oracle: plain pointers only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RMR_LIB: another build of the same library (tools/experiments: `make EXPERIMENTS=1` writes _build_exp/librmr.so)
LIB_PATH = os.environ.get("RMR_LIB") or os.path.join(_HERE, "_build", "librmr.so")

OK, ERR_INVALID_ARGUMENT, ERR_RUNTIME, ERR_LOGIC, ERR_DEVICE, ERR_CAPACITY = range(6)
MEM_HOST, MEM_DEVICE = 0, 1
FMT_U8_HWC, FMT_F32_NCHW = 0, 1
MAX_ARMORS = 64


class RmrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librmr status {code}: {msg}")
        self.code = code


class InvalidArgument(RmrError, ValueError):
    """reference: std::invalid_argument"""


class DeviceError(RmrError):
    """reference: CUDA_CHECK failure"""


class CapacityError(RmrError):
    pass


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int), ("height", C.c_int),
                ("stride", C.c_int), ("mem", C.c_int)]


class Detection(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("x", "y", "width", "height", "label", "confidence")]


DET_DTYPE = np.dtype([(n, np.float32) for n in ("x", "y", "width", "height", "label", "confidence")])


class PreParam(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("width", "height", "ratio", "dw", "dh")]

    def astuple(self):
        return (self.width, self.height, self.ratio, self.dw, self.dh)


class Robot(C.Structure):
    _fields_ = [("rect", C.c_float * 4), ("has_label", C.c_int), ("label", C.c_int),
                ("confidence", C.c_float), ("n_armors", C.c_int),
                ("armors", Detection * MAX_ARMORS), ("has_location", C.c_int),
                ("location", C.c_float * 3), ("track_state", C.c_int)]


class RobotRecord(C.Structure):
    _fields_ = [("rect", C.c_float * 4), ("location", C.c_float * 3), ("confidence", C.c_float),
                ("label", C.c_int32), ("flags", C.c_int32), ("stream_id", C.c_int32),
                ("frame_id", C.c_int32)]


class DetectorCfg(C.Structure):
    _fields_ = [("engine_path", C.c_char_p), ("classes", C.c_int), ("image_width", C.c_int),
                ("image_height", C.c_int), ("max_batch_size", C.c_int),
                ("opt_batch_size", C.c_int), ("nms_thresh", C.c_float),
                ("conf_thresh", C.c_float), ("input_width", C.c_int), ("input_height", C.c_int),
                ("input_channels", C.c_int), ("device", C.c_int), ("precision", C.c_int)]


class RobotDetectorCfg(C.Structure):
    _fields_ = [("car_engine_path", C.c_char_p), ("armor_engine_path", C.c_char_p),
                ("image_width", C.c_int), ("image_height", C.c_int), ("armor_classes", C.c_int),
                ("max_cars", C.c_int), ("opt_cars", C.c_int), ("iou_thresh", C.c_float),
                ("car_nms_thresh", C.c_float), ("car_conf_thresh", C.c_float),
                ("armor_nms_thresh", C.c_float), ("armor_conf_thresh", C.c_float),
                ("input_width", C.c_int), ("input_height", C.c_int), ("input_channels", C.c_int),
                ("device", C.c_int), ("max_frames", C.c_int), ("precision", C.c_int)]


class LocatorCfg(C.Structure):
    _fields_ = [("image_width", C.c_int), ("image_height", C.c_int),
                ("intrinsic", C.c_float * 9), ("lidar_to_camera", C.c_float * 16),
                ("world_to_camera", C.c_float * 16), ("zoom_factor", C.c_float),
                ("queue_size", C.c_int), ("min_depth_diff", C.c_float),
                ("max_depth_diff", C.c_float), ("cluster_tolerance", C.c_float),
                ("min_cluster_size", C.c_int), ("max_cluster_size", C.c_int),
                ("max_distance", C.c_float), ("device", C.c_int), ("max_points", C.c_int),
                ("max_foreground", C.c_int), ("max_frames", C.c_int)]


class TrackerCfg(C.Structure):
    _fields_ = [("observation_noise", C.c_float * 3), ("class_num", C.c_int), ("init_thresh", C.c_int),
                ("miss_thresh", C.c_int), ("max_acceleration", C.c_float),
                ("acceleration_correlation_time", C.c_float), ("distance_weight", C.c_float),
                ("feature_weight", C.c_float), ("max_iter", C.c_int), ("distance_thresh", C.c_float)]


class TrackInfo(C.Structure):
    _fields_ = [("id", C.c_int), ("state", C.c_int), ("label", C.c_int), ("init_count", C.c_int),
                ("miss_count", C.c_int), ("location", C.c_float * 3), ("state_vector", C.c_float * 9)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_longlong), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/rmr.h declares: (restype, argtypes)
_fp, _ip, _vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
_P = C.POINTER
SYMBOLS = {
    "rmr_last_error": (C.c_char_p, []),
    "rmr_abi_version": (C.c_int, []),
    "rmr_device_count": (C.c_int, []),
    "rmr_preparam_make": (C.c_int, [C.c_int] * 4 + [_P(PreParam)]),
    "rmr_letterbox_geometry": (C.c_int, [_P(PreParam), _ip, _ip, _ip, _ip]),
    "rmr_restore_detection": (C.c_int, [_P(Detection), _P(PreParam)]),
    "rmr_letterbox": (C.c_int, [C.c_int, _P(Image), _ip, C.c_int] + [C.c_int] * 7 +
                      [C.c_float, C.c_int, _vp]),
    "rmr_preprocess": (C.c_int, [C.c_int, _P(Image), _ip, C.c_int, C.c_int, C.c_int, _fp,
                                 _P(PreParam)]),
    "rmr_postprocess": (C.c_int, [C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_float, _P(PreParam), _vp, _ip, C.c_int]),
    "rmr_transpose": (C.c_int, [C.c_int, _fp, _fp, C.c_int, C.c_int]),
    "rmr_conv2d": (C.c_int, [C.c_int, _fp] + [C.c_int] * 4 + [_fp, _fp] + [C.c_int] * 6 +
                   [_fp, _fp, C.c_int]),
    "rmr_conv_bench": (C.c_int, [C.c_int] * 11 + [_fp]),
    "rmr_f32_to_e4m3": (C.c_int, [_fp, C.c_int, _vp]),
    "rmr_quant_e4m3": (C.c_int, [C.c_int, _fp, C.c_int, _vp]),
    "rmr_stream_owner": (C.c_int, [C.c_int, C.c_int]),
    "rmr_streams_of_rank": (C.c_int, [C.c_int, C.c_int, C.c_int, _ip, C.c_int]),
    "rmr_comm_unique_id": (C.c_int, [C.c_int, C.c_char_p]),
    "rmr_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, _P(_vp)]),
    "rmr_comm_destroy": (None, [_vp]),
    "rmr_comm_all_gather_records": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "rmr_pack_robot_records": (C.c_int, [_vp, _ip, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "rmr_detector_cfg_default": (None, [_P(DetectorCfg)]),
    "rmr_detector_create": (C.c_int, [_P(DetectorCfg), _P(_vp)]),
    "rmr_detector_destroy": (None, [_vp]),
    "rmr_detector_detect": (C.c_int, [_vp, _P(Image), _ip, C.c_int, _vp, _ip, C.c_int]),
    "rmr_detector_infer": (C.c_int, [_vp, _P(Image), _ip, C.c_int, _fp, _P(PreParam)]),
    "rmr_detector_read_feature": (C.c_int, [_vp, C.c_char_p, C.c_int, _fp, _ip]),
    "rmr_detector_arena_bytes": (C.c_double, [_vp]),
    "rmr_robot_detector_arena_bytes": (C.c_double, [_vp]),
    "rmr_detector_chunk": (C.c_int, [_vp]),
    "rmr_detector_anchors": (C.c_int, [_vp]),
    "rmr_detector_channels": (C.c_int, [_vp]),
    "rmr_detector_flops_per_image": (C.c_double, [_vp]),
    "rmr_robot_detector_cfg_default": (None, [_P(RobotDetectorCfg)]),
    "rmr_robot_detector_create": (C.c_int, [_P(RobotDetectorCfg), _P(_vp)]),
    "rmr_robot_detector_destroy": (None, [_vp]),
    "rmr_robot_detector_detect": (C.c_int, [_vp, _P(Image), _P(Robot), _ip, C.c_int]),
    "rmr_robot_detector_detect_batch": (C.c_int, [_vp, _P(Image), C.c_int, _ip, C.c_int,
                                                  _P(Robot), _ip, C.c_int]),
    "rmr_tune_file_version": (C.c_int, []),
    "rmr_pinned_alloc": (C.c_int, [C.c_size_t, _P(_vp)]),
    "rmr_pinned_alloc_on": (C.c_int, [C.c_int, C.c_size_t, _P(_vp)]),
    "rmr_pinned_free": (None, [_vp]),
    "rmr_upload_create": (C.c_int, [C.c_int, C.c_int, C.c_size_t, _P(_vp)]),
    "rmr_upload_destroy": (None, [_vp]),
    "rmr_upload_begin": (C.c_int, [_vp, C.c_int, _P(_vp), _P(C.c_size_t), C.c_int, _P(_vp)]),
    "rmr_upload_wait": (C.c_int, [_vp, C.c_int]),
    "rmr_robot_detector_read_heads": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _fp, _P(PreParam), _ip]),
    "rmr_robot_set_detection": (C.c_int, [_P(Robot), _P(Detection), _vp, C.c_int]),
    "rmr_compute_iou": (C.c_float, [_fp, _fp]),
    "rmr_group_robots": (C.c_int, [_P(Robot), C.c_int, C.c_float, _P(Robot), _ip]),
    "rmr_locator_cfg_default": (None, [_P(LocatorCfg)]),
    "rmr_locator_create": (C.c_int, [_P(LocatorCfg), _P(_vp)]),
    "rmr_locator_destroy": (None, [_vp]),
    "rmr_locator_update": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    "rmr_locator_cluster": (C.c_int, [_vp]),
    "rmr_locator_search": (C.c_int, [_vp, _P(Robot), C.c_int]),
    "rmr_locator_keep": (C.c_int, [_vp, C.c_int]),
    "rmr_locator_search_kept": (C.c_int, [_vp, C.c_int, _P(Robot), C.c_int]),
    "rmr_locator_search_batch": (C.c_int, [_vp, _P(Robot), _ip, C.c_int, C.c_int]),
    "rmr_locator_update_cluster_batch": (C.c_int, [_vp, _P(_vp), _ip, C.c_int, C.c_int, C.c_int]),
    "rmr_locator_width": (C.c_int, [_vp]),
    "rmr_locator_height": (C.c_int, [_vp]),
    "rmr_locator_read_image": (C.c_int, [_vp, C.c_int, _fp]),
    "rmr_locator_write_image": (C.c_int, [_vp, C.c_int, _fp]),
    "rmr_locator_state_bytes": (C.c_int, [_vp, _P(C.c_size_t)]),
    "rmr_locator_save_state": (C.c_int, [_vp, _vp, C.c_size_t]),
    "rmr_locator_load_state": (C.c_int, [_vp, _vp, C.c_size_t]),
    "rmr_locator_transform": (C.c_int, [_vp, C.c_int, _fp, _fp]),
    "rmr_locator_zoom": (C.c_int, [_vp, _ip, _ip]),
    "rmr_locator_foreground": (C.c_int, [_vp, _fp, _ip, _ip, C.c_int, _ip]),
    "rmr_locator_num_clusters": (C.c_int, [_vp]),
    "rmr_pipeline_run_batch": (C.c_int, [_vp, _vp, _P(Image), _P(_fp), _ip, C.c_int, C.c_int, C.c_int, _ip, C.c_int,
                                         _P(Robot), _ip, C.c_int]),
    "rmr_pipeline_run_streams": (C.c_int, [_vp, _P(_vp), C.c_int, _P(Image), _P(_fp), _ip, C.c_int, C.c_int, C.c_int, _ip,
                                           C.c_int, _P(Robot), _ip, C.c_int]),
    "rmr_profile_enable": (C.c_int, [C.c_int, C.c_int]),
    "rmr_profile_reset": (C.c_int, [C.c_int]),
    "rmr_profile_read": (C.c_int, [C.c_int, _P(KernelStat), C.c_int, _ip]),
    # tracker stage (host only)
    "rmr_kalman_create": (C.c_int, [C.c_int, C.c_int] + [_fp] * 6 + [_P(C.c_void_p)]),
    "rmr_kalman_destroy": (None, [C.c_void_p]),
    "rmr_kalman_predict": (C.c_int, [C.c_void_p]),
    "rmr_kalman_update": (C.c_int, [C.c_void_p, _fp]),
    "rmr_kalman_predict_ekf": (C.c_int, [C.c_void_p, _fp, _fp]),
    "rmr_kalman_update_ekf": (C.c_int, [C.c_void_p, _fp, _fp, _fp]),
    "rmr_kalman_state": (C.c_int, [C.c_void_p, _fp, _fp]),
    "rmr_singer_create": (C.c_int, [_fp, _fp, C.c_float, C.c_float, _fp, _P(C.c_void_p)]),
    "rmr_singer_destroy": (None, [C.c_void_p]),
    "rmr_singer_predict": (C.c_int, [C.c_void_p, C.c_float]),
    "rmr_singer_update": (C.c_int, [C.c_void_p, _fp]),
    "rmr_singer_state": (C.c_int, [C.c_void_p, _fp, _fp]),
    "rmr_auction": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _ip]),
    "rmr_robot_feature": (C.c_int, [_P(Robot), C.c_int, _fp]),
    "rmr_tracker_cfg_default": (None, [_P(TrackerCfg)]),
    "rmr_tracker_create": (C.c_int, [_P(TrackerCfg), _P(C.c_void_p)]),
    "rmr_tracker_destroy": (None, [C.c_void_p]),
    "rmr_tracker_update": (C.c_int, [C.c_void_p, _P(Robot), C.c_int, C.c_int64]),
    "rmr_tracker_tracks": (C.c_int, [C.c_void_p, _P(TrackInfo), C.c_int, _ip]),
}

_lib = None
MISSING: list = []


def lib():
    """Load librmr.so; fail loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()')")
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.  Two HIP
        # runtimes in one process cannot both open the GPU, so when torch is installed load it
        # FIRST: librmr.so's NEEDED libamdhip64.so.7 then resolves to the runtime torch uses and
        # device pointers / streams can be shared with torch tensors and torch.distributed.
        if os.environ.get("RMR_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                f = getattr(L, name)
            except AttributeError:  # tests/test_abi.py requires this list to stay empty
                MISSING.append(name)
                continue
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(status: int):
    if status == OK:
        return
    msg = lib().rmr_last_error().decode(errors="replace")
    cls = {ERR_INVALID_ARGUMENT: InvalidArgument, ERR_DEVICE: DeviceError,
           ERR_CAPACITY: CapacityError}.get(status, RmrError)
    raise cls(status, msg)


def fp(a):
    return a.ctypes.data_as(_fp)


def ip(a):
    return a.ctypes.data_as(_ip)
