"""ONNX initialiser -> weight pack: what replaces the reference's engine build from
``car.onnx`` / ``armor.onnx`` (src/detect/detector.cpp:177-243) and its on-disk cache
(detector.cpp:74-99, 281-311).

The ONNX Runtime / onnx packages are not needed: an ONNX file is a protobuf message and only three
message types matter here (ModelProto.graph = 7, GraphProto.initializer = 5, TensorProto), so this
module walks the wire format directly.  Ultralytics exports fuse BatchNorm into the convolutions and
keep module-path tensor names (``model.0.conv.weight`` ...), which are exactly the names a pack
uses; files whose initialisers were renamed by a graph optimiser are rejected with a clear error.
"""
from __future__ import annotations

import os
import struct
from collections import OrderedDict

import numpy as np

from . import weights as W

_FLOAT, _FLOAT16, _DOUBLE = 1, 10, 11


def _varint(buf, pos):
    val, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf):
    """Yield (field_number, wire_type, value) over one protobuf message."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fn, wt, v


def _tensor(buf):
    """TensorProto -> (name, ndarray float32) or (name, None) for non-float tensors."""
    dims, dtype, name, raw = [], None, "", None
    floats, external = [], False
    for fn, wt, v in _fields(buf):
        if fn == 1:  # dims: repeated int64, packed or not
            if wt == 2:
                p = 0
                while p < len(v):
                    d, p = _varint(v, p)
                    dims.append(d)
            else:
                dims.append(v)
        elif fn == 2:
            dtype = v
        elif fn == 4:  # float_data
            if wt == 2:
                floats.append(np.frombuffer(v, "<f4"))
            else:
                floats.append(np.frombuffer(v, "<f4", 1))
        elif fn == 8:
            name = bytes(v).decode()
        elif fn == 9:
            raw = bytes(v)
        elif fn == 14 and v == 1:  # data_location = EXTERNAL
            external = True
    if external:
        raise ValueError(f"tensor '{name}' uses external data, which is not supported")
    if dtype not in (_FLOAT, _FLOAT16, _DOUBLE):
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, {_FLOAT: "<f4", _FLOAT16: "<f2", _DOUBLE: "<f8"}[dtype])
    elif floats:
        arr = np.concatenate(floats)
    else:
        arr = np.zeros(0, np.float32)
    return name, arr.astype(np.float32).reshape(dims)


def read_initializers(path):
    """All floating-point initialisers of an ONNX file, by name."""
    data = memoryview(open(path, "rb").read())
    out = OrderedDict()
    graph = None
    for fn, wt, v in _fields(data):
        if fn == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError(f"{path}: no graph found (not an ONNX model?)")
    for fn, wt, v in _fields(graph):
        if fn == 5 and wt == 2:
            name, arr = _tensor(v)
            if arr is not None:
                out[name] = arr
    return out


def _infer_scale_nc(tensors):
    w0 = tensors["model.0.conv.weight"].shape[0]
    w9 = tensors["model.8.cv2.conv.weight"].shape[0]
    n2 = sum(1 for k in tensors if k.startswith("model.2.m.") and k.endswith(".cv1.conv.weight"))
    nc = tensors["model.22.cv3.0.2.weight"].shape[0]
    for scale in W.SCALES:
        a = W.arch(scale, nc)
        if a["ch"][0] == w0 and a["ch"][4] == w9 and a["n"][0] == n2:
            return scale, nc
    raise ValueError("the ONNX weights do not match a supported YOLOv8 scale (s, m, l, x)")


def onnx_to_pack(onnx_path, pack_path):
    """Convert an Ultralytics YOLOv8 detection export to a weight pack."""
    init = read_initializers(onnx_path)
    if "model.0.conv.weight" not in init:
        raise ValueError(f"{onnx_path}: initialisers do not carry Ultralytics module names "
                         "(model.0.conv.weight ...); export without renaming graph optimisations")
    scale, nc = _infer_scale_nc(init)
    tensors = OrderedDict()
    for name, cout, cin, k, _ in W.conv_specs(scale, nc):
        for suffix, shape in ((".weight", (cout, cin, k, k)), (".bias", (cout,))):
            key = name + suffix
            if key not in init:
                raise ValueError(f"{onnx_path}: missing tensor '{key}' (BatchNorm not fused into the convs?)")
            if tuple(init[key].shape) != shape:
                raise ValueError(f"{onnx_path}: tensor '{key}' has shape {tuple(init[key].shape)}, expected {shape}")
            tensors[key] = init[key]
    W.save_pack(pack_path, tensors, scale, nc)
    return scale, nc


def ensure_pack(engine_path):
    """The reference's engine-cache logic (detector.cpp:74-99) for packs: use `engine_path` when
    it exists; otherwise build it from the sibling ``.onnx`` file."""
    engine_path = str(engine_path)
    if os.path.exists(engine_path):
        return engine_path
    onnx_path = os.path.splitext(engine_path)[0] + ".onnx"
    if os.path.exists(onnx_path):
        onnx_to_pack(onnx_path, engine_path)
    return engine_path


if __name__ == "__main__":
    import sys

    if len(sys.argv) != 3:
        sys.exit("usage: python -m rm_radar_amd.onnx_import <model.onnx> <out.rmrw>")
    s, n = onnx_to_pack(sys.argv[1], sys.argv[2])
    print(f"wrote {sys.argv[2]}: YOLOv8{s}, {n} classes")
