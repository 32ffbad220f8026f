"""Weight packs (*.rmrw) -- what ``engine_path`` names for this library.

The reference builds a TensorRT engine from ``car.onnx`` / ``armor.onnx`` (detector.cpp:177-243)
and caches it on disk (detector.cpp:74-99).  Both ONNX files are absent from the reference tree,
so this module (a) defines the flat tensor file librmr.so loads, and (b) synthesises seeded
YOLOv8 weights of the public Ultralytics architecture (BatchNorm already folded into conv
weight + bias) for parity tests and benchmarks.  numpy only.

File layout (little endian):
    magic  'RMRW'  u32 version(=1)
    f32 depth_multiple, f32 width_multiple, u32 max_channels, u32 nc, u32 reg_max
    u32 n_tensors
    n_tensors x { u32 name_len, name bytes, u32 ndim, u32 dims[ndim], f32 data[prod(dims)] }
Tensor names follow Ultralytics' module paths with BN folded away, e.g.
``model.0.conv.weight`` / ``model.0.conv.bias``, ``model.2.m.0.cv1.conv.weight``,
``model.22.cv2.0.2.weight`` (plain conv with bias).
"""
from __future__ import annotations

import math
import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"RMRW"
SCALES = {  # depth, width, max_channels  (Ultralytics yolov8.yaml)
    "s": (0.33, 0.50, 1024),
    "m": (0.67, 0.75, 768),
    "l": (1.00, 1.00, 512),
    "x": (1.00, 1.25, 512),
}


# per-layer gains from tools/calibrate_gains.py (scale m): unit post-activation std on
# uniform-noise input, DFL / class logits std 1.5
GAINS = {
    "model.0.conv": 5.1261,
    "model.1.conv": 2.4990,
    "model.2.cv1.conv": 2.3675,
    "model.2.m.0.cv1.conv": 1.3338,
    "model.2.m.0.cv2.conv": 1.3132,
    "model.2.m.1.cv1.conv": 0.5729,
    "model.2.m.1.cv2.conv": 7.6461,
    "model.2.cv2.conv": 0.8697,
    "model.3.conv": 2.2601,
    "model.4.cv1.conv": 1.8350,
    "model.4.m.0.cv1.conv": 2.2417,
    "model.4.m.0.cv2.conv": 1.5082,
    "model.4.m.1.cv1.conv": 0.8218,
    "model.4.m.1.cv2.conv": 2.1527,
    "model.4.m.2.cv1.conv": 0.3883,
    "model.4.m.2.cv2.conv": 1.9630,
    "model.4.m.3.cv1.conv": 0.4410,
    "model.4.m.3.cv2.conv": 1.6301,
    "model.4.cv2.conv": 0.4377,
    "model.5.conv": 2.2674,
    "model.6.cv1.conv": 2.0937,
    "model.6.m.0.cv1.conv": 1.8660,
    "model.6.m.0.cv2.conv": 1.6527,
    "model.6.m.1.cv1.conv": 1.1910,
    "model.6.m.1.cv2.conv": 1.8289,
    "model.6.m.2.cv1.conv": 0.5989,
    "model.6.m.2.cv2.conv": 1.1163,
    "model.6.m.3.cv1.conv": 0.3110,
    "model.6.m.3.cv2.conv": 2.4019,
    "model.6.cv2.conv": 0.5499,
    "model.7.conv": 2.0894,
    "model.8.cv1.conv": 1.9785,
    "model.8.m.0.cv1.conv": 2.3682,
    "model.8.m.0.cv2.conv": 1.5393,
    "model.8.m.1.cv1.conv": 0.8045,
    "model.8.m.1.cv2.conv": 1.9544,
    "model.8.cv2.conv": 1.1495,
    "model.9.cv1.conv": 2.0009,
    "model.9.cv2.conv": 1.2752,
    "model.12.cv1.conv": 1.8071,
    "model.12.m.0.cv1.conv": 1.9639,
    "model.12.m.0.cv2.conv": 1.7842,
    "model.12.m.1.cv1.conv": 1.9643,
    "model.12.m.1.cv2.conv": 1.7475,
    "model.12.cv2.conv": 2.2504,
    "model.15.cv1.conv": 2.0883,
    "model.15.m.0.cv1.conv": 2.3896,
    "model.15.m.0.cv2.conv": 1.4702,
    "model.15.m.1.cv1.conv": 1.4743,
    "model.15.m.1.cv2.conv": 2.7930,
    "model.15.cv2.conv": 1.8973,
    "model.16.conv": 2.5577,
    "model.18.cv1.conv": 2.3968,
    "model.18.m.0.cv1.conv": 1.7029,
    "model.18.m.0.cv2.conv": 1.8357,
    "model.18.m.1.cv1.conv": 2.4765,
    "model.18.m.1.cv2.conv": 1.9819,
    "model.18.cv2.conv": 2.1528,
    "model.19.conv": 2.2015,
    "model.21.cv1.conv": 2.1947,
    "model.21.m.0.cv1.conv": 1.9585,
    "model.21.m.0.cv2.conv": 1.9546,
    "model.21.m.1.cv1.conv": 2.3910,
    "model.21.m.1.cv2.conv": 1.8386,
    "model.21.cv2.conv": 1.9997,
    "model.22.cv2.0.0.conv": 1.6220,
    "model.22.cv2.0.1.conv": 2.3455,
    "model.22.cv2.0.2": 1.8183,
    "model.22.cv3.0.0.conv": 2.0360,
    "model.22.cv3.0.1.conv": 1.8492,
    "model.22.cv3.0.2": 1.4662,
    "model.22.cv2.1.0.conv": 2.7710,
    "model.22.cv2.1.1.conv": 2.6901,
    "model.22.cv2.1.2": 2.3255,
    "model.22.cv3.1.0.conv": 2.6705,
    "model.22.cv3.1.1.conv": 2.6942,
    "model.22.cv3.1.2": 1.8508,
    "model.22.cv2.2.0.conv": 2.0483,
    "model.22.cv2.2.1.conv": 1.3081,
    "model.22.cv2.2.2": 2.6834,
    "model.22.cv3.2.0.conv": 2.4459,
    "model.22.cv3.2.1.conv": 2.0904,
    "model.22.cv3.2.2": 6.2443,
}


def make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


def arch(scale="m", nc=1):
    """Channel plan of YOLOv8 at a scale: returns dict(ch=[c1..c5], n=[n1..n4], nh, c2, c3)."""
    depth, width, max_ch = SCALES[scale]
    ch = [make_divisible(min(c, max_ch) * width, 8) for c in (64, 128, 256, 512, 1024)]
    n = [max(round(r * depth), 1) for r in (3, 6, 6, 3)]
    nh = max(round(3 * depth), 1)
    c2 = max(16, ch[2] // 4, 64)
    c3 = max(ch[2], min(nc, 100))
    return dict(ch=ch, n=n, nh=nh, c2=c2, c3=c3, depth=depth, width=width, max_ch=max_ch)


def conv_specs(scale="m", nc=1):
    """Ordered list of (name, cout, cin, k, has_act) for every conv of the network."""
    a = arch(scale, nc)
    c1, c2_, c3_, c4, c5 = a["ch"]
    specs = []

    def conv(name, cin, cout, k):
        specs.append((name + ".conv", cout, cin, k, True))

    def c2f(name, cin, cout, n):
        c = cout // 2
        conv(f"{name}.cv1", cin, 2 * c, 1)
        for i in range(n):
            conv(f"{name}.m.{i}.cv1", c, c, 3)
            conv(f"{name}.m.{i}.cv2", c, c, 3)
        conv(f"{name}.cv2", (2 + n) * c, cout, 1)

    conv("model.0", 3, c1, 3)
    conv("model.1", c1, c2_, 3)
    c2f("model.2", c2_, c2_, a["n"][0])
    conv("model.3", c2_, c3_, 3)
    c2f("model.4", c3_, c3_, a["n"][1])
    conv("model.5", c3_, c4, 3)
    c2f("model.6", c4, c4, a["n"][2])
    conv("model.7", c4, c5, 3)
    c2f("model.8", c5, c5, a["n"][3])
    conv("model.9.cv1", c5, c5 // 2, 1)
    conv("model.9.cv2", c5 // 2 * 4, c5, 1)
    c2f("model.12", c5 + c4, c4, a["nh"])
    c2f("model.15", c4 + c3_, c3_, a["nh"])
    conv("model.16", c3_, c3_, 3)
    c2f("model.18", c3_ + c4, c4, a["nh"])
    conv("model.19", c4, c4, 3)
    c2f("model.21", c4 + c5, c5, a["nh"])
    for i, cin in enumerate((c3_, c4, c5)):
        conv(f"model.22.cv2.{i}.0", cin, a["c2"], 3)
        conv(f"model.22.cv2.{i}.1", a["c2"], a["c2"], 3)
        specs.append((f"model.22.cv2.{i}.2", 64, a["c2"], 1, False))
        conv(f"model.22.cv3.{i}.0", cin, a["c3"], 3)
        conv(f"model.22.cv3.{i}.1", a["c3"], a["c3"], 3)
        specs.append((f"model.22.cv3.{i}.2", nc, a["c3"], 1, False))
    return specs


def flops_per_image(scale="m", nc=1, size=640):
    """2*MAC over every conv at size x size (78.681 GFLOP for m, nc=1)."""
    total = 0.0
    strides = {"model.0": 2, "model.1": 4, "model.2": 4, "model.3": 8, "model.4": 8, "model.5": 16,
               "model.6": 16, "model.7": 32, "model.8": 32, "model.9": 32, "model.12": 16,
               "model.15": 8, "model.16": 16, "model.18": 16, "model.19": 32, "model.21": 32}
    for name, cout, cin, k, _ in conv_specs(scale, nc):
        parts = name.split(".")
        if parts[1] == "22":
            st = (8, 16, 32)[int(parts[3])]
        else:
            st = strides["model." + parts[1]]
        hw = (size // st) ** 2
        total += 2.0 * hw * cout * cin * k * k
    return total


DEFAULT_GAIN = 2.2
# wide conv biases: part of every layer's variance then comes from a constant rather than from
# the (rounding-noise carrying) input, which keeps the random network from amplifying f16
# rounding differences chaotically -- trained networks are likewise robust to f16
BIAS_AMP = 1.2


def synthesize(scale="m", nc=1, seed=0, cls_bias=-5.0, gains=None):
    """Seeded weights that keep f16 activations in a healthy range through the whole network.

    weight ~ U(-b, b) with b = sqrt(3 * gain / fan_in); gain ~ 1/E[silu(z)^2] keeps the second
    moment of conv+SiLU layers near 1.  The final class conv gets a negative bias so that a few
    hundred of the 8400 anchors clear the confidence threshold (exercises decode + NMS)."""
    rng = np.random.default_rng(seed)
    tensors = OrderedDict()
    for name, cout, cin, k, act in conv_specs(scale, nc):
        fan_in = cin * k * k
        table = GAINS if gains is None else gains
        gain = table.get(name, DEFAULT_GAIN)
        b = math.sqrt(3.0 * gain / fan_in)
        w = rng.uniform(-b, b, (cout, cin, k, k)).astype(np.float32)
        if act:
            bias = rng.uniform(-BIAS_AMP, BIAS_AMP, cout).astype(np.float32)
        elif ".cv2." in name:  # DFL logits
            bias = rng.uniform(0.5, 1.5, cout).astype(np.float32)
        else:  # class logits
            bias = np.full(cout, cls_bias, np.float32) + rng.uniform(-0.2, 0.2, cout).astype(np.float32)
        tensors[name + ".weight"] = w
        tensors[name + ".bias"] = bias
    return tensors


def save_pack(path, tensors, scale="m", nc=1, reg_max=16):
    depth, width, max_ch = SCALES[scale]
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", 1))
        f.write(struct.pack("<ffIII", depth, width, max_ch, nc, reg_max))
        f.write(struct.pack("<I", len(tensors)))
        for name, t in tensors.items():
            nb = name.encode()
            t = np.ascontiguousarray(t, np.float32)
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", t.ndim))
            f.write(struct.pack(f"<{t.ndim}I", *t.shape))
            f.write(t.tobytes())


def load_pack(path):
    """-> (tensors OrderedDict, meta dict)"""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != MAGIC:
        raise ValueError(f"{path}: not an RMRW weight pack")
    off = 4
    (version,) = struct.unpack_from("<I", data, off)
    off += 4
    depth, width, max_ch, nc, reg_max = struct.unpack_from("<ffIII", data, off)
    off += 20
    (n,) = struct.unpack_from("<I", data, off)
    off += 4
    tensors = OrderedDict()
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", data, off)
        off += 4
        name = data[off:off + ln].decode()
        off += ln
        (nd,) = struct.unpack_from("<I", data, off)
        off += 4
        dims = struct.unpack_from(f"<{nd}I", data, off)
        off += 4 * nd
        cnt = int(np.prod(dims))
        tensors[name] = np.frombuffer(data, np.float32, cnt, off).reshape(dims).copy()
        off += 4 * cnt
    scale = next((k for k, v in SCALES.items() if abs(v[0] - depth) < 1e-6 and abs(v[1] - width) < 1e-6
                  and v[2] == max_ch), None)
    return tensors, dict(version=version, depth=depth, width=width, max_ch=max_ch, nc=nc,
                         reg_max=reg_max, scale=scale)


def make_synthetic_pack(path, scale="m", nc=1, seed=0, cls_bias=-5.0):
    save_pack(path, synthesize(scale, nc, seed, cls_bias), scale, nc)
    return path
