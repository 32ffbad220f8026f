"""CPU ORACLE for the network -- TEST INFRASTRUCTURE ONLY.

In the reference the network forward is TensorRT (closed source, >=8.5; call sites
src/detect/detector.cpp:102-107,187-231, src/detect/detector.h:122) executing ``car.onnx`` /
``armor.onnx`` -- YOLOv8m exports that are ABSENT from the reference tree
(.MISSING_LARGE_BLOBS:2-3).  Parity of the weights is therefore UNPINNED; what this module
restates is the PUBLISHED algorithm: the Ultralytics YOLOv8 architecture (yolov8.yaml +
nn/modules: Conv = conv+BN+SiLU with BN folded, C2f, SPPF, Detect with DFL)
as plain PyTorch fp32 functional ops on the CPU, reading the same weight pack the product loads.

Independence from the product: the only thing taken from ``rm_radar_amd`` is the pack FILE READER
(``weights.load_pack``: names -> arrays).  The layer plan is this module's own -- the depth / width / max-channel
multiples of the published yolov8.yaml (``_ULTRALYTICS_SCALES``) -- and ``_check_against_ultralytics`` asserts that the
tensors of the pack carry the Ultralytics module names with the shapes that plan implies (for scale "m" additionally
against the literal channel counts of the published YOLOv8m: 48 / 96 / 192 / 384 / 576, repeats 2 / 4 / 4 / 2, head
repeats 2), so a wrong table in the pack maker cannot agree with a wrong table here.

``emulate_f16=True`` rounds weights, the input and every stored activation to f16 exactly where
the HIP engine stores f16 (f32 accumulate, f32 bias/SiLU/residual, one rounding per stored
tensor; the two final 1x1 head convs stay f32), which isolates accumulation-order noise from
precision loss.

``fp8=True`` (implies emulate_f16) restates the engine's RMR_PRECISION_FP8 plan (BASELINE configs[4]): the
3x3 / stride-1 convolutions with >= 64 input channels (a multiple of 16) see their f16 input rounded to OCP
e4m3 (unit scale) and their f16 weights divided by max|w| / 448 of the output channel, rounded to e4m3 and
scaled back; everything else is as in the f16 mode (the Detect head's 3x3 convolutions included, unless
RMR_FP8_HEAD=1, the engine's switch).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _r16(t):
    return t.half().float()


def _e4m3(t):
    """OCP e4m3fn, round to nearest even, saturating at 448 (what the weight packer and v_cvt_pk_fp8_f32 do)"""
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


# depth multiple, width multiple, max channels -- ultralytics/cfg/models/v8/yolov8.yaml `scales:`
_ULTRALYTICS_SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768), "l": (1.00, 1.00, 512),
                       "x": (1.00, 1.25, 512)}


def _plan(scale):
    """Channels of P1..P5 and the C2f repeat counts of yolov8.yaml at a scale (parse_model: make_divisible(min(c, max) * w, 8),
    n = max(round(n * d), 1))"""
    depth, width, max_ch = _ULTRALYTICS_SCALES[scale]
    ch = [int(-(-(min(c, max_ch) * width) // 8) * 8) for c in (64, 128, 256, 512, 1024)]
    return {"ch": ch, "n": [max(round(r * depth), 1) for r in (3, 6, 6, 3)], "nh": max(round(3 * depth), 1)}


def _check_against_ultralytics(tensors, scale, nc):
    """The pack's tensors, by Ultralytics module name, must have the shapes of the published architecture."""
    a = _plan(scale)
    c1, c2, c3, c4, c5 = a["ch"]
    if scale == "m":   # the literal published YOLOv8m
        assert (a["ch"], a["n"], a["nh"]) == ([48, 96, 192, 384, 576], [2, 4, 4, 2], 2)
    cb, cc = max(16, c3 // 4, 64), max(c3, min(nc, 100))   # Detect: c2 (box branch), c3 (class branch)
    want = {"model.0.conv.weight": (c1, 3, 3, 3), "model.1.conv.weight": (c2, c1, 3, 3), "model.3.conv.weight": (c3, c2, 3, 3),
            "model.5.conv.weight": (c4, c3, 3, 3), "model.7.conv.weight": (c5, c4, 3, 3),
            "model.9.cv1.conv.weight": (c5 // 2, c5, 1, 1), "model.9.cv2.conv.weight": (c5, c5 * 2, 1, 1),
            "model.16.conv.weight": (c3, c3, 3, 3), "model.19.conv.weight": (c4, c4, 3, 3)}
    for name, cin, cout, n in (("model.2", c2, c2, a["n"][0]), ("model.4", c3, c3, a["n"][1]), ("model.6", c4, c4, a["n"][2]),
                               ("model.8", c5, c5, a["n"][3]), ("model.12", c5 + c4, c4, a["nh"]), ("model.15", c4 + c3, c3, a["nh"]),
                               ("model.18", c3 + c4, c4, a["nh"]), ("model.21", c4 + c5, c5, a["nh"])):
        c = cout // 2
        want[f"{name}.cv1.conv.weight"] = (2 * c, cin, 1, 1)
        want[f"{name}.cv2.conv.weight"] = (cout, (2 + n) * c, 1, 1)
        for i in range(n):
            want[f"{name}.m.{i}.cv1.conv.weight"] = (c, c, 3, 3)
            want[f"{name}.m.{i}.cv2.conv.weight"] = (c, c, 3, 3)
        assert f"{name}.m.{n}.cv1.conv.weight" not in tensors, f"{name}: more bottlenecks than yolov8.yaml has at scale {scale}"
    for i, cf in enumerate((c3, c4, c5)):
        want[f"model.22.cv2.{i}.0.conv.weight"] = (cb, cf, 3, 3)
        want[f"model.22.cv2.{i}.1.conv.weight"] = (cb, cb, 3, 3)
        want[f"model.22.cv2.{i}.2.weight"] = (64, cb, 1, 1)
        want[f"model.22.cv3.{i}.0.conv.weight"] = (cc, cf, 3, 3)
        want[f"model.22.cv3.{i}.1.conv.weight"] = (cc, cc, 3, 3)
        want[f"model.22.cv3.{i}.2.weight"] = (nc, cc, 1, 1)
    for name, shape in want.items():
        assert name in tensors, f"weight pack lacks {name}"
        assert tuple(tensors[name].shape) == shape, f"{name}: pack has {tuple(tensors[name].shape)}, Ultralytics YOLOv8{scale} has {shape}"
    n_conv = sum(1 for k in tensors if k.endswith(".weight"))
    assert n_conv == len(want), f"pack has {n_conv} convolutions, the architecture {len(want)}"
    return a


class YoloV8Ref:
    def __init__(self, tensors, meta, emulate_f16=False, fp8=False, jitter=0.0, jitter_seed=0):
        self.meta = meta
        # jitter > 0: every convolution's f32 result is multiplied by 1 + jitter * u, u uniform in [-1, 1) -- ANOTHER exact
        # implementation of the same plan (another f32 summation order moves a result by a few 2^-24 of its value): how far
        # two such implementations drift apart is the reproducibility floor of a plan (tools/fp8_parity_study.py)
        self.jitter = float(jitter)
        self._jg = torch.Generator().manual_seed(int(jitter_seed))
        self.f16 = emulate_f16 or fp8
        self.fp8 = fp8
        import os
        self.fp8_head = os.environ.get("RMR_FP8_HEAD", "0") not in ("", "0")   # the engine's switch: Detect convs too
        emulate_f16 = self.f16
        self.arch = _check_against_ultralytics(tensors, meta["scale"], meta["nc"])   # this module's own plan, not the pack maker's
        self.nc = meta["nc"]
        self.t = {}
        for k, v in tensors.items():
            tv = torch.from_numpy(np.ascontiguousarray(v))
            if emulate_f16 and k.endswith(".weight"):
                tv = _r16(tv)
            self.t[k] = tv

    # Conv = Conv2d(bias=False) + BN + SiLU, BN folded  [Ultralytics nn/modules/conv.py]
    def conv(self, name, x, k, s=1, act=True, residual=None, keep_f32=False):
        w = self.t[name + ".weight"]
        b = self.t[name + ".bias"]
        if self.fp8 and k == 3 and s == 1 and w.shape[1] >= 64 and w.shape[1] % 16 == 0 and (self.fp8_head or not name.startswith("model.22.")):
            x = _e4m3(x)
            scale = (w.abs().flatten(1).max(1).values / 448.0).clamp_min(1e-30).view(-1, 1, 1, 1)
            w = _e4m3(w / scale) * scale
        y = F.conv2d(x, w, b, stride=s, padding=k // 2)
        if self.jitter:
            y = y * (1.0 + self.jitter * (2.0 * torch.rand(y.shape, generator=self._jg) - 1.0))
        if act:
            y = y * torch.sigmoid(y)
        if residual is not None:
            y = y + residual
        if self.f16 and not keep_f32:
            y = _r16(y)
        return y

    # C2f.forward: y = cv1(x).chunk(2); y.append(m(y[-1])) ...; cv2(cat(y))
    def c2f(self, name, x, n, shortcut):
        y = list(self.conv(f"{name}.cv1.conv", x, 1).chunk(2, 1))
        for i in range(n):
            h = self.conv(f"{name}.m.{i}.cv1.conv", y[-1], 3)
            y.append(self.conv(f"{name}.m.{i}.cv2.conv", h, 3, residual=y[-1] if shortcut else None))
        return self.conv(f"{name}.cv2.conv", torch.cat(y, 1), 1)

    # SPPF.forward: three chained MaxPool2d(5,1,2)
    def sppf(self, name, x):
        x = self.conv(f"{name}.cv1.conv", x, 1)
        y1 = F.max_pool2d(x, 5, 1, 2)
        y2 = F.max_pool2d(y1, 5, 1, 2)
        y3 = F.max_pool2d(y2, 5, 1, 2)
        return self.conv(f"{name}.cv2.conv", torch.cat([x, y1, y2, y3], 1), 1)

    def backbone_neck(self, x, keep=None):
        """keep: a dict that receives every stage output by Ultralytics module name ("model.0" ... "model.21")"""
        a = self.arch
        n = a["n"]

        def k(name, t):
            if keep is not None:
                keep[name] = t
            return t
        x = k("model.0", self.conv("model.0.conv", x, 3, 2))
        x = k("model.1", self.conv("model.1.conv", x, 3, 2))
        x = k("model.2", self.c2f("model.2", x, n[0], True))
        x = k("model.3", self.conv("model.3.conv", x, 3, 2))
        p3 = k("model.4", self.c2f("model.4", x, n[1], True))
        x = k("model.5", self.conv("model.5.conv", p3, 3, 2))
        p4 = k("model.6", self.c2f("model.6", x, n[2], True))
        x = k("model.7", self.conv("model.7.conv", p4, 3, 2))
        x = k("model.8", self.c2f("model.8", x, n[3], True))
        p5 = k("model.9", self.sppf("model.9", x))
        up = F.interpolate(p5, scale_factor=2, mode="nearest")
        h4 = k("model.12", self.c2f("model.12", torch.cat([up, p4], 1), a["nh"], False))
        up = F.interpolate(h4, scale_factor=2, mode="nearest")
        o3 = k("model.15", self.c2f("model.15", torch.cat([up, p3], 1), a["nh"], False))
        x = self.conv("model.16.conv", o3, 3, 2)
        o4 = k("model.18", self.c2f("model.18", torch.cat([x, h4], 1), a["nh"], False))
        x = self.conv("model.19.conv", o4, 3, 2)
        o5 = k("model.21", self.c2f("model.21", torch.cat([x, p5], 1), a["nh"], False))
        return [o3, o4, o5]

    @torch.no_grad()
    def features(self, blob):
        """Every stage output of the backbone and neck for blob [B,3,H,W]: name -> [B,H,W,C] numpy (NHWC)."""
        x = torch.from_numpy(np.ascontiguousarray(blob, np.float32))
        if self.f16:
            x = _r16(x)
        keep = {}
        self.backbone_neck(x, keep)
        return {name: t.permute(0, 2, 3, 1).contiguous().numpy() for name, t in keep.items()}

    def head_logits(self, feats):
        """-> (box logits [B,64,A], class logits [B,nc,A]) in P3,P4,P5 anchor order"""
        box, cls = [], []
        for i, f in enumerate(feats):
            b = self.conv(f"model.22.cv2.{i}.0.conv", f, 3)
            b = self.conv(f"model.22.cv2.{i}.1.conv", b, 3)
            b = self.conv(f"model.22.cv2.{i}.2", b, 1, act=False, keep_f32=True)
            c = self.conv(f"model.22.cv3.{i}.0.conv", f, 3)
            c = self.conv(f"model.22.cv3.{i}.1.conv", c, 3)
            c = self.conv(f"model.22.cv3.{i}.2", c, 1, act=False, keep_f32=True)
            box.append(b.flatten(2))
            cls.append(c.flatten(2))
        return torch.cat(box, 2), torch.cat(cls, 2), [f.shape[2:] for f in feats]

    @staticmethod
    def decode_head(box, cls, shapes, strides=(8, 16, 32)):
        """Detect inference form: DFL + dist2bbox(xywh) * stride, sigmoid(cls)  [Ultralytics
        nn/modules/head.py] -> [B, 4+nc, A]"""
        B, _, A = box.shape
        ax, ay, st = [], [], []
        for (h, w), s in zip(shapes, strides):
            yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5,
                                    torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
            ax.append(xx.flatten())
            ay.append(yy.flatten())
            st.append(torch.full((h * w,), float(s)))
        ax, ay, st = torch.cat(ax), torch.cat(ay), torch.cat(st)
        p = box.view(B, 4, 16, A).softmax(2)
        dist = (p * torch.arange(16, dtype=torch.float32).view(1, 1, 16, 1)).sum(2)  # l,t,r,b
        x1, y1 = ax - dist[:, 0], ay - dist[:, 1]
        x2, y2 = ax + dist[:, 2], ay + dist[:, 3]
        cx, cy, w, h = (x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1
        xywh = torch.stack([cx, cy, w, h], 1) * st
        return torch.cat([xywh, cls.sigmoid()], 1)

    @torch.no_grad()
    def forward(self, blob):
        """blob: [B,3,H,W] f32 RGB in [0,1] (the reference's network input, Q5)."""
        x = torch.from_numpy(np.ascontiguousarray(blob, np.float32))
        if self.f16:
            x = _r16(x)
        box, cls, shapes = self.head_logits(self.backbone_neck(x))
        return self.decode_head(box, cls, shapes).numpy()


def load(path, emulate_f16=False, fp8=False, jitter=0.0, jitter_seed=0):
    from rm_radar_amd import weights as W
    tensors, meta = W.load_pack(path)
    return YoloV8Ref(tensors, meta, emulate_f16, fp8, jitter, jitter_seed)
