"""Shared helpers for the network parity tests: tuned synthetic packs, test images, matching."""
import os

import numpy as np


def test_image(seed, w=640, h=640):
    """Structured BGR u8 image: smooth gradients + blobs + a little noise (uniform noise alone
    averages out in the deep layers and gives no detections)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        fx, fy = rng.uniform(0.01, 0.06, 2)
        img[:, :, c] = 127 + 90 * np.sin(xx * fx + rng.uniform(0, 6)) * np.cos(yy * fy + rng.uniform(0, 6))
    for _ in range(12):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        r = rng.uniform(0.03, 0.15) * min(w, h)
        m = ((xx - cx) ** 2 + (yy - cy) ** 2) < r * r
        img[m] = rng.uniform(0, 255, 3)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def tuned_pack(path, nc, seed, conf_target, frac, images):
    """Synthetic YOLOv8m pack whose class bias is set so that ~frac of the anchors of `images`
    clear conf_target (the final class conv is linear in its bias)."""
    from oracle import yolov8_ref as R
    from rm_radar_amd import weights as W
    import oracle
    t = W.synthesize("m", nc, seed=seed, cls_bias=0.0)
    for i in range(3):
        t[f"model.22.cv3.{i}.2.bias"][:] = 0.0
    ref = R.YoloV8Ref(t, dict(scale="m", nc=nc))
    blobs = np.stack([oracle.preprocess(im)[0] for im in images])
    import torch
    with torch.no_grad():
        box, cls, _ = ref.head_logits(ref.backbone_neck(torch.from_numpy(blobs)))
    best = cls.max(1).values.flatten().numpy()
    q = np.quantile(best, 1.0 - frac)
    bias = np.log(conf_target / (1 - conf_target)) - q
    for i in range(3):
        t[f"model.22.cv3.{i}.2.bias"][:] = bias
    W.save_pack(path, t, "m", nc)
    return path


def iou_xywh(a, b):
    x1, y1 = max(a[0], b[0]), max(a[1], b[1])
    x2, y2 = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
    inter = max(0.0, x2 - x1) * max(0.0, y2 - y1)
    uni = a[2] * a[3] + b[2] * b[3] - inter
    return inter / uni if uni > 0 else 1.0


def match_detections(got, want, conf_thresh, margin=0.02, iou_min=0.99):
    """Every reference detection whose confidence is clear of the threshold by `margin` must have
    a partner (same label, IoU >= iou_min) and vice versa.  Returns (#matched, #skipped)."""
    def side(a, b):
        matched = skipped = 0
        for d in a:
            ok = any(int(e["label"]) == int(d["label"]) and
                     iou_xywh((d["x"], d["y"], d["width"], d["height"]),
                              (e["x"], e["y"], e["width"], e["height"])) >= iou_min for e in b)
            if ok:
                matched += 1
            elif abs(float(d["confidence"]) - conf_thresh) < margin:
                skipped += 1
            else:
                raise AssertionError(f"unmatched detection {d} (thresh {conf_thresh})")
        return matched, skipped
    m1, s1 = side(want, got)
    m2, s2 = side(got, want)
    return min(m1, m2), s1 + s2
