#!/usr/bin/env python3
"""Winograd F(2x2, 3x3) gate, decided on the CPU before any kernel is written.

An f16 MFMA engine running F(2x2, 3x3) has to round two things a direct convolution never rounds: the
transformed input tiles V = B^T d B and the transformed filters U = G g G^T (both must be f16 to be MFMA
operands; the products still accumulate in f32).  This script restates exactly that engine inside the
f16-emulating oracle -- V computed in f32 from the f16 activations and rounded ONCE (the best case: a
packed-f16 transform rounds twice), U computed in f32 from the f16 weights and rounded once, the 16 GEMMs and
the output transform A^T M A in f32, then the usual bias / SiLU / shortcut / one f16 rounding -- for the
stride-1 3x3 layers with at least MIN_CIN input channels of backbone and neck, and measures what the gate
asks for on the fixtures of tests/test_gpu_network.py:

  * every stage output against the plain f16-emulating oracle, as a multiple of STAGE_BUDGET (the gate: <= 1.5x);
  * the head tensor against the plain f16-emulating oracle (the gate: boxes <= 2 px, scores <= 1e-2).

usage: winograd_gate.py [min_cin=192] [dim=2|1] [head]      (CPU only; ~1 min)
dim = 1: F(2, 3) along x only (4 transformed planes, the filter rows direct: 1.5x fewer MACs), the variant conv_w1d.hip
implements; "head": the Detect head's 3x3 convolutions as well."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import netutil  # noqa: E402
import oracle  # noqa: E402
from oracle import yolov8_ref as R  # noqa: E402
from rm_radar_amd import weights as Wt  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd_conv(x, w, round_v=True, round_u=True):
    """3x3 / stride 1 / pad 1 convolution of x [B,C,H,W] (H, W even) by w [K,C,3,3] through F(2x2, 3x3)"""
    B, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                       # [B,C,H/2,W/2,4,4]
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", BT, d, BT)          # B^T d B
    U = torch.einsum("ij,kcjl,ml->kcim", G, w, G)                # G g G^T
    if round_v:
        V = V.half().float()
    if round_u:
        U = U.half().float()
    M = torch.einsum("kcim,bcyxim->bkyxim", U, V)
    Y = torch.einsum("ij,bkyxjl,ml->bkyxim", AT, M, AT)          # [B,K,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, w.shape[0], H, W)


def winograd_conv_1d(x, w):
    """the same through F(2, 3) along x only (the three filter rows stay direct): V = d B per row, U = g G^T per filter row;
    a V element is ONE f16 subtraction / addition of two f16 values (exactly what v_pk_add_f16 computes), U is rounded once"""
    B, C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    d = xp.unfold(3, 4, 2)                                        # [B,C,H+2,W/2,4]
    V = torch.einsum("ij,bcyxj->bcyxi", BT, d).half().float()
    U = torch.einsum("ij,kcrj->kcri", G, w).half().float()       # [K,C,3,4]
    M = sum(torch.einsum("kci,bcyxi->bkyxi", U[:, :, r], V[:, :, r:r + H]) for r in range(3))
    Y = torch.einsum("ij,bkyxj->bkyxi", AT, M)                   # [B,K,H,W/2,2]
    return Y.reshape(B, w.shape[0], H, W)


class WinogradRef(R.YoloV8Ref):
    min_cin = 192
    hits = 0
    dim = 2
    head = False

    def conv(self, name, x, k, s=1, act=True, residual=None, keep_f32=False):
        w = self.t[name + ".weight"]
        if not (k == 3 and s == 1 and w.shape[1] >= self.min_cin and (self.head or not name.startswith("model.22."))):
            return super().conv(name, x, k, s, act, residual, keep_f32)
        WinogradRef.hits += 1
        y = (winograd_conv(x, w) if self.dim == 2 else winograd_conv_1d(x, w)) + self.t[name + ".bias"].view(1, -1, 1, 1)
        if act:
            y = y * torch.sigmoid(y)
        if residual is not None:
            y = y + residual
        return y.half().float() if self.f16 and not keep_f32 else y


def main():
    min_cin = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    head = len(sys.argv) > 3 and sys.argv[3] == "head"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_network import STAGE_BUDGET
    torch.manual_seed(0)
    # the transform itself is exact in f32
    x, w = torch.randn(1, 8, 6, 8), torch.randn(4, 8, 3, 3)
    assert (winograd_conv(x, w, False, False) - F.conv2d(x, w, padding=1)).abs().max() < 1e-4
    assert (winograd_conv_1d(x.half().float(), w.half().float()) - F.conv2d(x.half().float(), w.half().float(), padding=1)).abs().max() < 5e-2
    images = [netutil.test_image(1), netutil.test_image(2, 810, 1080), netutil.test_image(3, 1280, 720)]
    blobs = np.stack([oracle.preprocess(im)[0] for im in images])
    for nc, seed, conf in ((1, 11, 0.25), (12, 12, 0.5)):
        path = f"/tmp/wino_{nc}.rmrw"
        netutil.tuned_pack(path, nc, seed, conf, 0.01, images)
        tensors, meta = Wt.load_pack(path)
        plain = R.YoloV8Ref(tensors, meta, True)
        wino = WinogradRef(tensors, meta, True)
        wino.min_cin, wino.dim, wino.head = min_cin, dim, head
        WinogradRef.hits = 0
        fp, fw = plain.features(blobs[:2]), wino.features(blobs[:2])
        print(f"nc={nc}: {'F(2x2,3x3)' if dim == 2 else 'F(2,3) along x'} on the stride-1 3x3 layers with Cin >= {min_cin} "
              f"({WinogradRef.hits} layers of backbone + neck{' + Detect head' if head else ''})")
        print(f"{'stage':10s} {'mean |wino - f16 oracle|':>26s} {'budget':>9s} {'x budget':>9s} {'max':>9s}")
        worst = 0.0
        for name, budget in STAGE_BUDGET.items():
            e = np.abs(fw[name] - fp[name])
            worst = max(worst, e.mean() / budget)
            print(f"{name:10s} {e.mean():26.6f} {budget:9.1e} {e.mean() / budget:9.2f} {e.max():9.4f}")
        hp, hw = plain.forward(blobs), wino.forward(blobs)
        eb, es = np.abs(hw[:, :4] - hp[:, :4]), np.abs(hw[:, 4:] - hp[:, 4:])
        print(f"head: boxes max {eb.max():.2f} px mean {eb.mean():.3f} px (gate 2.0 / 0.25), scores max {es.max():.4f} (gate 1e-2)")
        ok = worst <= 1.5 and eb.max() <= 2.0 and eb.mean() <= 0.25 and es.max() <= 1e-2
        print(f"gate (stages <= 1.5x budget, head tolerance): {'PASS' if ok else 'FAIL'} (worst stage {worst:.2f}x)\n")


if __name__ == "__main__":
    main()
