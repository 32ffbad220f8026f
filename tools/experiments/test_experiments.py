"""Tests of the measured-and-lost kernels kept out of the product library (tools/experiments/): the Winograd F(2, 3)
kernel conv_w1d.hip and (compile check only) the ping-pong conv_t32.  They need the EXPERIMENTS build:

    make -C rm_radar_amd/csrc -j8 EXPERIMENTS=1
    RMR_LIB=rm_radar_amd/_build_exp/librmr.so python -m pytest tools/experiments -m gpu -q

Not collected by `pytest tests/`; skipped when the loaded library was built without them."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = torch.nn.functional

from test_gpu_conv import r16, ref_conv, rmr  # noqa: E402,F401  (fixture + helpers of the product tests)
from test_gpu_network import _check_head, images, packs, refs  # noqa: E402,F401


@pytest.fixture(autouse=True)
def _needs_experiments_build(rmr):
    try:
        rmr.conv2d(np.zeros((1, 20, 20, 64), np.float32), np.zeros((96, 64, 3, 3), np.float32), None, 1, 1, False, tile=980)
    except rmr.InvalidArgument as e:
        if "cannot run" in str(e):
            pytest.skip("librmr.so was built without EXPERIMENTS=1")


def wino_ref(x_nhwc, w, b, silu, res):
    """F(2, 3) along x restated in torch with the kernel's roundings: V = B^T d (one f16 add of two f16 values), U = g G^T
    rounded to f16 once, f32 accumulation, f32 output transform (tools/winograd_gate.py::winograd_conv_1d)"""
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
    G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
    x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2)
    wt = torch.from_numpy(w)
    B, C, H, W = x.shape
    d = F.pad(x, (1, 1, 1, 1)).unfold(3, 4, 2)
    V = torch.einsum("ij,bcyxj->bcyxi", BT, d).half().float()
    U = torch.einsum("ij,kcrj->kcri", G, wt).half().float()
    M = sum(torch.einsum("kci,bcyxi->bkyxi", U[:, :, r], V[:, :, r:r + H]) for r in range(3))
    y = torch.einsum("ij,bkyxj->bkyxi", AT, M).reshape(B, wt.shape[0], H, W) + torch.from_numpy(b).view(1, -1, 1, 1)
    if silu:
        y = y * torch.sigmoid(y)
    y = y.permute(0, 2, 3, 1).numpy()
    return y + res if res is not None else y


def run_wino_case(rmr, n, h, w, cin, cout, silu, res, tile, seed):
    rng = np.random.default_rng(seed)
    x = r16(rng.normal(0, 1, (n, h, w, cin)).astype(np.float32))
    wt = r16((rng.normal(0, 1, (cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
    b = rng.normal(0, 0.5, cout).astype(np.float32)
    r = r16(rng.normal(0, 1, (n, h, w, cout)).astype(np.float32)) if res else None
    got = rmr.conv2d(x, wt, b, 1, 1, silu, r, tile=tile)
    scale = max(1.0, np.abs(got).max())
    # (a) against its own arithmetic: only the f32 accumulation order differs -- the bar of every other kernel
    err = np.abs(got - wino_ref(x, wt, b, silu, r)).max()
    assert err <= 2e-3 * scale, f"vs the Winograd restatement: max err {err}"
    # (b) against the direct convolution: what rounding V and U to f16 costs (2^-11 relative per operand, K = 12 Cin terms)
    err = np.abs(got - ref_conv(x, wt, b, 1, 1, silu, r)).max()
    assert err <= 1.2e-2 * scale, f"vs the direct convolution: max err {err}"


def test_conv_w1d_every_tile(rmr):
    # Winograd F(2, 3) along x on the conv_t32 skeleton (conv_w1d.hip, ids 980..): GEMM rows are 2-pixel tiles, four
    # accumulator sets per wave tile, 12 (filter row, xi) slices per 32-channel chunk, the input transform in LDS with the
    # left / right padding as lane masks, two strided epilogue passes
    tiles = [(512, 96), (256, 192), (512, 64)]
    for t, (bm, bn) in enumerate(tiles):
        run_wino_case(rmr, 3, 20, 20, 64, bn, True, True, 980 + t, seed=t)              # 3 images, ~3 tiles, 2 chunks
        run_wino_case(rmr, 1, 19, 22, 32, bn * 2, True, False, 980 + t, seed=40 + t)    # odd H, ragged M, 1 chunk, 2 channel tiles
        run_wino_case(rmr, 2, 7, 6, 32, bn, False, False, 980 + t, seed=50 + t)         # 3 tiles per row; tile far larger than the images
    run_wino_case(rmr, 2, 40, 40, 192, 192, True, True, 980, seed=70)    # 6 chunks, 72 taps
    run_wino_case(rmr, 2, 40, 40, 192, 192, True, False, 981, seed=71)
    run_wino_case(rmr, 1, 80, 80, 96, 96, True, True, 980, seed=72)      # W = 80: 43 raw blocks, 152 KiB of LDS
    run_wino_case(rmr, 1, 80, 80, 192, 192, True, False, 981, seed=73)
    run_wino_case(rmr, 2, 20, 20, 288, 288, True, True, 980, seed=74)    # 9 chunks, 3 channel tiles
    run_wino_case(rmr, 1, 80, 80, 192, 64, True, False, 982, seed=75)    # Detect box branch shape
    # more tiles than workgroups: every workgroup walks several tiles (the streams cross tile boundaries)
    run_wino_case(rmr, 330, 20, 20, 32, 192, True, True, 980, seed=76)   # 258 x 2 tiles on 256 workgroups
    run_wino_case(rmr, 330, 20, 20, 64, 192, True, False, 981, seed=77)  # 516 tiles of 256 x 192
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 5, 5, 32), np.float32), np.zeros((96, 32, 3, 3), np.float32), None, 1, 1, False, tile=980)  # odd width



def test_network_on_the_winograd_kernels(rmr, packs, refs, images, oracle, monkeypatch, tmp_path):
    """conv_w1d.hip (Winograd F(2, 3) along x) under the whole network: RMR_TUNE_ONLY=980-999 makes EVERY 3x3 / stride-1
    layer with Cin % 32 == 0 run on it -- backbone, neck and the Detect head's convolutions, 40 of the 83 layers -- at a batch
    size where the autotuner would not offer it.  The gate of the experiment: the same f16-emulating
    oracle and the same head tolerance as the direct kernels (2 px / 1e-2, mean 0.25 px); tools/winograd_gate.py is the
    CPU restatement of this arithmetic (operands V = B^T d and U = g G^T rounded to f16 once) that predicted it.  And it must
    not be the direct plan under another name: the outputs differ."""
    import shutil
    pack = str(tmp_path / "armor_w1d.rmrw")  # its own tuning cache
    shutil.copy(packs[1], pack)
    monkeypatch.setenv("RMR_WINOGRAD", "1")      # off by default: the kernel is slower than the direct one (conv_w1d.hip)
    monkeypatch.setenv("RMR_TUNE_ONLY", "980-999")
    n = 5
    det = rmr.Detector(pack, 12, (2592, 2048), n, conf_thresh=0.5)
    batch = [images[i % 3] for i in range(n)]
    got, _ = det.infer(batch)
    det.close()
    blobs = np.stack([oracle.preprocess(im)[0] for im in images])
    want = refs["armor"][1].forward(blobs)
    for i in range(n):
        _check_head(got[i:i + 1], want[i % 3:i % 3 + 1], 2.0, 1e-2)
    tuned = [l.split() for l in open(pack + ".tune").read().splitlines()[1:]]
    assert sum(1 for t in tuned if 980 <= int(t[2]) < 1000) >= 30
    monkeypatch.setenv("RMR_WINOGRAD", "0")
    monkeypatch.delenv("RMR_TUNE_ONLY")
    direct = rmr.Detector(packs[1], 12, (2592, 2048), n, conf_thresh=0.5)
    ref, _ = direct.infer(batch)
    direct.close()
    d = np.abs(got - ref)
    print(f"winograd plan vs direct plan: boxes max {d[:, :4].max():.3f} px mean {d[:, :4].mean():.4f} px, scores max {d[:, 4:].max():.5f}")
    assert d.max() > 0


