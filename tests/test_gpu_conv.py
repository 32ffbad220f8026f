"""GPU parity of the MFMA implicit-GEMM conv kernel (one layer at a time, every tile shape)
against plain PyTorch fp32 conv2d on the CPU.  Inputs and weights are pre-rounded to f16, so the
only difference is f32 accumulation order: tolerance 2e-3 relative to the output scale."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(scope="module")
def rmr():
    import rm_radar_amd as r
    assert r.device_count() >= 1
    return r


def r16(a):
    return a.astype(np.float16).astype(np.float32)


def ref_conv(x_nhwc, w, b, stride, pad, silu, res):
    y = F.conv2d(torch.from_numpy(x_nhwc).permute(0, 3, 1, 2), torch.from_numpy(w),
                 torch.from_numpy(b), stride=stride, padding=pad)
    if silu:
        y = y * torch.sigmoid(y)
    y = y.permute(0, 2, 3, 1).numpy()
    if res is not None:
        y = y + res
    return y


def run_case(rmr, n, h, w, cin, cout, k, stride, silu, res, tile=-1, seed=0):
    rng = np.random.default_rng(seed)
    x = r16(rng.normal(0, 1, (n, h, w, cin)).astype(np.float32))
    wt = r16((rng.normal(0, 1, (cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    b = rng.normal(0, 0.5, cout).astype(np.float32)
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    r = r16(rng.normal(0, 1, (n, ho, wo, cout)).astype(np.float32)) if res else None
    got = rmr.conv2d(x, wt, b, stride, pad, silu, r, tile=tile)
    want = ref_conv(x, wt, b, stride, pad, silu, r)
    err = np.abs(got - want).max()
    assert err <= 2e-3 * max(1.0, np.abs(want).max()), f"max err {err}"


CASES = [
    # n, h, w, cin, cout, k, stride, silu, residual
    (1, 32, 32, 3, 48, 3, 2, True, False),     # stem: Cin 3 padded to 8, K = 72 padded to 96
    (2, 16, 16, 48, 96, 3, 2, True, False),    # model.1: K = 432 (not a multiple of 32)
    (1, 20, 20, 96, 96, 1, 1, True, False),    # C2f cv1
    (1, 20, 20, 48, 48, 3, 1, True, True),     # bottleneck with shortcut
    (3, 10, 10, 192, 96, 1, 1, True, False),
    (1, 9, 13, 288, 288, 3, 1, True, True),    # odd spatial size, M not a tile multiple
    (1, 8, 8, 1152, 576, 1, 1, True, False),
    (1, 12, 12, 192, 256, 3, 1, True, False),  # fused head conv (64 + 192)
    (1, 12, 12, 64, 64, 1, 1, False, False),   # DFL conv: bias, no activation
    (1, 12, 12, 192, 12, 1, 1, False, False),  # class conv nc=12 -> padded to 16
    (1, 12, 12, 192, 1, 1, 1, False, False),   # class conv nc=1
    (2, 40, 40, 96, 192, 3, 2, True, False),
    (1, 7, 5, 8, 16, 3, 1, False, False),
    (1, 5, 5, 16, 32, 5, 1, True, False),      # 5x5 window
]


@pytest.mark.parametrize("case", CASES)
def test_conv_auto_tile(rmr, case):
    run_case(rmr, *case)


def test_conv_every_tile(rmr):
    from rm_radar_amd import _lib
    tiles = [(256, 96), (128, 96), (64, 96), (256, 48), (128, 48), (64, 48), (256, 64), (128, 64),
             (64, 64), (128, 128), (64, 128), (256, 32), (64, 32), (256, 16), (64, 16),
             (128, 96), (128, 128), (128, 64), (128, 48), (256, 48), (64, 96), (64, 128)]  # 15.. : BK = 64
    for t, (bm, bn) in enumerate(tiles):
        cout = bn * 2
        run_case(rmr, 1, 19, 23, 48, cout, 3, 1, True, True, tile=t, seed=t)   # M = 437: ragged
        run_case(rmr, 2, 16, 16, 96, bn, 1, 1, False, False, tile=t, seed=100 + t)


def test_conv_dma_every_tile(rmr):
    # the LDS-DMA pipelined kernel (conv_dma.hip): tile ids 100.., needs Cin % 32 == 0
    tiles = [(128, 96), (256, 96), (64, 96), (128, 128), (64, 128), (128, 64), (64, 64), (256, 48),
             (128, 48), (64, 48), (256, 16), (64, 16), (256, 32), (64, 32),
             (256, 192), (256, 96), (512, 96), (256, 128), (256, 256), (256, 288), (128, 288), (128, 192),  # 8 waves
             (128, 96), (256, 96), (64, 96), (128, 128), (64, 128), (128, 64), (256, 48), (256, 192),
             (256, 96), (256, 128), (128, 288), (128, 192)]  # BK = 64
    for t, (bm, bn) in enumerate(tiles):
        run_case(rmr, 1, 19, 23, 64, bn * 2, 3, 1, True, True, tile=100 + t, seed=t)    # ragged M, padding taps
        run_case(rmr, 2, 16, 16, 96, bn, 1, 1, False, False, tile=100 + t, seed=50 + t)  # 1x1, 3 K slices
        run_case(rmr, 1, 20, 20, 32, bn, 3, 2, True, False, tile=100 + t, seed=90 + t)   # stride 2
    run_case(rmr, 3, 40, 40, 192, 192, 3, 1, True, True, tile=100, seed=7)              # 54 K slices, 38 blocks
    run_case(rmr, 1, 9, 9, 64, 96, 5, 1, True, False, tile=102, seed=8)                 # 5x5 window
    run_case(rmr, 2, 20, 20, 96, 96, 3, 1, True, True, tile=122, seed=9)                # BK 64, Cin 96: slices straddle taps
    run_case(rmr, 1, 20, 20, 288, 288, 3, 1, True, True, tile=132, seed=10)             # K = 2592 = 40.5 slices of 64
    for t, bn in ((144, 192), (145, 192), (146, 128)):                                   # 320-row tiles
        run_case(rmr, 1, 19, 23, 64, bn * 2, 3, 1, True, True, tile=t, seed=t)
        run_case(rmr, 2, 40, 40, 96, bn, 3, 2, True, False, tile=t, seed=t + 1)          # stride 2, 800 rows
        run_case(rmr, 2, 16, 16, 96, bn, 1, 1, False, False, tile=t, seed=t + 2)
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 4, 4, 48), np.float32), np.zeros((96, 48, 3, 3), np.float32), None, 1, 1,
                   False, tile=100)  # Cin = 48 is not a multiple of 32


def test_conv_dma_split_k(rmr):
    # split-K: several workgroups share an output tile, the last arriver reduces (ticket counter
    # re-arms itself: the hook launches twice).  tile = 1000 * split + 100 + dma tile
    for split in (2, 3, 4, 9):
        run_case(rmr, 1, 20, 20, 192, 192, 3, 1, True, True, tile=1000 * split + 100 + 6, seed=split)   # 64 x 64
        run_case(rmr, 1, 40, 40, 96, 96, 3, 1, True, False, tile=1000 * split + 100 + 2, seed=10 + split)  # 64 x 96
        run_case(rmr, 2, 9, 11, 64, 32, 1, 1, False, False, tile=1000 * min(split, 2) + 100 + 13, seed=20 + split)
    run_case(rmr, 1, 20, 20, 288, 288, 3, 1, True, True, tile=1000 * 6 + 100 + 24, seed=31)            # BK = 64
    run_case(rmr, 4, 40, 40, 192, 192, 3, 1, True, True, tile=1000 * 3 + 100 + 0, seed=32)             # 128 x 96, 100 tiles


def test_conv_halo_every_tile(rmr):
    # halo-staged 3x3 / stride-1 kernel (conv_halo.hip): tile ids 200..; input range staged once,
    # taps are row shifts, image borders (and image-to-image boundaries inside a tile) are masked
    tiles = [(256, 192), (256, 96), (256, 288), (256, 128), (256, 256), (256, 64), (128, 192), (128, 288),
             (128, 96), (128, 128), (256, 96), (256, 48), (256, 48), (256, 96), (384, 48),
             # whole-chunk slices (9 taps per barrier), then one filter row per barrier
             (64, 48), (128, 48), (256, 48), (64, 96), (128, 96), (64, 32), (128, 32), (64, 64), (128, 64), (32, 96),
             (128, 192), (256, 192), (256, 96), (256, 192), (256, 96), (256, 288), (512, 96), (512, 96),
             (320, 192), (320, 96)]
    for t, (bm, bn) in enumerate(tiles):
        run_case(rmr, 3, 20, 20, 64, bn, 3, 1, True, True, tile=200 + t, seed=t)          # 3 images per ~5 tiles
        run_case(rmr, 1, 19, 23, 32, bn * 2, 3, 1, True, False, tile=200 + t, seed=40 + t)  # odd W, ragged M
    run_case(rmr, 2, 40, 40, 192, 192, 3, 1, True, True, tile=200, seed=70)   # 6 chunks, 54 slices
    run_case(rmr, 1, 80, 80, 96, 96, 3, 1, True, True, tile=201, seed=71)     # W = 80: 418-row input range
    run_case(rmr, 2, 20, 20, 288, 288, 3, 1, True, True, tile=207, seed=72)   # 9 chunks
    run_case(rmr, 1, 5, 5, 32, 96, 3, 1, False, False, tile=208, seed=73)     # tile far larger than the image
    run_case(rmr, 1, 160, 160, 48, 48, 3, 1, True, True, tile=211, seed=74)   # Cin 48: partial channel chunk, W = 160
    run_case(rmr, 2, 33, 29, 48, 96, 3, 1, True, False, tile=213, seed=75)    # Cin 48, odd sizes
    run_case(rmr, 1, 24, 24, 40, 48, 3, 1, False, False, tile=212, seed=76)   # Cin 40 (8-channel granularity)
    run_case(rmr, 4, 40, 40, 192, 192, 3, 1, True, True, tile=216, seed=77)   # batch-4 P4 bottleneck, 6 slices of 9 taps
    run_case(rmr, 1, 80, 80, 96, 96, 3, 1, True, True, tile=219, seed=78)     # W = 80 at 9 taps per slice
    run_case(rmr, 1, 20, 20, 288, 288, 3, 1, True, True, tile=224, seed=79)   # 32 x 96 tiles, 9 chunks
    run_case(rmr, 2, 40, 40, 192, 192, 3, 1, True, True, tile=225, seed=80)   # one barrier per filter row
    run_case(rmr, 1, 33, 29, 48, 96, 3, 1, True, False, tile=218, seed=81)    # Cin 48: partial chunk at 9 taps per slice
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 8, 32), np.float32), np.zeros((96, 32, 3, 3), np.float32), None, 2, 1,
                   False, tile=201)  # stride 2 is not a halo case
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 8, 32), np.float32), np.zeros((96, 32, 1, 1), np.float32), None, 1, 0,
                   False, tile=201)  # 1x1


def test_conv_weights_stationary(rmr):
    # conv_ws.hip (ids 300..): 3x3 / s1, 48 -> 48 channels on 160-wide maps; the filter stays in
    # registers and a workgroup walks a strip of rows through an LDS ring.  Variants = strip heights.
    for v, rows in enumerate([40, 20, 10, 8, 4, 2] * 2):   # ids 0..5: 4 waves x all channels; 6..11: 12 waves x one channel tile
        run_case(rmr, 2, 40, 160, 48, 48, 3, 1, True, True, tile=300 + v, seed=80 + v)   # 2 images, H = 40
    run_case(rmr, 1, 160, 160, 48, 48, 3, 1, True, False, tile=300, seed=90)   # 4 strips of 40 rows, no residual
    run_case(rmr, 3, 6, 160, 48, 48, 3, 1, False, True, tile=305, seed=91)     # strips of 2 rows, H = 6: ring wraps at image ends
    run_case(rmr, 1, 10, 160, 48, 48, 3, 1, True, True, tile=302, seed=92)     # one strip = whole image
    run_case(rmr, 1, 20, 160, 48, 41, 3, 1, True, True, tile=301, seed=93)     # 41 channels padded to 48
    run_case(rmr, 1, 160, 160, 48, 48, 3, 1, True, True, tile=306, seed=94)    # 12-wave layout, 4 strips, residual
    run_case(rmr, 3, 6, 160, 48, 48, 3, 1, False, False, tile=311, seed=95)    # 12-wave layout, 2-row strips
    run_case(rmr, 1, 20, 160, 48, 41, 3, 1, True, False, tile=307, seed=96)    # 12-wave layout, padded channels
    # ids 12..16: the pipelined kernel (conv_wsp_kernel: tile-outer, a tile's SiLU and drain under the next tiles' MFMAs);
    # 12-15 two waves per SIMD (strips of 160 / 80 / 40 / 20 rows), 16 one wave per SIMD (160)
    for v, rows in enumerate([160, 80, 40, 20, 160]):
        run_case(rmr, 2, 160, 160, 48, 48, 3, 1, True, True, tile=312 + v, seed=100 + v)
        run_case(rmr, 1, rows, 160, 48, 41, 3, 1, v % 2 == 0, False, tile=312 + v, seed=110 + v)   # one strip, padded channels
    run_case(rmr, 3, 20, 160, 48, 48, 3, 1, False, True, tile=315, seed=120)    # H = 20: the strip is the image, rows above / below arrive as zeros
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 80, 48), np.float32), np.zeros((48, 48, 3, 3), np.float32), None, 1, 1,
                   False, tile=300)  # W != 160
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 160, 48), np.float32), np.zeros((96, 48, 3, 3), np.float32), None, 1, 1,
                   False, tile=300)  # 96 output channels
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 6, 160, 48), np.float32), np.zeros((48, 48, 3, 3), np.float32), None, 1, 1,
                   False, tile=300)  # H = 6 is not a multiple of the 40-row strip


def test_conv_direct_every_tile(rmr):
    # conv_direct.hip (ids 400..): fragments straight from L2, K split across the waves of a
    # workgroup, partial tiles reduced in LDS in wave order.  Every tile on 3x3 (borders, ragged M,
    # several images), stride 2, 1x1, and K-step counts below / not divisible by the wave count.
    tiles = [(64, 96, 4), (32, 96, 8), (64, 48, 8), (32, 48, 8), (16, 96, 8), (16, 48, 8), (32, 96, 4),
             (64, 64, 4), (32, 64, 8), (16, 64, 8), (64, 48, 4), (16, 16, 8)]
    for t, (bm, bn, nw) in enumerate(tiles):
        run_case(rmr, 2, 13, 11, 64, bn * 2, 3, 1, True, True, tile=400 + t, seed=100 + t)    # 18 K steps, ragged M
        run_case(rmr, 1, 9, 9, 32, bn, 1, 1, True, False, tile=400 + t, seed=120 + t)         # 1x1, ONE K step: idle waves
    run_case(rmr, 4, 40, 40, 192, 192, 3, 1, True, True, tile=400, seed=140)    # the batch-4 P4 bottleneck
    run_case(rmr, 1, 20, 20, 288, 288, 3, 1, True, True, tile=401, seed=141)    # 81 K steps over 8 waves
    run_case(rmr, 1, 41, 37, 96, 192, 3, 2, True, False, tile=406, seed=142)    # stride 2, odd sizes
    run_case(rmr, 1, 20, 20, 576, 16, 1, 1, False, False, tile=411, seed=143)   # 16-channel head conv, no activation
    run_case(rmr, 3, 8, 8, 160, 64, 3, 1, False, True, tile=407, seed=144)      # 5 chunks per tap
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 8, 48), np.float32), np.zeros((96, 48, 3, 3), np.float32), None, 1, 1,
                   False, tile=400)  # Cin % 32 != 0
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 8, 32), np.float32), np.zeros((48, 32, 3, 3), np.float32), None, 1, 1,
                   False, tile=400)  # 48 output channels on a 96-wide tile


def test_conv_stem(rmr):
    # conv_stem.hip (id 500): 3x3 / stride 2, 3 -> 48 channels; the input patch of an 8 x 32 output
    # tile is staged once, image borders arrive as zeros from the buffer bounds check
    run_case(rmr, 2, 64, 128, 3, 48, 3, 2, True, False, tile=500, seed=150)    # 4 x 2 tiles per image, every border
    run_case(rmr, 1, 16, 64, 3, 48, 3, 2, False, False, tile=500, seed=151)    # one tile, no activation
    run_case(rmr, 1, 160, 192, 3, 41, 3, 2, True, False, tile=500, seed=152)   # 41 channels padded to 48
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 16, 48, 3), np.float32), np.zeros((48, 3, 3, 3), np.float32), None, 2, 1,
                   False, tile=500)  # output width 24 is not a multiple of the 32-pixel tile
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 16, 64, 3), np.float32), np.zeros((48, 3, 3, 3), np.float32), None, 1, 1,
                   False, tile=500)  # stride 1


def test_conv_weights_stationary_f16_views(rmr, monkeypatch):
    """The same layers through their PRODUCTION epilogues (RMR_CONV2D_OUT16: the f16 output view the network uses -- LDS
    stages, 16-byte stores, the shortcut added in registers at the drain), old kernel and pipelined kernel: each within
    the f16 rounding of the torch reference, and bit-identical to each other (same f32 operation order per value)."""
    monkeypatch.setenv("RMR_CONV2D_OUT16", "1")
    rng = np.random.default_rng(7)
    for res in (False, True):
        for silu in (True, False):
            x = r16(rng.normal(0, 1, (2, 160, 160, 48)).astype(np.float32))
            wt = r16((rng.normal(0, 1, (48, 48, 3, 3)) / np.sqrt(48 * 9)).astype(np.float32))
            b = rng.normal(0, 0.5, 48).astype(np.float32)
            r = r16(rng.normal(0, 1, (2, 160, 160, 48)).astype(np.float32)) if res else None
            want = ref_conv(x, wt, b, 1, 1, silu, r)
            outs = {t: rmr.conv2d(x, wt, b, 1, 1, silu, r, tile=t) for t in (300, 306, 312, 313, 314, 315, 316)}
            for t, got in outs.items():
                err = np.abs(got - want).max()
                assert err <= 2e-3 * max(1.0, np.abs(want).max()), f"tile {t} res {res} silu {silu}: max err {err}"
                assert np.array_equal(got, outs[300]), f"tile {t} res {res} silu {silu}: not bit-identical to tile 300"


def test_fused_bottleneck_matches_torch_and_the_two_launches(rmr, monkeypatch):
    """conv_wsf (ids 340..): out = x + SiLU(conv(SiLU(conv(x) + b)) + b), 3x3 / 48 -> 48 on 160-wide maps, one launch with the
    hidden tensor in LDS (rmr_conv2d runs it with one filter for both convolutions).  Against torch with the hidden tensor
    rounded to f16 where the two-launch plan stores it; and bit-identical to those two launches through their f16 views."""
    rng = np.random.default_rng(11)
    for n, h, cout in ((2, 160, 48), (3, 40, 48), (1, 20, 41)):
        x = r16(rng.normal(0, 1, (n, h, 160, 48)).astype(np.float32))
        wt = r16((rng.normal(0, 1, (cout, 48, 3, 3)) / np.sqrt(48 * 9)).astype(np.float32))
        if cout < 48:   # the second convolution consumes the first's output: pad the filter to 48 x 48 by hand
            wt = np.concatenate([wt, np.zeros((48 - cout, 48, 3, 3), np.float32)])
        b = rng.normal(0, 0.5, 48).astype(np.float32)
        hid = r16(ref_conv(x, wt, b, 1, 1, True, None))
        want = ref_conv(hid, wt, b, 1, 1, True, x)
        for v, rows in enumerate([160, 80, 40, 20]):
            if h % rows:
                continue
            got = rmr.conv2d(x, wt, b, 1, 1, True, None, tile=340 + v)      # f32 view
            err = np.abs(got - want).max()
            assert err <= 2e-3 * max(1.0, np.abs(want).max()), f"variant {v} n {n} h {h}: max err {err}"
    monkeypatch.setenv("RMR_CONV2D_OUT16", "1")
    x = r16(rng.normal(0, 1, (2, 160, 160, 48)).astype(np.float32))
    wt = r16((rng.normal(0, 1, (48, 48, 3, 3)) / np.sqrt(48 * 9)).astype(np.float32))
    b = rng.normal(0, 0.5, 48).astype(np.float32)
    hid = rmr.conv2d(x, wt, b, 1, 1, True, None, tile=312)        # first launch, f16 output
    two = rmr.conv2d(hid, wt, b, 1, 1, True, x, tile=312)         # second launch with the shortcut
    for v in range(4):
        one = rmr.conv2d(x, wt, b, 1, 1, True, None, tile=340 + v)
        assert np.array_equal(one, two), f"variant {v}: fused bottleneck differs from the two launches"
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 80, 48), np.float32), np.zeros((48, 48, 3, 3), np.float32), None, 1, 1, True, tile=340)  # W != 160


def test_conv_ws_stride2(rmr):
    # conv_ws_s2.hip (ids 600..): 3x3 / stride 2, 48 -> 96 channels, 320-wide input; ring rows are
    # de-interleaved by column parity, each workgroup takes one half of the map's width
    for v, rows in enumerate([40, 20, 10, 8, 4, 2]):
        run_case(rmr, 2, 80, 320, 48, 96, 3, 2, True, False, tile=600 + v, seed=160 + v)   # Ho = 40
    run_case(rmr, 1, 320, 320, 48, 96, 3, 2, True, False, tile=600, seed=170)              # the real layer, 4 strips x 2 halves
    run_case(rmr, 3, 12, 320, 48, 89, 3, 2, False, False, tile=605, seed=171)              # 89 channels padded to 96, no activation
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 16, 160, 48), np.float32), np.zeros((96, 48, 3, 3), np.float32), None, 2, 1,
                   False, tile=600)  # input width 160
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 16, 320, 48), np.float32), np.zeros((96, 48, 3, 3), np.float32), None, 1, 1,
                   False, tile=600)  # stride 1


def test_conv_pointwise_weights_stationary(rmr):
    # conv_pw.hip (ids 700..): 1x1 layers with K = 96 / 192 / 384 / 576, N = 96 / 192; the filter stays in
    # registers and each workgroup walks many row blocks through a DMA ring whose waits count the
    # stores of recent epilogues.  Ragged M (the last block is partly out of range), fewer blocks than
    # ring stages, and more blocks than workgroups (the persistent walk proper, several blocks per CU).
    shapes = {0: (96, 96), 1: (96, 96), 8: (96, 96), 2: (192, 96), 3: (192, 96), 9: (192, 96), 4: (192, 192),
              5: (192, 192), 6: (384, 192), 7: (384, 192), 10: (576, 192), 11: (384, 192)}
    for v, (k, n) in shapes.items():
        run_case(rmr, 2, 13, 11, k, n, 1, 1, True, False, tile=700 + v, seed=180 + v)        # 286 rows: 2-3 blocks
        run_case(rmr, 1, 5, 7, k, n - 7, 1, 1, False, False, tile=700 + v, seed=190 + v)     # one partial block, padded channels
    run_case(rmr, 2, 240, 200, 96, 96, 1, 1, True, False, tile=700, seed=200)    # 750 blocks of 128 over 512 workgroups
    run_case(rmr, 1, 300, 257, 96, 96, 1, 1, True, False, tile=701, seed=201)    # 302 blocks of 256, ragged
    run_case(rmr, 1, 250, 200, 192, 192, 1, 1, True, False, tile=704, seed=202)  # 391 blocks, two stages per block
    run_case(rmr, 1, 200, 180, 384, 192, 1, 1, True, False, tile=706, seed=203)  # 282 blocks, four stages per block
    run_case(rmr, 1, 160, 130, 576, 192, 1, 1, True, False, tile=710, seed=204)  # 325 blocks of 64, six stages per block
    # layers wider than a workgroup's 192 channels: 2-3 workgroups walk the same rows side by side
    run_case(rmr, 2, 13, 11, 768, 384, 1, 1, True, False, tile=712, seed=205)
    run_case(rmr, 1, 150, 120, 768, 384, 1, 1, True, False, tile=712, seed=206)  # 282 blocks x 2 channel tiles, eight stages per block
    run_case(rmr, 1, 130, 100, 384, 384, 1, 1, True, False, tile=706, seed=207)
    run_case(rmr, 1, 9, 9, 576, 570, 1, 1, False, False, tile=710, seed=208)     # three channel tiles, padded channels
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 8, 96), np.float32), np.zeros((96, 96, 3, 3), np.float32), None, 1, 1,
                   False, tile=700)  # 3x3
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 8, 192), np.float32), np.zeros((96, 192, 1, 1), np.float32), None, 1, 0,
                   False, tile=700)  # variant 0 is K = 96


def test_conv_matches_c_oracle(rmr, oracle):
    # the plain-C direct convolution (oracle/rmr_oracle.c) agrees with both
    rng = np.random.default_rng(5)
    x = r16(rng.normal(0, 1, (1, 10, 12, 16)).astype(np.float32))
    wt = r16((rng.normal(0, 1, (32, 16, 3, 3)) / 12).astype(np.float32))
    b = rng.normal(0, 0.5, 32).astype(np.float32)
    got = rmr.conv2d(x, wt, b, 1, 1, True)
    want = oracle.conv2d_nchw(x.transpose(0, 3, 1, 2), wt, b, 1, 1, True).transpose(0, 2, 3, 1)
    assert np.abs(got - want).max() <= 2e-3


def test_conv_bad_arguments(rmr):
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 4, 4, 8), np.float32), np.zeros((48, 8, 3, 3), np.float32), None, 1, 1,
                   False, tile=0)  # tile 0 has BN = 96, 48 is not a multiple


def test_conv_t32_every_tile(rmr):
    # 32x32x16-MFMA 3x3 / stride-1 kernel (conv_t32.hip): ids 800..; persistent workgroups walking tiles with a
    # continuous DMA stream, weights pre-packed as LDS images, fragment reads ahead of the MFMAs across the
    # barrier, lane masks for the border taps, 16-byte stores (lane-pair exchange or rows staged through LDS)
    tiles = [(256, 192), (256, 192), (256, 192), (512, 96), (256, 256), (512, 64), (256, 96), (256, 64), (128, 192),
             (128, 192), (256, 96), (128, 128), (256, 64),   # 9..12: two four-wave workgroups per CU
             None, None,                                      # 13, 14: tiles 10 / 9 with rows staged through an input buffer (wide maps only)
             (128, 96)]                                       # 15: three two-wave workgroups per CU (round 6)
    for t, shape in enumerate(tiles):
        if shape is None:
            continue
        bm, bn = shape
        run_case(rmr, 3, 20, 20, 64, bn, 3, 1, True, True, tile=800 + t, seed=t)            # 3 images in ~5 tiles
        run_case(rmr, 1, 19, 23, 32, bn * 2, 3, 1, True, False, tile=800 + t, seed=40 + t)  # odd W, ragged M, 1 chunk
    run_case(rmr, 2, 40, 40, 192, 192, 3, 1, True, True, tile=800, seed=70)   # 6 chunks, 54 taps
    run_case(rmr, 1, 80, 80, 192, 192, 3, 1, True, False, tile=801, seed=71)  # W = 80: 27 input blocks
    run_case(rmr, 1, 80, 80, 96, 96, 3, 1, True, True, tile=803, seed=72)     # 512-row tiles on 80-wide maps
    run_case(rmr, 2, 20, 20, 288, 288, 3, 1, True, True, tile=806, seed=73)   # 9 chunks, 3 channel tiles
    run_case(rmr, 1, 5, 5, 32, 96, 3, 1, False, False, tile=806, seed=74)     # tile far larger than the image
    run_case(rmr, 5, 40, 40, 96, 256, 3, 1, True, False, tile=804, seed=75)   # fused head conv shape
    # more tiles than workgroups: every workgroup walks several tiles (the stream crosses tile boundaries,
    # the channel tile changes under it)
    run_case(rmr, 160, 20, 20, 32, 192, 3, 1, True, True, tile=806, seed=76)  # 250 x 2 tiles of 256 x 96 on 512 slots... and
    run_case(rmr, 300, 20, 20, 32, 192, 3, 1, True, True, tile=802, seed=77)  # 469 tiles of 256 x 192 on 256 workgroups
    run_case(rmr, 330, 20, 20, 32, 96, 3, 1, True, False, tile=806, seed=78)  # 516 tiles on 512 workgroups: a few walk two
    run_case(rmr, 200, 20, 20, 64, 192, 3, 1, True, True, tile=809, seed=79)  # 625 tiles of 128 x 192 on 512 four-wave workgroups
    run_case(rmr, 1, 80, 80, 96, 96, 3, 1, True, True, tile=810, seed=80)     # four-wave 256 x 96 on 80-wide maps
    run_case(rmr, 2, 80, 80, 96, 96, 3, 1, True, True, tile=813, seed=81)     # ... with rows staged through the spent input buffer
    run_case(rmr, 3, 80, 80, 64, 192, 3, 1, True, False, tile=813, seed=82)   # 150 tiles, two channel tiles
    run_case(rmr, 40, 80, 80, 32, 96, 3, 1, True, True, tile=813, seed=83)    # 1000 tiles on 512 workgroups: the stage meets the next tile's DMAs
    run_case(rmr, 64, 20, 20, 288, 288, 3, 1, True, True, tile=815, seed=84)  # the layer tile 15 is for: 600 tiles on 768 two-wave workgroups
    run_case(rmr, 300, 20, 20, 32, 96, 3, 1, True, False, tile=815, seed=85)  # 938 tiles: some workgroups walk two
    run_case(rmr, 2, 40, 40, 96, 96, 3, 1, True, True, tile=810, seed=86)     # rows of HALF weight blocks (two waves share a KiB), 3 chunks
    # split-K (ids 1000 * split + 800 + tile; batches of 1-4 images): one workgroup per (tile, range of chunks), the partial
    # tiles summed in split order by the last arriver; launched twice by the test entry point (the tickets re-arm themselves)
    run_case(rmr, 4, 40, 40, 192, 192, 3, 1, True, True, tile=3000 + 812, seed=90)    # 25 x 3 tiles x 3 splits of 2 chunks
    run_case(rmr, 1, 40, 40, 192, 192, 3, 1, True, False, tile=6000 + 812, seed=91)   # one chunk per workgroup
    run_case(rmr, 4, 20, 20, 288, 288, 3, 1, True, True, tile=9000 + 810, seed=92)    # 7 x 3 tiles x 9 splits
    run_case(rmr, 1, 20, 20, 288, 288, 3, 1, True, False, tile=4000 + 810, seed=93)   # 9 chunks over 4 workgroups: 2, 2, 2, 3
    run_case(rmr, 4, 80, 80, 96, 96, 3, 1, True, True, tile=3000 + 810, seed=94)      # 100 tiles x 3 on 512 slots
    run_case(rmr, 1, 80, 80, 96, 96, 3, 1, True, False, tile=2000 + 803, seed=95)     # 13 tiles of 512 x 96, chunks 1 + 2
    run_case(rmr, 1, 19, 23, 64, 192, 3, 1, True, True, tile=2000 + 809, seed=96)     # ragged M
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((64, 40, 40, 64), np.float32), np.zeros((192, 64, 3, 3), np.float32), None, 1, 1,
                   False, tile=2000 + 812)  # 1200 workgroups: no slot per (tile, split)
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 4, 4, 48), np.float32), np.zeros((96, 48, 3, 3), np.float32), None, 1, 1,
                   False, tile=806)  # Cin = 48 is not a multiple of 32


SB = 100000   # kernel ids of the small-batch family (conv_sb.hip): SB + variant


def test_conv_sb_every_variant(rmr):
    # small batches (conv_sb.hip): tiles of 32 x 32 ... 128 x 128 whose whole operand set is in flight at once; even variants =
    # halo form (3x3 / stride 1), odd variants = gathered form (1x1, strided 3x3); 3-12 waves, those of a wave tile share K
    tiles = [(32, 32), (64, 32), (32, 64), (64, 64), (128, 32), (64, 32), (64, 32), (128, 32), (128, 64), (128, 64), (128, 64),
             (128, 96), (128, 96), (64, 32), (128, 64), (32, 32), (64, 32), (64, 64), (128, 32), (128, 64),
             (64, 32), (64, 32), (32, 32), (128, 32), (64, 64), (128, 64), (128, 96), (128, 64),   # 20..: four loader waves beside the computing waves
             (64, 32), (64, 32), (32, 32), (128, 32), (128, 64)]                               # 28..: eight
    def ran(*args, **kw):   # a variant whose ring cannot hold two stages of a layer refuses it (the tuner never offers it there)
        try:
            run_case(rmr, *args, **kw)
            return 1
        except rmr.InvalidArgument:
            return 0

    for t, (bm, bn) in enumerate(tiles):
        h, g = SB + 2 * t, SB + 2 * t + 1
        n_h = ran(3, 20, 20, 64, bn, 3, 1, True, True, tile=h, seed=t)                  # two chunks, shortcut
        n_h += ran(1, 19, 23, 32, bn * 2, 3, 1, True, False, tile=h, seed=20 + t)       # odd W, ragged M, one chunk
        n_g = ran(2, 20, 20, 192, bn, 1, 1, True, True, tile=g, seed=40 + t)            # 1x1, six units
        n_g += ran(1, 19, 23, 96, bn * 2, 1, 1, False, False, tile=g, seed=60 + t)      # 1x1, f32-style epilogue (no activation)
        n_g += ran(2, 20, 20, 64, bn, 3, 2, True, False, tile=g, seed=80 + t)           # 3x3 / stride 2, even size
        n_g += ran(1, 19, 23, 32, bn, 3, 2, True, False, tile=g, seed=100 + t)          # 3x3 / stride 2, odd size
        assert n_h >= 1 and n_g >= 3, (t, n_h, n_g)
    # the layers of a batch-1 frame (car: one image, armor: four crops)
    run_case(rmr, 1, 40, 40, 192, 192, 3, 1, True, True, tile=SB + 2, seed=200)     # six stages of 28 KiB: the whole K range in LDS
    run_case(rmr, 1, 20, 20, 288, 288, 3, 1, True, True, tile=SB + 0, seed=201)     # nine stages: more than the ring holds (refills)
    run_case(rmr, 4, 80, 80, 96, 96, 3, 1, True, True, tile=SB + 24, seed=202)      # 128 x 96 on 80-wide maps
    run_case(rmr, 1, 80, 80, 96, 96, 3, 1, True, False, tile=SB + 36, seed=203)     # two workgroups per CU: a ring of two stages
    run_case(rmr, 4, 40, 40, 192, 256, 3, 1, True, False, tile=SB + 18, seed=204)   # fused head conv shape
    run_case(rmr, 1, 20, 20, 1152, 576, 1, 1, True, False, tile=SB + 3, seed=205)   # SPPF cv2: 36 units
    run_case(rmr, 4, 40, 40, 768, 384, 1, 1, True, False, tile=SB + 19, seed=206)   # C2f cv2
    run_case(rmr, 1, 80, 80, 192, 384, 3, 2, True, False, tile=SB + 7, seed=207)    # model.5: 54 units, strided
    run_case(rmr, 1, 40, 40, 64, 64, 1, 1, False, False, tile=SB + 7, seed=208)     # DFL conv: bias only
    run_case(rmr, 1, 5, 5, 32, 96, 3, 1, False, False, tile=SB + 22, seed=209)      # tile far larger than the image
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 4, 4, 48), np.float32), np.zeros((96, 48, 3, 3), np.float32), None, 1, 1, False, tile=SB)   # Cin % 32
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 8, 8, 64), np.float32), np.zeros((64, 64, 3, 3), np.float32), None, 2, 1, False, tile=SB)   # halo form, strided layer


def test_conv_sb_seeded_sweep_of_shapes(rmr):
    """The variants of conv_sb on layers nobody hand-picked: 120 seeded draws of (images, map size, channels, kernel, stride,
    activation, shortcut, variant) -- odd maps, maps smaller than a tile, channel counts that leave a remainder of the channel
    tile, one to nine 32-channel chunks -- each against torch in f32 (run_case's tolerance).  A variant may refuse a layer (ring
    too small, Cout not a multiple of its tile); it must never run one wrong.  At least 70 of the draws must run."""
    rng = np.random.default_rng(5)
    n_var = 66
    ran = 0
    for case in range(120):
        k = int(rng.choice([1, 3]))
        stride = int(rng.choice([1, 2])) if k == 3 else 1
        n = int(rng.integers(1, 6))
        h, w = int(rng.integers(3, 45)), int(rng.integers(3, 45))
        cin = 32 * int(rng.integers(1, 10))
        cout = 32 * int(rng.integers(1, 9))
        silu = bool(rng.integers(0, 2))
        res = bool(rng.integers(0, 2)) and silu and stride == 1 and cin == cout
        v = int(rng.integers(0, n_var // 2)) * 2 + (0 if (k == 3 and stride == 1) else 1)
        try:
            run_case(rmr, n, h, w, cin, cout, k, stride, silu, res, tile=SB + v, seed=300 + case)
            ran += 1
        except rmr.InvalidArgument:
            pass
    assert ran >= 70, ran


def test_conv_g32_every_tile(rmr):
    # the gathered form of conv_t32 (conv_g32.hip, ids 950..): 1x1 layers and 3x3 layers of any stride, one
    # (tap, 32-channel chunk) stage = the tile's pixel rows at that tap + the weight slice, padding as
    # out-of-range DMA offsets
    tiles = [(256, 192, 1), (512, 96, 1), (256, 192, 3), (512, 96, 3), (256, 128, 1), (256, 128, 3),
             (128, 192, 1), (128, 192, 3)]   # 6, 7: two four-wave workgroups per CU
    for t, (bm, bn, k) in enumerate(tiles):
        cmin = 32 * (6 if k == 1 else 1)   # a tile has at least a ring of stages
        run_case(rmr, 3, 20, 20, cmin, bn, k, 1, True, True, tile=950 + t, seed=t)             # ~5 tiles
        run_case(rmr, 1, 19, 23, cmin + 32, bn * 2, k, 1, True, False, tile=950 + t, seed=20 + t)  # odd W, ragged M, 2 channel tiles
        if k == 3:
            run_case(rmr, 2, 20, 20, 64, bn, 3, 2, True, False, tile=950 + t, seed=40 + t)     # stride 2, even size
            run_case(rmr, 1, 19, 23, 32, bn, 3, 2, True, False, tile=950 + t, seed=50 + t)     # stride 2, odd size
    run_case(rmr, 2, 40, 40, 768, 384, 1, 1, True, False, tile=950, seed=70)    # C2f cv2 shape: 24 stages
    run_case(rmr, 1, 20, 20, 1152, 576, 1, 1, True, False, tile=950, seed=71)   # SPPF cv2 shape: 3 channel tiles
    run_case(rmr, 1, 80, 80, 192, 384, 3, 2, True, False, tile=952, seed=72)    # model.5 shape: 54 stages
    run_case(rmr, 1, 5, 5, 192, 96, 1, 1, False, False, tile=951, seed=73)      # tile far larger than the image
    # more tiles than workgroups: the stream crosses tile boundaries, the channel tile changes under it
    run_case(rmr, 300, 20, 20, 192, 192, 1, 1, True, True, tile=950, seed=74)   # 469 tiles on 256 workgroups
    run_case(rmr, 150, 40, 40, 32, 192, 3, 2, True, False, tile=952, seed=75)   # 235 x ... strided, one chunk per tile
    run_case(rmr, 330, 20, 20, 128, 192, 1, 1, True, False, tile=951, seed=76)  # 258 x 2 tiles of 512 x 96
    run_case(rmr, 200, 20, 20, 192, 192, 1, 1, True, True, tile=956, seed=77)   # 625 tiles on 512 four-wave workgroups
    run_case(rmr, 200, 40, 40, 64, 192, 3, 2, True, False, tile=957, seed=78)   # the same, strided
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 4, 4, 64), np.float32), np.zeros((192, 64, 1, 1), np.float32), None, 1, 0,
                   False, tile=950)  # two stages: shorter than the ring
    with pytest.raises(rmr.InvalidArgument):
        rmr.conv2d(np.zeros((1, 4, 4, 192), np.float32), np.zeros((192, 192, 3, 3), np.float32), None, 1, 1,
                   False, tile=950)  # a 1x1 tile on a 3x3 layer


def _e4m3(a):
    """OCP e4m3fn rounding (nearest even, saturating), as the packer and the device quantiser do it"""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).clamp(-448, 448).to(torch.float8_e4m3fn).float().numpy()


def run_case_f8(rmr, n, h, w, cin, cout, silu, res, tile, seed=0):
    """conv_t32f8 against plain PyTorch fp32 conv2d on the SAME e4m3 operands: inputs rounded to f16 and then
    to e4m3 (unit scale), weights rounded to f16, divided by their row's scale max|w| / 448, rounded to e4m3 --
    so that only the f32 accumulation order differs (2e-3 of the output scale, as for the f16 kernels)."""
    rng = np.random.default_rng(seed)
    x = r16(rng.normal(0, 1, (n, h, w, cin)).astype(np.float32))
    wt = r16((rng.normal(0, 1, (cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
    b = rng.normal(0, 0.5, cout).astype(np.float32)
    r = r16(rng.normal(0, 1, (n, h, w, cout)).astype(np.float32)) if res else None
    got = rmr.conv2d(x, wt, b, 1, 1, silu, r, tile=tile)
    scale = (np.abs(wt).reshape(cout, -1).max(1) / np.float32(448.0)).astype(np.float32)
    wq = _e4m3(wt / scale[:, None, None, None]) * scale[:, None, None, None]
    want = ref_conv(_e4m3(x), wq.astype(np.float32), b, 1, 1, silu, r)
    err = np.abs(got - want).max()
    assert err <= 2e-3 * max(1.0, np.abs(want).max()), f"max err {err}"
    # and the quantisation itself is not the identity: the f16 result differs visibly
    assert np.abs(got - ref_conv(x, wt, b, 1, 1, silu, r)).max() > 5e-3


def test_conv_t32f8_every_tile(rmr):
    # the fp8 form of conv_t32 (conv_t32f8.hip, ids 900..): e4m3 operands on v_mfma_scale_f32_32x32x64_f8f6f4
    tiles = [(256, 192), (256, 192), (512, 96), (256, 256), (512, 64), (256, 96), (128, 192), (256, 64)]
    for t, (bm, bn) in enumerate(tiles):
        run_case_f8(rmr, 3, 20, 20, 64, bn, True, True, 900 + t, seed=t)             # one 64-channel chunk
        run_case_f8(rmr, 1, 19, 23, 96, bn * 2, True, False, 900 + t, seed=40 + t)   # 1.5 chunks: zero-padded tail, odd W
    run_case_f8(rmr, 2, 40, 40, 192, 192, True, True, 900, seed=70)     # 3 chunks
    run_case_f8(rmr, 1, 80, 80, 96, 96, True, True, 902, seed=71)       # 512-row tiles on 80-wide maps
    run_case_f8(rmr, 2, 20, 20, 288, 288, True, True, 905, seed=72)     # 4.5 chunks, 3 channel tiles
    run_case_f8(rmr, 1, 5, 5, 32, 96, False, False, 905, seed=73)       # half a chunk, tile far larger than the image
    run_case_f8(rmr, 300, 20, 20, 64, 192, True, True, 906, seed=74)    # workgroups walk several tiles
