#!/usr/bin/env python3
"""An outside yardstick for the convolution engine (VERDICT r04 "missing" #3): the layers that carry the batch-64 step
(`layer_roofline.top` of the bench line), each timed three ways on the SAME box, in the SAME session, on the same kind of data
(uniform random f16):

  (a) torch.matmul on the layer's GEMM shape M x N x K -- hipBLASLt / rocBLAS, the vendor's GEMM with NO im2col work at all
      (the im2col matrix is materialised beforehand and not timed: an upper bound of what a convolution of that shape can reach);
  (b) torch.nn.functional.conv2d, channels_last f16 -- MIOpen, the vendor's convolution (benchmark mode: it picks its solver);
  (c) rmr_conv_bench with the kernel the committed plan runs the layer on.

Measurement only: nothing here is imported by the package or by bench.py.  Under `rocprofv3 --kernel-trace` the same run names
the vendor kernels (tools/rocpd_summary.py on the database gives their VGPRs / LDS / durations).

usage: python tools/yardstick.py [reps]          (GPU box; ~2 minutes)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import rm_radar_amd as rmr  # noqa: E402

# n, h, w, cin, cout, k, stride, residual, the plan's kernel id, the row of layer_roofline.top it stands for
LAYERS = [
    (256, 40, 40, 192, 192, 3, 1, 1, 810, "conv n256 M409600 N192 K1728 k3 s1 g10"),
    (256, 80, 80, 96, 96, 3, 1, 1, 806, "conv n256 M1638400 N96 K864 k3 s1 g6"),
    (64, 40, 40, 192, 192, 3, 1, 1, 812, "conv n64 M102400 N192 K1728 k3 s1 g12"),
    (256, 80, 80, 192, 256, 3, 1, 0, 804, "conv n256 M1638400 N256 K1728 k3 s1 g4"),
    (256, 80, 80, 96, 96, 3, 1, 1, 810, "conv n256 M1638400 N96 K864 k3 s1 g10"),
    (256, 20, 20, 288, 288, 3, 1, 1, 810, "conv n256 M102400 N288 K2592 k3 s1 g10"),
    (256, 80, 80, 192, 192, 3, 1, 0, 810, "conv n256 M1638400 N192 K1728 k3 s1 g10"),
    (256, 320, 320, 48, 96, 3, 2, 0, 600, "conv n256 M6553600 N96 K432 k3 s2 v0"),
]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
torch.backends.cudnn.benchmark = True   # MIOpen's find mode: the best solver it has for the shape
dev = torch.device("cuda:0")


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best   # ms


print(f"{'layer':46s} {'GFLOP':>7s} | {'hipBLASLt GEMM':>15s} | {'MIOpen conv2d':>15s} | {'rmr (plan kernel)':>18s} | verdict")
for (n, h, w, cin, cout, k, s, res, kid, name) in LAYERS:
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    M, N, K = n * ho * wo, cout, cin * k * k
    flops = 2.0 * M * N * K
    # (a) the GEMM of the shape, operands resident, uniform random f16 (what rmr_conv_bench feeds its kernels)
    gemm_ms = None
    try:
        a = (torch.rand(M, K, device=dev, dtype=torch.float16) * 2 - 1)
        b = (torch.rand(K, N, device=dev, dtype=torch.float16) * 2 - 1) / (K ** 0.5)
        gemm_ms = timed(lambda: torch.matmul(a, b), reps)
        del a, b
    except RuntimeError as e:   # out of memory on the largest shapes: say so
        print(f"  ({name}: GEMM operands do not fit: {str(e)[:60]})")
    torch.cuda.empty_cache()
    # (b) the vendor's convolution, channels_last f16, bias and SiLU not included (a bare convolution: its best case)
    x = (torch.rand(n, cin, h, w, device=dev, dtype=torch.float16) * 2 - 1).contiguous(memory_format=torch.channels_last)
    wt = ((torch.rand(cout, cin, k, k, device=dev, dtype=torch.float16) * 2 - 1) / (K ** 0.5)).contiguous(memory_format=torch.channels_last)
    conv_ms = timed(lambda: F.conv2d(x, wt, None, stride=s, padding=k // 2), reps)
    del x, wt
    torch.cuda.empty_cache()
    # (c) this engine: conv + bias + SiLU (+ shortcut where the layer has one), the plan's kernel
    rmr_ms = min(rmr.conv_bench(n, h, w, cin, cout, k, s, kid, bool(res), reps) for _ in range(3))
    tf = lambda ms: flops / ms / 1e9 if ms else float("nan")   # noqa: E731
    best_vendor = max(tf(gemm_ms), tf(conv_ms)) if gemm_ms else tf(conv_ms)
    ratio = best_vendor / tf(rmr_ms)
    verdict = ("vendor ahead by %.0f %%: its tile shape is the next experiment" % ((ratio - 1) * 100)) if ratio > 1.10 else \
              ("within 10 %% of the vendor's best (%.2fx)" % ratio) if ratio > 0.95 else ("ahead of both vendor paths (%.2fx)" % (1 / ratio))
    g = f"{gemm_ms * 1e3:7.0f} us {tf(gemm_ms):5.0f} TF" if gemm_ms else "      n/a      "
    print(f"{name:46s} {flops / 1e9:7.1f} | {g} | {conv_ms * 1e3:7.0f} us {tf(conv_ms):5.0f} TF | {rmr_ms * 1e3:7.0f} us {tf(rmr_ms):5.0f} TF    | {verdict}", flush=True)
