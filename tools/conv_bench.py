#!/usr/bin/env python3
"""Time one conv layer with given kernels on device-resident data (rmr_conv_bench).
usage: conv_bench.py n,h,w,cin,cout[,k[,stride[,res]]] kernel[,kernel...] [reps]
e.g.   conv_bench.py 64,40,40,192,192 228,233,800,801"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rm_radar_amd as rmr  # noqa: E402

shape = [int(v) for v in sys.argv[1].split(",")]
n, h, w, cin, cout = shape[:5]
k = shape[5] if len(shape) > 5 else 3
stride = shape[6] if len(shape) > 6 else 1
res = bool(shape[7]) if len(shape) > 7 else False
kernels = [int(v) for v in sys.argv[2].split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
flops = 2.0 * n * ho * wo * cout * cin * k * k
for kid in kernels:
    try:
        best = min(rmr.conv_bench(n, h, w, cin, cout, k, stride, kid, res, reps) for _ in range(3))
        print(f"M{n * ho * wo} N{cout} K{cin * k * k} k{k} s{stride} res{int(res)} kernel {kid}: {best * 1e3:8.1f} us  {flops / best / 1e9:7.1f} TFLOP/s", flush=True)
    except rmr.RmrError as e:
        print(f"kernel {kid}: {e}", flush=True)
