#!/usr/bin/env python3
"""Every variant of conv_sb on one layer, hot operands: sb_variants.py n h w cin cout k stride res"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rm_radar_amd as rmr  # noqa: E402

n, h, w, cin, cout, k, s, res = [int(v) for v in sys.argv[1:9]]
ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
flops = 2.0 * n * ho * wo * cout * cin * k * k
print(f"--- n{n} M{n * ho * wo} N{cout} K{cin * k * k} k{k} s{s} res{res}  ({flops / 1e9:.2f} GFLOP)")
out = []
for v in range(int(os.environ.get("SB_NVAR", "80"))):
    try:
        t = min(rmr.conv_bench(n, h, w, cin, cout, k, s, 100000 + v, bool(res), 30) for _ in range(3)) * 1e3
    except rmr.RmrError:
        continue
    out.append((t, v))
for t, v in sorted(out):
    print(f"    sb {v:3d}: {t:7.2f} us  {flops / t / 1e6:7.1f} TFLOP/s")
