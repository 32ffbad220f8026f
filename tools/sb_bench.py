#!/usr/bin/env python3
"""The layers of a batch-1 frame (car: one image, armor: four crops), one at a time: the kernel the round-4 plan runs them on
against every variant of the small-batch family (conv_sb.hip, ids 100000 + variant), on operands that are hot in the L2
(back-to-back launches of one layer) and cold (RMR_BENCH_COLD replicas: what a layer of a network sees).
usage: sb_bench.py [cold copies, default 40]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rm_radar_amd as rmr  # noqa: E402

SB = 100000
LAYERS = [  # n, h, w, cin, cout, k, stride, res, the plan's kernels
    (1, 40, 40, 192, 192, 3, 1, 1, [220, 2812]),
    (1, 80, 80, 96, 96, 3, 1, 1, [220]),
    (1, 20, 20, 288, 288, 3, 1, 1, [3109, 411]),
    (1, 40, 40, 192, 256, 3, 1, 0, [6807]),
    (1, 80, 80, 192, 256, 3, 1, 0, [2812]),
    (1, 20, 20, 576, 256, 3, 1, 0, [6106]),
    (1, 80, 80, 96, 64, 3, 1, 0, [220]),
    (1, 20, 20, 1152, 576, 1, 1, 0, [111]),
    (1, 40, 40, 768, 384, 1, 1, 0, [109]),
    (1, 80, 80, 192, 192, 1, 1, 0, [705]),
    (1, 40, 40, 384, 576, 3, 2, 0, [4106]),
    (1, 80, 80, 192, 384, 3, 2, 0, [109]),
    (4, 40, 40, 192, 192, 3, 1, 1, [215]),
    (4, 80, 80, 96, 96, 3, 1, 1, [212]),
    (4, 20, 20, 288, 288, 3, 1, 1, [220]),
    (4, 20, 20, 1152, 576, 1, 1, 0, [106]),
    (4, 40, 40, 768, 384, 1, 1, 0, [20]),
]
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nvar = 66


def best(n, h, w, cin, cout, k, s, res, kid, reps=30):
    try:
        return min(rmr.conv_bench(n, h, w, cin, cout, k, s, kid, bool(res), reps) for _ in range(3)) * 1e3
    except rmr.RmrError:
        return None


for (n, h, w, cin, cout, k, s, res, plan) in LAYERS:
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    flops = 2.0 * n * ho * wo * cout * cin * k * k
    rows = {}
    for mode in ("hot", "cold"):
        if mode == "cold":
            os.environ["RMR_BENCH_COLD"] = str(copies)
        else:
            os.environ.pop("RMR_BENCH_COLD", None)
        for kid in plan + [SB + v for v in range(nvar)]:
            t = best(n, h, w, cin, cout, k, s, res, kid)
            if t is not None:
                rows.setdefault(kid, {})[mode] = t
    os.environ.pop("RMR_BENCH_COLD", None)
    print(f"--- n{n} M{n * ho * wo} N{cout} K{cin * k * k} k{k} s{s} res{res}  ({flops / 1e9:.2f} GFLOP)")
    sb = {kid: r for kid, r in rows.items() if kid >= SB}
    for kid in plan:
        if kid in rows:
            r = rows[kid]
            print(f"    plan {kid:6d}: hot {r.get('hot', 0):7.1f} us  cold {r.get('cold', 0):7.1f} us")
    for kid, r in sorted(sb.items(), key=lambda kv: kv[1].get("cold", 1e9))[:6]:
        print(f"    sb {kid - SB:3d}      : hot {r.get('hot', 0):7.1f} us  cold {r.get('cold', 0):7.1f} us   ({flops / r['cold'] / 1e6:6.1f} TFLOP/s cold)")
    sys.stdout.flush()
