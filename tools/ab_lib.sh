#!/bin/bash
# Same-box A/B of two BUILDS under ONE bench.py (this tree's): <old librmr.so> with <its plan directory> against this tree's
# library and plans, runs alternating.  The measurement is the same program either way (round 5's A/B ran each tree
# with its own bench.py, whose latency legs marshalled differently).
# usage: bash tools/ab_lib.sh _ab_r05/rm_radar_amd/_build/librmr.so _ab_r05/profiles/plans r05 r06 > profiles/r06_ab_r05_vs_r06.txt
OLDLIB=$1; OLDPLANS=$2; A=$3; B=$4
OUT=$(pwd)/gpurun_out/ab; mkdir -p $OUT
CMD="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --seconds 5"
for i in 1 2; do
    RMR_LIB=$OLDLIB $CMD --plan $OLDPLANS > $OUT/${A}_$i.json 2> $OUT/${A}_$i.err
    $CMD > $OUT/${B}_$i.json 2> $OUT/${B}_$i.err
done
echo "# Same-box A/B (one gpurun call, one MI355X, runs alternating), BOTH builds driven by this tree's bench.py: the $A library ($OLDLIB, plans $OLDPLANS)"
echo "# against the $B library under its committed plans.  $CMD"
python - $OUT $A $B <<'P'
import json, sys, os
out, a, b = sys.argv[1:4]
print("# file            frames/s  steady(5 s)  p50_ms  p99_ms  car_ms  armor_ms  first_layer_ms  dominant TFLOP/s  all-conv TFLOP/s  parity  network")
for tag in (a, b):
    for i in (1, 2):
        p = os.path.join(out, f"{tag}_{i}.json")
        try:
            d = [json.loads(l) for l in open(p) if l.startswith("{")][-1]
        except Exception as e:   # noqa: BLE001
            print(f"{tag}_{i}.json  unreadable: {e}")
            continue
        st = d["stage_ms_per_step"]
        print(f"{tag}_{i}.json{'':7s}{d['value']:9.1f} {d['steady_state']['value']:11.1f} {d['p50_ms_batch1']:7.3f} {d['p99_ms_batch1']:7.3f} "
              f"{st['network, car stage']:7.3f} {st['network, armor stage']:9.3f} {st['first layer + letterbox sampling (car + armor)']:9.3f} "
              f"{d['roofline']['achieved']:14.1f} {d['roofline_all_conv_launches']['achieved']:16.1f}  {d.get('parity_checked')}  {(d.get('parity') or {}).get('network_checked')}   {d['config']['kernel_plan'][:40]}")
P
