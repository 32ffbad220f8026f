#!/bin/bash
# Same-box A/B of two trees of this repo (one gpurun call, runs alternating): <old tree dir> against the tree of the repo root,
# each under its own committed plan.  The old tree: git archive <commit> | tar -x -C _ab_r04 && (cd _ab_r04 && python -c "import __graft_entry__ as g; g.build()").
# usage: bash tools/ab_rounds.sh _ab_r04 r04 r05 > profiles/r05_ab_r04_vs_r05.txt
OLD=$1; A=$2; B=$3
OUT=$(pwd)/gpurun_out/ab; mkdir -p $OUT   # (scratch on the GPU box; redirect the table there or to stdout)
CMD="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --seconds 5"
for i in 1 2; do
    (cd $OLD && $CMD > $OUT/${A}_$i.json 2> $OUT/${A}_$i.err)
    $CMD > $OUT/${B}_$i.json 2> $OUT/${B}_$i.err
done
echo "# Same-box A/B (one gpurun call, one MI355X, runs alternating): the $A tree ($OLD, its own committed plan) against the $B tree under its committed plan."
echo "# $CMD"
python - $OUT $A $B <<'P'
import json, sys, glob, os
out, a, b = sys.argv[1:4]
print("# file                 frames/s  steady(5 s)  p50_ms  p99_ms  car_ms  armor_ms  first_layer_ms")
for tag in (a, b):
    for i in (1, 2):
        p = os.path.join(out, f"{tag}_{i}.json")
        try:
            d = [json.loads(l) for l in open(p) if l.startswith("{")][-1]
        except Exception as e:   # noqa: BLE001
            print(f"{tag}_{i}.json  unreadable: {e}")
            continue
        st = d["stage_ms_per_step"]
        print(f"{tag}_{i}.json{'':12s}{d['value']:9.1f} {d['steady_state']['value']:11.1f} {d['p50_ms_batch1']:7.3f} {d['p99_ms_batch1']:7.3f} "
              f"{st['network, car stage']:7.3f} {st['network, armor stage']:9.3f} {st['first layer + letterbox sampling (car + armor)']:9.3f}")
P
