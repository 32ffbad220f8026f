#!/bin/bash
# SQ counters of single convolution kernels under tools/conv_bench.py (separate rocprofv3 --pmc passes, kernel trace only).
# usage: bash tools/conv_bench_pmc.sh "<conv_bench shape>" "<kernel ids>" "<kernel name substring>" [out.txt]
export TMPDIR=/tmp
SHAPE=$1; IDS=$2; K=$3
OUT=gpurun_out/cbpmc; rm -rf $OUT; mkdir -p $OUT
echo "# rocprofv3 --pmc <set> --kernel-trace -- RMR_BENCH_DATA=2 python tools/conv_bench.py $SHAPE $IDS 3   (per dispatch: instances summed; mean over dispatches)"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$OUT/$(echo $set | tr ' ' '_' | cut -c1-40)
  RMR_BENCH_DATA=2 rocprofv3 --pmc $set --kernel-trace -d $d -- python tools/conv_bench.py $SHAPE $IDS 3 > $d.log 2>&1
  python - "$(find $d -name '*.db' | head -1)" "$K" <<'PY'
import sqlite3, sys
from collections import defaultdict
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
ix = {k: i for i, k in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ix else "name"
disp = next(k for k in ("dispatch_id", "dispatch_idx", "event_id", "id") if k in ix)
acc = defaultdict(lambda: defaultdict(float))
for r in c.execute("select * from counters_collection"):
    if sys.argv[2] in r[ix[name_col]]:
        acc[(r[ix[name_col]], r[ix["counter_name"]])][r[ix[disp]]] += float(r[ix["value"]])
dur = {n: (k, t) for n, k, t in c.execute("select name, count(*), sum(end-start) from kernels group by name")}
for (k, cn), d in sorted(acc.items()):
    n, t = dur.get(k, (0, 0))
    print(f"{k[:90]:90s} {cn:28s} {sum(d.values()) / len(d):18.0f}  n={len(d)}  avg_us={t / max(n, 1) / 1e3:.1f}")
PY
done
rm -rf $OUT
