# effective clock and MFMA-busy fraction of one conv_bench kernel: GRBM_GUI_ACTIVE / duration
# usage: bash tools/conv_clock.sh "<shape>" <kernel id> "<kernel name substring>"
export TMPDIR=/tmp
SHAPE=${1:-256,40,40,192,192}; KID=${2:-800}; K=${3:-conv_t32_kernel}
OUT=gpurun_out/cclk; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT -- python tools/conv_bench.py $SHAPE $KID 3 > $OUT.log 2>&1
python - "$(find $OUT -name '*.db' | head -1)" "$K" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t.startswith('kernels')][0] if any(t.startswith('kernels') for t in tabs) else None
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
ix = {k: i for i, k in enumerate(cols)}
rows = [r for r in c.execute("select * from counters_collection") if sys.argv[2] in (r[ix['kernel_name']] if 'kernel_name' in ix else r[ix['name']])]
from collections import defaultdict
agg = defaultdict(list)
for r in rows:
    agg[r[ix['counter_name']]].append(r[ix['value']])
dur = None
for cand in ('duration', 'dur'):
    if cand in ix:
        dur = [r[ix[cand]] for r in rows]
if dur is None and 'start' in ix and 'end' in ix:
    dur = [r[ix['end']] - r[ix['start']] for r in rows]
m = {k: sum(v) / len(v) for k, v in agg.items()}
print({k: round(v) for k, v in m.items()}, 'cols', [k for k in cols if k not in ('kernel_name',)][:30])
if dur:
    d = sum(dur) / len(dur)
    print(f"mean duration {d / 1e3:.1f} us  clock {m.get('GRBM_GUI_ACTIVE', 0) / d:.3f} GHz  mfma busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / max(m.get('GRBM_GUI_ACTIVE', 1), 1):.3f} of SIMD cycles")
PY
