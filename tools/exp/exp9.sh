export TMPDIR=/tmp
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --seconds 6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', round(d['value'], 1), 'p50', d.get('p50_ms_batch1'), 'p99', d.get('p99_ms_batch1'), 'steady', d.get('steady_state'), 'all_conv', d['roofline_all_conv_launches']['achieved'])
"
python tools/latency_probe.py 4 200 2>&1 | tail -2
grep -c " [0-9]*8[0-9][0-9]$" /tmp/rmr_packs/*.tune | head; awk '$3 >= 1000 {print FILENAME, $0}' /tmp/rmr_packs/*.tune | head -40
