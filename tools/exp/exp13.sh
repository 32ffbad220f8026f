export TMPDIR=/tmp
rm -f /tmp/rmr_packs/*.tune
RMR_WINOGRAD=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --seconds 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('RMR_WINOGRAD=1: value', round(d['value'], 1), 'all_conv', d['roofline_all_conv_launches']['achieved'])
"
echo "conv_w1d choices in the tuning files (third column 980..999):"; awk '$3 >= 980 && $3 < 1000 {n++} END {print n+0}' /tmp/rmr_packs/*.tune
rm -f /tmp/rmr_packs/*.tune
