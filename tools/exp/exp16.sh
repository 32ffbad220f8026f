export TMPDIR=/tmp
for P in 0 1 0 1; do
  rm -f /tmp/rmr_packs/*.tune
  echo "== RMR_T32_WALK=$P"
  RMR_T32_WALK=$P python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --seconds 6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', round(d['value'], 1), 'steady', d['steady_state']['value'], 'all_conv', d['roofline_all_conv_launches']['achieved'], 'g10', d['roofline_all_conv_launches']['by_instantiation'].get('g10'))
"
done
