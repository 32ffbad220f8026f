export TMPDIR=/tmp
for P in 0 1 0 1; do
  rm -f /tmp/rmr_packs/*.tune
  echo "== RMR_T32_PRIO=$P"
  RMR_T32_PRIO=$P python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-latency --seconds 6 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('value', round(d['value'], 1), 'steady', d.get('steady_state'), 'all_conv', d['roofline_all_conv_launches']['achieved'], {k: (v['ms_per_step'], v['tflops']) for k, v in d['roofline_all_conv_launches']['by_instantiation'].items() if k[0] == 'g'})
"
done
