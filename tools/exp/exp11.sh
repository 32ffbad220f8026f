export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_network.py tests/test_gpu_conv.py -x -q -k "fp8 or f8" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
RMR_FP8=1 python tools/layer_profile.py 256 12 2>&1 | grep -v amdgpu | head -14
python bench.py --config 4 --steps 5 --warmup 1 --no-cpu-baseline --no-latency --seconds 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('config4 value', round(d['value'], 1), 'steady', d.get('steady_state'), 'parity', d.get('parity_checked'), d.get('parity'))
"
RMR_FP8_FUSE=0 python bench.py --config 4 --steps 5 --warmup 1 --no-cpu-baseline --no-latency --seconds 4 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('config4 (passes) value', round(d['value'], 1), 'steady', d.get('steady_state'))
"
