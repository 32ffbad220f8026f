export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "t32_every" 2>&1 | tail -3
python tools/conv_bench.py 4,40,40,192,192 220,812,2812,3812,6812,3810 200 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py 1,40,40,192,192 220,812,3812,6812,6807 200 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py 4,20,20,288,288 220,810,3810,9810,9806 200 2>&1 | grep -v amdgpu.ids
