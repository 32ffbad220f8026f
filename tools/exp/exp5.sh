export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "t32_every" 2>&1 | tail -5
python tools/conv_bench.py 256,40,40,192,192 800,802,810,813,814,815,816 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py 256,80,80,96,96 803,810,815,816 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py 256,20,20,288,288 806,810,815,816 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py 256,80,80,192,256 804,817 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py 256,80,80,192,192 801,803,814,816 2>&1 | grep -v amdgpu.ids
