export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "t32_every" 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/conv_bench.py 256,80,80,96,96 810,813,810,813 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py 256,80,80,96,96,3,1,1 810,813 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py 256,80,80,192,192 810,813,809,814 2>&1 | grep -v amdgpu.ids
