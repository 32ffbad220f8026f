export TMPDIR=/tmp
python tools/conv_bench.py 256,40,40,192,192 802,810,813,814,815,819,820,824,827 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py 256,80,80,96,96 803,810,815,816 2>&1 | grep -v amdgpu.ids
