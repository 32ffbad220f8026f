export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "t32_every" 2>&1 | grep -E "passed|failed|rror" | tail -3
for W in 0 1; do
echo "== RMR_T32_WALK=$W"
RMR_T32_WALK=$W python tools/conv_bench.py 256,80,80,96,96 803,810 2>&1 | grep -v amdgpu
RMR_T32_WALK=$W python tools/conv_bench.py 256,40,40,192,192 800,802,809,810 2>&1 | grep -v amdgpu
RMR_T32_WALK=$W python tools/conv_bench.py 256,20,20,288,288 806,810 2>&1 | grep -v amdgpu
RMR_T32_WALK=$W python tools/conv_bench.py 64,40,40,192,192 800,810 2>&1 | grep -v amdgpu
RMR_T32_WALK=$W python tools/conv_bench.py 256,80,80,192,256 804 2>&1 | grep -v amdgpu
done
