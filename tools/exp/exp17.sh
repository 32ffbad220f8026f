export TMPDIR=/tmp
echo "Cin = 32: a pixel's chunk is the whole 64-byte pixel, an input DMA instruction reads one contiguous KiB (810 full | 818 all DMAs out of range | 823 weight DMAs | 824 input DMAs | 816 no epilogue | 820 no epilogue, no DMA data)"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,32,96 810,818,823,824,816,820 20 2>&1 | grep -v amdgpu
echo "Cin = 96 (192-byte pixels: 64-byte segments)"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,96,96 810,818,823,824,816,820 20 2>&1 | grep -v amdgpu
echo "Cin = 64 (128-byte pixels)"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,64,96 810,818,823,824,816,820 20 2>&1 | grep -v amdgpu
