export TMPDIR=/tmp
echo "tile 3 (512 x 96, eight waves, one workgroup per CU) on M1638400 N96 K864, SiLU-like data: 803 full | 843 no epilogue | 844 DMAs out of range | 845 neither | 846 weight DMAs out of range | 847 input DMAs out of range | 848 MFMAs + barriers only ; tile 10 for reference: 810 | 816 no epilogue | 818 DMAs out of range | 820 neither"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,96,96 803,843,844,845,846,847,848,810,816,818,820 20 2>&1 | grep -v amdgpu
