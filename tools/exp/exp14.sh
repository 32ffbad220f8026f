export TMPDIR=/tmp
echo "tile 10 (256 x 96, two four-wave workgroups per CU) with parts removed, SiLU-like data: 810 full | 815 no MFMA | 816 no epilogue | 817 epilogue without stores | 818 DMAs out of range | 819 no fragment reads | 820 no epilogue, DMAs out of range | 821 MFMAs + barriers only | 823 weight DMAs out of range | 824 input DMAs out of range | 825 no vmcnt waits"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,96,96 810,815,816,817,818,819,820,821,823,824,825 20 2>&1 | grep -v amdgpu
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,40,40,192,192 810,815,816,817,818,819,820,821,823,824,825 20 2>&1 | grep -v amdgpu
