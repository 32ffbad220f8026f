mkdir -p gpurun_out/abl
{
echo "== 96ch layer (810 full, 814 no epi, 815 no stores, 823 no waits, 824 no waits+no stores, 825 no waits+no epilogue, 813 no MFMA, 819 MFMA+barriers)"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,96,96 810,814,815,823,824,825,813,819 20
echo "== same, act=0 (narrow path)"
echo "== stagger"
for s in 0 2 4 8; do RMR_T32_STAGGER=$s RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,96,96 810 20; done
echo "== zeros"
RMR_BENCH_DATA=1 python tools/conv_bench.py 256,80,80,96,96 810,814 20
} > gpurun_out/abl/abl3.txt 2>&1
cat gpurun_out/abl/abl3.txt
