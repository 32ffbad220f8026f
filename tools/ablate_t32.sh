mkdir -p gpurun_out/abl
{
echo "== 96ch layer (810 full, 826 nt stores, 827 nt input DMAs, 828 both)"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,96,96 810,826,827,828 20
echo "== 192ch 40x40 (801 full, 829 nt stores, 830 both)"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,40,40,192,192 801,829,830 20
echo "== 192ch 80x80"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,192,192 801,829,830 10
} > gpurun_out/abl/abl4.txt 2>&1
cat gpurun_out/abl/abl4.txt
