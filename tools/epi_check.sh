mkdir -p gpurun_out/epi
{
python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -5
echo "== 96ch layer"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,96,96 810,806,803 20
echo "== 96ch layer with residual"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,96,96,3,1,1 810,806,803 20
echo "== 192ch 40x40"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,40,40,192,192 800,801,802,809,808 20
echo "== 192ch 40x40 res"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,40,40,192,192,3,1,1 800,801,802,809 20
echo "== 288ch 20x20"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,20,20,288,288 800,802,810,806,803 20
echo "== g32: s2 3x3 and 1x1"
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,192,384,3,2 952,957 20
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,40,40,768,384,1,1 950,956 20
python bench.py > gpurun_out/epi/bench.json 2> gpurun_out/epi/bench.err
python -c "
import json;d=json.loads(open('gpurun_out/epi/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'], d['steady_state'])"
} > gpurun_out/epi/epi1.txt 2>&1
cat gpurun_out/epi/epi1.txt
