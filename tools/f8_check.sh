mkdir -p gpurun_out/f8
{
python -m pytest tests/test_gpu_conv.py tests/test_gpu_network.py -x -q -k "f8 or fp8" 2>&1 | tail -3
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,40,40,192,192 900,901,906 20
RMR_BENCH_DATA=2 python tools/conv_bench.py 256,80,80,96,96 902,905 20
python bench.py --dtype fp8 --no-cpu-baseline --no-latency > gpurun_out/f8/bench_fp8.json 2> gpurun_out/f8/bench_fp8.err
python bench.py --config 4 --no-cpu-baseline --no-latency > gpurun_out/f8/bench_c4.json 2> gpurun_out/f8/bench_c4.err
python - <<'P'
import json
for f in ("gpurun_out/f8/bench_fp8.json","gpurun_out/f8/bench_c4.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]);print(f, d['value'],d['ms_per_step'],d['dtype'],d['roofline']['achieved'],d['roofline']['frac'])
P
} > gpurun_out/f8/f8.txt 2>&1
cat gpurun_out/f8/f8.txt
