export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_network.py tests/test_gpu_prepost.py tests/test_gpu_sample.py tests/test_gpu_bench_step.py -x -q 2>&1 | tail -4
python tools/latency_probe.py 4 300 2>&1 | tail -1
python tools/latency_probe.py 4 300 2>&1 | tail -1
