#!/bin/bash
# Artefacts beside tools/round_profile.sh: the power-limit microbenchmarks, counters and clocks of the 3x3 kernels on
# one layer, the fp8 plan's profile and bench lines, configs[3].  usage: bash tools/round_profile_extra.sh <tag> [round]
set -u
TAG=${1:-v1}
R=${2:-r06}
export TMPDIR=/tmp
OUT=gpurun_out/round
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power > $OUT/${R}_mfma_power_f16.txt
hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_fp8_power.hip -o /tmp/mfma_fp8 && /tmp/mfma_fp8 > $OUT/${R}_mfma_power_fp8.txt
bash tools/conv_pmc.sh 256,40,40,192,192 800 conv_t32 > $OUT/${R}_conv_t32_pmc.txt 2>&1
bash tools/conv_pmc.sh 256,80,80,192,384,3,2 952 conv_g32 > $OUT/${R}_conv_g32_pmc.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_dma_rate.hip -o /tmp/lds_dma && /tmp/lds_dma > $OUT/${R}_lds_dma_rate.txt
{
  echo "# effective clock (GRBM_GUI_ACTIVE summed over 8 XCDs / duration: divide the printed GHz by 8) and MFMA-busy fraction"
  echo "# (printed fraction x 8) of one layer, M409600 N192 K1728, random operands unless noted"
  echo "conv_halo 320x192 (233):";       bash tools/conv_clock.sh 256,40,40,192,192 233 conv_halo | tail -1
  echo "conv_t32 256x192 (800):";        bash tools/conv_clock.sh 256,40,40,192,192 800 conv_t32 | tail -1
  echo "conv_t32 256x96 x2/CU (806):";   bash tools/conv_clock.sh 256,40,40,192,192 806 conv_t32 | tail -1
  echo "conv_t32 256x192, zero input:";  RMR_BENCH_DATA=1 bash tools/conv_clock.sh 256,40,40,192,192 800 conv_t32 | tail -1
  echo "conv_t32f8 256x192 (900):";      bash tools/conv_clock.sh 256,40,40,192,192 900 conv_t32f8 | tail -1
  echo "# isolated layer bench (tools/conv_bench.py), TFLOP/s:"
  python tools/conv_bench.py 256,40,40,192,192 233,800,806,900,906 2>/dev/null
  python tools/conv_bench.py 256,80,80,96,96 234,803,806,902,905 2>/dev/null
  python tools/conv_bench.py 256,20,20,288,288 214,806,905 2>/dev/null
  echo "# strided 3x3 and 1x1 layers: conv_dma (1xx) against conv_g32 (95x)"
  python tools/conv_bench.py 256,160,160,96,192,3,2 144,952 2>/dev/null
  python tools/conv_bench.py 256,80,80,192,384,3,2 144,952 2>/dev/null
  python tools/conv_bench.py 256,40,40,384,576,3,2 144,952 2>/dev/null
  python tools/conv_bench.py 256,40,40,768,384,1,1 131,950 2>/dev/null
  python tools/conv_bench.py 256,20,20,1152,576,1,1 144,950 2>/dev/null
  echo "# rocm-smi during a sustained run of kernel 800:"
  python tools/conv_bench.py 256,40,40,192,192 800 20000 > /dev/null 2>&1 &
  BG=$!
  for i in 1 2 3 4 5; do sleep 2; rocm-smi --showpower --showmaxpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | tr "\n" " "; echo; done
  wait $BG
} > $OUT/${R}_conv_clock.txt 2>&1
RMR_FP8=1 python tools/layer_profile.py 256 12 > $OUT/${R}_layer_profile_b256_fp8_${TAG}.txt 2>&1
python bench.py --config 4 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${R}_bench_config4_fp8_${TAG}.json 2> $OUT/bench_fp8.log
python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${R}_bench_config3_${TAG}.json 2>> $OUT/bench_fp8.log
python bench.py --dtype fp8 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${R}_bench_b64_fp8_${TAG}.json 2>> $OUT/bench_fp8.log
# configs[1]: batch 1 on the reference sample's 2592 x 2048 frames + 10 k-point clouds from host memory
python bench.py --config 1 > $OUT/${R}_bench_config1.json 2>> $OUT/bench_fp8.log
# the bounds SURVEY 8d asks for beside the K = 4 headline: K = 0 (car stage only) and K = 20 (kMaxBatchSize)
python bench.py --crops 0 --steps 20 --warmup 20 --no-cpu-baseline --no-latency > $OUT/${R}_bench_crops0_${TAG}.json 2>> $OUT/bench_fp8.log
python bench.py --crops 20 --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $OUT/${R}_bench_crops20_${TAG}.json 2>> $OUT/bench_fp8.log
tail -c 700 $OUT/${R}_bench_config4_fp8_${TAG}.json
