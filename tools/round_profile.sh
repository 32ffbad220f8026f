#!/bin/bash
# Regenerates the round's measurement artefacts on the GPU box into gpurun_out/round/ (copy the
# summaries into profiles/ afterwards).  usage: bash tools/round_profile.sh <tag>
set -u
TAG=${1:-v4}
export TMPDIR=/tmp
OUT=gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --seconds 0 > /dev/null 2>&1   # warms the tuning cache
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-profile --seconds 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) \
    profiles/r03_pmc_conv_traffic.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- $CMD  (second pass: --pmc WRITE_SIZE)" > $OUT/pmc_traffic.log 2>&1
cp profiles/r03_pmc_conv_traffic.json $OUT/
rm -rf $OUT/pmc_fetch $OUT/pmc_write
SCMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-latency --seconds 0"
rocprofv3 --kernel-trace --stats -d $OUT/stats -- $SCMD > $OUT/r03_bench_b64_${TAG}_under_rocprof.json 2> $OUT/stats.log
python tools/rocpd_summary.py $(find $OUT/stats -name "*.db" | head -1) > $OUT/r03_bench_b64_kernel_stats_${TAG}.txt
rm -rf $OUT/stats
python bench.py --steps 5 --warmup 2 > $OUT/r03_bench_b64_${TAG}.json 2> $OUT/bench.log
python tools/layer_profile.py 64 12 > $OUT/r03_layer_profile_${TAG}.txt 2>&1
python tools/layer_profile.py 256 12 > $OUT/r03_layer_profile_b256_${TAG}.txt 2>&1
python tools/layer_profile.py 4 12 > $OUT/r03_layer_profile_b4_${TAG}.txt 2>&1
python tools/layer_profile.py 1 1 > $OUT/r03_layer_profile_b1_${TAG}.txt 2>&1
python tools/latency_probe.py 4 30 > /dev/null 2>&1
RMR_GRAPH=0 rocprofv3 --kernel-trace -d $OUT/lat -- python tools/latency_probe.py 4 40 > $OUT/latency_probe.log 2>&1
python tools/trace_gaps.py $(find $OUT/lat -name "*.db" | head -1) > $OUT/r03_latency_trace_${TAG}.txt 2>&1
rm -rf $OUT/lat
python tools/latency_probe.py 4 100 > $OUT/r03_latency_probe_${TAG}.txt 2>&1
tail -2 $OUT/r03_bench_b64_${TAG}.json | cut -c1-600
