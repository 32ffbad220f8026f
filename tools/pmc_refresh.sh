#!/bin/bash
# Re-measures profiles/rNN_pmc_conv_traffic.json (FETCH_SIZE / WRITE_SIZE per convolution launch AND per layer, two rocprofv3
# --pmc passes, separate as MI355X_MICROARCH.md prescribes) on the current kernel sources, so that bench.py's traffic_stale is
# false for the build that is judged.  The passes and the launch-order run all use the committed pinned plan.
# usage: bash tools/pmc_refresh.sh [round]
set -u
R=${1:-r04}
export TMPDIR=/tmp
OUT=gpurun_out/round
mkdir -p $OUT
# the enqueue order of one step's convolution launches (layer name, algorithmic bytes and FLOPs) from the bench's own profiled steps
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-parity --seconds 0 --launch-order $OUT/launch_order.txt > $OUT/order_run.json 2> $OUT/order_run.log
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-parity --no-profile --seconds 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) \
    profiles/${R}_pmc_conv_traffic.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- $CMD  (second pass: --pmc WRITE_SIZE)" $OUT/launch_order.txt > $OUT/pmc_traffic.log 2>&1
cp profiles/${R}_pmc_conv_traffic.json $OUT/
rm -rf $OUT/pmc_fetch $OUT/pmc_write
