#!/bin/bash
# Re-measures profiles/r03_pmc_conv_traffic.json (FETCH_SIZE / WRITE_SIZE per convolution launch, two rocprofv3 --pmc passes)
# on the current kernel sources, so that bench.py's traffic_stale is false for the build that is judged.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/round
mkdir -p $OUT
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --seconds 0 > /dev/null 2>&1   # warms the tuning cache
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-profile --seconds 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*.db" | head -1) $(find $OUT/pmc_write -name "*.db" | head -1) \
    profiles/r03_pmc_conv_traffic.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- $CMD  (second pass: --pmc WRITE_SIZE)" > $OUT/pmc_traffic.log 2>&1
cp profiles/r03_pmc_conv_traffic.json $OUT/
rm -rf $OUT/pmc_fetch $OUT/pmc_write
