#!/bin/bash
# Regenerates the round's measurement artefacts on the GPU box into gpurun_out/round/ (copy the summaries into profiles/
# afterwards).  Everything runs under the committed pinned plan (profiles/plans/, bench.py --plan auto; the layer profiles
# through RMR_PLAN), so kernel stats, PMC traffic and the bench line describe the same launches.
# usage: bash tools/round_profile.sh <tag> [round]
set -u
TAG=${1:-v1}
R=${2:-r06}
export TMPDIR=/tmp
OUT=gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
bash tools/pmc_refresh.sh $R
SCMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-latency --no-parity --seconds 0"
rocprofv3 --kernel-trace --stats -d $OUT/stats -- $SCMD > $OUT/${R}_bench_b64_${TAG}_under_rocprof.json 2> $OUT/stats.log
python tools/rocpd_summary.py $(find $OUT/stats -name "*.db" | head -1) > $OUT/${R}_bench_b64_kernel_stats_${TAG}.txt
rm -rf $OUT/stats
# the driver's own command form: 20 steps behind 5 warm-up steps, every leg
python bench.py --steps 20 --warmup 5 > $OUT/${R}_bench_b64_${TAG}.json 2> $OUT/bench.log
cp $OUT/${R}_bench_b64_${TAG}.json $OUT/${R}_bench_default_final.json
for spec in "64 12 " "256 12 _b256" "4 12 _b4" "1 1 _b1"; do
    set -- $spec
    python tools/layer_profile.py $1 $2 > $OUT/${R}_layer_profile${3:-}_${TAG}.txt 2>&1
done
python tools/latency_probe.py 4 30 > /dev/null 2>&1
RMR_GRAPH=0 rocprofv3 --kernel-trace -d $OUT/lat -- python tools/latency_probe.py 4 40 > $OUT/latency_probe.log 2>&1
python tools/trace_gaps.py $(find $OUT/lat -name "*.db" | head -1) > $OUT/${R}_latency_trace_${TAG}.txt 2>&1
rm -rf $OUT/lat
python tools/latency_probe.py 4 100 > $OUT/${R}_latency_probe_${TAG}.txt 2>&1
tail -2 $OUT/${R}_bench_b64_${TAG}.json | cut -c1-600
