#!/bin/bash
# A/B of two builds of the library on the 3x3 layers conv_t32 carries (isolated launches, random f16 data):
#   tools/ab_t32.sh <lib A> <lib B> [reps]   (alternating A, B, A, B per layer so that clock drift hits both)
A=$1; B=$2; REPS=${3:-20}
run() { RMR_LIB=$1 RMR_BENCH_DATA=2 python tools/conv_bench.py $2 $3 $REPS 2>&1 | grep -v amdgpu.ids | sed "s|^|$4 |"; }
for spec in "256,40,40,192,192 810,806,801" "256,40,40,192,192,3,1,1 810,806" "256,80,80,96,96 810,813,806" "256,80,80,96,96,3,1,1 810,813" \
            "256,20,20,288,288 810" "256,80,80,192,192 810,801" "64,40,40,192,192 810,812" "64,80,80,96,96 810,813" "256,80,80,192,256 804" "256,40,40,384,256 804"; do
  set -- $spec
  run $A $1 $2 A; run $B $1 $2 B; run $A $1 $2 A; run $B $1 $2 B
done
