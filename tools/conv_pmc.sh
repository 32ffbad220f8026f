# SQ counters of one conv_bench run (separate rocprofv3 --pmc passes, kernel trace only).
# usage: bash tools/conv_pmc.sh "<shape>" <kernel id> "<kernel name substring>"
export TMPDIR=/tmp
SHAPE=${1:-256,40,40,192,192}
KID=${2:-800}
K=${3:-conv_t32_kernel}
OUT=gpurun_out/cpmc; rm -rf $OUT; mkdir -p $OUT
echo "# SQ counters of kernels matching '$K' in: rocprofv3 --pmc <set> --kernel-trace -- python tools/conv_bench.py $SHAPE $KID 3"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL"; do
  d=$OUT/$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d $d -- python tools/conv_bench.py $SHAPE $KID 3 > $d.log 2>&1
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) "$K" | head -12
done
rm -rf $OUT
