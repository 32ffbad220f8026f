# Vector-memory path counters (TA / TCP / TCC) of one conv_bench run: separate rocprofv3 --pmc passes, kernel trace only.
# usage: bash tools/conv_vmem_pmc.sh "<shape>" <kernel id> "<kernel name substring>"
export TMPDIR=/tmp
SHAPE=${1:-256,80,80,96,96}
KID=${2:-810}
K=${3:-conv_t32_kernel}
OUT=gpurun_out/vpmc; rm -rf $OUT; mkdir -p $OUT
echo "# TA / TCP / TCC counters (summed over the chip) of kernels matching '$K' in: rocprofv3 --pmc <set> --kernel-trace -- python tools/conv_bench.py $SHAPE $KID 3"
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY TA_BUFFER_READ_LDS_WAVEFRONTS" "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ" "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" "TCP_TCP_TA_DATA_STALL_CYCLES TCP_TOTAL_ACCESSES" "TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ" "TCC_HIT TCC_MISS" "TCC_EA0_RDREQ TCC_TAG_STALL" "TD_TD_BUSY TD_TC_STALL"; do
  d=$OUT/$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d $d -- python tools/conv_bench.py $SHAPE $KID 3 > $d.log 2>&1
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) "$K" | head -8
done
rm -rf $OUT
