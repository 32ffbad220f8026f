# SQ counters of one kernel family of the 256-image armor forward (separate rocprofv3 --pmc passes, kernel
# trace only).  usage: bash tools/kernel_pmc.sh "<kernel name substring>" > profiles/r01_pmc_<name>.txt
export TMPDIR=/tmp
K=${1:-conv_pw_kernel}
OUT=gpurun_out/kpmc; rm -rf $OUT; mkdir -p $OUT
python tools/layer_profile.py 256 12 > /dev/null 2>&1
echo "# SQ counters of kernels matching '$K' in: rocprofv3 --pmc <set> --kernel-trace -- python tools/layer_profile.py 256 12"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$OUT/$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d $d -- python tools/layer_profile.py 256 12 > $d.log 2>&1
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) "$K" | head -12
done
rm -rf $OUT
