export TMPDIR=/tmp
OUT=gpurun_out/halopmc; rm -rf $OUT; mkdir -p $OUT
python tools/layer_profile.py 256 12 > /dev/null 2>&1
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; do
  d=$OUT/$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace -d $d -- python tools/layer_profile.py 256 12 > $d.log 2>&1
  python tools/pmc_summary.py $(find $d -name "*.db" | head -1) "conv_halo_kernel<4, 2, 4, 3, 4, 1>" | head -8
done
rm -rf $OUT
