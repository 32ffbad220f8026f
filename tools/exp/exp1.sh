export TMPDIR=/tmp
for P in 0 1 2; do
  echo "== RMR_T32_PRIO=$P"
  RMR_T32_PRIO=$P python tools/conv_bench.py 256,40,40,192,192 800,802,806,809,810 2>/dev/null
  RMR_T32_PRIO=$P python tools/conv_bench.py 256,80,80,96,96 803,806,810 2>/dev/null
  RMR_T32_PRIO=$P python tools/conv_bench.py 256,20,20,288,288 806,810 2>/dev/null
done
echo "== winograd-1D proxy (M/2 pixels, 4/3 Cin): MFMA + DMA + LDS-read structure only"
python tools/conv_bench.py 256,40,20,256,192 800,802,806,809,810 2>/dev/null
python tools/conv_bench.py 256,80,40,128,96 803,806,810 2>/dev/null
