export TMPDIR=/tmp
python tools/conv_bench.py 256,40,40,192,192 800,810,980,981 2>/dev/null
python tools/conv_bench.py 256,80,80,96,96 803,810,980 2>/dev/null
python tools/conv_bench.py 256,20,20,288,288 806,810,980 2>/dev/null
python tools/conv_bench.py 256,80,80,192,192 801,980,981 2>/dev/null
python tools/conv_bench.py 256,80,80,192,64 812,982 2>/dev/null
python tools/conv_bench.py 64,40,40,192,192 800,810,980,981 2>/dev/null
python tools/conv_bench.py 64,80,80,96,96 803,810,980 2>/dev/null
