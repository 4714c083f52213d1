export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
timeout 1500 python tools/ts_probe.py > $O/ts_probe.txt 2>&1; cat $O/ts_probe.txt
