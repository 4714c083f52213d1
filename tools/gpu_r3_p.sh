export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_powmod.py -m gpu -x -q -k "timesliced or segments" ) > $O/pytest_ts.log 2>&1; tail -3 $O/pytest_ts.log
timeout 1500 python tools/ts_probe.py > $O/ts_probe.txt 2>&1; cat $O/ts_probe.txt
