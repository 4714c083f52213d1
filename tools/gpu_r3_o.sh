# round 3, call O: time-sliced form of the two-wavefront kernel — parity, then the shape sweep
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_powmod.py tests/test_gpu_instances.py -m gpu -x -q ) > $O/pytest_ts.log 2>&1; tail -6 $O/pytest_ts.log
timeout 900 python tools/sweep_shapes.py 2048 > $O/sweep_2048.txt 2>&1; cat $O/sweep_2048.txt
timeout 600 python tools/sweep_shapes.py 4096 > $O/sweep_4096.txt 2>&1; cat $O/sweep_4096.txt
timeout 600 python tools/ts_probe.py 4096 > $O/ts_probe_4096.txt 2>&1; cat $O/ts_probe_4096.txt
