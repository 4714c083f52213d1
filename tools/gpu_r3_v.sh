# round 3, call V: after the latency-instance work — full parity suite, shape sweep
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-200
timeout 600 python tools/sweep_shapes.py 2048 4096 > $O/sweep.txt 2>&1; cat $O/sweep.txt
