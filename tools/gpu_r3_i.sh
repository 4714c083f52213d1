# round 3, call I: role alternation + block unrolling: parity, sweep
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_instances.py tests/test_gpu_stress.py -x -q ) > $O/pytest_a.log 2>&1; tail -3 $O/pytest_a.log
timeout 900 python tools/sweep_shapes.py 2048 4096 > $O/sweep_shapes.txt 2>&1; tail -30 $O/sweep_shapes.txt
