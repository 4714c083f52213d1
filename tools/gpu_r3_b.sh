# round 3, call B: parity of the two-wavefront pair kernels (every instance), then the launch-shape sweep
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_instances.py -x -q ) > $O/pytest_instances.log 2>&1; tail -5 $O/pytest_instances.log
timeout 600 python tools/sweep_shapes.py 2048 4096 > $O/sweep_shapes.txt 2>&1; tail -32 $O/sweep_shapes.txt
