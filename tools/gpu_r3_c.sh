# round 3, call C: parity of all pair-kernel instances incl. L = 2, launch-shape sweep with L = 2
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_instances.py -q -k "instance_parity" ) > $O/pytest_instances.log 2>&1; tail -5 $O/pytest_instances.log
timeout 900 python tools/sweep_shapes.py 2048 4096 1024 > $O/sweep_shapes.txt 2>&1; tail -50 $O/sweep_shapes.txt
