# round 3, call D: 2 pairs per workgroup parity + sweep; composition tests (stand-in package, dist with the real engine, C4 full size)
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_instances.py -q -k "instance_parity" ) > $O/pytest_instances.log 2>&1; tail -3 $O/pytest_instances.log
timeout 900 python tools/sweep_shapes.py 2048 4096 > $O/sweep_shapes.txt 2>&1; tail -28 $O/sweep_shapes.txt
( time timeout 1200 python -m pytest tests/test_gpu_standin.py tests/test_gpu_dist.py -x -q ) > $O/pytest_compositions.log 2>&1; tail -30 $O/pytest_compositions.log
( time timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k c4_full ) > $O/pytest_c4.log 2>&1; tail -30 $O/pytest_c4.log
