# round 3, call E: composition tests (stand-in package, dist with the real engine, C4 full size)
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_standin.py tests/test_gpu_dist.py -q ) > $O/pytest_compositions.log 2>&1; tail -30 $O/pytest_compositions.log
( time timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k c4_full ) > $O/pytest_c4.log 2>&1; tail -30 $O/pytest_c4.log
