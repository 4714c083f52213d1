# round 3, call V: friendly instances of the 9-limb two-wavefront kernel — parity (incl. four streams), sweep, probes
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_powmod.py tests/test_gpu_instances.py tests/test_gpu_stress.py tests/test_gpu_fullsize.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-200
timeout 600 python tools/sweep_shapes.py 2048 > $O/sweep.txt 2>&1; cat $O/sweep.txt | cut -c1-200
timeout 300 python tools/variant_probe.py 2>/dev/null | tail -1
