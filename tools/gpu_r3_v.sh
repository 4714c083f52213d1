# round 3, call V: friendly-modulus passes in the 3-limb instances — parity first, then latency
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_powmod.py tests/test_gpu_instances.py tests/test_gpu_stress.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -25 $O/pytest.log | cut -c1-200
timeout 300 python tools/latency_probe.py > $O/latency.txt 2>&1; cat $O/latency.txt
timeout 300 python tools/variant_probe.py 2>/dev/null | tail -1
