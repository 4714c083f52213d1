export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python tools/soak_kernels.py 20261003 3 > $O/soak.log 2>&1; tail -4 $O/soak.log
