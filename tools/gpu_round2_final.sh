export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/profile_round.sh r02 > $O/profile_round.log 2>&1; tail -2 $O/profile_round.log
python tools/latency_probe.py > $O/latency.txt 2>&1
