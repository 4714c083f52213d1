# round 3, call Q: after the time-sliced launch form — full GPU suite, smoke, then the whole profile round (r03)
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
( time bash tools/profile_round.sh r03 ) > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
cp profiles/r03_instr_model.json $O/ 2>/dev/null
cp profiles/r03_hbm_traffic*.json $O/ 2>/dev/null
ls gpurun_out/prof_r03 | head -80
